// Index kernels: sample->batch shift (kgcnn/ops/partition.py:97-162), the one-pass index preparation shared by
// every gather / pooling call, CSR construction and the stable argsort of kgcnn/layers/pooling.py:66.
//
// All of these are integer / byte work bound by HBM (or, at QM9 batch sizes, by launch latency): coalesced
// 16-byte reads of the (M,2) int64 index rows, a binary search over the L2-resident row_splits per edge,
// coalesced int32 column writes.  No LDS staging is needed: row_splits for 10^5 graphs is 800 KB (L2 resident).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mp_common.h"
#include "mp_edge_prepare.h"

namespace {

// largest g in [0, G) with splits[g] <= e  (graph owning flat element e); splits has G+1 entries.
__device__ __forceinline__ int64_t owner_of(const int64_t* __restrict__ splits, int64_t G, int64_t e) {
  int64_t lo = 0, hi = G;  // invariant: splits[lo] <= e < splits[hi]
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (splits[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void shift_index_kernel(const int64_t* __restrict__ idx, int64_t M, int K,
                                   const int64_t* __restrict__ node_splits, const int64_t* __restrict__ edge_splits,
                                   int64_t G, int64_t sign, int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < M; e += stride) {
    const int64_t shift = sign * node_splits[owner_of(edge_splits, G, e)];
    if (K == 2) {
      // one 16-byte row per lane: dwordx4 load / store
      const longlong2 v = reinterpret_cast<const longlong2*>(idx)[e];
      reinterpret_cast<longlong2*>(out)[e] = make_longlong2(v.x + shift, v.y + shift);
    } else {
      for (int k = 0; k < K; ++k) out[e * K + k] = idx[e * K + k] + shift;
    }
  }
}

__global__ void index_prepare_kernel(const int64_t* __restrict__ idx, int64_t M, int K,
                                     const int64_t* __restrict__ node_splits,
                                     const int64_t* __restrict__ edge_splits, int64_t G, int64_t N,
                                     int32_t* __restrict__ cols, int32_t* __restrict__ flags) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  int local_flags = 0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e - (threadIdx.x & 63) < M;
       e += stride) {
    // one wave-uniform owner search (scalar loads) for the wave's first edge, then a short per-lane walk
    const int64_t e_wave = __builtin_amdgcn_readfirstlane(static_cast<int>(e - (threadIdx.x & 63)));
    int64_t g = owner_of(edge_splits, G, e_wave < M ? e_wave : M - 1);
    if (e >= M) continue;
    while (g + 1 < G && edge_splits[g + 1] <= e) ++g;
    const int64_t base = node_splits[g];
    const int64_t n_g = node_splits[g + 1] - base;
    // shift of the previous edge (for the sortedness check of the batch-level ids)
    int64_t base_prev = base;
    if (e > 0 && edge_splits[g] > e - 1) base_prev = node_splits[owner_of(edge_splits, G, e - 1)];
    for (int k = 0; k < K; ++k) {
      int64_t v = idx[e * K + k];
      if (v < 0 || v >= n_g) {
        local_flags |= MP_FLAG_OOB;
        v = v < 0 ? 0 : (n_g > 0 ? n_g - 1 : 0);
      }
      int64_t s = v + base;
      if (s >= N) s = N > 0 ? N - 1 : 0;  // only reachable for an empty graph with edges (already flagged)
      cols[static_cast<int64_t>(k) * M + e] = static_cast<int32_t>(s);
      if (e > 0 && k < 2) {
        const int64_t sp = idx[(e - 1) * K + k] + base_prev;
        if (sp > s) local_flags |= (k == 0 ? MP_FLAG_UNSORTED_COL0 : MP_FLAG_UNSORTED_COL1);
      }
    }
  }
  mp_publish_flags(flags, local_flags);
}

// ptr[n] = first position e with seg[e] >= n; seg sorted ascending.  ptr is pre-zeroed so that a caller who
// wrongly claims sortedness still gets in-range offsets (rows may then be wrong, never out of bounds).
__global__ void csr_from_sorted_kernel(const int32_t* __restrict__ seg, int64_t M, int64_t N,
                                       int32_t* __restrict__ ptr) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e <= M; e += stride) {
    int64_t prev = e > 0 ? seg[e - 1] : -1;
    int64_t cur = e < M ? seg[e] : N;
    if (prev < -1) prev = -1;
    if (cur > N) cur = N;
    for (int64_t n = prev + 1; n <= cur; ++n) ptr[n] = static_cast<int32_t>(e);
  }
}

// index_prepare (K = 2) fused with the geometry pre-step NodePosition -> LazySubtract -> EuclideanNorm
// (kgcnn/literature/Schnet.py:116-117): one pass over the (M,2) int64 rows yields receiver / sender ids and the
// edge distance the Gauss expansion starts from.
template <bool LDS_SPLITS, bool COL1 = false>
__global__ __launch_bounds__(256) void edge_prepare_kernel(mp_prep::EdgePrepArgs p) {
  mp_prep::edge_prepare_body<LDS_SPLITS, COL1>(p, blockIdx.x, gridDim.x);
}

__global__ void iota_kernel(int32_t* __restrict__ out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = static_cast<int32_t>(i);
}

inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

}  // namespace

extern "C" {

int mp_shift_index_i64(const int64_t* idx, int64_t M, int K, const int64_t* node_splits, const int64_t* edge_splits,
                       int64_t G, int direction, int64_t* out, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && K >= 1 && G >= 0, "mp_shift_index_i64: bad sizes M=%lld K=%d G=%lld", (long long)M, K,
             (long long)G);
  MP_REQUIRE(direction == 1 || direction == -1, "mp_shift_index_i64: direction must be +1 or -1");
  if (M == 0) return MP_OK;
  MP_REQUIRE(idx && out && node_splits && edge_splits && G > 0, "mp_shift_index_i64: null pointer / no graphs");
  shift_index_kernel<<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(idx, M, K, node_splits, edge_splits, G,
                                                                          direction, out);
  return mp::check_launch("mp_shift_index_i64");
}

int mp_index_prepare_i64(const int64_t* idx, int64_t M, int K, const int64_t* node_splits, const int64_t* edge_splits,
                         int64_t G, int64_t N, int32_t* cols, int32_t* flags, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && K >= 1 && G >= 0 && N >= 0, "mp_index_prepare_i64: bad sizes");
  MP_REQUIRE(N < (int64_t{1} << 31) && M < (int64_t{1} << 31), "mp_index_prepare_i64: N, M must fit int32");
  MP_REQUIRE(flags != nullptr, "mp_index_prepare_i64: null flags");
  if (M == 0) return MP_OK;
  MP_REQUIRE(idx && cols && node_splits && edge_splits && G > 0, "mp_index_prepare_i64: null pointer / no graphs");
  if (K == 2) {
    // (receiver, sender) pairs - every conv of the reference: one 16-B load per edge, row_splits searched in LDS for
    // batches up to 1023 graphs (the edge-preparation body of the fused forward, with both sortedness flags)
    mp_prep::EdgePrepArgs p{idx, M, node_splits, edge_splits, G, N, nullptr, cols, cols + M, nullptr, flags};
    if (G <= mp_prep::PREP_LDS_GRAPHS) {
      edge_prepare_kernel<true, true><<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(p);
    } else {
      edge_prepare_kernel<false, true><<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(p);
    }
    return mp::check_launch("mp_index_prepare_i64");
  }
  index_prepare_kernel<<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(idx, M, K, node_splits, edge_splits, G, N,
                                                                            cols, flags);
  return mp::check_launch("mp_index_prepare_i64");
}

int mp_edge_prepare_i64_f32(const int64_t* idx, int64_t M, const int64_t* node_splits, const int64_t* edge_splits,
                            int64_t G, int64_t N, const float* xyz, int32_t* recv, int32_t* send, float* dist,
                            int32_t* flags, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && G >= 0 && N >= 0, "mp_edge_prepare_i64_f32: bad sizes");
  MP_REQUIRE(N < (int64_t{1} << 31) && M < (int64_t{1} << 31), "mp_edge_prepare_i64_f32: N, M must fit int32");
  MP_REQUIRE(flags != nullptr, "mp_edge_prepare_i64_f32: null flags");
  MP_REQUIRE((dist == nullptr) || (xyz != nullptr), "mp_edge_prepare_i64_f32: dist requested without coordinates");
  if (M == 0) return MP_OK;
  MP_REQUIRE(idx && recv && send && node_splits && edge_splits && G > 0, "mp_edge_prepare_i64_f32: null pointer");
  mp_prep::EdgePrepArgs p{idx, M, node_splits, edge_splits, G, N, xyz, recv, send, dist, flags};
  if (G <= mp_prep::PREP_LDS_GRAPHS) {
    edge_prepare_kernel<true><<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(p);
  } else {
    edge_prepare_kernel<false><<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(p);
  }
  return mp::check_launch("mp_edge_prepare_i64_f32");
}

int mp_csr_from_sorted_i32(const int32_t* seg, int64_t M, int64_t N, int32_t* ptr, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && N >= 0 && ptr, "mp_csr_from_sorted_i32: bad arguments");
  MP_REQUIRE(M == 0 || seg, "mp_csr_from_sorted_i32: null seg");
  MP_HIP(hipMemsetAsync(ptr, 0, sizeof(int32_t) * (N + 1), mp::as_stream(stream)));
  csr_from_sorted_kernel<<<mp::grid_for(M + 1), 256, 0, mp::as_stream(stream)>>>(seg, M, N, ptr);
  return mp::check_launch("mp_csr_from_sorted_i32");
}

int mp_sort_workspace_bytes(int64_t M, size_t* bytes_out_host) {
  MP_REQUIRE(bytes_out_host && M >= 0, "mp_sort_workspace_bytes: bad arguments");
  size_t temp = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, temp, static_cast<const int32_t*>(nullptr),
                                           static_cast<int32_t*>(nullptr), static_cast<const int32_t*>(nullptr),
                                           static_cast<int32_t*>(nullptr), static_cast<size_t>(M > 0 ? M : 1), 0, 32,
                                           hipStream_t{nullptr}, false);
  if (e != hipSuccess) {
    mp::set_error("rocprim temp-size query failed: %s", hipGetErrorString(e));
    return MP_EHIP;
  }
  *bytes_out_host = align256(temp) + align256(sizeof(int32_t) * static_cast<size_t>(M > 0 ? M : 1));
  return MP_OK;
}

int mp_sort_segments_i32(const int32_t* seg, int64_t M, int32_t* seg_sorted, int32_t* perm, void* ws, size_t ws_bytes,
                         mpStream_t stream) {
  MP_REQUIRE(M >= 0, "mp_sort_segments_i32: bad M");
  if (M == 0) return MP_OK;
  MP_REQUIRE(seg && seg_sorted && perm && ws, "mp_sort_segments_i32: null pointer");
  size_t need = 0;
  int rc = mp_sort_workspace_bytes(M, &need);
  if (rc != MP_OK) return rc;
  MP_REQUIRE(ws_bytes >= need, "mp_sort_segments_i32: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = mp::as_stream(stream);
  int32_t* iota = static_cast<int32_t*>(ws);
  void* temp = static_cast<char*>(ws) + align256(sizeof(int32_t) * static_cast<size_t>(M));
  size_t temp_bytes = ws_bytes - align256(sizeof(int32_t) * static_cast<size_t>(M));
  iota_kernel<<<mp::grid_for(M), 256, 0, s>>>(iota, M);
  // LSD radix sort is stable: equal receivers keep their original edge order, as tf.argsort(stable=True) does.
  MP_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, seg, seg_sorted, iota, perm, static_cast<size_t>(M), 0, 32, s,
                                   false));
  return mp::check_launch("mp_sort_segments_i32");
}

}  // extern "C"
