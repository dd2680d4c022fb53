// Segment reductions: the sorted tf.math.segment_{sum,mean,max,min} of kgcnn/ops/segment.py:39-46 as used by
// PoolingLocalEdges / PoolingWeightedLocalEdges (kgcnn/layers/pooling.py:63-76, :150-170) and PoolingNodes
// (pooling.py:215-218), segment_softmax (ops/segment.py:5-24) and the relational scatter (ops/scatter.py:18-23).
//
// Design (HBM-bound, ~1 flop/B): receiver-parallel over a CSR of the sorted receiver ids.  One work item owns
// (output row, 16-byte feature chunk) and walks its segment [ptr[n], ptr[n+1]) front to back, so
//  * every message row is read exactly once with dwordx4 loads, 32 adjacent lanes per 512-byte row,
//  * the float accumulation order is the edge order - the order TF's CPU kernel uses after the stable sort -
//    which makes the result deterministic and lets the parity test ask for equality with the sequential oracle,
//  * no atomics, no zero-init pass: rows without edges are written as 0 here (the has_unconnected pad,
//    pooling.py:74-76, fused), interior gaps likewise (TF fills missing segment ids with 0).
// For an unsorted index list the rows are visited through the stable argsort permutation instead of
// materialising tf.gather(dens, node_order) (pooling.py:68) - the (M,F) copy of the reference never exists.
#include <type_traits>

#include "mp_common.h"

namespace {

template <int W>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[W]) {
  if constexpr (W == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}

template <int W>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[W]) {
  if constexpr (W == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *p = v[0];
  }
}

// PERM / GATHER / WEIGHT: whether perm, row_index, weight are given - compile-time, because a run-time null test in front
// of every load of the unrolled rounds becomes a branch per load, and the compiler then waits for each load before the
// next one is issued (seen in the ISA: global_load, s_waitcnt vmcnt(0), s_cbranch, ... - one round trip per edge).
template <int W, typename PtrT, bool PERM, bool GATHER, bool WEIGHT>
__global__ __launch_bounds__(256) void segment_reduce_csr_kernel(int op, const float* __restrict__ data, int64_t M, int64_t row_elems,
                                          const PtrT* __restrict__ ptr, const int32_t* __restrict__ perm,
                                          int64_t n_out, const float* __restrict__ weight, int normalize,
                                          float* __restrict__ out, const int32_t* __restrict__ row_index,
                                          int64_t n_rows, int act, float act_alpha) {
  const int64_t chunks = row_elems / W;
  const int64_t total = n_out * chunks;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % chunks;
    const int64_t n = t / chunks;
    int64_t a = static_cast<int64_t>(ptr[n]);
    int64_t b = static_cast<int64_t>(ptr[n + 1]);
    a = a < 0 ? 0 : (a > M ? M : a);
    b = b < a ? a : (b > M ? M : b);
    float acc[W];
#pragma unroll
    for (int i = 0; i < W; ++i) acc[i] = 0.0f;
    float wsum = 0.0f;
    const float* base = data + c * W;
    // rows are fetched UB at a time (independent loads in flight), then folded in edge order.  Eight at a time suits the
    // short segments of molecular graphs; a segment with 16 or more rows left (hub nodes of a citation graph - Cora's
    // largest receiver has 129 edges - or a whole graph under PoolingNodes) takes 16 per round: the walk of such a
    // segment is a chain of dependent round trips (index -> row), and it alone set the kernel's time (37 us at config 5).
    auto fold = [&](int64_t e0, auto ub_tag) {
      constexpr int UB = decltype(ub_tag)::value;
      // Three unconditional phases (tail slots repeat the segment's last row: loads only, masked in the fold): every
      // phase's loads are independent, so UB of them are in flight at once.  Written as one predicated block per row the
      // index load of row u+1 sits behind the row load of row u and - vmcnt retires in order - waits for it: measured one
      // full round trip per edge (0.2 us), 35 us for Cora's 129-edge hub.
      float v[UB][W];
      float wv[UB];
      int64_t rr[UB], src[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t ec = e0 + u < b ? e0 + u : b - 1;
        rr[u] = PERM ? static_cast<int64_t>(perm[ec]) : ec;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        src[u] = rr[u];
        if constexpr (GATHER) {
          // gather-on-read: the row of edge r is x[row_index[r]] (GatherNodesOutgoing fused into the reduce)
          const int64_t j = row_index[rr[u]];
          src[u] = j < 0 ? 0 : (j >= n_rows ? n_rows - 1 : j);
        }
        wv[u] = 1.0f;
        if constexpr (WEIGHT) wv[u] = weight[rr[u]];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) load_vec<W>(base + src[u] * row_elems, v[u]);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (e0 + u < b) {
          if constexpr (WEIGHT) {
            wsum += wv[u];
#pragma unroll
            for (int i = 0; i < W; ++i) v[u][i] *= wv[u];
          }
          if (e0 + u == a) {
#pragma unroll
            for (int i = 0; i < W; ++i) acc[i] = v[u][i];
          } else if (op == MP_MAX) {
#pragma unroll
            for (int i = 0; i < W; ++i) acc[i] = fmaxf(acc[i], v[u][i]);
          } else if (op == MP_MIN) {
#pragma unroll
            for (int i = 0; i < W; ++i) acc[i] = fminf(acc[i], v[u][i]);
          } else {
#pragma unroll
            for (int i = 0; i < W; ++i) acc[i] += v[u][i];
          }
        }
      }
    };
    int64_t e0 = a;
    for (; b - e0 >= 16; e0 += 16) fold(e0, std::integral_constant<int, 16>());
    for (; e0 < b; e0 += 8) fold(e0, std::integral_constant<int, 8>());
    if (op == MP_MEAN && b > a) {
      const float cnt = static_cast<float>(b - a);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i] = acc[i] / cnt;
    }
    if (normalize && WEIGHT) {
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i] = wsum == 0.0f ? 0.0f : acc[i] / wsum;  // tf.math.divide_no_nan
    }
    if (act != 0) {
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i] = mp_apply_act(act, act_alpha, acc[i]);
    }
    store_vec<W>(out + n * row_elems + c * W, acc);
  }
}

// segment_softmax: one work item per (segment, feature); three sequential sweeps (max, sum of exp, normalise).
template <typename PtrT>
__global__ void segment_softmax_csr_kernel(const float* __restrict__ a, int64_t M, int64_t row_elems,
                                           const PtrT* __restrict__ ptr, const int32_t* __restrict__ perm,
                                           int64_t n_seg, float* __restrict__ out) {
  const int64_t total = n_seg * row_elems;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % row_elems;
    const int64_t n = t / row_elems;
    int64_t lo = static_cast<int64_t>(ptr[n]);
    int64_t hi = static_cast<int64_t>(ptr[n + 1]);
    lo = lo < 0 ? 0 : (lo > M ? M : lo);
    hi = hi < lo ? lo : (hi > M ? M : hi);
    if (hi == lo) continue;
    float mx = 0.0f;
    for (int64_t e = lo; e < hi; ++e) {
      const int64_t r = perm ? static_cast<int64_t>(perm[e]) : e;
      const float v = a[r * row_elems + c];
      mx = e == lo ? v : fmaxf(mx, v);
    }
    float sum = 0.0f;
    for (int64_t e = lo; e < hi; ++e) {
      const int64_t r = perm ? static_cast<int64_t>(perm[e]) : e;
      sum += expf(a[r * row_elems + c] - mx);
    }
    for (int64_t e = lo; e < hi; ++e) {
      const int64_t r = perm ? static_cast<int64_t>(perm[e]) : e;
      out[r * row_elems + c] = expf(a[r * row_elems + c] - mx) / sum;
    }
  }
}

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  // ordered-int trick: valid for all non-NaN floats
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void scatter_relational_kernel(int op, const float* __restrict__ edges, int64_t M, int64_t row_elems,
                                          const int32_t* __restrict__ recv, const int32_t* __restrict__ relation,
                                          int64_t N, int64_t R, float* __restrict__ out) {
  const int64_t total = M * row_elems;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % row_elems;
    const int64_t e = t / row_elems;
    const int64_t n = recv[e];
    const int64_t r = relation[e];
    if (n < 0 || n >= N || r < 0 || r >= R) continue;
    float* dst = out + (n * R + r) * row_elems + c;
    const float v = edges[t];
    if (op == MP_SUM) atomicAdd(dst, v);
    else if (op == MP_MAX) atomic_max_f32(dst, v);
    else atomic_min_f32(dst, v);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// dispatch of the three compile-time flags
template <int W, typename PtrT>
void launch_reduce_flags(unsigned grid, hipStream_t s, int op, const float* data, int64_t M, int64_t row_elems,
                         const PtrT* ptr, const int32_t* perm, int64_t n_out, const float* weight, int normalize,
                         float* out, const int32_t* row_index, int64_t n_rows, int act, float alpha) {
#define MP_SEG_LAUNCH(P, G, WT)                                                                                   \
  segment_reduce_csr_kernel<W, PtrT, P, G, WT><<<grid, 256, 0, s>>>(op, data, M, row_elems, ptr, perm, n_out, weight, \
                                                                    normalize, out, row_index, n_rows, act, alpha)
  const int key = (perm ? 4 : 0) | (row_index ? 2 : 0) | (weight ? 1 : 0);
  switch (key) {
    case 0: MP_SEG_LAUNCH(false, false, false); break;
    case 1: MP_SEG_LAUNCH(false, false, true); break;
    case 2: MP_SEG_LAUNCH(false, true, false); break;
    case 3: MP_SEG_LAUNCH(false, true, true); break;
    case 4: MP_SEG_LAUNCH(true, false, false); break;
    case 5: MP_SEG_LAUNCH(true, false, true); break;
    case 6: MP_SEG_LAUNCH(true, true, false); break;
    default: MP_SEG_LAUNCH(true, true, true); break;
  }
#undef MP_SEG_LAUNCH
}

template <typename PtrT>
int launch_segment_reduce(int op, const float* data, int64_t M, int64_t row_elems, const PtrT* ptr,
                          const int32_t* perm, int64_t N_out, const float* weight, int normalize, float* out,
                          hipStream_t s, const char* what, const int32_t* row_index = nullptr, int64_t n_rows = 0,
                          int act = 0, float alpha = 0.0f) {
  if (row_elems % 4 == 0 && aligned16(data) && aligned16(out)) {
    launch_reduce_flags<4, PtrT>(mp::grid_for(N_out * (row_elems / 4)), s, op, data, M, row_elems, ptr, perm, N_out,
                                 weight, normalize, out, row_index, n_rows, act, alpha);
  } else {
    launch_reduce_flags<1, PtrT>(mp::grid_for(N_out * row_elems), s, op, data, M, row_elems, ptr, perm, N_out, weight,
                                 normalize, out, row_index, n_rows, act, alpha);
  }
  return mp::check_launch(what);
}

}  // namespace

extern "C" {

int mp_segment_reduce_csr_f32(int op, const float* data, int64_t M, int64_t row_elems, const int32_t* ptr,
                              const int32_t* perm, int64_t N_out, const float* weight, int normalize_by_weight,
                              float* out, mpStream_t stream) {
  MP_REQUIRE(op >= MP_SUM && op <= MP_MIN, "mp_segment_reduce_csr_f32: unknown op %d", op);
  MP_REQUIRE(M >= 0 && N_out >= 0 && row_elems >= 1, "mp_segment_reduce_csr_f32: bad sizes");
  if (N_out == 0) return MP_OK;
  MP_REQUIRE(ptr && out && (M == 0 || data), "mp_segment_reduce_csr_f32: null pointer");
  return launch_segment_reduce<int32_t>(op, data, M, row_elems, ptr, perm, N_out, weight, normalize_by_weight, out,
                                        mp::as_stream(stream), "mp_segment_reduce_csr_f32");
}

int mp_gather_segment_reduce_csr_f32(int op, const float* x, int64_t N, int64_t row_elems, const int32_t* send,
                                     int64_t M, const int32_t* ptr, const int32_t* perm, int64_t N_out,
                                     const float* weight, int normalize_by_weight, int act, float act_alpha, float* out,
                                     mpStream_t stream) {
  MP_REQUIRE(op >= MP_SUM && op <= MP_MIN, "mp_gather_segment_reduce_csr_f32: unknown op %d", op);
  MP_REQUIRE(M >= 0 && N >= 0 && N_out >= 0 && row_elems >= 1, "mp_gather_segment_reduce_csr_f32: bad sizes");
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST, "mp_gather_segment_reduce_csr_f32: unknown activation");
  if (N_out == 0) return MP_OK;
  MP_REQUIRE(ptr && out && (M == 0 || (x && send && N > 0)), "mp_gather_segment_reduce_csr_f32: null pointer");
  return launch_segment_reduce<int32_t>(op, x, M, row_elems, ptr, perm, N_out, weight, normalize_by_weight, out,
                                        mp::as_stream(stream), "mp_gather_segment_reduce_csr_f32", send, N, act,
                                        act_alpha);
}

int mp_pool_graph_f32(int op, const float* x, const int64_t* row_splits, int64_t G, int64_t row_elems,
                      const float* weight, float* out, mpStream_t stream) {
  MP_REQUIRE(op >= MP_SUM && op <= MP_MIN, "mp_pool_graph_f32: unknown op %d", op);
  MP_REQUIRE(G >= 0 && row_elems >= 1, "mp_pool_graph_f32: bad sizes");
  if (G == 0) return MP_OK;
  MP_REQUIRE(row_splits && out, "mp_pool_graph_f32: null pointer");
  // M is only a clamp for the offsets; row_splits is trusted to end at the value count (ragged_validate=False).
  const int64_t M = INT64_MAX;
  return launch_segment_reduce<int64_t>(op, x, M, row_elems, row_splits, nullptr, G, weight, 0, out,
                                        mp::as_stream(stream), "mp_pool_graph_f32");
}

int mp_segment_softmax_csr_f32(const float* a, int64_t M, int64_t row_elems, const int32_t* ptr, const int32_t* perm,
                               int64_t N, float* out, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && N >= 0 && row_elems >= 1, "mp_segment_softmax_csr_f32: bad sizes");
  if (M == 0 || N == 0) return MP_OK;
  MP_REQUIRE(a && ptr && out, "mp_segment_softmax_csr_f32: null pointer");
  segment_softmax_csr_kernel<int32_t><<<mp::grid_for(N * row_elems), 256, 0, mp::as_stream(stream)>>>(
      a, M, row_elems, ptr, perm, N, out);
  return mp::check_launch("mp_segment_softmax_csr_f32");
}

int mp_scatter_relational_f32(int op, const float* edges, int64_t M, int64_t row_elems, const int32_t* recv,
                              const int32_t* relation, int64_t N, int64_t R, float* out, mpStream_t stream) {
  MP_REQUIRE(op == MP_SUM || op == MP_MAX || op == MP_MIN, "mp_scatter_relational_f32: op must be sum/max/min");
  MP_REQUIRE(M >= 0 && N >= 0 && R >= 1 && row_elems >= 1, "mp_scatter_relational_f32: bad sizes");
  if (M == 0) return MP_OK;
  MP_REQUIRE(edges && recv && relation && out, "mp_scatter_relational_f32: null pointer");
  scatter_relational_kernel<<<mp::grid_for(M * row_elems), 256, 0, mp::as_stream(stream)>>>(op, edges, M, row_elems,
                                                                                           recv, relation, N, R, out);
  return mp::check_launch("mp_scatter_relational_f32");
}

}  // extern "C"
