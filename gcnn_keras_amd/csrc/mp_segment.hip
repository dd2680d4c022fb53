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

template <int W, typename PtrT>
__global__ void segment_reduce_csr_kernel(int op, const float* __restrict__ data, int64_t M, int64_t row_elems,
                                          const PtrT* __restrict__ ptr, const int32_t* __restrict__ perm,
                                          int64_t n_out, const float* __restrict__ weight, int normalize,
                                          float* __restrict__ out, const int32_t* __restrict__ row_index = nullptr,
                                          int64_t n_rows = 0, int act = 0, float act_alpha = 0.0f) {
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
    constexpr int UB = 8;
    // rows are fetched UB (eight) at a time (independent loads in flight), then folded in edge order
    for (int64_t e0 = a; e0 < b; e0 += UB) {
      float v[UB][W];
      float wv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        wv[u] = 1.0f;
        if (e0 + u < b) {
          const int64_t r = perm ? static_cast<int64_t>(perm[e0 + u]) : (e0 + u);
          int64_t src = r;
          if (row_index) {
            // gather-on-read: the row of edge r is x[row_index[r]] (GatherNodesOutgoing fused into the reduce)
            src = row_index[r];
            src = src < 0 ? 0 : (src >= n_rows ? n_rows - 1 : src);
          }
          load_vec<W>(base + src * row_elems, v[u]);
          if (weight) wv[u] = weight[r];
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (e0 + u < b) {
          if (weight) {
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
    }
    if (op == MP_MEAN && b > a) {
      const float cnt = static_cast<float>(b - a);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i] = acc[i] / cnt;
    }
    if (normalize && weight) {
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

template <typename PtrT>
int launch_segment_reduce(int op, const float* data, int64_t M, int64_t row_elems, const PtrT* ptr,
                          const int32_t* perm, int64_t N_out, const float* weight, int normalize, float* out,
                          hipStream_t s, const char* what) {
  if (row_elems % 4 == 0 && aligned16(data) && aligned16(out)) {
    segment_reduce_csr_kernel<4, PtrT><<<mp::grid_for(N_out * (row_elems / 4)), 256, 0, s>>>(
        op, data, M, row_elems, ptr, perm, N_out, weight, normalize, out);
  } else {
    segment_reduce_csr_kernel<1, PtrT><<<mp::grid_for(N_out * row_elems), 256, 0, s>>>(
        op, data, M, row_elems, ptr, perm, N_out, weight, normalize, out);
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
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_SOFTPLUS2, "mp_gather_segment_reduce_csr_f32: unknown activation");
  if (N_out == 0) return MP_OK;
  MP_REQUIRE(ptr && out && (M == 0 || (x && send && N > 0)), "mp_gather_segment_reduce_csr_f32: null pointer");
  hipStream_t s = mp::as_stream(stream);
  if (row_elems % 4 == 0 && aligned16(x) && aligned16(out)) {
    segment_reduce_csr_kernel<4, int32_t><<<mp::grid_for(N_out * (row_elems / 4)), 256, 0, s>>>(
        op, x, M, row_elems, ptr, perm, N_out, weight, normalize_by_weight, out, send, N, act, act_alpha);
  } else {
    segment_reduce_csr_kernel<1, int32_t><<<mp::grid_for(N_out * row_elems), 256, 0, s>>>(
        op, x, M, row_elems, ptr, perm, N_out, weight, normalize_by_weight, out, send, N, act, act_alpha);
  }
  return mp::check_launch("mp_gather_segment_reduce_csr_f32");
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
