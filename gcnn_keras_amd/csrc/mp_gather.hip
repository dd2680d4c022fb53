// Row gathers: tf.gather(node, idx, axis=0) of kgcnn/layers/gather.py:83,228, GatherState's tf.repeat
// (gather.py:363-369) and the Keras Embedding lookup (kgcnn/layers/modules.py:526-528).
//
// HBM/L2-bound copies.  Work item = (edge, selected column, 16-byte chunk of the row); consecutive lanes take
// consecutive chunks of one row, so a 512-byte F=128 row is read and written by 32 adjacent lanes with dwordx4
// accesses (2 rows per wave64 instruction).  The node table (1.2 MB at config 2, 58 MB per GPU at config 4) is
// L2 / Infinity-Cache resident, the (M, .) output stream is the HBM traffic.
#include "mp_common.h"

namespace {

// Four independent (index -> row -> store) chains per thread and iteration: a single dependent chain per wave leaves
// the memory system latency-bound (3.3 TB/s measured); with four in flight the copy streams.
template <typename VecT>
__global__ void gather_cols_kernel(const VecT* __restrict__ x, int64_t N, int64_t chunks,
                                   const int32_t* __restrict__ cols, int64_t M, int ncols, int c0, int c1, int c2,
                                   int c3, VecT* __restrict__ out) {
  constexpr int U = 4;
  const int64_t total = M * ncols * chunks;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t0 < total; t0 += U * stride) {
    int64_t row[U];
    int64_t cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = t0 + u * stride;
      row[u] = -1;
      cc[u] = 0;
      if (t < total) {
        cc[u] = t % chunks;
        const int64_t ec = t / chunks;
        const int k = static_cast<int>(ec % ncols);
        const int64_t e = ec / ncols;
        const int sel = k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3));
        row[u] = cols[static_cast<int64_t>(sel) * M + e];
      }
    }
    VecT v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = VecT{};
      if (row[u] >= 0 && row[u] < N) v[u] = x[row[u] * chunks + cc[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = t0 + u * stride;
      if (t < total) out[t] = v[u];
    }
  }
}

__device__ __forceinline__ int64_t owner_of(const int64_t* __restrict__ splits, int64_t G, int64_t e) {
  int64_t lo = 0, hi = G;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (splits[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename VecT>
__global__ void gather_i64_kernel(const VecT* __restrict__ x, int64_t N, int64_t chunks,
                                  const int64_t* __restrict__ idx, int64_t M, int K, int col,
                                  const int64_t* __restrict__ node_splits, const int64_t* __restrict__ edge_splits,
                                  int64_t G, VecT* __restrict__ out) {
  const int ncols = col < 0 ? K : 1;
  const int64_t total = M * ncols * chunks;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % chunks;
    const int64_t ec = t / chunks;
    const int k = static_cast<int>(ec % ncols);
    const int64_t e = ec / ncols;
    int64_t row = idx[e * K + (col < 0 ? k : col)];
    if (node_splits) row += node_splits[owner_of(edge_splits, G, e)];
    VecT v{};
    if (row >= 0 && row < N) v = x[row * chunks + c];  // TF-GPU semantics: out-of-range rows read as zeros
    out[t] = v;
  }
}

template <typename VecT>
__global__ void repeat_rows_kernel(const VecT* __restrict__ state, const int64_t* __restrict__ splits, int64_t G,
                                   int64_t chunks, int64_t N, VecT* __restrict__ out) {
  const int64_t total = N * chunks;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % chunks;
    const int64_t n = t / chunks;
    out[t] = state[owner_of(splits, G, n) * chunks + c];
  }
}

template <typename VecT>
__global__ void embedding_kernel(const VecT* __restrict__ table, int64_t vocab, int64_t chunks,
                                 const float* __restrict__ numbers, int64_t N, VecT* __restrict__ out,
                                 int32_t* __restrict__ flags) {
  const int64_t total = N * chunks;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % chunks;
    const int64_t n = t / chunks;
    const int64_t row = static_cast<int32_t>(numbers[n]);  // Keras casts non-integer input to int32 (truncation)
    VecT v{};
    if (row >= 0 && row < vocab) v = table[row * chunks + c];
    else if (flags && c == 0) atomicOr(flags, MP_FLAG_OOB);
    out[t] = v;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int mp_gather_rows_f32(const float* x, int64_t N, int64_t row_elems, const int32_t* cols, int64_t M, int ncols,
                       const int32_t* colsel_host, float* out, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && row_elems >= 1, "mp_gather_rows_f32: bad sizes");
  MP_REQUIRE(ncols >= 1 && ncols <= 4 && colsel_host, "mp_gather_rows_f32: ncols must be 1..4 with a selection list");
  if (M == 0) return MP_OK;
  MP_REQUIRE(x && cols && out, "mp_gather_rows_f32: null pointer");
  int sel[4] = {0, 0, 0, 0};
  for (int i = 0; i < ncols; ++i) {
    MP_REQUIRE(colsel_host[i] >= 0, "mp_gather_rows_f32: negative column");
    sel[i] = colsel_host[i];
  }
  hipStream_t s = mp::as_stream(stream);
  if (row_elems % 4 == 0 && aligned16(x) && aligned16(out)) {
    const int64_t chunks = row_elems / 4;
    gather_cols_kernel<float4><<<mp::grid_for(M * ncols * chunks), 256, 0, s>>>(
        reinterpret_cast<const float4*>(x), N, chunks, cols, M, ncols, sel[0], sel[1], sel[2], sel[3],
        reinterpret_cast<float4*>(out));
  } else {
    gather_cols_kernel<float><<<mp::grid_for(M * ncols * row_elems), 256, 0, s>>>(x, N, row_elems, cols, M, ncols,
                                                                                 sel[0], sel[1], sel[2], sel[3], out);
  }
  return mp::check_launch("mp_gather_rows_f32");
}

int mp_gather_rows_i64_f32(const float* x, int64_t N, int64_t row_elems, const int64_t* idx, int64_t M, int K, int col,
                           const int64_t* node_splits, const int64_t* edge_splits, int64_t G, float* out,
                           mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && row_elems >= 1 && K >= 1, "mp_gather_rows_i64_f32: bad sizes");
  MP_REQUIRE(col < K, "mp_gather_rows_i64_f32: column %d out of range for K=%d", col, K);
  MP_REQUIRE((node_splits == nullptr) == (edge_splits == nullptr), "mp_gather_rows_i64_f32: pass both splits or none");
  if (M == 0) return MP_OK;
  MP_REQUIRE(x && idx && out, "mp_gather_rows_i64_f32: null pointer");
  MP_REQUIRE(node_splits == nullptr || G > 0, "mp_gather_rows_i64_f32: no graphs");
  const int ncols = col < 0 ? K : 1;
  hipStream_t s = mp::as_stream(stream);
  if (row_elems % 4 == 0 && aligned16(x) && aligned16(out)) {
    const int64_t chunks = row_elems / 4;
    gather_i64_kernel<float4><<<mp::grid_for(M * ncols * chunks), 256, 0, s>>>(
        reinterpret_cast<const float4*>(x), N, chunks, idx, M, K, col, node_splits, edge_splits, G,
        reinterpret_cast<float4*>(out));
  } else {
    gather_i64_kernel<float><<<mp::grid_for(M * ncols * row_elems), 256, 0, s>>>(x, N, row_elems, idx, M, K, col,
                                                                                node_splits, edge_splits, G, out);
  }
  return mp::check_launch("mp_gather_rows_i64_f32");
}

int mp_repeat_rows_f32(const float* state, const int64_t* splits, int64_t G, int64_t row_elems, int64_t N, float* out,
                       mpStream_t stream) {
  MP_REQUIRE(G >= 0 && N >= 0 && row_elems >= 1, "mp_repeat_rows_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(state && splits && out && G > 0, "mp_repeat_rows_f32: null pointer / no graphs");
  hipStream_t s = mp::as_stream(stream);
  if (row_elems % 4 == 0 && aligned16(state) && aligned16(out)) {
    const int64_t chunks = row_elems / 4;
    repeat_rows_kernel<float4><<<mp::grid_for(N * chunks), 256, 0, s>>>(reinterpret_cast<const float4*>(state), splits,
                                                                       G, chunks, N, reinterpret_cast<float4*>(out));
  } else {
    repeat_rows_kernel<float><<<mp::grid_for(N * row_elems), 256, 0, s>>>(state, splits, G, row_elems, N, out);
  }
  return mp::check_launch("mp_repeat_rows_f32");
}

int mp_embedding_f32(const float* table, int64_t vocab, int64_t dim, const float* numbers, int64_t N, float* out,
                     int32_t* flags, mpStream_t stream) {
  MP_REQUIRE(vocab >= 1 && dim >= 1 && N >= 0, "mp_embedding_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(table && numbers && out, "mp_embedding_f32: null pointer");
  hipStream_t s = mp::as_stream(stream);
  if (dim % 4 == 0 && aligned16(table) && aligned16(out)) {
    const int64_t chunks = dim / 4;
    embedding_kernel<float4><<<mp::grid_for(N * chunks), 256, 0, s>>>(reinterpret_cast<const float4*>(table), vocab,
                                                                     chunks, numbers, N,
                                                                     reinterpret_cast<float4*>(out), flags);
  } else {
    embedding_kernel<float><<<mp::grid_for(N * dim), 256, 0, s>>>(table, vocab, dim, numbers, N, out, flags);
  }
  return mp::check_launch("mp_embedding_f32");
}

}  // extern "C"
