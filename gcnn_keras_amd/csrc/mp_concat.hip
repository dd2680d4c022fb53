// Disjoint union of several resident ragged batches in ONE launch: the device-side analogue of concatenating the lists two
// `MemoryGraphList.tensor()` calls would have produced (kgcnn/data/base.py:203-239; np.concatenate + row_lengths,
// kgcnn/data/utils.py:129-157).  Per-graph "sample" edge indices need no rewriting (kgcnn/layers/base.py:27: the shift into
// the batch happens from the row splits, kgcnn/ops/partition.py:140-155); only the row splits are rebased.
//
// Why: a 128-graph SchNet forward leaves most of the chip idle (819 edge tiles on 1024 SIMDs, 144 node tiles on 256 CUs, a
// kernel boundary of ~4 us per launch); k independent batches served by one launch sequence pay the eight boundaries and
// the weight staging once.  Measured (bench.py): four 128-graph batches per launch group, two or three groups in flight:
// 860-900 M edges/s against 757 M for four separate forwards in flight.
#include "mp_common.h"

namespace {

struct ConcatArgs {
  mp_concat_desc d;
  int64_t n_off[MP_CONCAT_MAX + 1], m_off[MP_CONCAT_MAX + 1], g_off[MP_CONCAT_MAX + 1];
};

typedef long long i64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void concat_batches_kernel(ConcatArgs a) {
  const int k = a.d.k;
  const int64_t M = a.m_off[k], N = a.n_off[k], G = a.g_off[k];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < M || t < N || t <= G; t += stride) {
    if (t < M) {           // one (i, j) pair = 16 B
      int b = 0;
#pragma unroll
      for (int q = 1; q < MP_CONCAT_MAX; ++q) b += (q < k && t >= a.m_off[q]) ? 1 : 0;
      const i64x2 v = reinterpret_cast<const i64x2*>(a.d.src[b].idx)[t - a.m_off[b]];
      reinterpret_cast<i64x2*>(a.d.idx)[t] = v;
    }
    if (t < N) {
      int b = 0;
#pragma unroll
      for (int q = 1; q < MP_CONCAT_MAX; ++q) b += (q < k && t >= a.n_off[q]) ? 1 : 0;
      const int64_t l = t - a.n_off[b];
      const float* x = a.d.src[b].xyz + l * 3;
      a.d.xyz[t * 3 + 0] = x[0];
      a.d.xyz[t * 3 + 1] = x[1];
      a.d.xyz[t * 3 + 2] = x[2];
      if (a.d.z_is_i64) static_cast<int64_t*>(a.d.z)[t] = static_cast<const int64_t*>(a.d.src[b].z)[l];
      else static_cast<float*>(a.d.z)[t] = static_cast<const float*>(a.d.src[b].z)[l];
    }
    if (t <= G) {
      int b = 0;
#pragma unroll
      for (int q = 1; q < MP_CONCAT_MAX; ++q) b += (q < k && t >= a.g_off[q]) ? 1 : 0;
      // graph t of the union = graph t - g_off[b] of batch b; the closing entry (t == G) is the last batch's last split
      const int64_t l = t - a.g_off[b];
      a.d.node_splits[t] = a.d.src[b].node_splits[l] + a.n_off[b];
      a.d.edge_splits[t] = a.d.src[b].edge_splits[l] + a.m_off[b];
    }
  }
}

}  // namespace

extern "C" int mp_concat_batches(const mp_concat_desc* d, mpStream_t stream) {
  MP_REQUIRE(d != nullptr && d->k >= 1 && d->k <= MP_CONCAT_MAX, "mp_concat_batches: 1..%d batches", MP_CONCAT_MAX);
  ConcatArgs a{};
  a.d = *d;
  for (int b = 0; b < d->k; ++b) {
    const mp_batch_src& s = d->src[b];
    MP_REQUIRE(s.N >= 0 && s.M >= 0 && s.G >= 0, "mp_concat_batches: negative size in batch %d", b);
    MP_REQUIRE(s.node_splits && s.edge_splits && (s.N == 0 || (s.z && s.xyz)) && (s.M == 0 || s.idx),
               "mp_concat_batches: null pointer in batch %d", b);
    a.n_off[b + 1] = a.n_off[b] + s.N;
    a.m_off[b + 1] = a.m_off[b] + s.M;
    a.g_off[b + 1] = a.g_off[b] + s.G;
  }
  const int64_t N = a.n_off[d->k], M = a.m_off[d->k], G = a.g_off[d->k];
  MP_REQUIRE(d->node_splits && d->edge_splits && (N == 0 || (d->z && d->xyz)) && (M == 0 || d->idx),
             "mp_concat_batches: null output pointer");
  int64_t work = M > N ? M : N;
  if (G + 1 > work) work = G + 1;
  concat_batches_kernel<<<mp::grid_for(work), 256, 0, mp::as_stream(stream)>>>(a);
  return mp::check_launch("mp_concat_batches");
}
