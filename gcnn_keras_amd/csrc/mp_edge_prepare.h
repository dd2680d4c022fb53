// Edge preparation body shared by mp_index.hip (stand-alone kernel) and mp_schnet_node.hip (stage-0 kernel that runs it
// beside the node-input chain): shift + receiver/sender split + sortedness/range flags + edge distance in one pass
// (kgcnn/ops/partition.py:140-155, kgcnn/layers/gather.py:228, kgcnn/literature/Schnet.py:116-117).
#pragma once
#include "mp_common.h"

namespace mp_prep {

// largest g in [0, G) with splits[g] <= e
__device__ __forceinline__ int64_t owner_of(const int64_t* __restrict__ splits, int64_t G, int64_t e) {
  int64_t lo = 0, hi = G;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (splits[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// owner search on a staged (LDS) copy of the splits
__device__ __forceinline__ int owner_of_lds(const int64_t* splits, int G, int64_t e) {
  int lo = 0, hi = G;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (splits[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

constexpr int PREP_LDS_GRAPHS = 1023;  // batches up to this many graphs search their row_splits in LDS

struct EdgePrepArgs {
  const int64_t* idx;
  int64_t M;
  const int64_t* node_splits;
  const int64_t* edge_splits;
  int64_t G, N;
  const float* xyz;
  int32_t* recv;
  int32_t* send;
  float* dist;
  int32_t* flags;
};

// LDS_SPLITS: every workgroup first copies both row_splits arrays (G+1 <= 1024 entries each) into LDS with one
// coalesced round trip; the per-edge owner search then never leaves the CU.  At QM9 batch sizes this is a chain of
// dependent round trips (search steps -> offsets -> index row -> coordinates), so removing the ~7 search trips is what
// matters.  Larger batches search the L2-resident arrays directly (one wave-uniform search + a short walk).
// `block` / `nblocks`: this workgroup's position among the workgroups doing edge preparation.
// COL1: also track the sortedness of the sender column (the generic index plan reports both columns).
// Extra: per-edge continuation extra(e, dx, dy, dz, dist) for callers that derive more from the edge vector in the same
// pass (PaiNN: direction, Bessel basis and its derivative, cosine envelope).
struct NoEdgeExtra {
  static constexpr bool active = false;
  __device__ __forceinline__ void operator()(int64_t, float, float, float, float) const {}
};

template <bool LDS_SPLITS, bool COL1 = false, class Extra = NoEdgeExtra>
__device__ __forceinline__ void edge_prepare_body(const EdgePrepArgs& p, int64_t block, int64_t nblocks,
                                                  const Extra& extra = Extra()) {
  const int64_t* __restrict__ idx = p.idx;
  const int64_t* __restrict__ node_splits = p.node_splits;
  const int64_t* __restrict__ edge_splits = p.edge_splits;
  const float* __restrict__ xyz = p.xyz;
  const int64_t M = p.M, G = p.G, N = p.N;
  __shared__ int64_t s_es[LDS_SPLITS ? PREP_LDS_GRAPHS + 1 : 1];
  __shared__ int64_t s_ns[LDS_SPLITS ? PREP_LDS_GRAPHS + 1 : 1];
  // The index row of an edge (and the row before it, for the sortedness test) does not depend on the owner search: it
  // is requested first - together with the row_splits staging, one round trip instead of three in a row (staging ->
  // index row -> previous row) - and the next iteration's rows are requested while this one is processed.  Rows past
  // the end re-read the last row (unused).
  const longlong2* __restrict__ idx2 = reinterpret_cast<const longlong2*>(idx);
  const int64_t e_first = block * blockDim.x + threadIdx.x;
  longlong2 v_pre = {0, 0}, pv_pre = {0, 0};
  auto fetch_rows = [&](int64_t e) {
    const int64_t ec = e < M ? e : M - 1;
    v_pre = idx2[ec];
    pv_pre = idx2[ec > 0 ? ec - 1 : 0];
  };
  if (M > 0) fetch_rows(e_first);
  if constexpr (LDS_SPLITS) {
    for (int i = threadIdx.x; i <= G; i += blockDim.x) {
      s_es[i] = edge_splits[i];
      s_ns[i] = node_splits[i];
    }
    __syncthreads();
  }
  const int64_t stride = nblocks * blockDim.x;
  int local_flags = 0;
  for (int64_t e = e_first; e - (threadIdx.x & 63) < M; e += stride) {
    const longlong2 v = v_pre, pv = pv_pre;
    fetch_rows(e + stride);
    int64_t g, base, n_g, g_start;
    if constexpr (LDS_SPLITS) {
      if (e >= M) continue;
      g = owner_of_lds(s_es, static_cast<int>(G), e);
      base = s_ns[g];
      n_g = s_ns[g + 1] - base;
      g_start = s_es[g];
    } else {
      const int64_t e_wave = __builtin_amdgcn_readfirstlane(static_cast<int>(e - (threadIdx.x & 63)));
      g = owner_of(edge_splits, G, e_wave < M ? e_wave : M - 1);
      if (e >= M) continue;
      while (g + 1 < G && edge_splits[g + 1] <= e) ++g;
      base = node_splits[g];
      n_g = node_splits[g + 1] - base;
      g_start = edge_splits[g];
    }
    int64_t i = v.x, j = v.y;
    if (i < 0 || i >= n_g || j < 0 || j >= n_g) {
      local_flags |= MP_FLAG_OOB;
      const int64_t hi = n_g > 0 ? n_g - 1 : 0;
      i = i < 0 ? 0 : (i > hi ? hi : i);
      j = j < 0 ? 0 : (j > hi ? hi : j);
    }
    int64_t si = i + base, sj = j + base;
    if (si >= N) si = N > 0 ? N - 1 : 0;
    if (sj >= N) sj = N > 0 ? N - 1 : 0;
    p.recv[e] = static_cast<int32_t>(si);
    p.send[e] = static_cast<int32_t>(sj);
    // receiver of the previous edge: only the same graph can break the order (an earlier graph's ids are smaller
    // because node offsets grow with the graph index)
    if (g_start < e) {
      if (pv.x + base > si) local_flags |= MP_FLAG_UNSORTED_COL0;
      if constexpr (COL1) {
        if (pv.y + base > sj) local_flags |= MP_FLAG_UNSORTED_COL1;
      }
    }
    if (p.dist || Extra::active) {
      const float dx = xyz[si * 3 + 0] - xyz[sj * 3 + 0];
      const float dy = xyz[si * 3 + 1] - xyz[sj * 3 + 1];
      const float dz = xyz[si * 3 + 2] - xyz[sj * 3 + 2];
      const float dist = sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 0.0f));
      if (p.dist) p.dist[e] = dist;
      if constexpr (Extra::active) extra(e, dx, dy, dz, dist);
    }
  }
  mp_publish_flags(p.flags, local_flags);
}

}  // namespace mp_prep
