// On-GPU SetRange (kgcnn/graph/preprocessor.py:288-314 -> define_adjacency_from_distance, kgcnn/graph/adj.py:537-593,
// exclusive mode, no self loops): connect i -> j when dist(i,j) < max_distance AND j is among the max_neighbours + 1
// nearest entries of row i (the +1 is i itself at distance 0).  The reference builds an n x n distance matrix per
// molecule in NumPy on the host; here a whole ragged batch is processed on the device in two passes (count, fill)
// around one prefix sum, and the fill pass emits - besides the API's (M,2) int64 sample indices, row-major (i, j)
// order = receiver-sorted - the int32 receiver / sender ids, the edge distance (range_attributes) and the receiver CSR,
// so the message-passing kernels can start without any further index preparation.
//
// One thread owns one receiving atom i and scans the atoms j of its molecule (coordinates of the molecule are read
// from L1/L2: 29 atoms = 348 B).  Rank test: j qualifies if fewer than max_neighbours + 1 atoms k (k != j) are
// strictly closer to i than j is, or equally close with a smaller index (the order a stable argsort would give;
// exact float ties do not occur for generic coordinates).  Distances are sqrt(dx^2 + dy^2 + dz^2) in float32, the same
// arithmetic as kgcnn/graph/adj.py:481-482.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mp_common.h"

namespace {

__device__ __forceinline__ int64_t owner_of(const int64_t* __restrict__ splits, int64_t G, int64_t e) {
  int64_t lo = 0, hi = G;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (splits[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float dist_f32(const float* __restrict__ xyz, int64_t a, int64_t b) {
  const float dx = xyz[a * 3 + 0] - xyz[b * 3 + 0];
  const float dy = xyz[a * 3 + 1] - xyz[b * 3 + 1];
  const float dz = xyz[a * 3 + 2] - xyz[b * 3 + 2];
  return sqrtf(dx * dx + dy * dy + dz * dz);
}

__device__ __forceinline__ bool qualifies(const float* __restrict__ xyz, int64_t base, int64_t n, int64_t i, int64_t j,
                                          float max_distance, int max_neighbours, bool use_distance, float* d_out) {
  if (i == j) return false;
  const float d = dist_f32(xyz, base + i, base + j);
  *d_out = d;
  if (use_distance && !(d < max_distance)) return false;
  if (max_neighbours >= 0 && max_neighbours + 1 < n) {
    int closer = 0;  // entries of row i sorted before j (includes i itself at distance 0)
    for (int64_t k = 0; k < n; ++k) {
      if (k == j) continue;
      const float dk = dist_f32(xyz, base + i, base + k);
      if (dk < d || (dk == d && k < j)) ++closer;
    }
    if (closer > max_neighbours) return false;
  }
  return true;
}

template <bool FILL>
__global__ void radius_graph_kernel(const float* __restrict__ xyz, const int64_t* __restrict__ node_splits, int64_t G,
                                    int64_t N, float max_distance, int max_neighbours, int use_distance,
                                    int32_t* __restrict__ counts, const int32_t* __restrict__ node_ptr,
                                    int64_t* __restrict__ idx_out, int32_t* __restrict__ recv, int32_t* __restrict__ send,
                                    float* __restrict__ dist) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t a = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; a < N; a += stride) {
    const int64_t g = owner_of(node_splits, G, a);
    const int64_t base = node_splits[g];
    const int64_t n = node_splits[g + 1] - base;
    const int64_t i = a - base;
    int32_t cnt = 0;
    int64_t pos = FILL ? static_cast<int64_t>(node_ptr[a]) : 0;
    for (int64_t j = 0; j < n; ++j) {
      float d;
      if (!qualifies(xyz, base, n, i, j, max_distance, max_neighbours, use_distance != 0, &d)) continue;
      if constexpr (FILL) {
        idx_out[pos * 2] = i;
        idx_out[pos * 2 + 1] = j;
        if (recv) recv[pos] = static_cast<int32_t>(a);
        if (send) send[pos] = static_cast<int32_t>(base + j);
        if (dist) dist[pos] = d;
        ++pos;
      } else {
        ++cnt;
      }
    }
    if constexpr (!FILL) counts[a] = cnt;
  }
}

__global__ void edge_splits_kernel(const int32_t* __restrict__ node_ptr, const int64_t* __restrict__ node_splits,
                                   int64_t G, int64_t* __restrict__ edge_splits) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g <= G; g += stride)
    edge_splits[g] = node_ptr[node_splits[g]];
}

inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

}  // namespace

extern "C" {

int mp_radius_graph_workspace_bytes(int64_t N, size_t* bytes_out_host) {
  MP_REQUIRE(bytes_out_host && N >= 0, "mp_radius_graph_workspace_bytes: bad arguments");
  size_t temp = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, temp, static_cast<const int32_t*>(nullptr),
                                         static_cast<int32_t*>(nullptr), int32_t{0}, static_cast<size_t>(N + 1),
                                         rocprim::plus<int32_t>(), hipStream_t{nullptr}, false);
  if (e != hipSuccess) {
    mp::set_error("rocprim temp-size query failed: %s", hipGetErrorString(e));
    return MP_EHIP;
  }
  *bytes_out_host = align256(temp) + align256(sizeof(int32_t) * static_cast<size_t>(N + 1));
  return MP_OK;
}

int mp_radius_graph_count_f32(const float* xyz, const int64_t* node_splits, int64_t G, int64_t N, float max_distance,
                              int max_neighbours, int32_t* node_ptr /* (N+1) */, int64_t* edge_splits /* (G+1) */,
                              void* ws, size_t ws_bytes, mpStream_t stream) {
  MP_REQUIRE(G >= 0 && N >= 0 && node_ptr && edge_splits && ws, "mp_radius_graph_count_f32: bad arguments");
  MP_REQUIRE(N < (int64_t{1} << 31), "mp_radius_graph_count_f32: N must fit int32");
  size_t need = 0;
  int rc = mp_radius_graph_workspace_bytes(N, &need);
  if (rc != MP_OK) return rc;
  MP_REQUIRE(ws_bytes >= need, "mp_radius_graph_count_f32: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = mp::as_stream(stream);
  int32_t* counts = static_cast<int32_t*>(ws);
  void* temp = static_cast<char*>(ws) + align256(sizeof(int32_t) * static_cast<size_t>(N + 1));
  size_t temp_bytes = ws_bytes - align256(sizeof(int32_t) * static_cast<size_t>(N + 1));
  MP_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * static_cast<size_t>(N + 1), s));
  if (N > 0) {
    MP_REQUIRE(xyz && node_splits && G > 0, "mp_radius_graph_count_f32: null pointer / no graphs");
    const int use_distance = max_distance >= 0.0f ? 1 : 0;
    radius_graph_kernel<false><<<mp::grid_for(N, 64), 64, 0, s>>>(xyz, node_splits, G, N, max_distance, max_neighbours,
                                                                  use_distance, counts, nullptr, nullptr, nullptr,
                                                                  nullptr, nullptr);
  }
  MP_HIP(rocprim::exclusive_scan(temp, temp_bytes, counts, node_ptr, int32_t{0}, static_cast<size_t>(N + 1),
                                 rocprim::plus<int32_t>(), s, false));
  if (G > 0) edge_splits_kernel<<<mp::grid_for(G + 1), 256, 0, s>>>(node_ptr, node_splits, G, edge_splits);
  else MP_HIP(hipMemsetAsync(edge_splits, 0, sizeof(int64_t), s));
  return mp::check_launch("mp_radius_graph_count_f32");
}

int mp_radius_graph_fill_f32(const float* xyz, const int64_t* node_splits, int64_t G, int64_t N, float max_distance,
                             int max_neighbours, const int32_t* node_ptr, int64_t M, int64_t* idx_out, int32_t* recv,
                             int32_t* send, float* dist, mpStream_t stream) {
  MP_REQUIRE(G >= 0 && N >= 0 && M >= 0, "mp_radius_graph_fill_f32: bad sizes");
  if (N == 0 || M == 0) return MP_OK;
  MP_REQUIRE(xyz && node_splits && node_ptr && idx_out && G > 0, "mp_radius_graph_fill_f32: null pointer");
  const int use_distance = max_distance >= 0.0f ? 1 : 0;
  radius_graph_kernel<true><<<mp::grid_for(N, 64), 64, 0, mp::as_stream(stream)>>>(
      xyz, node_splits, G, N, max_distance, max_neighbours, use_distance, nullptr, node_ptr, idx_out, recv, send, dist);
  return mp::check_launch("mp_radius_graph_fill_f32");
}

}  // extern "C"
