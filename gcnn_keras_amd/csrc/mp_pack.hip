// Host-side ragged batch packer: the native counterpart of
//   kgcnn.data.utils.ragged_tensor_from_nested_numpy (kgcnn/data/utils.py:129-157: np.concatenate + row lengths) and
//   MemoryGraphList.tensor (kgcnn/data/base.py:203-239),
// writing straight into (pinned) staging memory from which one asynchronous copy per tensor feeds the engine.  While it
// concatenates the per-graph edge-index lists it also emits what the device would otherwise recompute per batch
// (mp_index_prepare_i64 + mp_csr_from_sorted_i32): the shifted int32 index columns, the MP_FLAG_* word and - when the
// receivers are sorted - the CSR offsets.
//
// No kernels in this file: plain C++ threads over contiguous graph ranges (the work is memcpy-bound; a batch of config
// 2 is 0.5 MB).  Compiled with the rest of the library so that it shares the error / status conventions.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mp_common.h"

namespace {

template <typename Fn>
void parallel_ranges(int64_t G, int threads, const int64_t* weight_prefix, Fn fn) {
  // contiguous graph ranges of (roughly) equal payload, by the prefix sums already at hand
  const int64_t total = weight_prefix[G];
  int t = threads < 1 ? 1 : threads;
  if (total < (int64_t{1} << 16) || G < 2) t = 1;  // small batches: thread start-up costs more than the copy
  if (t > G) t = static_cast<int>(G);
  if (t <= 1) {
    fn(0, G);
    return;
  }
  std::vector<std::thread> pool;
  int64_t g0 = 0;
  for (int i = 0; i < t; ++i) {
    const int64_t target = total * (i + 1) / t;
    int64_t g1 = i == t - 1 ? G : std::upper_bound(weight_prefix + g0, weight_prefix + G + 1, target) - weight_prefix;
    if (g1 > G) g1 = G;
    if (g1 < g0) g1 = g0;
    if (g1 > g0) pool.emplace_back(fn, g0, g1);
    g0 = g1;
  }
  for (auto& th : pool) th.join();
}

template <typename S, typename D>
void copy_cast(const void* src, void* dst, int64_t n) {
  const S* s = static_cast<const S*>(src);
  D* d = static_cast<D*>(dst);
  for (int64_t i = 0; i < n; ++i) d[i] = static_cast<D>(s[i]);
}

size_t kind_size(int kind) {
  switch (kind) {
    case MP_DT_F32: return 4;
    case MP_DT_F64: return 8;
    case MP_DT_I32: return 4;
    case MP_DT_I64: return 8;
    default: return 0;
  }
}

using cast_fn = void (*)(const void*, void*, int64_t);
cast_fn pick_cast(int src, int dst) {
  if (src == MP_DT_F64 && dst == MP_DT_F32) return copy_cast<double, float>;
  if (src == MP_DT_F32 && dst == MP_DT_F64) return copy_cast<float, double>;
  if (src == MP_DT_I32 && dst == MP_DT_I64) return copy_cast<int32_t, int64_t>;
  if (src == MP_DT_I64 && dst == MP_DT_I32) return copy_cast<int64_t, int32_t>;
  if (src == MP_DT_I64 && dst == MP_DT_F32) return copy_cast<int64_t, float>;
  if (src == MP_DT_I32 && dst == MP_DT_F32) return copy_cast<int32_t, float>;
  return nullptr;
}

}  // namespace

extern "C" {

int mp_host_alloc(size_t bytes, int pinned, void** out_host) {
  MP_REQUIRE(out_host != nullptr, "mp_host_alloc: null output");
  *out_host = nullptr;
  if (bytes == 0) bytes = 64;
  if (pinned) {
    void* p = nullptr;
    MP_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    *out_host = p;
    return MP_OK;
  }
  void* p = nullptr;
  if (posix_memalign(&p, 256, bytes) != 0 || p == nullptr) {
    mp::set_error("mp_host_alloc: out of host memory (%zu bytes)", bytes);
    return MP_EINVAL;
  }
  *out_host = p;
  return MP_OK;
}

int mp_host_free(void* p, int pinned) {
  if (p == nullptr) return MP_OK;
  if (pinned) {
    MP_HIP(hipHostFree(p));
  } else {
    free(p);
  }
  return MP_OK;
}

int mp_memcpy_h2d_async(void* dst_device, const void* src_host, size_t bytes, mpStream_t stream) {
  if (bytes == 0) return MP_OK;
  MP_REQUIRE(dst_device && src_host, "mp_memcpy_h2d_async: null pointer");
  MP_HIP(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, mp::as_stream(stream)));
  return MP_OK;
}

int mp_pack_rows_host(const void* const* rows_host, const int64_t* counts_host, int64_t G, int64_t row_elems,
                      int src_kind, int dst_kind, void* dst_host, int64_t* splits_out_host, int threads) {
  MP_REQUIRE(G >= 0 && row_elems >= 0, "mp_pack_rows_host: bad sizes");
  MP_REQUIRE(splits_out_host != nullptr, "mp_pack_rows_host: null splits");
  const size_t ssz = kind_size(src_kind), dsz = kind_size(dst_kind);
  MP_REQUIRE(ssz && dsz, "mp_pack_rows_host: unknown dtype kind");
  cast_fn cast = nullptr;
  if (src_kind != dst_kind) {
    cast = pick_cast(src_kind, dst_kind);
    if (!cast) {
      mp::set_error("mp_pack_rows_host: unsupported conversion %d -> %d", src_kind, dst_kind);
      return MP_ENOTSUP;
    }
  }
  splits_out_host[0] = 0;
  for (int64_t g = 0; g < G; ++g) {
    MP_REQUIRE(counts_host && counts_host[g] >= 0, "mp_pack_rows_host: negative row count in graph %lld",
               static_cast<long long>(g));
    splits_out_host[g + 1] = splits_out_host[g] + counts_host[g];
  }
  if (G == 0 || splits_out_host[G] == 0 || row_elems == 0) return MP_OK;
  MP_REQUIRE(rows_host && dst_host, "mp_pack_rows_host: null pointer");
  for (int64_t g = 0; g < G; ++g)
    MP_REQUIRE(counts_host[g] == 0 || rows_host[g] != nullptr, "mp_pack_rows_host: null rows for graph %lld",
               static_cast<long long>(g));
  char* dst = static_cast<char*>(dst_host);
  parallel_ranges(G, threads, splits_out_host, [&](int64_t g0, int64_t g1) {
    for (int64_t g = g0; g < g1; ++g) {
      const int64_t n = counts_host[g] * row_elems;
      if (n == 0) continue;
      char* d = dst + static_cast<size_t>(splits_out_host[g]) * row_elems * dsz;
      if (cast) cast(rows_host[g], d, n);
      else memcpy(d, rows_host[g], static_cast<size_t>(n) * dsz);
    }
  });
  return MP_OK;
}

int mp_pack_edge_index_host(const void* const* idx_rows_host, int idx_kind, const int64_t* edge_counts_host,
                            const int64_t* node_counts_host, int64_t G, int K, int64_t* idx_out_host,
                            int64_t* edge_splits_out_host, int64_t* node_splits_out_host, int32_t* cols_out_host,
                            int32_t* csr_ptr_out_host, int32_t* flags_out_host, int threads) {
  MP_REQUIRE(G >= 0 && K >= 1, "mp_pack_edge_index_host: bad sizes");
  MP_REQUIRE(idx_kind == MP_DT_I64 || idx_kind == MP_DT_I32, "mp_pack_edge_index_host: indices must be int32/int64");
  MP_REQUIRE(edge_splits_out_host && node_splits_out_host && flags_out_host, "mp_pack_edge_index_host: null output");
  edge_splits_out_host[0] = 0;
  node_splits_out_host[0] = 0;
  for (int64_t g = 0; g < G; ++g) {
    MP_REQUIRE(edge_counts_host && node_counts_host && edge_counts_host[g] >= 0 && node_counts_host[g] >= 0,
               "mp_pack_edge_index_host: negative count in graph %lld", static_cast<long long>(g));
    edge_splits_out_host[g + 1] = edge_splits_out_host[g] + edge_counts_host[g];
    node_splits_out_host[g + 1] = node_splits_out_host[g] + node_counts_host[g];
  }
  const int64_t M = edge_splits_out_host[G], N = node_splits_out_host[G];
  MP_REQUIRE(N < (int64_t{1} << 31) && M < (int64_t{1} << 31), "mp_pack_edge_index_host: N, M must fit int32");
  *flags_out_host = 0;
  if (csr_ptr_out_host) {
    for (int64_t n = 0; n <= N; ++n) csr_ptr_out_host[n] = 0;
  }
  if (M == 0) return MP_OK;
  MP_REQUIRE(idx_rows_host && idx_out_host && cols_out_host, "mp_pack_edge_index_host: null pointer");
  std::atomic<int> flags{0};
  parallel_ranges(G, threads, edge_splits_out_host, [&](int64_t g0, int64_t g1) {
    int local = 0;
    for (int64_t g = g0; g < g1; ++g) {
      const int64_t m = edge_counts_host[g], e0 = edge_splits_out_host[g];
      const int64_t base = node_splits_out_host[g], n_g = node_counts_host[g];
      if (m == 0) continue;
      for (int64_t e = 0; e < m; ++e) {
        for (int k = 0; k < K; ++k) {
          const int64_t raw = idx_kind == MP_DT_I64 ? static_cast<const int64_t*>(idx_rows_host[g])[e * K + k]
                                                     : static_cast<const int32_t*>(idx_rows_host[g])[e * K + k];
          idx_out_host[(e0 + e) * K + k] = raw;  // the API's sample indices, untouched
          int64_t v = raw;
          if (v < 0 || v >= n_g) {  // same clamp + flag as mp_index_prepare_i64
            local |= MP_FLAG_OOB;
            v = v < 0 ? 0 : (n_g > 0 ? n_g - 1 : 0);
          }
          int64_t s = v + base;
          if (s >= N) s = N > 0 ? N - 1 : 0;
          cols_out_host[static_cast<int64_t>(k) * M + e0 + e] = static_cast<int32_t>(s);
        }
      }
    }
    if (local) flags.fetch_or(local);
  });
  // sortedness of the batch-level ids is judged on the UNCLAMPED shifted values, as on the device
  int f = flags.load();
  {
    int64_t g = 0, gp = 0;
    for (int64_t e = 1; e < M && (f & (MP_FLAG_UNSORTED_COL0 | MP_FLAG_UNSORTED_COL1)) !=
                                     (MP_FLAG_UNSORTED_COL0 | MP_FLAG_UNSORTED_COL1); ++e) {
      while (edge_splits_out_host[g + 1] <= e) ++g;
      while (edge_splits_out_host[gp + 1] <= e - 1) ++gp;
      for (int k = 0; k < K && k < 2; ++k) {
        const int64_t cur = idx_out_host[e * K + k] + node_splits_out_host[g];
        const int64_t prev = idx_out_host[(e - 1) * K + k] + node_splits_out_host[gp];
        if (prev > cur) f |= (k == 0 ? MP_FLAG_UNSORTED_COL0 : MP_FLAG_UNSORTED_COL1);
      }
    }
  }
  *flags_out_host = f;
  if (csr_ptr_out_host && !(f & MP_FLAG_UNSORTED_COL0)) {
    // ptr[n] = first edge position with receiver >= n (what mp_csr_from_sorted_i32 produces)
    const int32_t* recv = cols_out_host;
    int64_t e = 0;
    for (int64_t n = 0; n <= N; ++n) {
      while (e < M && recv[e] < n) ++e;
      csr_ptr_out_host[n] = static_cast<int32_t>(e);
    }
  }
  return MP_OK;
}

}  // extern "C"
