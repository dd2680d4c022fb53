// Reverse pass of the SchNet node-side chains and the whole energy + force pass behind one C-ABI call
// (kgcnn/model/force.py:159-201 - tape.gradient(energy, coordinates) - around kgcnn/literature/Schnet.py:104-148).
//
// The forward chains of csrc/mp_schnet_node.hip keep d act / d pre-activation of every shifted softplus (SAVE builds);
// with those the reverse of a chain is again a chain of GEMMs on 16-node tiles, against the TRANSPOSED kernels:
//
//   head  : g_pl1 = g_h * dl1 ; g_pl0 = (g_pl1 Wl1^T) * dl0 ; g_n = g_pl0 Wl0^T ; g_pre2 = (g_n W3^T) * d2 ; g_agg = g_pre2 W2^T
//   block : g_n += g_x Wx_i^T ;                                                   g_pre2 = (g_n W3^T) * d2 ; g_agg = g_pre2 W2^T
//
// (g_h = dE/dh is a per-graph row for the MLP head - written by the readout kernel - or the last_mlp kernel itself for
// the linear head).  Same tile machinery as the forward: four waves per 16-node tile, each wave a 32-column slice of
// every weight matrix in registers (mp_schnet_node_pack_f32 images of the transposed kernels), activations handed from
// GEMM to GEMM through LDS - 5 (head) or 3 (block) GEMMs per launch instead of one launch each.
// Between the chains run the edge kernels: mp_cfconv_gauss_dist_grad_f32 (dE/dd) and the forward cfconv kernel with
// swapped index columns (dE/dx_j); the force kernel at the end turns dE/dd into -dE/dx over both CSRs.
#include "mp_common.h"
#include "mp_node_tile.h"

namespace {

struct BwdArgs {
  int64_t N;
  int ntiles;
  // head input: tile[row][k] = gh[(gh_row ? gh_row[node] : 0) * 64 + k] * dl1[node * 64 + k]
  const float* gh;
  const int32_t* gh_row;
  const float* dl1;
  const float* Wl1T;   // packed (64 -> 128)
  const float* dl0;    // (N, F)
  const float* Wl0T;   // packed (128 -> 128)
  // block input: g_x (N, F), consumed rows re-zeroed (the next swapped cfconv accumulates into them)
  float* gx;
  const float* WxT;    // packed
  // both
  float* g_n;          // (N, F) head: written; block: updated in place
  const float* W3T;
  const float* d2;     // (N, F)
  const float* W2T;
  float* g_agg;        // (N, F) written
};

#define MP_BWD_FOR_OUT(cb, r, row, col, body)                       \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                 \
      const int row = 4 * (lane >> 4) + r;                          \
      const int col = wave * 32 + 16 * cb + (lane & 15);            \
      body                                                          \
    }                                                               \
  }

template <bool HEAD>
__global__ __launch_bounds__(256, HEAD ? 1 : 2) void schnet_bwd_chain_kernel(BwdArgs a) {
  __shared__ float Xa[16 * X_LD];
  __shared__ float Xb[16 * X_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nblocks = gridDim.x;

  // ---- input tile through registers: requested before the weight slices, next tile's during the GEMMs ---------------
  float4 stg[HEAD ? 1 : 2];
  auto stage_load = [&](int t) {
    const int64_t n0 = static_cast<int64_t>(t) * 16;
    if constexpr (HEAD) {
      const int r = tid >> 4, k4 = tid & 15;
      const int64_t node = n0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < a.ntiles && node < a.N) {
        const int64_t src = a.gh_row ? a.gh_row[node] : 0;
        const float4 g = reinterpret_cast<const float4*>(a.gh + src * 64)[k4];
        const float4 d = reinterpret_cast<const float4*>(a.dl1 + node * 64)[k4];
        v = make_float4(g.x * d.x, g.y * d.y, g.z * d.z, g.w * d.w);
      }
      stg[0] = v;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = tid + j * 256;
        const int r = i / (F / 4), k4 = i % (F / 4);
        const int64_t node = n0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < a.ntiles && node < a.N) {
          float4* p = reinterpret_cast<float4*>(a.gx + node * F) + k4;
          v = *p;
          *p = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        stg[j] = v;
      }
    }
  };
  auto stage_store = [&]() {
    if constexpr (HEAD) {
      float* d = Xa + (tid >> 4) * X_LD + 4 * (tid & 15);
      d[0] = stg[0].x; d[1] = stg[0].y; d[2] = stg[0].z; d[3] = stg[0].w;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = tid + j * 256;
        float* d = Xa + (i / (F / 4)) * X_LD + 4 * (i % (F / 4));
        d[0] = stg[j].x; d[1] = stg[j].y; d[2] = stg[j].z; d[3] = stg[j].w;
      }
    }
  };
  stage_load(blockIdx.x);

  // ---- weight slices -> registers ------------------------------------------------------------------------------------
  float w_a[2][(HEAD ? 64 : F) / 4];   // head: Wl1^T ; block: Wx^T
  float w_b[2][(HEAD ? F : 4) / 4];    // head: Wl0^T
  float w_c[2][F / 4];                 // W3^T
  float w_d[2][F / 4];                 // W2^T
  if constexpr (HEAD) {
    load_wslice_packed<64, 2>(a.Wl1T, wave, lane, w_a);
    load_wslice_packed<F, 2>(a.Wl0T, wave, lane, w_b);
  } else {
    load_wslice_packed<F, 2>(a.WxT, wave, lane, w_a);
  }
  load_wslice_packed<F, 2>(a.W3T, wave, lane, w_c);
  load_wslice_packed<F, 2>(a.W2T, wave, lane, w_d);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += nblocks) {
    const int64_t node0 = static_cast<int64_t>(tile) * 16;
    stage_store();
    __syncthreads();
    stage_load(tile + nblocks);
    // the saved derivatives / the running node gradient of this thread's output elements: asked for now, used after the
    // GEMMs that hide their latency
    float first[2][4], d2v[2][4];
    MP_BWD_FOR_OUT(cb, r, row, col, {
      const bool ok = node0 + row < a.N;
      first[cb][r] = ok ? (HEAD ? a.dl0[(node0 + row) * F + col] : a.g_n[(node0 + row) * F + col]) : 0.0f;
    })
    floatx4 acc[1][2];
#define MP_BWD_ZERO acc[0][0] = floatx4{0.f, 0.f, 0.f, 0.f}; acc[0][1] = floatx4{0.f, 0.f, 0.f, 0.f};
    float* cur;   // the LDS tile that holds g_n of this tile after the first phase
    if constexpr (HEAD) {
      // g_pl0 = (g_pl1 Wl1^T) * dl0
      MP_BWD_ZERO
      gemm_tile<64, 2, 1>(Xa, lane, w_a, acc);
      MP_BWD_FOR_OUT(cb, r, row, col, { Xb[row * X_LD + col] = acc[0][cb][r] * first[cb][r]; })
      __syncthreads();
      // g_n = g_pl0 Wl0^T
      MP_BWD_ZERO
      gemm_tile<F, 2, 1>(Xb, lane, w_b, acc);
      MP_BWD_FOR_OUT(cb, r, row, col, {
        const float v = acc[0][cb][r];
        Xa[row * X_LD + col] = v;
        if (node0 + row < a.N) a.g_n[(node0 + row) * F + col] = v;
      })
      cur = Xa;
    } else {
      // g_n += g_x Wx^T
      MP_BWD_ZERO
      gemm_tile<F, 2, 1>(Xa, lane, w_a, acc);
      MP_BWD_FOR_OUT(cb, r, row, col, {
        const float v = acc[0][cb][r] + first[cb][r];
        Xb[row * X_LD + col] = v;
        if (node0 + row < a.N) a.g_n[(node0 + row) * F + col] = v;
      })
      cur = Xb;
    }
    float* oth = cur == Xa ? Xb : Xa;
    MP_BWD_FOR_OUT(cb, r, row, col, {   // in flight during the W3^T GEMM (asked for earlier it costs spills)
      d2v[cb][r] = (node0 + row < a.N) ? a.d2[(node0 + row) * F + col] : 0.0f;
    })
    __syncthreads();
    // g_pre2 = (g_n W3^T) * d2
    MP_BWD_ZERO
    gemm_tile<F, 2, 1>(cur, lane, w_c, acc);
    MP_BWD_FOR_OUT(cb, r, row, col, { oth[row * X_LD + col] = acc[0][cb][r] * d2v[cb][r]; })
    __syncthreads();
    // g_agg = g_pre2 W2^T
    MP_BWD_ZERO
    gemm_tile<F, 2, 1>(oth, lane, w_d, acc);
    MP_BWD_FOR_OUT(cb, r, row, col, {
      if (node0 + row < a.N) a.g_agg[(node0 + row) * F + col] = acc[0][cb][r];
    })
    __syncthreads();   // Xa / Xb are reused by the next tile
  }
}

__device__ __forceinline__ float wave_sum_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// out_n = scale * dE/dx_n with dE/dx_n = sum over the edges that touch n of g_d[e] (x_n - x_other) / d_e
// (d = |x_recv - x_send|: the same expression on the receiver and on the sender side; divide_no_nan - nothing at d = 0).
// One wave per node, lanes stride the node's receiver-side and sender-side edge lists (CSR offsets + the stable-sort
// permutations of the index plan), fixed-shape wave reduction: deterministic.
__global__ __launch_bounds__(256) void schnet_force_kernel(const float* __restrict__ g_d, const float* __restrict__ xyz,
                                                           const float* __restrict__ dist,
                                                           const int32_t* __restrict__ recv,
                                                           const int32_t* __restrict__ send,
                                                           const int32_t* __restrict__ ptr0,
                                                           const int32_t* __restrict__ perm0,
                                                           const int32_t* __restrict__ ptr1,
                                                           const int32_t* __restrict__ perm1, int64_t N, int64_t M,
                                                           float scale, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t n = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; n < N; n += nwaves) {
    const float x0 = xyz[n * 3], x1 = xyz[n * 3 + 1], x2 = xyz[n * 3 + 2];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int side = 0; side < 2; ++side) {
      const int32_t* ptr = side == 0 ? ptr0 : ptr1;
      const int32_t* perm = side == 0 ? perm0 : perm1;
      const int32_t* other = side == 0 ? send : recv;
      int lo = ptr[n], hi = ptr[n + 1];
      lo = lo < 0 ? 0 : lo;
      hi = hi > M ? static_cast<int>(M) : hi;
      for (int e = lo + lane; e < hi; e += 64) {
        const int64_t r = perm ? perm[e] : e;
        const float d = dist[r];
        const int64_t o = other[r];
        if (d == 0.0f || o < 0 || o >= N) continue;
        const float w = g_d[r] / d;
        acc[0] += w * (x0 - xyz[o * 3]);
        acc[1] += w * (x1 - xyz[o * 3 + 1]);
        acc[2] += w * (x2 - xyz[o * 3 + 2]);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = wave_sum_f(acc[k]);
    if (lane == 0) {
      out[n * 3 + 0] = scale * acc[0];
      out[n * 3 + 1] = scale * acc[1];
      out[n * 3 + 2] = scale * acc[2];
    }
  }
}

int launch_chain(bool head, BwdArgs a, hipStream_t s, const char* what) {
  a.ntiles = static_cast<int>((a.N + 15) / 16);
  const int grid = a.ntiles < 512 ? a.ntiles : 512;
  if (head) schnet_bwd_chain_kernel<true><<<grid, 256, 0, s>>>(a);
  else schnet_bwd_chain_kernel<false><<<grid, 256, 0, s>>>(a);
  return mp::check_launch(what);
}

}  // namespace

extern "C" {

int mp_schnet_bwd_head_f32(const float* gh, const int32_t* gh_row, const float* dl1, int64_t N, const float* Wl1T,
                           const float* dl0, const float* Wl0T, const float* W3T, const float* d2, const float* W2T,
                           float* g_n, float* g_agg, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_schnet_bwd_head_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(gh && dl1 && Wl1T && dl0 && Wl0T && W3T && d2 && W2T && g_n && g_agg, "mp_schnet_bwd_head_f32: null pointer");
  BwdArgs a{};
  a.N = N;
  a.gh = gh; a.gh_row = gh_row; a.dl1 = dl1; a.Wl1T = Wl1T; a.dl0 = dl0; a.Wl0T = Wl0T;
  a.g_n = g_n; a.W3T = W3T; a.d2 = d2; a.W2T = W2T; a.g_agg = g_agg;
  return launch_chain(true, a, mp::as_stream(stream), "mp_schnet_bwd_head_f32");
}

int mp_schnet_bwd_block_f32(float* g_x, int64_t N, const float* WxT, float* g_n, const float* W3T, const float* d2,
                            const float* W2T, float* g_agg, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_schnet_bwd_block_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(g_x && WxT && g_n && W3T && d2 && W2T && g_agg, "mp_schnet_bwd_block_f32: null pointer");
  BwdArgs a{};
  a.N = N;
  a.gx = g_x; a.WxT = WxT; a.g_n = g_n; a.W3T = W3T; a.d2 = d2; a.W2T = W2T; a.g_agg = g_agg;
  return launch_chain(false, a, mp::as_stream(stream), "mp_schnet_bwd_block_f32");
}

int mp_schnet_force_from_gd_f32(const float* g_d, const float* xyz, const float* dist, const int32_t* recv,
                                const int32_t* send, const int32_t* ptr0, const int32_t* perm0, const int32_t* ptr1,
                                const int32_t* perm1, int64_t N, int64_t M, float scale, float* out, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0, "mp_schnet_force_from_gd_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(ptr0 && ptr1 && xyz && out && (M == 0 || (g_d && dist && recv && send)),
             "mp_schnet_force_from_gd_f32: null pointer");
  schnet_force_kernel<<<mp::grid_for(N * 64), 256, 0, mp::as_stream(stream)>>>(g_d, xyz, dist, recv, send, ptr0, perm0,
                                                                               ptr1, perm1, N, M, scale, out);
  return mp::check_launch("mp_schnet_force_from_gd_f32");
}

int mp_schnet_force_launch(const mp_schnet_force_desc* f, mpStream_t stream) {
  MP_REQUIRE(f != nullptr, "mp_schnet_force_launch: null descriptor");
  const mp_schnet_forward_desc* d = &f->fwd;
  MP_REQUIRE(d->depth >= 1 && d->depth <= MP_SCHNET_MAX_DEPTH, "mp_schnet_force_launch: depth %d not in 1..%d", d->depth,
             MP_SCHNET_MAX_DEPTH);
  MP_REQUIRE(d->flags & 2, "mp_schnet_force_launch: node-side weights must be packed images (flags bit 1)");
  MP_REQUIRE(f->xs && f->d2 && f->dl0 && f->dl1 && f->g_n && f->g_agg && f->g_x && f->g_d && f->force && f->ptr0 &&
                 f->ptr1 && f->Wl0T && f->Wl1T,
             "mp_schnet_force_launch: null pointer");
  MP_REQUIRE(d->Wo0 == nullptr || (f->g_pool && f->node_graph), "mp_schnet_force_launch: the MLP head needs g_pool and "
             "node_graph");
  const int64_t NF = d->N * 128;
  const int32_t* seg0 = f->seg0 ? f->seg0 : d->recv;
  const int32_t* seg1 = f->seg1 ? f->seg1 : d->send;
  // ------------------------------------------------------------------------------------------------------- forward
  int rc = mp_schnet_stage0_f32(d->numbers, d->N, d->embedding, d->vocab, d->emb_dim == 128 ? 128 : 64, d->W0, d->b0,
                                d->Wx[0], d->n, f->xs, d->idx, d->M, d->node_splits, d->edge_splits, d->G, d->xyz,
                                d->recv, d->send, d->dist, d->flags_word, d->flags & (3 | 256), stream);
  if (rc != MP_OK) return rc;
  for (int i = 0; i < d->depth; ++i) {
    rc = mp_cfconv_gauss_fused_f32(f->xs + i * NF, d->N, d->dist, d->bins, d->g_distance, d->g_sigma, d->g_offset,
                                   d->packed[i], seg0, d->send, f->perm0, d->M, d->flags, d->agg, stream);
    if (rc != MP_OK) return rc;
    if (i + 1 < d->depth) {
      rc = mp_schnet_node_update_save_f32(d->agg, d->N, d->W2[i], d->b2[i], d->W3[i], d->b3[i], d->n, d->Wx[i + 1],
                                          f->xs + (i + 1) * NF, f->d2 + i * NF, d->flags & 3, stream);
    } else {
      rc = mp_schnet_node_last_save_f32(d->agg, d->N, d->W2[i], d->b2[i], d->W3[i], d->b3[i], d->n, d->Wl0, d->bl0,
                                        d->Wl1, d->bl1, d->h, f->d2 + i * NF, f->dl0, f->dl1, d->flags & 3, stream);
    }
    if (rc != MP_OK) return rc;
  }
  rc = mp_schnet_readout_grad_f32(d->h, d->node_splits, d->G, d->Wo0, d->bo0, d->Wo1, d->bo1, d->out,
                                  d->Wo0 ? f->g_pool : nullptr, stream);
  if (rc != MP_OK) return rc;
  // ------------------------------------------------------------------------------------------------------- reverse
  const int last = d->depth - 1;
  rc = mp_schnet_bwd_head_f32(d->Wo0 ? f->g_pool : d->Wo1, d->Wo0 ? f->node_graph : nullptr, f->dl1, d->N, f->Wl1T,
                              f->dl0, f->Wl0T, f->W3T[last], f->d2 + last * NF, f->W2T[last], f->g_n, f->g_agg, stream);
  if (rc != MP_OK) return rc;
  for (int i = last; i >= 0; --i) {
    rc = mp_cfconv_gauss_dist_grad_f32(f->xs + i * NF, f->g_agg, d->N, d->dist, d->bins, d->g_distance, d->g_sigma,
                                       d->g_offset, f->packed_bwd[i], d->recv, d->send, d->M, i == last ? 0 : 1, f->g_d,
                                       stream);
    if (rc != MP_OK) return rc;
    if (i == 0) break;   // block 0's sender features come from the embedding: no path to the coordinates
    rc = mp_cfconv_gauss_fused_f32(f->g_agg, d->N, d->dist, d->bins, d->g_distance, d->g_sigma, d->g_offset, d->packed[i],
                                   seg1, d->recv, f->perm1, d->M, d->flags, f->g_x, stream);
    if (rc != MP_OK) return rc;
    rc = mp_schnet_bwd_block_f32(f->g_x, d->N, f->WxT[i], f->g_n, f->W3T[i - 1], f->d2 + (i - 1) * NF, f->W2T[i - 1],
                                 f->g_agg, stream);
    if (rc != MP_OK) return rc;
  }
  return mp_schnet_force_from_gd_f32(f->g_d, d->xyz, d->dist, d->recv, d->send, f->ptr0, f->perm0, f->ptr1, f->perm1,
                                     d->N, d->M, f->force_scale, f->force, stream);
}

}  // extern "C"
