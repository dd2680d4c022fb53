// Fused SchNet continuous-filter convolution (kgcnn/layers/conv/schnet_conv.py:73-79):
//
//     w   = Dense(F, linear)(Dense(F, ssp)(rbf))         per-edge filter MLP  (B -> F -> F)
//     x_j = GatherNodesOutgoing([x, idx])                 sender rows
//     out = PoolingLocalEdges(sum)([x, x_j * w, idx])     segment-sum at the receiver
//
// in ONE kernel: neither the (M,F) hidden activations, the (M,F) filter, the gathered (M,F) sender rows nor
// the (M,F) messages ever reach HBM (the reference materialises all of them, ~5.7 KB/edge/block; this kernel
// reads 4 B (distance) or 4B B (rbf) + 8 B of indices per edge and adds each node row once or twice).
//
// Roofline: 2(BF + F^2) + 2F = 38.1 kflop per edge at F=128, B=20 against <= 100 B of compulsory traffic:
// MFMA-bound (FP32 matrix peak 157.3 TF).  FP32-in/FP32-acc MFMA only (v_mfma_f32_32x32x2_f32) to hold the
// 1e-5 budget - there is no TF32 on gfx950.
//
// Structure (F = 128 fixed; wave64; one wave owns a tile of 32 consecutive edges of the receiver-sorted list):
//  * W1 (+ its bias as an extra input row) and W2 live in LDS for the whole persistent workgroup, stored
//    [k][4*c + blk] so that one ds_read_b128 yields the operands of all four 32-wide feature blocks.
//  * GEMM1 is computed TRANSPOSED, hT[f][e] = sum_b W1[b][f] rbf[e][b]: its accumulator then has the edge on the
//    lane and the feature in the register, which is exactly the A-operand layout of GEMM2 (edge rows, k = feature)
//    - the 128x32 hidden tile goes from one MFMA chain to the next in registers, no LDS round trip, no shuffles.
//    The k order of GEMM2 is the accumulator's register order (a permutation of 0..127), the weights are read in
//    that order.
//  * GEMM2 w[e][j] leaves the output feature on the lane and the edge in the register: the sender gather
//    x[send[e]][j] is a coalesced 128-B read per half wave, the multiply is in place.
//  * Segment-sum: the 32x128 message tile is transposed through a private LDS slab ([feature][edge], padded),
//    each lane then owns two features and walks the 32 edges in order with the (wave-uniform, scalar) receiver
//    ids: interior segments are stored, the first and last segment of a tile - which may continue in the
//    neighbouring tile - are added with one 256-B float atomic per 64 features.  A node whose edges span two
//    tiles receives two adds onto a zero row, so the result is order independent (a + b == b + a); only
//    receivers spanning three or more tiles (in-degree > 32) can differ in the last bit between runs.
//  * The output buffer must be zero on entry (unconnected nodes keep 0 = the has_unconnected pad of
//    kgcnn/layers/pooling.py:74-76).
#include "mp_common.h"

namespace {

using floatx16 = __attribute__((ext_vector_type(16))) float;

constexpr int F = 128;          // feature width of the fused kernel
constexpr int TE = 32;          // edges per wave tile
constexpr int MAX_KROWS = 34;   // W1 rows in LDS: B inputs + 1 bias row, padded to even (B <= 32)

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// shifted softplus, kgcnn/ops/activ.py:15, in the form max(x,0) + log1p(exp(-|x|)) - log(2): identical to TF's
// thresholded log1p(exp(x)) up to float rounding (|delta| < 2e-7 absolute) and free of overflow.
__device__ __forceinline__ float ssp_fast(float x) {
  // v_exp_f32 / v_log_f32 are base-2: t = 2^(-|x| log2 e), ssp = max(x,0) + ln2 * (log2(1 + t) - 1)
  const float t = __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f);
  const float l = __builtin_amdgcn_logf(1.0f + t);
  return fmaf(l - 1.0f, 0.6931471805599453f, fmaxf(x, 0.0f));
}
__device__ __forceinline__ float ssp_exact(float x) { return mp_softplus(x) - 0.6931471805599453f; }

struct CfconvArgs {
  const float* x;         // (N, F) sender-side node features
  const float* edge_in;   // GAUSS: dist (M) ; else rbf (M, B)
  const float* packed;    // mp_cfconv_pack_f32 output: [MAX_KROWS][F] W1|b1 rows, [F][F] W2, [F] b2 (LDS image order)
  const int32_t* recv;    // (M) receiver ids, ascending (already permuted if perm != null)
  const int32_t* send;    // (M) sender ids in original edge order
  const int32_t* perm;    // (M) sorted position -> original edge, or null
  float* out;             // (N, F) zero-initialised
  int64_t M, N;
  int B;
  float g_distance, g_gamma, g_offset;  // Gauss basis parameters (geom.py:567-571)
  int ntiles;
  unsigned long long* diag;  // optional [8] cycle sums per phase (diagnostic build only)
};

#define MP_STAMP(idx)                                                                  \
  if constexpr (DIAG) {                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();                     \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    diag_sum[idx] += t_now - t_prev;                                                   \
    t_prev = t_now;                                                                    \
  }

template <int WAVES, bool GAUSS, bool FAST_SSP, int NKT, bool DIAG = false>
__global__ __launch_bounds__(WAVES * 64) void cfconv_fused_kernel(CfconvArgs a) {
  unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = 0;
  if constexpr (DIAG) t_prev = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) float lds[];
  float* W1s = lds;                          // [MAX_KROWS][F] packed
  float* W2s = lds + MAX_KROWS * F;          // [F][F] packed

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31;
  const int hh = lane >> 5;
  const int B = a.B;
  // k-steps of GEMM1: B inputs + 1 bias row, two k per step; NKT > 0 fixes it at compile time (B = 20 -> 11)
  const int nk = NKT > 0 ? NKT : ((B + 2) >> 1);
  constexpr int NKMAX = NKT > 0 ? NKT : MAX_KROWS / 2;

  // ---- per-tile edge data is fetched one tile ahead (the first tile's before the weight staging) so that its
  //      latency (perm -> send / dist is a dependent chain) hides under staging resp. the previous tile's MFMAs -----
  const int tile_first = blockIdx.x * WAVES + wave;
  const int tile_step = gridDim.x * WAVES;
  int nx_send = 0, nx_recv = 0;
  float nx_d = 0.0f;
  int64_t nx_ep = 0;
  // Lane c does not take edge c of the tile but edge eps(c): with that assignment the accumulator rows of GEMM2
  // (row = (r&3) + 8(r>>2) + 4hh) hold, per lane half, SIXTEEN CONSECUTIVE edges in register order (edge 16hh + r), so
  // the segmented sum below runs in registers.  Receiver ids are additionally kept in plain edge order (lane c <-> edge c)
  // for the segment mask.
  const int eps_c = 16 * ((c >> 2) & 1) + (c & 3) + 4 * (c >> 3);
  auto prefetch_tile = [&](int t) {
    if (t < a.ntiles) {
      const int64_t e0n = static_cast<int64_t>(t) * TE;
      const int64_t e = e0n + eps_c;
      const int64_t ec = e < a.M ? e : a.M - 1;
      nx_ep = a.perm ? static_cast<int64_t>(a.perm[ec]) : ec;
      nx_send = a.send[nx_ep];
      const int64_t es = (e0n + c) < a.M ? (e0n + c) : a.M - 1;
      nx_recv = a.recv[es];
      if constexpr (GAUSS) nx_d = a.edge_in[nx_ep];
    }
  };
  prefetch_tile(tile_first);

  // ---- stage the pre-packed weights once per workgroup.  The small W1 image (17 KB) goes through registers and is
  //      published by the first barrier - GEMM1 needs nothing else.  The 64 KB of W2 follow by LDS-DMA
  //      (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPRs), issued after that barrier so that no ordinary
  //      load is pending beside them, and are awaited only right before the first GEMM2: their flight hides under the
  //      first tile's sender-row loads, Gauss basis, GEMM1 and softplus. -----------------------------------------------
  {
    const float4* src = reinterpret_cast<const float4*>(a.packed);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < (MAX_KROWS * F) / 4; i += WAVES * 64) dst[i] = src[i];
  }
  float bias2[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) bias2[jb] = a.packed[MAX_KROWS * F + F * F + jb * 32 + c];
  __syncthreads();
  {
    constexpr int CHUNKS_PER_WAVE = (F * F / 256) / WAVES;  // 1-KB chunks of the W2 image per wave
    const float* src = a.packed + MAX_KROWS * F;
#pragma unroll
    for (int i = 0; i < CHUNKS_PER_WAVE; ++i) {
      const int chunk = wave * CHUNKS_PER_WAVE + i;
      __builtin_amdgcn_global_load_lds(src + chunk * 256 + lane * 4,
                                       (__attribute__((address_space(3))) void*)(W2s + chunk * 256), 16, 0, 0);
    }
  }
  bool w2_ready = false;  // wave-uniform: every wave passes the publishing barrier exactly once

  MP_STAMP(0)
  const float* w1_lane = W1s + (nk * hh) * F + 4 * c;  // + s*F           : rows s (lo half) / nk+s (hi half)
  const float* w2_lane = W2s + (4 * hh) * F + 4 * c;   // + (ib*32 + (r&3) + 8*(r>>2))*F

  for (int tile0 = tile_first; tile0 < a.ntiles; tile0 += tile_step) {
    const int tile = __builtin_amdgcn_readfirstlane(tile0);
    const int64_t e0 = static_cast<int64_t>(tile) * TE;
    const int64_t e_mine = e0 + eps_c;
    const bool valid = e_mine < a.M;
    const int64_t ep = nx_ep;
    const int my_send = nx_send;
    const int my_recv = nx_recv;
    const float d_mine = nx_d;

    // ---- sender rows of this tile: issued first, consumed after GEMM2 (coalesced 128-B reads per half wave) --------
    float xv[4][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rowmap(r, hh);               // MFMA row; its lane handles edge 16 hh + r
      const int snode = __shfl(my_send, row, 64);
      const bool row_valid = (e0 + 16 * hh + r) < a.M;
      const float* xrow = a.x + static_cast<int64_t>(snode) * F + c;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const float t = xrow[jb * 32];  // always in range: padding rows reuse the last valid edge's sender
        xv[jb][r] = row_valid ? t : 0.0f;
      }
    }
    prefetch_tile(tile0 + tile_step);  // next tile's edge data, in flight during this tile's GEMMs

    // ---- B operand of GEMM1: this lane's half of its edge's basis row (+ the constant 1 of the bias row) ----
    float rb[NKMAX];
    if constexpr (GAUSS) {
      const float d = d_mine;
      const float fbins = static_cast<float>(B);
#pragma unroll
      for (int s = 0; s < NKMAX; ++s) {
        const int k = s + nk * hh;
        const float mu = static_cast<float>(k) / fbins * a.g_distance;
        const float v = (d - a.g_offset) - mu;
        float val;
        if constexpr (FAST_SSP) {
          val = __builtin_amdgcn_exp2f((v * v) * (a.g_gamma * -1.4426950408889634f));  // v_exp_f32: 2^(x log2 e)
        } else {
          val = expf((v * v) * (a.g_gamma * -1.0f));
        }
        val = k < B ? val : (k == B ? 1.0f : 0.0f);
        rb[s] = (s < nk && valid) ? val : 0.0f;
      }
    } else {
#pragma unroll
      for (int s = 0; s < NKMAX; ++s) {
        const int k = s + nk * hh;
        float val = 0.0f;
        if (s < nk && valid) val = k < B ? a.edge_in[ep * B + k] : (k == B ? 1.0f : 0.0f);
        rb[s] = val;
      }
    }

    MP_STAMP(1)
    // ---- GEMM1 (transposed): hT[f][e] = sum_k W1p[k][f] * rb[e][k]; lane = edge, register = feature ---------
    floatx16 h[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[ib][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NKMAX; ++s) {
      if (s < nk) {
        const float4 wv = *reinterpret_cast<const float4*>(w1_lane + s * F);
        h[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, rb[s], h[0], 0, 0, 0);
        h[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, rb[s], h[1], 0, 0, 0);
        h[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, rb[s], h[2], 0, 0, 0);
        h[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, rb[s], h[3], 0, 0, 0);
      }
    }
    MP_STAMP(2)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[ib][r] = FAST_SSP ? ssp_fast(h[ib][r]) : ssp_exact(h[ib][r]);

    if (!w2_ready) {
      __syncthreads();  // drains this wave's LDS-DMA (vmcnt) and publishes all four waves' parts of W2
      w2_ready = true;
    }
    MP_STAMP(3)
    // ---- GEMM2: w[e][j] = sum_f h[e][f] W2[f][j] + b2[j]; A = the accumulator registers of GEMM1 ------------
    floatx16 w[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) w[jb][r] = bias2[jb];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 bv =
            *reinterpret_cast<const float4*>(w2_lane + (ib * 32 + (r & 3) + 8 * (r >> 2)) * F);
        const float av = h[ib][r];
        w[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, w[0], 0, 0, 0);
        w[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, w[1], 0, 0, 0);
        w[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, w[2], 0, 0, 0);
        w[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, w[3], 0, 0, 0);
      }
    }

    MP_STAMP(4)
    // ---- multiply by the sender row and sum the segments IN REGISTERS.  Each lane half holds 16 consecutive edges of
    //      one feature column per block; it walks them with lane-varying predicates (v_cndmask, no LDS, no scalar
    //      branches per edge): a segment that closes inside the half is stored at once (it is interior to the tile)
    //      except the half's first one, which is kept; afterwards the low half's open tail is handed to the high half
    //      if the segment continues across edge 15|16, and the (at most four) boundary segments are written: plain
    //      stores for tile-interior ones, one float atomic for the tile's first and last segment, which may continue in
    //      the neighbouring tiles.  Padding edges of the last tile repeat the last receiver and carry zeros. -----------
    const int prev_recv = __shfl_up(my_recv, 1, 64);  // all lanes take part; lanes 0 / 32 are masked out below
    const unsigned start_mask =
        static_cast<unsigned>(__ballot((c > 0) & (my_recv != prev_recv)) & 0xffffffffull);
    const unsigned half_mask = hh ? (start_mask >> 16) : (start_mask & 0xffffu);  // bit r: edge 16hh + r opens a segment
    int my_rows[16];  // receiver of each of this half's edges
#pragma unroll
    for (int r = 0; r < 16; ++r) my_rows[r] = __shfl(my_recv, 16 * hh + r, 64);
    float acc[4], first[4];
    int closed = 0;  // segments closed so far in this half
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      acc[jb] = w[jb][0] * xv[jb][0];
      first[jb] = 0.0f;
    }
    float* const out_col = a.out + c;
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const bool opens = (half_mask >> r) & 1u;
      if (opens && closed > 0) {  // a segment strictly inside the half: exclusive to this tile
        float* dst = out_col + static_cast<int64_t>(my_rows[r - 1]) * F;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) dst[jb * 32] = acc[jb];
      }
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const float m = w[jb][r] * xv[jb][r];
        first[jb] = (opens && closed == 0) ? acc[jb] : first[jb];
        acc[jb] = opens ? m : acc[jb] + m;
      }
      closed += opens ? 1 : 0;
    }
    MP_STAMP(5)
    // wave-uniform shape of the tile
    const int nlo = __builtin_popcount(start_mask & 0xfffeu) + 1;   // segments touching the low half
    const int nhi = __builtin_popcount(start_mask >> 17) + 1;        // segments touching the high half
    const bool cont = ((start_mask >> 16) & 1u) == 0;                // edge 16 continues the segment of edge 15
    // hand the low half's open tail to the high half (lanes c+32) if the segment continues
    float tail[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) tail[jb] = __shfl_xor(acc[jb], 32, 64);  // in hi lanes: the low half's tail
    MP_STAMP(6)
    if (hh == 0) {
      if (nlo > 1) {  // the tile's first segment closed inside the low half
        float* dst = out_col + static_cast<int64_t>(__builtin_amdgcn_readlane(my_recv, 0)) * F;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) atomicAdd(dst + jb * 32, first[jb]);
      }
      if (!cont) {    // the low half's last segment ends at edge 15
        float* dst = out_col + static_cast<int64_t>(__builtin_amdgcn_readlane(my_recv, 15)) * F;
        if (nlo == 1) {
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) atomicAdd(dst + jb * 32, acc[jb]);   // it is also the tile's first
        } else {
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) dst[jb * 32] = acc[jb];
        }
      }
    } else {
      // the high half's first segment (the whole half if nhi == 1), plus the low half's tail when it continues
      float v0[4];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const float mine = nhi > 1 ? first[jb] : acc[jb];
        v0[jb] = cont ? tail[jb] + mine : mine;   // edge order: low-half part first
      }
      {
        float* dst = out_col + static_cast<int64_t>(__builtin_amdgcn_readlane(my_recv, 16)) * F;
        const bool boundary = (cont && nlo == 1) || nhi == 1;   // contains the tile's first or last edge
        if (boundary) {
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) atomicAdd(dst + jb * 32, v0[jb]);
        } else {
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) dst[jb * 32] = v0[jb];
        }
      }
      if (nhi > 1) {  // the tile's last segment
        float* dst = out_col + static_cast<int64_t>(__builtin_amdgcn_readlane(my_recv, 31)) * F;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) atomicAdd(dst + jb * 32, acc[jb]);
      }
    }
    MP_STAMP(7)
  }
  if (!w2_ready) __syncthreads();  // a wave without tiles still owes the workgroup its barrier
  if constexpr (DIAG) {
    if (lane == 0 && a.diag) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(a.diag + i, diag_sum[i]);
    }
  }
}

constexpr int PACKED_FLOATS = MAX_KROWS * F + F * F + F;

// LDS image of the filter-MLP weights: row k of W1 (k < B), the bias b1 as row B, zero rows up to MAX_KROWS, then W2,
// every row stored [4*c + blk] = W[k][blk*32 + c]; finally b2 in natural order.
__global__ void cfconv_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int B,
                                   const float* __restrict__ W2, const float* __restrict__ b2,
                                   float* __restrict__ packed) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PACKED_FLOATS; i += stride) {
    float v = 0.0f;
    if (i < MAX_KROWS * F) {
      const int k = i / F, col = (i % F) / 4 + 32 * (i % 4);
      if (k < B) v = W1[k * F + col];
      else if (k == B && b1) v = b1[col];
    } else if (i < MAX_KROWS * F + F * F) {
      const int j = i - MAX_KROWS * F;
      const int k = j / F, col = (j % F) / 4 + 32 * (j % 4);
      v = W2[k * F + col];
    } else {
      v = b2 ? b2[i - MAX_KROWS * F - F * F] : 0.0f;
    }
    packed[i] = v;
  }
}

template <int WAVES>
size_t cfconv_lds_bytes() {
  return sizeof(float) * (MAX_KROWS * F + F * F);
}

template <int WAVES, bool GAUSS, bool FAST, int NKT, bool DIAG>
int launch_cfconv(const CfconvArgs& args, int grid, hipStream_t s) {
  const size_t lds = cfconv_lds_bytes<WAVES>();
  static bool attr_set = false;
  if (!attr_set) {
    MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_kernel<WAVES, GAUSS, FAST, NKT, DIAG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    attr_set = true;
  }
  cfconv_fused_kernel<WAVES, GAUSS, FAST, NKT, DIAG><<<grid, WAVES * 64, lds, s>>>(args);
  return mp::check_launch("mp_cfconv_fused_f32");
}

template <bool GAUSS, bool FAST>
int launch_by_basis(const CfconvArgs& args, int waves, int grid, hipStream_t s) {
  if (waves == 8) {
    if (args.B == 20) return launch_cfconv<8, GAUSS, FAST, 11, false>(args, grid, s);
    return launch_cfconv<8, GAUSS, FAST, 0, false>(args, grid, s);
  }
  if (args.B == 20) return launch_cfconv<4, GAUSS, FAST, 11, false>(args, grid, s);  // SchNet default: 20 bins
  return launch_cfconv<4, GAUSS, FAST, 0, false>(args, grid, s);
}

int cfconv_dispatch(CfconvArgs args, bool gauss, int flags, hipStream_t s) {
  MP_REQUIRE(args.M >= 0 && args.N >= 0, "mp_cfconv: bad sizes");
  MP_REQUIRE(args.B >= 1 && args.B <= MAX_KROWS - 2, "mp_cfconv: basis size B=%d must be in 1..%d", args.B,
             MAX_KROWS - 2);
  if (args.M == 0 || args.N == 0) return MP_OK;
  MP_REQUIRE(args.x && args.edge_in && args.packed && args.recv && args.send && args.out, "mp_cfconv: null pointer");
  MP_REQUIRE(args.M < (int64_t{1} << 31), "mp_cfconv: M must fit int32");
  args.ntiles = static_cast<int>((args.M + TE - 1) / TE);
  const bool fast = (flags & 1) != 0;
  // One workgroup per CU (the LDS holds the weights), persistent over the tiles; four waves = one per SIMD, each
  // with a matrix pipe of its own.  An 8-wave build (two per SIMD, 256-VGPR cap, two-pass epilogue) exists behind
  // flag bit 2; measured on MI355X it gains < 1 % at 2.5 M edges (the 4-wave kernel already keeps the pipe ~80 %
  // busy at the clock the chip holds under this load) and loses at small M, so it is not selected automatically.
  int waves = (flags & 4) ? 8 : 4;
  int grid = (args.ntiles + waves - 1) / waves;
  if (grid > 256) grid = 256;
  if (args.diag) {
    MP_REQUIRE(gauss && args.B == 20, "mp_cfconv: the diagnostic build exists for the 20-bin Gauss variant only");
    grid = (args.ntiles + 3) / 4 > 256 ? 256 : (args.ntiles + 3) / 4;
    return launch_cfconv<4, true, true, 11, true>(args, grid, s);
  }
  if (gauss) {
    return fast ? launch_by_basis<true, true>(args, waves, grid, s) : launch_by_basis<true, false>(args, waves, grid, s);
  }
  return fast ? launch_by_basis<false, true>(args, waves, grid, s) : launch_by_basis<false, false>(args, waves, grid, s);
}

}  // namespace

extern "C" {

int mp_cfconv_packed_floats(void) { return PACKED_FLOATS; }

int mp_cfconv_pack_f32(const float* W1, const float* b1, int B, const float* W2, const float* b2, float* packed,
                       mpStream_t stream) {
  MP_REQUIRE(B >= 1 && B <= MAX_KROWS - 2, "mp_cfconv_pack_f32: basis size B=%d must be in 1..%d", B, MAX_KROWS - 2);
  MP_REQUIRE(W1 && W2 && packed, "mp_cfconv_pack_f32: null pointer");
  cfconv_pack_kernel<<<64, 256, 0, mp::as_stream(stream)>>>(W1, b1, B, W2, b2, packed);
  return mp::check_launch("mp_cfconv_pack_f32");
}

int mp_cfconv_fused_f32(const float* x, int64_t N, const float* rbf, int B, const float* packed,
                        const int32_t* recv_sorted, const int32_t* send, const int32_t* perm, int64_t M, int flags,
                        float* out_zeroed, mpStream_t stream) {
  CfconvArgs a{};
  a.x = x; a.edge_in = rbf; a.packed = packed;
  a.recv = recv_sorted; a.send = send; a.perm = perm; a.out = out_zeroed;
  a.M = M; a.N = N; a.B = B;
  return cfconv_dispatch(a, false, flags, mp::as_stream(stream));
}

int mp_cfconv_gauss_fused_f32(const float* x, int64_t N, const float* dist, int bins, float distance, float sigma,
                              float offset, const float* packed, const int32_t* recv_sorted, const int32_t* send,
                              const int32_t* perm, int64_t M, int flags, float* out_zeroed, mpStream_t stream) {
  MP_REQUIRE(sigma != 0.0f, "mp_cfconv_gauss_fused_f32: sigma must be non-zero");
  CfconvArgs a{};
  a.x = x; a.edge_in = dist; a.packed = packed;
  a.recv = recv_sorted; a.send = send; a.perm = perm; a.out = out_zeroed;
  a.M = M; a.N = N; a.B = bins;
  a.g_distance = distance;
  a.g_gamma = static_cast<float>(1.0 / static_cast<double>(sigma) / static_cast<double>(sigma) / 2.0);
  a.g_offset = offset;
  return cfconv_dispatch(a, true, flags, mp::as_stream(stream));
}

// Diagnostic build of the 20-bin Gauss variant (fast softplus): adds per-phase s_memtime sums (8 x uint64,
// caller-zeroed) for lane 0 of every wave.  Its run time is not representative (the stamps fence the schedule); read
// the SHARES only.
int mp_cfconv_gauss_diag_f32(const float* x, int64_t N, const float* dist, int bins, float distance, float sigma,
                             float offset, const float* packed, const int32_t* recv_sorted, const int32_t* send,
                             const int32_t* perm, int64_t M, float* out_zeroed, unsigned long long* diag8,
                             mpStream_t stream) {
  MP_REQUIRE(sigma != 0.0f && diag8, "mp_cfconv_gauss_diag_f32: bad arguments");
  CfconvArgs a{};
  a.x = x; a.edge_in = dist; a.packed = packed;
  a.recv = recv_sorted; a.send = send; a.perm = perm; a.out = out_zeroed;
  a.M = M; a.N = N; a.B = bins;
  a.g_distance = distance;
  a.g_gamma = static_cast<float>(1.0 / static_cast<double>(sigma) / static_cast<double>(sigma) / 2.0);
  a.g_offset = offset;
  a.diag = diag8;
  return cfconv_dispatch(a, true, 1, mp::as_stream(stream));
}

}  // extern "C"
