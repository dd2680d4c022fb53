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
// MFMA-bound.  FP32 results throughout (the 1e-5 budget; there is no TF32 on gfx950), computed on the BF16 matrix pipe as
// an exact FP32 emulation - every FP32 operand is split into three bf16 pieces (8 + 8 + 8 mantissa bits: hi = bf16(x),
// mid = bf16(x - hi), lo = bf16(x - hi - mid), each difference exact in FP32), and the six leading cross products
// (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi) are accumulated in FP32 by v_mfma_f32_32x32x16_bf16; the three dropped
// products are below 2^-24 of |a||b|, the rounding of an FP32 product itself.  Measured on a 32x128x128 tile against
// float64 (scripts/probes/bf16x3_probe.hip): max error 2.55e-7 of the output scale, the FP32 MFMA chain 2.56e-7.
// GEMM2 (K = 128, 85 % of the matrix work): six 32-cycle MFMAs replace eight 64-cycle ones per 16 k, 2.67x the FP32
// matrix rate for the same result.  GEMM1 (K = B + 1): the same for the basis sizes with their own build (20, 25: the k
// slots of both lane halves fill two k blocks); other sizes keep v_mfma_f32_32x32x2_f32.
//
// Structure (F = 128 fixed; wave64; one wave owns a tile of 32 consecutive edges of the receiver-sorted list):
//  * W1 (+ its bias as an extra input row) and W2 live in LDS for the whole persistent workgroup, as bf16-piece images
//    in MFMA operand order (one ds_read_b128 per piece and step; staged by LDS-DMA); lane c of GEMM2 owns the four output
//    features 4c..4c+3 (one per accumulator block).  Generic basis sizes: W1 as FP32 rows [4*c + blk], one ds_read_b128
//    yields the A operands of all four 32-wide hidden blocks.
//  * GEMM1 is computed TRANSPOSED, hT[f][e] = sum_b W1[b][f] rbf[e][b]: its accumulator then has the edge on the
//    lane and the feature in the register, which is exactly the A-operand layout of GEMM2 (edge rows, k = feature)
//    - the 128x32 hidden tile goes from one MFMA chain to the next in registers, no LDS round trip, no shuffles.
//    The k order of GEMM2 is the accumulator's register order (a permutation of 0..127), the weights are read in
//    that order.
//  * Lane c of the wave does not take edge c of the tile but edge eps(c), a permutation chosen so that the accumulator
//    of GEMM2 holds, per lane half, SIXTEEN CONSECUTIVE edges in register order.  The sender gather x[send[e]][4c..4c+3]
//    is then one 16-B load per lane and edge (a half wave reads a whole 512-B row), the multiply is in place, and the
//  * segment-sum is a register walk: the segment structure of a tile is one 32-bit start mask (a ballot over the
//    sorted receiver ids), i.e. wave-uniform, so the walk is driven by scalar tests - a step without a boundary is
//    two v_pk_mul_f32 + two v_pk_add_f32, a step with one branches (s_cbranch) into code that stores the closed
//    segment (16 B per lane) under a lane mask built on the scalar unit.  Segments interior to the tile are stored;
//    the first and last segment of a tile - which may continue in the neighbouring tile - are added with float
//    atomics, after a 512-B lane transposition through LDS that makes each atomic instruction cover one contiguous
//    128-B line (device-scope atomics are paid per line).  A node whose edges span two tiles receives two adds onto a
//    zero row, so the result is order independent (a + b == b + a); only receivers spanning three or more tiles
//    (in-degree > 32) can differ in the last bit between runs.
//  * A bf16 MFMA holds the SIMD's vector issue for 8 of its 32 cycles: the vector work between the GEMMs (softplus,
//    operand splits) is cut into tasks of <= 24 issue cycles and pinned to MFMA slots (see PIPE below); FP32 MFMAs (the
//    generic builds' GEMM1) block the vector lanes entirely (scripts/probes/mfma_probe.hip), so the phases outside the
//    matrix work are written for instruction count (packed FP32, scalar control, 6-instruction softplus).
//  * The output buffer must be zero on entry (unconnected nodes keep 0 = the has_unconnected pad of
//    kgcnn/layers/pooling.py:74-76).
#include <mutex>
#include <type_traits>

#include "mp_common.h"

namespace {

using floatx16 = __attribute__((ext_vector_type(16))) float;
using floatx2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int F = 128;          // feature width of the fused kernel
constexpr int TE = 32;          // edges per wave tile
constexpr int MAX_KROWS = 34;   // W1 rows in LDS: B inputs + 1 bias row, padded to even (B <= 32)
// W2 in LDS: three bf16 images (hi, mid, lo pieces) in MFMA-operand order, 16 B per (piece, k block m, column block jb,
// lane): element i of that entry = piece(W2[f(m, i, hh)][4 c + jb]), f = 32 (m >> 1) + rowmap(8 (m & 1) + i, hh) - the
// feature that GEMM1's accumulator register (ib = m >> 1, r = 8 (m & 1) + i) holds in lane half hh.
constexpr int W2_PIECE_FLOATS = F * F / 2;             // 16384 bf16 = 32 KB per piece
constexpr int W2_IMG_FLOATS = 3 * W2_PIECE_FLOATS;     // 96 KB
// W1 (+ bias row) as bf16 pieces for GEMM1 on the bf16 pipe (basis sizes whose 2 nk <= 32 slots fit two k blocks of 16):
// 16 B per (piece, k block kb, hidden block ib, lane): element i = piece(W1ext[k][32 ib + c]) with s = 8 kb + i,
// k = s + nk hh for s < nk (the k assignment of the FP32 build: lane half hh holds rows nk hh .. nk hh + nk - 1), zero
// beyond; W1ext = W1 rows, then b1, then zeros.
constexpr int W1B_FLOATS = 3 * 2 * 4 * 64 * 4;         // 24 KB
constexpr int G1B_MAX_NK = 16;

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// shifted softplus, kgcnn/ops/activ.py:15, in the form max(x,0) + log1p(exp(-|x|)) - log(2): identical to TF's
// thresholded log1p(exp(x)) up to float rounding (|delta| < 2e-7 absolute) and free of overflow.
__device__ __forceinline__ float ssp_fast(float x) {
  // v_exp_f32 / v_log_f32 are base-2: t = 2^(-|x| log2 e), ssp = max(x,0) + ln2 * log2((1 + t) / 2).  Six VALU
  // instructions: the halving is exact and folded into one fma, max(x, 0) is an integer max on the bit pattern (no
  // canonicalising v_max in front of it) - FP32 MFMA and VALU share the SIMD's FP32 lanes on gfx950, so every VALU
  // instruction of this kernel is time taken from the matrix pipe (scripts/probes/mfma_probe.hip).
  const float t = __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f);
  const float l = __builtin_amdgcn_logf(__builtin_fmaf(t, 0.5f, 0.5f));
  const int xi = __float_as_int(x);
  const float relu = __int_as_float(xi > 0 ? xi : 0);
  return __builtin_fmaf(l, 0.6931471805599453f, relu);
}
__device__ __forceinline__ float ssp_exact(float x) { return mp_softplus(x) - 0.6931471805599453f; }

struct CfconvArgs {
  const float* x;         // (N, F) sender-side node features
  const float* edge_in;   // GAUSS: dist (M) ; else rbf (M, B)
  const float* packed;    // mp_cfconv_pack_f32 output: [MAX_KROWS][F] W1|b1 rows, 3 bf16 images of W2, [F] b2 (LDS image order)
  const int32_t* recv;    // (M) receiver ids, ascending (already permuted if perm != null)
  const int32_t* send;    // (M) sender ids in original edge order
  const int32_t* perm;    // (M) sorted position -> original edge, or null
  float* out;             // (N, F) zero-initialised
  int64_t M, N;
  int B;
  float g_distance, g_gamma, g_offset;  // Gauss basis parameters (geom.py:567-571)
  int ntiles;
  // deterministic mode (flags bit 5): the partial sums of a tile's first and last segment - which may continue in the
  // neighbouring tiles - are not added to `out` with float atomics but stored to bnd_val[2 tile + {0,1}][F] with their
  // receiver in bnd_node[...] (-1 = none); cfconv_boundary_kernel then adds them per receiver in tile order.
  float* bnd_val;
  int32_t* bnd_node;
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

template <int WAVES, bool GAUSS, bool FAST_SSP, int NKT, bool DIAG = false, bool COMPACT = false>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 4 && COMPACT) ? 2 : 1) void cfconv_fused_kernel(CfconvArgs a) {
  unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = 0;
  if constexpr (DIAG) t_prev = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) float lds[];
  // rows of W1 kept in LDS: all MAX_KROWS in the generic build, the 2 nk that GEMM1 reads when the basis size is fixed
  // (22 for B = 20: the workgroup then needs 79.3 KB, so two workgroups - e.g. of two forwards in flight - share a CU)
  constexpr bool G1B = NKT > 0 && NKT <= G1B_MAX_NK;   // GEMM1 on the bf16 pipe (three pieces per operand, like GEMM2)
  // hand-placed pipeline (volatile-asm LDS reads one step ahead, vector tasks pinned to MFMA slots): one wave per SIMD.
  // (With two waves per SIMD the same placement measured equal to the compiler's within noise: 496 vs 503 us at 2.5 M edges.)
  constexpr bool PIPE = G1B && WAVES == 4;
  constexpr int W1ROWS = NKT > 0 ? 2 * NKT : MAX_KROWS;
  constexpr int W1_LDS_FLOATS = G1B ? W1B_FLOATS : W1ROWS * F;
  float* W1s = lds;                          // [W1ROWS][F] packed, or the bf16 piece image (W1B_FLOATS)
  float* W2s = lds + W1_LDS_FLOATS;          // three bf16 operand images of W2 (W2_IMG_FLOATS)
  float* Xs = W2s + W2_IMG_FLOATS;           // [WAVES][2][F] lane-transposition scratch for the boundary atomics

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
  // XCD-aware block order (mp_common.h): consecutive edge tiles - the same molecules' sender rows - share one XCD's L2
  const int tile_first = static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x)) * WAVES + wave;
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

  // ---- stage the pre-packed weights once per workgroup, all of them by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave
  //      instruction, no VGPRs, nothing to wait for at the point of issue): W1's image, then the three W2 images.  (Round 1
  //      sent W1 through registers in a rolled load -> wait -> ds_write loop: three serial round trips to L2 before the
  //      first tile could start.)  The 120 KB take ~2 k cycles to arrive (64 B per clock and CU) - about the round trip of
  //      the first tile's edge data, which was requested just before them and whose wait (vmcnt retires in order) is also
  //      theirs.  One barrier, right before the first GEMM1, publishes the weights; it does not wait for the sender rows.
  //      (Measured and not kept: W2 requested after the edge data's wait so that it flies under GEMM1 - the requests then
  //      start a round trip later and GEMM2 waits for them: 10.9 vs 10.7 us at config 2.) ---------------------------------
  float bias2[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) bias2[jb] = a.packed[MAX_KROWS * F + W2_IMG_FLOATS + 4 * c + jb];
  constexpr int W2_CHUNKS_PER_WAVE = (W2_IMG_FLOATS / 256) / WAVES;  // 1-KB chunks of the W2 images per wave: 24 or 12
  auto request_weights = [&]() {
    constexpr int W1_CHUNKS = W1_LDS_FLOATS / 256;
    static_assert(W1_LDS_FLOATS % 256 == 0, "the W1 image is staged in 1-KB chunks");
    const float* src1 = a.packed + (G1B ? MAX_KROWS * F + W2_IMG_FLOATS + F : 0);
#pragma unroll
    for (int i = 0; i < (W1_CHUNKS + WAVES - 1) / WAVES; ++i) {
      const int chunk = i * WAVES + wave;   // (few chunks, needed first: not rotated)
      if (chunk < W1_CHUNKS)
        __builtin_amdgcn_global_load_lds(src1 + chunk * 256 + lane * 4,
                                         (__attribute__((address_space(3))) void*)(W1s + chunk * 256), 16, 0, 0);
    }
    const float* src = a.packed + MAX_KROWS * F;
#pragma unroll
    for (int i = 0; i < W2_CHUNKS_PER_WAVE; ++i) {
      const int chunk = wave * W2_CHUNKS_PER_WAVE + i;
      __builtin_amdgcn_global_load_lds(src + chunk * 256 + lane * 4,
                                       (__attribute__((address_space(3))) void*)(W2s + chunk * 256), 16, 0, 0);
    }
  };
  // s_waitcnt vmcnt(N), N < 64: everything but the wave's N youngest vector-memory operations has completed (they retire
  // in order).  bf16-GEMM1 build with one wave per SIMD: every LDS read of the weights is volatile asm, so a bare
  // s_barrier behind an explicit, exact wait orders them; the other builds read weights in plain C++ and take a full
  // drain and the fenced __syncthreads().
  auto publish = [&](auto vm_outstanding) {
    constexpr int N = decltype(vm_outstanding)::value;
    if constexpr (WAVES == 4 && G1B) {
      __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
      __builtin_amdgcn_s_barrier();
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  };
  // A wave WITH a tile has, at the barrier, issued behind the weight requests: the loads of its edge data (waited for
  // at tile start - which drains the requests, too), then (sender rows at tile start) 16 row loads and 0-4 loads of the
  // next tile's edge data.  "All but the youngest 16" therefore covers the weights and leaves the rows in flight.
  constexpr int ROWS_EARLY = (WAVES == 4 && !COMPACT) ? 16 : 0;
  using AllButRows = std::integral_constant<int, ROWS_EARLY>;
  using Everything = std::integral_constant<int, 0>;
  request_weights();
  // (Measured and not kept, config 2, against 10.5 us: the first tile's edge data by volatile-asm loads awaited by hand
  //  with the exact count, so that the tile starts under the weight transfer - 10.7 us, the barrier then waits for the
  //  same transfer; a per-workgroup rotation of the chunk order against L2 channel queueing - 11.0 us.)
  bool w_ready = false;   // wave-uniform: every wave passes the publishing barrier exactly once

  MP_STAMP(0)
  const float* w1_lane = W1s + (nk * hh) * F + 4 * c;  // + s*F           : rows s (lo half) / nk+s (hi half)
  const char* w2_lane = reinterpret_cast<const char*>(W2s) + lane * 16;   // + ((piece * 8 + m) * 4 + jb) * 1024

  for (int tile0 = tile_first; tile0 < a.ntiles; tile0 += tile_step) {
    const int tile = __builtin_amdgcn_readfirstlane(tile0);
    const int64_t e0 = static_cast<int64_t>(tile) * TE;
    const int64_t e_mine = e0 + eps_c;
    const bool valid = e_mine < a.M;
    const int64_t ep = nx_ep;
    const int my_send = nx_send;
    const int my_recv = nx_recv;
    const float d_mine = nx_d;

    // ---- sender rows of this tile (one 16-B read per lane and edge: a half wave reads one whole 512-B row).  With
    //      one wave per SIMD they are requested first and consumed after GEMM2; with two waves per SIMD (WAVES = 8,
    //      256 registers per wave) they are requested only after GEMM2 - the sibling wave's MFMAs cover the latency; the
    //      256-register 4-wave build requests them right before GEMM2. ----
    constexpr int X_AT = WAVES > 4 ? 2 : (COMPACT ? 1 : 0);  // 0: tile start, 1: before GEMM2, 2: after GEMM2
    float xv[4][16];
    auto load_sender_rows = [&]() {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rowmap(r, hh);               // MFMA row; its lane handles edge 16 hh + r
        const int snode = __shfl(my_send, row, 64);
        // always in range: padding rows reuse the last valid edge's sender
        const float4 t = *reinterpret_cast<const float4*>(a.x + static_cast<int64_t>(snode) * F + 4 * c);
        xv[0][r] = t.x;
        xv[1][r] = t.y;
        xv[2][r] = t.z;
        xv[3][r] = t.w;
      }
    };
    if constexpr (X_AT == 0) load_sender_rows();
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

    if (!w_ready) {
      publish(AllButRows{});  // this wave's share of the weights has landed, and after the barrier every other wave's
      w_ready = true;
    }
    MP_STAMP(1)
    // ---- GEMM1 (transposed): hT[f][e] = sum_k W1p[k][f] * rb[e][k]; lane = edge, register = feature ---------
    floatx16 h[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[ib][r] = 0.0f;
    floatx16 w[4];   // GEMM2's accumulators: w[jb][r] = filter value of edge row r, feature 4 c + jb (b2 joins in the walk)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) w[jb][r] = 0.0f;
    auto ssp_val = [&](int ib, int r) { h[ib][r] = FAST_SSP ? ssp_fast(h[ib][r]) : ssp_exact(h[ib][r]); };
    // The softplus in two stages of at most 24 issue cycles (what one bf16 MFMA leaves free): A = the exponential,
    // B = logarithm and the rest.  The arithmetic is ssp_fast's.
    // (MP_PIN: an empty volatile asm that "rewrites" the value - instruction selection lets pure arithmetic float towards
    //  its use, across the placement fences below; a pinned result is computed where the source says.)
#define MP_PIN(x) asm volatile("" : "+v"(x))
    auto ssp_a = [&](int ib, int r, float& t) {
      if constexpr (FAST_SSP) {
        t = __builtin_amdgcn_exp2f(fabsf(h[ib][r]) * -1.4426950408889634f);
        MP_PIN(t);
      } else {
        float y = ssp_exact(h[ib][r]);
        MP_PIN(y);
        h[ib][r] = y;
      }
    };
    auto ssp_b = [&](int ib, int r, float t) {
      if constexpr (FAST_SSP) {
        const float l = __builtin_amdgcn_logf(__builtin_fmaf(t, 0.5f, 0.5f));
        const int xi = __float_as_int(h[ib][r]);
        float y = __builtin_fmaf(l, 0.6931471805599453f, __int_as_float(xi > 0 ? xi : 0));
        MP_PIN(y);
        h[ib][r] = y;
      }
    };
    // A pieces of GEMM2's k block m: the (softplus-ed) accumulator registers 8 (m & 1) .. + 7 of hidden block m >> 1,
    // split exactly into hi + mid + lo (bf16 each)
    auto split_into = [&](int m, int i, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
      const float x = h[m >> 1][8 * (m & 1) + i];
      const __bf16 p0 = static_cast<__bf16>(x);
      const float r1 = x - static_cast<float>(p0);        // exact: at most 17 significant bits
      const __bf16 p1 = static_cast<__bf16>(r1);
      const float r2 = r1 - static_cast<float>(p1);       // exact
      hi[i] = p0;
      mid[i] = p1;
      lo[i] = static_cast<__bf16>(r2);
    };
    // the same for the value pair (2 j, 2 j + 1) of k block m, in two stages: A = hi pieces and first remainders,
    // B = mid and lo pieces
    // the same for the value pair (2 j, 2 j + 1) of k block m - one dword of each piece vector - in two stages of packed
    // instructions: A = v_cvt_pk_bf16_f32, widen both halves back (shift / mask), v_pk_add_f32 -> hi dword and the two
    // first remainders; B = the same again for mid, one more conversion for lo
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    auto pack2 = [](floatx2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); };
    auto widen2 = [](unsigned u) { return floatx2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; };
    auto set_dword = [](bf16x8& v, int j, unsigned u) {
      uint4 t = __builtin_bit_cast(uint4, v);
      (j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w) = u;
      v = __builtin_bit_cast(bf16x8, t);
    };
    auto split2_a = [&](int m, int j, bf16x8& hi, floatx2& rem) {
      const floatx2 x = {h[m >> 1][8 * (m & 1) + 2 * j], h[m >> 1][8 * (m & 1) + 2 * j + 1]};
      unsigned u = pack2(x);
      rem = x - widen2(u);           // exact
      MP_PIN(u);
      MP_PIN(rem);
      set_dword(hi, j, u);
    };
    auto split2_b = [&](int j, bf16x8& mid, bf16x8& lo, floatx2 rem) {
      unsigned u = pack2(rem);
      unsigned l = pack2(rem - widen2(u));
      MP_PIN(u);
      MP_PIN(l);
      set_dword(mid, j, u);
      set_dword(lo, j, l);
    };
    // Placement fences: the compiler clusters MFMAs and pushes the vector work behind them (measured: 16 transcendentals
    // after GEMM1's last MFMA, runs of 10-25 MFMAs without a vector instruction in GEMM2).  With one wave per SIMD the
    // vector tasks are therefore pinned to their MFMA slot; with two waves per SIMD (WAVES = 8) the sibling wave's MFMAs
    // fill this wave's vector phases and only whole steps are pinned.
#define MP_SLOT_FENCE() do { if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0); } while (0)
#define MP_STEP_FENCE() do { if constexpr (G1B) __builtin_amdgcn_sched_barrier(0); } while (0)
    bf16x8 a_hi, a_mid, a_lo;
    if constexpr (G1B) {
      // On the bf16 pipe, FP32-exact like GEMM2: the lane's basis values (B operand; slot s = 8 kb + i of its half) are
      // split into three bf16 pieces in registers, W1's pieces come pre-split from LDS (A operand: row = hidden feature
      // 32 ib + c), six products per (hidden block, k block), smallest first: 48 MFMAs of 32 cycles replace 44 FP32
      // MFMAs of 64 that also block the vector lanes.  A bf16 MFMA holds the vector issue for 8 of its 32 cycles; the
      // vector work that follows GEMM1 is placed in those shadows: GEMM2's accumulator initialisation (b2) under hidden
      // block 0, the shifted softplus of hidden block 0 and the split of GEMM2's first k block under hidden blocks
      // 1..3, and the softplus of k block m + 2 next to the split of k block m + 1 under GEMM2's k block m.
      const unsigned w1b_addr = static_cast<unsigned>(reinterpret_cast<size_t>(
          (__attribute__((address_space(3))) const char*)(reinterpret_cast<const char*>(W1s) + lane * 16)));
#define MP_G1_READ6(dst, ibn)                                                                                              \
    asm volatile("ds_read_b128 %0, %6 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\t"     \
                 "ds_read_b128 %3, %6 offset:%10\n\tds_read_b128 %4, %6 offset:%11\n\tds_read_b128 %5, %6 offset:%12"       \
                 : "=&v"(dst[0][0]), "=&v"(dst[0][1]), "=&v"(dst[0][2]), "=&v"(dst[1][0]), "=&v"(dst[1][1]),               \
                   "=&v"(dst[1][2])                                                                                         \
                 : "v"(w1b_addr), "n"(((0 * 2 + 0) * 4 + (ibn)) * 1024), "n"(((1 * 2 + 0) * 4 + (ibn)) * 1024),            \
                   "n"(((2 * 2 + 0) * 4 + (ibn)) * 1024), "n"(((0 * 2 + 1) * 4 + (ibn)) * 1024),                           \
                   "n"(((1 * 2 + 1) * 4 + (ibn)) * 1024), "n"(((2 * 2 + 1) * 4 + (ibn)) * 1024))
#define MP_G1_WAIT6(dst)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                    \
                 : "+v"(dst[0][0]), "+v"(dst[0][1]), "+v"(dst[0][2]), "+v"(dst[1][0]), "+v"(dst[1][1]), "+v"(dst[1][2]))
      bf16x8 ga[2][3];   // (one wave per SIMD) A pieces of the current hidden block: block 0's fly under the basis split
      if constexpr (PIPE) {
        MP_G1_READ6(ga, 0);
        MP_SLOT_FENCE();
      }
      bf16x8 q[2][3];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sl = 8 * kb + i;
          const float x = sl < NKMAX ? rb[sl < NKMAX ? sl : 0] : 0.0f;
          const __bf16 p0 = static_cast<__bf16>(x);
          const float r1 = x - static_cast<float>(p0);
          const __bf16 p1 = static_cast<__bf16>(r1);
          const float r2 = r1 - static_cast<float>(p1);
          q[kb][0][i] = p0;
          q[kb][1][i] = p1;
          q[kb][2][i] = static_cast<__bf16>(r2);
        }
      // vector task t of hidden blocks 1..3 (36 MFMA slots, 40 tasks): 0..31 = softplus stage (t & 1) of h[0][t >> 1],
      // 32..39 = split stage (t & 1) of value pair (t - 32) >> 1 of k block 0
      float g1_t = 0.0f;
      floatx2 g1_rem = {0.0f, 0.0f};
      auto g1_task = [&](int t) {
        if (t < 32) {
          if (t & 1) ssp_b(0, t >> 1, g1_t);
          else ssp_a(0, t >> 1, g1_t);
        } else if (t & 1) {
          split2_b((t - 32) >> 1, a_mid, a_lo, g1_rem);
        } else {
          split2_a(0, (t - 32) >> 1, a_hi, g1_rem);
        }
      };
      auto g1_slot_tasks = [&](int ib, int sl) {   // sl = 6 kb + product within the hidden block
        if (ib > 0) {
          const int g = 12 * (ib - 1) + sl;
          for (int t = (40 * g) / 36; t < (40 * (g + 1)) / 36; ++t) g1_task(t);
        }
      };
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // (A piece, B piece), smallest product first
      if constexpr (!PIPE) {
        const char* w1b_lane = reinterpret_cast<const char*>(W1s) + lane * 16;  // + ((piece * 2 + kb) * 4 + ib) * 1024
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            bf16x8 ga[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              ga[pc] = *reinterpret_cast<const bf16x8*>(w1b_lane + ((pc * 2 + kb) * 4 + ib) * 1024);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
              h[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[PA[pr]], q[kb][PB[pr]], h[ib], 0, 0, 0);
              g1_slot_tasks(ib, 6 * kb + pr);
            }
          }
          MP_STEP_FENCE();
        }
      } else {
        // one wave per SIMD: the six A pieces of hidden block ib + 1 are requested (volatile asm, program order) ahead
        // of block ib's MFMAs; the wait that publishes them closes the block (cf. GEMM2 below)
        if constexpr (PIPE) {   // the basis pieces exist before the wait (and so before the first MFMA)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) asm volatile("" : "+v"(q[kb][pc]));
        }
        MP_SLOT_FENCE();
        MP_G1_WAIT6(ga);
        MP_SLOT_FENCE();
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          bf16x8 gn[2][3];
          if (ib < 3) {
            MP_G1_READ6(gn, ib + 1);
            MP_SLOT_FENCE();
          }
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
              h[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[kb][PA[pr]], q[kb][PB[pr]], h[ib], 0, 0, 0);
              g1_slot_tasks(ib, 6 * kb + pr);
              MP_SLOT_FENCE();
            }
          if (ib < 3) {
            MP_G1_WAIT6(gn);
            MP_SLOT_FENCE();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) ga[kb][pc] = gn[kb][pc];
          }
        }
#undef MP_G1_READ6
#undef MP_G1_WAIT6
      }
      MP_STAMP(2)
    } else {
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
        for (int r = 0; r < 16; ++r) ssp_val(ib, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) split_into(0, i, a_hi, a_mid, a_lo);
    }

    if constexpr (X_AT == 1) load_sender_rows();
    MP_STAMP(3)
    // ---- GEMM2: w[e][j] = sum_f h[e][f] W2[f][j] + b2[j]; A = the accumulator registers of GEMM1, split into three
    //      bf16 pieces per k block (16 features: 8 registers of each lane half); B = the pre-split images in LDS, one
    //      ds_read_b128 per piece; six v_mfma_f32_32x32x16_bf16 per (k block, column block), smallest products first.
    //      Vector tasks of step (k block m, column block jb), one per MFMA slot: the split of value pair jb of k block
    //      m + 1 (stages A, B) and - bf16 GEMM1 builds - the softplus of the same pair of k block m + 2 (stages A, B
    //      of each value): every value is ready one k block before its split. ----
    float st_t0 = 0.0f, st_t1 = 0.0f;
    floatx2 st_rem = {0.0f, 0.0f};
    auto step_task = [&](int m, int jb, int pr, bf16x8& n_hi, bf16x8& n_mid, bf16x8& n_lo) {
      if constexpr (G1B) {
        const int mm = m + 2, ib2 = (mm >> 1) & 3, r0 = 8 * (mm & 1) + 2 * jb;
        if (pr == 0 && m < 7) split2_a(m + 1, jb, n_hi, st_rem);
        if (pr == 1 && m < 6) ssp_a(ib2, r0, st_t0);
        if (pr == 2 && m < 6) ssp_b(ib2, r0, st_t0);
        if (pr == 3 && m < 7) split2_b(jb, n_mid, n_lo, st_rem);
        if (pr == 4 && m < 6) ssp_a(ib2, r0 + 1, st_t1);
        if (pr == 5 && m < 6) ssp_b(ib2, r0 + 1, st_t1);
      } else {
        if (pr == 5 && m < 7) {
          split_into(m + 1, 2 * jb, n_hi, n_mid, n_lo);
          split_into(m + 1, 2 * jb + 1, n_hi, n_mid, n_lo);
        }
      }
    };
    // one MFMA slot: product (A piece, B piece) - smallest products first - then the slot's vector task
#define MP_G2_SLOT(pr, AP, BP)                                               \
    w[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP, BP, w[jb], 0, 0, 0); \
    step_task(m, jb, pr, n_hi, n_mid, n_lo);                                 \
    MP_SLOT_FENCE()
    if constexpr (!PIPE) {
      // two waves per SIMD: the sibling wave's MFMAs cover this wave's LDS latency, and the 256-register budget has no
      // room for a second set of B pieces - plain reads, scheduled by the compiler
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        bf16x8 n_hi, n_mid, n_lo;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          bf16x8 bq[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            bq[pc] = *reinterpret_cast<const bf16x8*>(w2_lane + (pc * 32 + 4 * m + jb) * 1024);
          MP_G2_SLOT(0, a_lo, bq[0]);
          MP_G2_SLOT(1, a_hi, bq[2]);
          MP_G2_SLOT(2, a_mid, bq[1]);
          MP_G2_SLOT(3, a_mid, bq[0]);
          MP_G2_SLOT(4, a_hi, bq[1]);
          MP_G2_SLOT(5, a_hi, bq[0]);
          MP_STEP_FENCE();
        }
        a_hi = n_hi;
        a_mid = n_mid;
        a_lo = n_lo;
      }
    } else {
      // B pieces are read one (k block, column block) step ahead of the MFMAs that consume them.  Left to itself the
      // compiler (at its register cap) sinks every ds_read_b128 in front of its first MFMA and waits out the LDS latency
      // there, 2-3 times per step: the reads are therefore issued as volatile asm in program order, and the wait that
      // publishes them - tied to the destination registers so that no consumer can move above it - closes the step.
      const unsigned w2_addr = static_cast<unsigned>(reinterpret_cast<size_t>(
          (__attribute__((address_space(3))) const char*)(w2_lane)));
      const unsigned w2_addr_mid = w2_addr + 32 * 1024, w2_addr_lo = w2_addr + 64 * 1024;
#define MP_LDS_B128(dst, addr, t) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"((t) * 1024))
      bf16x8 bq[3];
      MP_LDS_B128(bq[0], w2_addr, 0);
      MP_LDS_B128(bq[1], w2_addr_mid, 0);
      MP_LDS_B128(bq[2], w2_addr_lo, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]));
      MP_SLOT_FENCE();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        bf16x8 n_hi, n_mid, n_lo;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          bf16x8 cq[3];
          if (4 * m + jb + 1 < 32) {
            // (the accumulator operand orders the reads AHEAD of this step's MFMAs; nothing touches it)
            if constexpr (G1B) {   // (the slot fences keep the reads ahead of the MFMAs)
              asm volatile("ds_read_b128 %0, %3 offset:%6\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %5 offset:%6"
                           : "=&v"(cq[0]), "=&v"(cq[1]), "=&v"(cq[2])
                           : "v"(w2_addr), "v"(w2_addr_mid), "v"(w2_addr_lo), "n"((4 * m + jb + 1) * 1024));
              MP_SLOT_FENCE();
            } else {
              asm volatile("ds_read_b128 %0, %3 offset:%6\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %5 offset:%6"
                           : "=&v"(cq[0]), "=&v"(cq[1]), "=&v"(cq[2])
                           : "v"(w2_addr), "v"(w2_addr_mid), "v"(w2_addr_lo), "n"((4 * m + jb + 1) * 1024), "a"(w[jb]));
              asm volatile("" : "+a"(w[jb]));
            }
          }
          MP_G2_SLOT(0, a_lo, bq[0]);
          MP_G2_SLOT(1, a_hi, bq[2]);
          MP_G2_SLOT(2, a_mid, bq[1]);
          MP_G2_SLOT(3, a_mid, bq[0]);
          MP_G2_SLOT(4, a_hi, bq[1]);
          MP_G2_SLOT(5, a_hi, bq[0]);
          if (4 * m + jb + 1 < 32) {
            // the accumulator operand only orders this wait behind the step's MFMAs (nothing reads it)
            if constexpr (G1B) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+a"(w[jb]));
            bq[0] = cq[0];
            bq[1] = cq[1];
            bq[2] = cq[2];
            MP_SLOT_FENCE();
          }
        }
        a_hi = n_hi;
        a_mid = n_mid;
        a_lo = n_lo;
      }
#undef MP_LDS_B128
    }
#undef MP_G2_SLOT
#undef MP_PIN
#undef MP_SLOT_FENCE
#undef MP_STEP_FENCE

    MP_STAMP(4)
    if constexpr (X_AT == 2) load_sender_rows();
    // ---- multiply by the sender row and sum the segments IN REGISTERS.  Each lane half holds 16 consecutive edges
    //      (register order) of the lane's four feature columns.  The segment structure is wave-uniform data (one
    //      32-bit start mask from a ballot), so the walk is driven by SCALAR tests: a step without a boundary in either
    //      half is four multiply-adds; a step with one takes a scalar branch into the general code, where lane masks
    //      built on the scalar unit select which half stores / restarts.  A segment that closes inside a half after
    //      another one opened there is exclusive to this tile: one 16-B store per lane.  The (at most four) segments that
    //      touch a half boundary are resolved afterwards: the low half's open tail moves to the high half if the
    //      segment continues across edge 15|16; tile-interior ones are stored, the tile's first and last segment -
    //      which may continue in the neighbouring tiles - are added with float atomics.  Padding edges of the last
    //      tile repeat the last receiver and carry zeros. -----------------------------------------------------------------
    const int prev_recv = __shfl_up(my_recv, 1, 64);  // all lanes take part; lanes 0 / 32 are masked out below
    const unsigned sm = static_cast<unsigned>(__ballot((c > 0) & (my_recv != prev_recv)) & 0xffffffffull);
    constexpr unsigned long long LO = 0x00000000ffffffffull, HI = 0xffffffff00000000ull;
    float* const out_lane = a.out + 4 * c;
    if (e0 + TE > a.M) {  // the one partial tile: its padding edges carry zeros
      const int rows_valid = static_cast<int>(a.M - e0) - 16 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) xv[jb][r] = r < rows_valid ? xv[jb][r] : 0.0f;
    }
    // features (4c, 4c+1) and (4c+2, 4c+3) as register pairs: v_pk_mul_f32 / v_pk_add_f32 do two columns per instruction
    const floatx2 b01 = {bias2[0], bias2[1]}, b23 = {bias2[2], bias2[3]};   // filter = h W2 + b2, as Dense adds it
    floatx2 acc01 = (floatx2{w[0][0], w[1][0]} + b01) * floatx2{xv[0][0], xv[1][0]};
    floatx2 acc23 = (floatx2{w[2][0], w[3][0]} + b23) * floatx2{xv[2][0], xv[3][0]};
    floatx2 first01 = {0.0f, 0.0f}, first23 = {0.0f, 0.0f};
    unsigned had = 0;  // scalar: bit 0 / 1 = a segment has opened inside the low / high half
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const unsigned o = (sm >> r) & 0x10001u;  // bit 0: edge r opens a segment; bit 16: edge 16 + r does
      const floatx2 m01 = (floatx2{w[0][r], w[1][r]} + b01) * floatx2{xv[0][r], xv[1][r]};
      const floatx2 m23 = (floatx2{w[2][r], w[3][r]} + b23) * floatx2{xv[2][r], xv[3][r]};
      if (o != 0) {
        const bool o_lo = (o & 1u) != 0, o_hi = (o >> 16) != 0;
        const unsigned long long open_mask = (o_lo ? LO : 0ull) | (o_hi ? HI : 0ull);
        const unsigned long long store_mask = ((o_lo && (had & 1u)) ? LO : 0ull) | ((o_hi && (had & 2u)) ? HI : 0ull);
        if (__builtin_amdgcn_inverse_ballot_w64(store_mask)) {
          const int row_lo = __builtin_amdgcn_readlane(my_recv, r - 1);
          const int row_hi = __builtin_amdgcn_readlane(my_recv, 16 + r - 1);
          *reinterpret_cast<float4*>(out_lane + static_cast<int64_t>(hh ? row_hi : row_lo) * F) =
              make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
        }
        if (__builtin_amdgcn_inverse_ballot_w64(open_mask & ~store_mask)) {  // the half's first segment closes: keep it
          first01 = acc01;
          first23 = acc23;
        }
        if (__builtin_amdgcn_inverse_ballot_w64(open_mask)) {  // restart (the add below then yields m)
          acc01 = floatx2{0.0f, 0.0f};
          acc23 = floatx2{0.0f, 0.0f};
        }
        had |= (o_lo ? 1u : 0u) | (o_hi ? 2u : 0u);
      }
      acc01 += m01;
      acc23 += m23;
    }
    float acc[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
    float first[4] = {first01.x, first01.y, first23.x, first23.y};
    MP_STAMP(5)
    // wave-uniform shape of the tile
    const int nlo = __builtin_popcount(sm & 0xfffeu) + 1;   // segments touching the low half
    const int nhi = __builtin_popcount(sm >> 17) + 1;        // segments touching the high half
    const bool cont = ((sm >> 16) & 1u) == 0;                // edge 16 continues the segment of edge 15
    const int node_e0 = __builtin_amdgcn_readlane(my_recv, 0), node_e15 = __builtin_amdgcn_readlane(my_recv, 15);
    const int node_e16 = __builtin_amdgcn_readlane(my_recv, 16), node_e31 = __builtin_amdgcn_readlane(my_recv, 31);
    // Float atomics leave the XCD (device scope) and are paid per 128-B line touched: the lane's four features
    // 4c..4c+3 are first transposed through 512 B of LDS so that each of the four atomic instructions of a half wave
    // covers ONE contiguous line (features 32 jb + c) instead of four (measured: 3.3 us -> <1 us per launch at config 2).
    float* const xs_half = Xs + (wave * 2 + hh) * F;
    auto atomic_row = [&](int node, int which, float v0, float v1, float v2, float v3) {
      if (a.bnd_val != nullptr) {  // deterministic mode: park the partial row, slot `which` (0 = tile's first segment)
        const int64_t slot = 2 * static_cast<int64_t>(tile) + which;
        *reinterpret_cast<float4*>(a.bnd_val + slot * F + 4 * c) = make_float4(v0, v1, v2, v3);
        if (c == 0) a.bnd_node[slot] = node;
        return;
      }
      *reinterpret_cast<float4*>(xs_half + 4 * c) = make_float4(v0, v1, v2, v3);
      float* dst = a.out + static_cast<int64_t>(node) * F + c;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) atomicAdd(dst + jb * 32, xs_half[jb * 32 + c]);
    };
    auto store_row = [&](int node, float v0, float v1, float v2, float v3) {
      *reinterpret_cast<float4*>(out_lane + static_cast<int64_t>(node) * F) = make_float4(v0, v1, v2, v3);
    };
    float tail[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (cont) {  // hand the low half's open tail to the high half (lanes c + 32)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) tail[jb] = __shfl_xor(acc[jb], 32, 64);
    }
    MP_STAMP(6)
    if (hh == 0) {
      if (nlo > 1) atomic_row(node_e0, 0, first[0], first[1], first[2], first[3]);  // first segment closed in the low half
      if (!cont) {    // the low half's last segment ends at edge 15
        if (nlo == 1) atomic_row(node_e15, 0, acc[0], acc[1], acc[2], acc[3]);      // it is also the tile's first
        else store_row(node_e15, acc[0], acc[1], acc[2], acc[3]);
      }
    } else {
      // the high half's first segment (the whole half if nhi == 1), plus the low half's tail when it continues
      float v0[4];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const float mine = nhi > 1 ? first[jb] : acc[jb];
        v0[jb] = cont ? tail[jb] + mine : mine;   // edge order: low-half part first
      }
      // tile's first segment (it reaches back to edge 0) or its last one (it reaches edge 31) - or the whole tile
      if ((cont && nlo == 1) || nhi == 1) atomic_row(node_e16, (cont && nlo == 1) ? 0 : 1, v0[0], v0[1], v0[2], v0[3]);
      else store_row(node_e16, v0[0], v0[1], v0[2], v0[3]);
      if (nhi > 1) atomic_row(node_e31, 1, acc[0], acc[1], acc[2], acc[3]);                  // the tile's last segment
    }
    MP_STAMP(7)
  }
  // a wave without tiles still owes the workgroup its barrier (and must not exit under its DMA)
  if (!w_ready) publish(Everything{});
  if constexpr (DIAG) {
    if (lane == 0 && a.diag) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(a.diag + i, diag_sum[i]);
    }
  }
}

constexpr int PACKED_FLOATS = MAX_KROWS * F + W2_IMG_FLOATS + F + W1B_FLOATS;

// LDS image of the filter-MLP weights: row k of W1 (k < B), the bias b1 as row B, zero rows up to MAX_KROWS, each row
// stored [4*c + blk] = W1[k][blk*32 + c]; then the three bf16 operand images of W2 (see W2_PIECE_FLOATS; lane c of GEMM2
// owns output features 4c .. 4c+3, one per accumulator block) and b2 in natural order.
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
    } else if (i < MAX_KROWS * F + W2_IMG_FLOATS) {
      // one float slot = two consecutive bf16 elements (2 q, 2 q + 1) of an entry
      const int j = i - MAX_KROWS * F;
      const int piece = j / W2_PIECE_FLOATS, t = j % W2_PIECE_FLOATS;
      const int q = t & 3, ln = (t >> 2) & 63, jb = (t >> 8) & 3, m = t >> 10;
      const int cc = ln & 31, hh = ln >> 5;
      unsigned bits[2];
      for (int e = 0; e < 2; ++e) {
        const int ii = 2 * q + e;
        const int f = 32 * (m >> 1) + rowmap(8 * (m & 1) + ii, hh);
        const float x = W2[f * F + 4 * cc + jb];
        const __bf16 p0 = static_cast<__bf16>(x);
        const float r1 = x - static_cast<float>(p0);
        const __bf16 p1 = static_cast<__bf16>(r1);
        const float r2 = r1 - static_cast<float>(p1);
        const __bf16 p2 = static_cast<__bf16>(r2);
        const __bf16 pick = piece == 0 ? p0 : (piece == 1 ? p1 : p2);
        bits[e] = static_cast<unsigned>(__builtin_bit_cast(unsigned short, pick));
      }
      v = __uint_as_float(bits[0] | (bits[1] << 16));
    } else if (i < MAX_KROWS * F + W2_IMG_FLOATS + F) {
      v = b2 ? b2[i - MAX_KROWS * F - W2_IMG_FLOATS] : 0.0f;
    } else {
      // bf16 pieces of W1 | b1 for GEMM1 on the bf16 pipe (see W1B_FLOATS); zeros when the basis needs a third k block
      const int t = i - (MAX_KROWS * F + W2_IMG_FLOATS + F);
      const int q = t & 3, ln = (t >> 2) & 63, e = t >> 8;
      const int ib = e & 3, kb = (e >> 2) & 1, piece = e >> 3;
      const int cc = ln & 31, hh = ln >> 5;
      const int nk = (B + 2) >> 1;
      unsigned bits[2] = {0u, 0u};
      if (nk <= G1B_MAX_NK) {
        for (int e2 = 0; e2 < 2; ++e2) {
          const int sl = 8 * kb + 2 * q + e2;
          const int k = sl < nk ? sl + nk * hh : MAX_KROWS;
          float x = 0.0f;
          if (k < B) x = W1[k * F + 32 * ib + cc];
          else if (k == B && b1) x = b1[32 * ib + cc];
          const __bf16 p0 = static_cast<__bf16>(x);
          const float r1 = x - static_cast<float>(p0);
          const __bf16 p1 = static_cast<__bf16>(r1);
          const float r2 = r1 - static_cast<float>(p1);
          const __bf16 p2 = static_cast<__bf16>(r2);
          const __bf16 pick = piece == 0 ? p0 : (piece == 1 ? p1 : p2);
          bits[e2] = static_cast<unsigned>(__builtin_bit_cast(unsigned short, pick));
        }
      }
      v = __uint_as_float(bits[0] | (bits[1] << 16));
    }
    packed[i] = v;
  }
}

// Deterministic mode, second pass.  Slot 2t holds tile t's first-segment partial (always present: the first segment
// touches the tile start), slot 2t+1 its last-segment partial if the tile has more than one segment.  Receivers are
// sorted, so the slots that belong to one receiver are consecutive; the first of them (the leader) sums the run in slot
// order - i.e. in edge order - and stores the row: nothing else writes that receiver (a receiver gets either plain
// stores, when its segment is interior to one tile, or boundary partials, never both), so the result does not depend on
// scheduling.  One 32-lane group per slot, lane = four features.
__global__ void cfconv_boundary_kernel(const float* __restrict__ bnd_val, const int32_t* __restrict__ bnd_node,
                                       int64_t nslots, float* __restrict__ out) {
  const int64_t group = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  const int64_t ngroups = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 5;
  const int c = threadIdx.x & 31;
  for (int64_t i = group; i < nslots; i += ngroups) {
    const int node = bnd_node[i];
    if (node < 0) continue;
    if (i > 0) {  // previous occupied slot: i-1 if occupied, else i-2 (even slots are always occupied)
      int prev = bnd_node[i - 1];
      if (prev < 0 && i > 1) prev = bnd_node[i - 2];
      if (prev == node) continue;  // not the leader of its run
    }
    float4 sum = reinterpret_cast<const float4*>(bnd_val + i * F)[c];
    for (int64_t j = i + 1; j < nslots; ++j) {
      const int nj = bnd_node[j];
      if (nj < 0) continue;
      if (nj != node) break;
      const float4 v = reinterpret_cast<const float4*>(bnd_val + j * F)[c];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    reinterpret_cast<float4*>(out + static_cast<int64_t>(node) * F)[c] = sum;
  }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and per kernel: remember, under a mutex, on which
// devices each instantiation has been opted in (a process may drive several GPUs, from several threads).
inline int ensure_dynamic_lds(const void* kernel, size_t lds, unsigned long long* done_mask, std::mutex* mu) {
  int dev = 0;
  MP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(*mu);
  if (dev < 64 && ((*done_mask >> dev) & 1ull)) return MP_OK;
  MP_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  if (dev < 64) *done_mask |= 1ull << dev;
  return MP_OK;
}

template <int WAVES, int NKT>
size_t cfconv_lds_bytes() {
  const int w1 = (NKT > 0 && NKT <= G1B_MAX_NK) ? W1B_FLOATS : (NKT > 0 ? 2 * NKT : MAX_KROWS) * F;
  return sizeof(float) * (w1 + W2_IMG_FLOATS + WAVES * 2 * F);
}

template <int WAVES, bool GAUSS, bool FAST, int NKT, bool DIAG, bool COMPACT = false>
int launch_cfconv(const CfconvArgs& args, int grid, hipStream_t s) {
  const size_t lds = cfconv_lds_bytes<WAVES, NKT>();
  static std::mutex mu;                    // one per instantiation, like the mask: guarded, per device
  static unsigned long long done_mask = 0;
  const int rc = ensure_dynamic_lds(
      reinterpret_cast<const void*>(&cfconv_fused_kernel<WAVES, GAUSS, FAST, NKT, DIAG, COMPACT>), lds, &done_mask, &mu);
  if (rc != MP_OK) return rc;
  cfconv_fused_kernel<WAVES, GAUSS, FAST, NKT, DIAG, COMPACT><<<grid, WAVES * 64, lds, s>>>(args);
  return mp::check_launch("mp_cfconv_fused_f32");
}

template <bool GAUSS, bool FAST>
int launch_by_basis(const CfconvArgs& args, int waves, int grid, bool compact, hipStream_t s) {
  (void)compact;   // flag bit 4 selects the 8-wave build in cfconv_dispatch (two waves per SIMD on ONE LDS image)
  // basis sizes with a compile-time k count run GEMM1 on the bf16 pipe: 20 (SchNet default) and 25 (the fork's
  // force_schnet.py configuration); everything else takes the generic FP32 GEMM1
  // the 8-wave build (256 registers per wave) exists for the fast-softplus, fixed-basis builds only: with the exact
  // softplus or the generic FP32 GEMM1 it spilled (2-27 registers, 12-112 B of scratch per lane: a scratch segment is paid
  // for at every launch) - cfconv_dispatch gives those the 4-wave build
  if constexpr (FAST) {
    if (waves == 8 && args.B == 20) return launch_cfconv<8, GAUSS, FAST, 11, false>(args, grid, s);
    if (waves == 8 && args.B == 25) return launch_cfconv<8, GAUSS, FAST, 13, false>(args, grid, s);
  }
  if (args.B == 20) return launch_cfconv<4, GAUSS, FAST, 11, false>(args, grid, s);
  if (args.B == 25) return launch_cfconv<4, GAUSS, FAST, 13, false>(args, grid, s);
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
  // One workgroup per CU (the LDS holds the weights), persistent over the tiles.  FP32 MFMA and FP32 VALU share the
  // SIMD's lanes, so a single wave per SIMD leaves the matrix pipe idle during its own vector phases (softplus,
  // segment walk); a second wave per SIMD fills those gaps once the waves run out of step, i.e. when every wave has
  // several tiles: 8 waves (256 registers each, sender rows loaded late) from 4096 tiles on - measured 812 vs 928 us at
  // 2.5 M edges, 87 vs 93 us at 0.2 M - and 4 waves (one tile per wave spread over more CUs) below.  Flag bit 2 forces
  // 8 waves, bit 3 forces 4, bit 4 selects the 256-register 4-wave build (Gauss variant, 20 bins).
  // Between the regimes the launch is a whole number of ROUNDS over the resident waves (256 workgroups: 1024 waves of the
  // 4-wave build, 2048 of the 8-wave one), and a round of the 8-wave build - two waves sharing a SIMD - costs 1.85 rounds
  // of the 4-wave build (4093 tiles, a launch group of five config-2 batches: 4 rounds x 8.35 us against 2 x 15.45 us;
  // 3274 tiles: 30.5 against 29.1 us; 2456 tiles: 24.3 against 26.5 us; 4910 tiles: 8 waves 7 % slower).  From 3072 tiles
  // on the build with the smaller rounds x cost product runs; far above, the 8-wave build's 0.925 wins.
  int waves = 4;
  if (args.ntiles >= 3072) {
    const int r4 = (args.ntiles + 1023) / 1024, r8 = (args.ntiles + 2047) / 2048;
    waves = (37 * r8 < 20 * r4) ? 8 : 4;
  }
  if (flags & 4) waves = 8;
  if (flags & 8) waves = 4;
  if (!(fast && (args.B == 20 || args.B == 25))) waves = 4;   // no spill-free 8-wave build for these (launch_by_basis)
  // Flag bit 4 ("several forwards in flight"): every workgroup claims a whole CU (all its registers, 107-120 KB of LDS), so
  // forwards of different batches share the GPU only CU by CU; with this flag a launch uses half as many workgroups, two
  // tiles per wave - the per-workgroup fixed cost (launch ramp, 107 KB of weight staging) is paid once per eight tiles
  // and the other half of the CUs is free for the other batches' kernels.  Costs a lone launch latency; not the default.
  const bool compact = (flags & 16) != 0;
  int grid = (args.ntiles + waves - 1) / waves;
  if (compact && grid > 1) grid = (grid + 1) / 2;
  if (grid > 256) grid = 256;
  if (args.diag) {
    MP_REQUIRE(gauss && args.B == 20, "mp_cfconv: the diagnostic build exists for the 20-bin Gauss variant only");
    grid = (args.ntiles + 3) / 4 > 256 ? 256 : (args.ntiles + 3) / 4;
    return launch_cfconv<4, true, true, 11, true>(args, grid, s);
  }
  if (args.bnd_val != nullptr)  // every slot starts empty; the main kernel fills the occupied ones
    MP_HIP(hipMemsetAsync(args.bnd_node, 0xff, sizeof(int32_t) * 2 * static_cast<size_t>(args.ntiles), s));
  int rc;
  if (gauss) {
    rc = fast ? launch_by_basis<true, true>(args, waves, grid, compact, s)
              : launch_by_basis<true, false>(args, waves, grid, compact, s);
  } else {
    rc = fast ? launch_by_basis<false, true>(args, waves, grid, compact, s)
              : launch_by_basis<false, false>(args, waves, grid, compact, s);
  }
  if (rc != MP_OK || args.bnd_val == nullptr) return rc;
  const int64_t nslots = 2 * static_cast<int64_t>(args.ntiles);
  cfconv_boundary_kernel<<<mp::grid_for(nslots * 32), 256, 0, s>>>(args.bnd_val, args.bnd_node, nslots, args.out);
  return mp::check_launch("mp_cfconv (deterministic boundary pass)");
}

inline size_t bnd_val_bytes(int64_t M) { return sizeof(float) * 2 * F * static_cast<size_t>((M + TE - 1) / TE); }
inline size_t det_ws_bytes(int64_t M) {
  return ((bnd_val_bytes(M) + 255) & ~static_cast<size_t>(255)) + sizeof(int32_t) * 2 * static_cast<size_t>((M + TE - 1) / TE);
}

// flags bit 5 (deterministic): carve the boundary buffers out of the caller's workspace
int attach_det_workspace(CfconvArgs* a, int flags, void* ws, size_t ws_bytes, const char* who) {
  if (!(flags & 32) || a->M <= 0) return MP_OK;
  MP_REQUIRE(ws != nullptr && ws_bytes >= det_ws_bytes(a->M), "%s: deterministic mode needs %zu workspace bytes (got %zu)",
             who, det_ws_bytes(a->M), ws_bytes);
  a->bnd_val = static_cast<float*>(ws);
  a->bnd_node = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + ((bnd_val_bytes(a->M) + 255) & ~static_cast<size_t>(255)));
  return MP_OK;
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

int mp_cfconv_det_workspace_bytes(int64_t M, size_t* bytes_out_host) {
  MP_REQUIRE(M >= 0 && bytes_out_host, "mp_cfconv_det_workspace_bytes: bad arguments");
  *bytes_out_host = det_ws_bytes(M);
  return MP_OK;
}

int mp_cfconv_fused_ws_f32(const float* x, int64_t N, const float* rbf, int B, const float* packed,
                           const int32_t* recv_sorted, const int32_t* send, const int32_t* perm, int64_t M, int flags,
                           float* out_zeroed, void* ws, size_t ws_bytes, mpStream_t stream) {
  CfconvArgs a{};
  a.x = x; a.edge_in = rbf; a.packed = packed;
  a.recv = recv_sorted; a.send = send; a.perm = perm; a.out = out_zeroed;
  a.M = M; a.N = N; a.B = B;
  const int rc = attach_det_workspace(&a, flags, ws, ws_bytes, "mp_cfconv_fused_ws_f32");
  if (rc != MP_OK) return rc;
  return cfconv_dispatch(a, false, flags, mp::as_stream(stream));
}

int mp_cfconv_gauss_fused_ws_f32(const float* x, int64_t N, const float* dist, int bins, float distance, float sigma,
                                 float offset, const float* packed, const int32_t* recv_sorted, const int32_t* send,
                                 const int32_t* perm, int64_t M, int flags, float* out_zeroed, void* ws, size_t ws_bytes,
                                 mpStream_t stream) {
  MP_REQUIRE(sigma != 0.0f, "mp_cfconv_gauss_fused_ws_f32: sigma must be non-zero");
  CfconvArgs a{};
  a.x = x; a.edge_in = dist; a.packed = packed;
  a.recv = recv_sorted; a.send = send; a.perm = perm; a.out = out_zeroed;
  a.M = M; a.N = N; a.B = bins;
  a.g_distance = distance;
  a.g_gamma = static_cast<float>(1.0 / static_cast<double>(sigma) / static_cast<double>(sigma) / 2.0);
  a.g_offset = offset;
  const int rc = attach_det_workspace(&a, flags, ws, ws_bytes, "mp_cfconv_gauss_fused_ws_f32");
  if (rc != MP_OK) return rc;
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
