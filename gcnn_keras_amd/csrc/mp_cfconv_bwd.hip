// Reverse pass of SchNetCFconv w.r.t. the edge distance (forces: kgcnn/model/force.py:159-177 differentiates the energy
// through kgcnn/layers/conv/schnet_conv.py:73-79 and kgcnn/layers/geom.py:567-571).
//
//   forward   out_i = sum_{e: recv(e)=i} x[send(e)] * w_e ,   w_e = ssp(g(d_e) W1 + b1) W2 + b2 ,  g = Gauss basis
//
// Given g_out = dE/d out (N,F) the reverse pass needs two things:
//   dE/dx_j   = sum_{e: send(e)=j} g_out[recv(e)] * w_e     - the SAME kernel as the forward with the roles of the two
//               index columns swapped (mp_cfconv_gauss_fused_f32 over the sender-sorted list): no new code;
//   dE/dd_e   = sum_h [ (v_e W2^T)_h * ssp'(pre1_e)_h * (g'(d_e) W1)_h ] ,  v_e = g_out[recv(e)] * x[send(e)]
//               - this kernel: per 32-edge tile three MFMA chains in the transposed form of the forward kernel (edge on
//               the lane, feature in the accumulator register):  P = pre1^T = W1p^T g(d),  Z = W1^T g'(d),
//               GH = W2 v^T (128 x 128 x 32, the same 256 MFMAs as the forward's GEMM2), then an in-register
//               contraction over the 128 hidden features and one cross-half shuffle.
//
// All three chains run on the bf16 matrix pipe as the exact FP32 emulation of the forward kernel (csrc/mp_cfconv.hip: three
// bf16 pieces per operand, six products, FP32 accumulate).  P and Z (K = B + 1 <= 32 slots; larger bases keep
// v_mfma_f32_32x32x2_f32): W1 | b1 pre-split in the image (the forward's layout), the basis values and their derivatives
// split in registers; GH (K = 128): W2's pieces pre-split in the image (A operand, one ds_read_b128 per piece), v built in
// REGISTERS - lane (edge e, k half) loads the 8 features of g_out[recv(e)] and x[send(e)] that are its k slots of the
// current k block, multiplies and splits them.  As in the forward kernel the vector work is cut into tasks of <= 24 issue
// cycles, pinned to MFMA slots (a bf16 MFMA leaves 24 of its 32 cycles to the vector unit): the splits of v and the
// sigmoid of P ride under GH's MFMAs.  LDS per workgroup: W1 as bf16 pieces (24 KB) and as FP32 rows (17 KB, large
// bases) + the three W2 images = 137 KB, staged by LDS-DMA.
#include <mutex>

#include "mp_common.h"

namespace {

using floatx16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using floatx2 = __attribute__((ext_vector_type(2))) float;

constexpr int F = 128;
constexpr int TE = 32;
constexpr int MAX_KROWS = 34;   // W1 rows in LDS: B inputs + 1 bias row, padded to even (B <= 32)
constexpr int WAVES = 4;
// W2 as A operand of v_mfma_f32_32x32x16_bf16: 16 B per (piece, k block kb, row block ib, lane): element i =
// piece(W2[32 ib + (lane & 31)][16 kb + 8 (lane >> 5) + i])
constexpr int W2_PIECE_FLOATS = F * F / 2;
constexpr int W2_IMG_FLOATS = 3 * W2_PIECE_FLOATS;
constexpr int W1B_FLOATS = 3 * 2 * 4 * 64 * 4;   // bf16 pieces of W1 | b1, the layout of csrc/mp_cfconv.hip (24 KB)
constexpr int G1B_MAX_NK = 16;
constexpr int PACKED_BWD_FLOATS = MAX_KROWS * F + W2_IMG_FLOATS + W1B_FLOATS;

struct CfconvBwdArgs {
  const float* x;        // (N, F) sender-side node features of the block (forward input)
  const float* g_out;    // (N, F) dE/d out
  const float* dist;     // (M)
  const float* packed;   // mp_cfconv_bwd_pack_f32 image
  const int32_t* recv;   // (M) original edge order
  const int32_t* send;   // (M)
  float* g_d;            // (M) dE/dd: written or added to
  int accumulate;
  int64_t M, N;
  int B;
  float g_distance, g_gamma, g_offset;
  int ntiles;
};

#define MP_PIN(x) asm volatile("" : "+v"(x))
#define MP_FENCE() __builtin_amdgcn_sched_barrier(0)

// G1B: the basis fits two k blocks of 16 slots (nk = (B + 2) / 2 <= 16): P and Z on the bf16 pipe
template <bool G1B>
__global__ __launch_bounds__(WAVES * 64, 1) void cfconv_dist_grad_kernel(CfconvBwdArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* W1s = lds;                        // [MAX_KROWS][F] packed like the forward (FP32 rows)
  float* W2s = lds + MAX_KROWS * F;        // three bf16 operand images of W2
  float* W1Bs = W2s + W2_IMG_FLOATS;       // bf16 pieces of W1 | b1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31;
  const int hh = lane >> 5;
  const int B = a.B;
  const int nk = (B + 2) >> 1;             // k pairs of the basis GEMMs (B inputs + bias row, padded to even)
  // the 137 KB image by LDS-DMA (1 KB per wave instruction, no VGPRs, all requests in flight at once; the first version
  // copied it through registers in a rolled load -> wait -> ds_write loop: 28 serial round trips to L2, a third of the
  // kernel's time at 64 graphs)
  {
    constexpr int CHUNKS = PACKED_BWD_FLOATS / 256;
    static_assert(PACKED_BWD_FLOATS % 256 == 0, "the image is staged in 1-KB chunks");
#pragma unroll
    for (int i = 0; i < (CHUNKS + WAVES - 1) / WAVES; ++i) {
      const int chunk = i * WAVES + wave;
      if (chunk < CHUNKS)
        __builtin_amdgcn_global_load_lds(a.packed + chunk * 256 + lane * 4,
                                         (__attribute__((address_space(3))) void*)(lds + chunk * 256), 16, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's chunks have landed
  __syncthreads();
  const float* w1_lane = W1s + (nk * hh) * F + 4 * c;   // + s*F : rows s (low half) / nk + s (high half)
  const unsigned w2_addr = static_cast<unsigned>(reinterpret_cast<size_t>(
      (__attribute__((address_space(3))) const char*)(reinterpret_cast<const char*>(W2s) + lane * 16)));
  const unsigned w2_addr_mid = w2_addr + 32 * 1024, w2_addr_lo = w2_addr + 64 * 1024;
  const unsigned w1b_addr = static_cast<unsigned>(reinterpret_cast<size_t>(
      (__attribute__((address_space(3))) const char*)(reinterpret_cast<const char*>(W1Bs) + lane * 16)));
  const float fbins = static_cast<float>(B);
  constexpr float LOG2E = 1.4426950408889634f;

  auto pack2 = [](floatx2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); };
  auto widen2 = [](unsigned u) { return floatx2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; };
  auto set_dword = [](bf16x8& v, int j, unsigned u) {
    uint4 t = __builtin_bit_cast(uint4, v);
    (j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w) = u;
    v = __builtin_bit_cast(bf16x8, t);
  };
  // exact split of a value pair into hi + mid + lo (bf16 each): dword j of the three piece vectors
  auto split_pair = [&](floatx2 x, int j, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const unsigned u0 = pack2(x);
    const floatx2 r1 = x - widen2(u0);
    const unsigned u1 = pack2(r1);
    const unsigned u2 = pack2(r1 - widen2(u1));
    set_dword(hi, j, u0);
    set_dword(mid, j, u1);
    set_dword(lo, j, u2);
  };

  // XCD-aware block order (mp_common.h): consecutive edge tiles share one XCD's L2
  for (int tile = static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x)) * WAVES + wave; tile < a.ntiles;
       tile += gridDim.x * WAVES) {
    const int64_t e0 = static_cast<int64_t>(tile) * TE;
    const int64_t e_mine = e0 + c;                      // lane c (both halves) <-> edge c of the tile
    const bool valid = e_mine < a.M;
    const int64_t ec = valid ? e_mine : a.M - 1;        // (padding lanes repeat the last edge; their column is never stored)
    const int my_recv = a.recv[ec];
    const int my_send = a.send[ec];
    const float d = a.dist[ec];

    const int i_node = my_recv < 0 ? 0 : (my_recv >= a.N ? static_cast<int>(a.N) - 1 : my_recv);
    const int j_node = my_send < 0 ? 0 : (my_send >= a.N ? static_cast<int>(a.N) - 1 : my_send);
    // this lane's k slots of k block kb: features 16 kb + 8 hh + (0..7) of its edge's two rows (32 B each)
    const float4* g_row = reinterpret_cast<const float4*>(a.g_out + static_cast<int64_t>(i_node) * F + 8 * hh);
    const float4* x_row = reinterpret_cast<const float4*>(a.x + static_cast<int64_t>(j_node) * F + 8 * hh);
    float4 gq[2][2], xq[2][2];           // [buffer = k block & 1][first / second four features]
    auto request_rows = [&](int kb) {
      gq[kb & 1][0] = g_row[4 * kb]; gq[kb & 1][1] = g_row[4 * kb + 1];
      xq[kb & 1][0] = x_row[4 * kb]; xq[kb & 1][1] = x_row[4 * kb + 1];
    };
    request_rows(0);
    request_rows(1);

    // v = g_out[recv] * x[send] of k block kb, value pair j, in three stages of <= 24 issue cycles (MFMA-slot tasks)
    bf16x8 nb_hi, nb_mid, nb_lo;          // B pieces of the NEXT k block of GH, under construction
    floatx2 vs_rem = {0.0f, 0.0f};
    auto vsplit_stage = [&](int kb, int j, int stage) {
      if (stage == 0) {
        const float4 g4 = gq[kb & 1][j >> 1], x4 = xq[kb & 1][j >> 1];
        const floatx2 v = (j & 1) ? floatx2{g4.z * x4.z, g4.w * x4.w} : floatx2{g4.x * x4.x, g4.y * x4.y};
        unsigned u = pack2(v);
        vs_rem = v - widen2(u);
        MP_PIN(u);
        MP_PIN(vs_rem);
        set_dword(nb_hi, j, u);
      } else if (stage == 1) {
        unsigned u = pack2(vs_rem);
        vs_rem = vs_rem - widen2(u);
        MP_PIN(u);
        MP_PIN(vs_rem);
        set_dword(nb_mid, j, u);
      } else {
        unsigned u = pack2(vs_rem);
        MP_PIN(u);
        set_dword(nb_lo, j, u);
      }
    };

    // ---- P' = log2(e) pre1^T (with bias row) and Z = (g'(d) W1)^T: B operands are this lane's half of its edge's basis
    //      row (scaled by log2 e: the sigmoid below is then 1 / (1 + 2^-P')) and of its derivative ----
    floatx16 P[4], Z[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        P[ib][r] = 0.0f;
        Z[ib][r] = 0.0f;
      }
    if constexpr (G1B) {
      bf16x8 qb[2][3], qd[2][3];          // [k block][piece] of the basis values / their derivatives (slot s = 8 kb + i)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          floatx2 rb2, rd2;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int sl = 8 * kb + 2 * j + h2;
            const int k = sl + nk * hh;
            const float mu = static_cast<float>(k) / fbins * a.g_distance;
            const float v = (d - a.g_offset) - mu;
            const float gv = __builtin_amdgcn_exp2f((v * v) * (a.g_gamma * -LOG2E));   // v_exp_f32
            const bool in = sl < nk;
            const float rb = (in && k < B) ? gv * LOG2E : ((in && k == B) ? LOG2E : 0.0f);
            const float rd = (in && k < B) ? gv * (-2.0f * a.g_gamma * v) : 0.0f;      // d/dd exp(-gamma v^2)
            if (h2) { rb2.y = rb; rd2.y = rd; } else { rb2.x = rb; rd2.x = rd; }
          }
          split_pair(rb2, j, qb[kb][0], qb[kb][1], qb[kb][2]);
          split_pair(rd2, j, qd[kb][0], qd[kb][1], qd[kb][2]);
        }
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
      // [k block][piece] of W1 (A operand: row = hidden feature 32 ib + c), hidden block ib + 1 requested ahead of ib's MFMAs
      bf16x8 ga[2][3];
      MP_G1_READ6(ga, 0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          MP_PIN(qb[kb][pc]);
          MP_PIN(qd[kb][pc]);
        }
      MP_FENCE();
      MP_G1_WAIT6(ga);
      MP_FENCE();
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // (A piece, B piece), smallest product first
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        bf16x8 gn[2][3];
        if (ib < 3) {
          MP_G1_READ6(gn, ib + 1);
          MP_FENCE();
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) {
            P[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[kb][PA[pr]], qb[kb][PB[pr]], P[ib], 0, 0, 0);
            Z[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[kb][PA[pr]], qd[kb][PB[pr]], Z[ib], 0, 0, 0);
            // the last hidden block's 12 slot pairs build GH's first B pieces (rows requested at tile start)
            if (ib == 3) vsplit_stage(0, (6 * kb + pr) / 3, (6 * kb + pr) % 3);
            MP_FENCE();
          }
        if (ib < 3) {
          MP_G1_WAIT6(gn);
          MP_FENCE();
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) ga[kb][pc] = gn[kb][pc];
        }
      }
#undef MP_G1_READ6
#undef MP_G1_WAIT6
    } else {
      for (int s = 0; s < nk; ++s) {
        const int k = s + nk * hh;
        const float mu = static_cast<float>(k) / fbins * a.g_distance;
        const float v = (d - a.g_offset) - mu;
        const float gv = __builtin_amdgcn_exp2f((v * v) * (a.g_gamma * -LOG2E));   // v_exp_f32
        const float rb = k < B ? gv * LOG2E : (k == B ? LOG2E : 0.0f);
        const float rbd = k < B ? gv * (-2.0f * a.g_gamma * v) : 0.0f;             // d/dd exp(-gamma v^2)
        const float4 wv = *reinterpret_cast<const float4*>(w1_lane + s * F);
        P[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, rb, P[0], 0, 0, 0);
        P[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, rb, P[1], 0, 0, 0);
        P[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, rb, P[2], 0, 0, 0);
        P[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, rb, P[3], 0, 0, 0);
        Z[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, rbd, Z[0], 0, 0, 0);
        Z[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, rbd, Z[1], 0, 0, 0);
        Z[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, rbd, Z[2], 0, 0, 0);
        Z[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, rbd, Z[3], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 12; ++t) vsplit_stage(0, t / 3, t % 3);
    }

    // ---- GH[h][e] = sum_j W2[h][j] v[e][j] on the bf16 pipe: A = W2 pieces from LDS (volatile-asm reads one step ahead),
    //      B = this lane's eight v values of the k block.  Slot tasks of k block kb (24 slots): the three split stages of
    //      the four value pairs of k block kb + 1 (rows requested one k block earlier), then eight values of
    //      q = sigmoid(pre1) * z = Z / (1 + 2^-P'), kept in P - 24 issue cycles each (v_exp_f32, add, v_rcp_f32, mul) ----
    floatx16 GH[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) GH[ib][r] = 0.0f;
    bf16x8 a_hi, a_mid, a_lo;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a_hi), "=&v"(a_mid), "=&v"(a_lo) : "v"(w2_addr), "v"(w2_addr_mid), "v"(w2_addr_lo));
    MP_FENCE();
    auto gh_task = [&](int kb, int sl) {
      if (sl < 12) {
        if (kb + 1 < 8) vsplit_stage(kb + 1, sl / 3, sl % 3);
      } else if (sl < 20) {
        const int idx = 8 * kb + (sl - 12), ib = idx >> 4, r = idx & 15;
        float q = Z[ib][r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-P[ib][r]));
        MP_PIN(q);
        P[ib][r] = q;
      }
    };
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const bf16x8 b_hi = nb_hi, b_mid = nb_mid, b_lo = nb_lo;   // this k block's B pieces (built one block earlier)
      if (kb + 2 < 8) request_rows(kb + 2);                      // (buffer kb & 1: its rows were consumed one block ago)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        bf16x8 n_hi, n_mid, n_lo;
        if (4 * kb + ib + 1 < 32) {
          asm volatile("ds_read_b128 %0, %3 offset:%6\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %5 offset:%6"
                       : "=&v"(n_hi), "=&v"(n_mid), "=&v"(n_lo)
                       : "v"(w2_addr), "v"(w2_addr_mid), "v"(w2_addr_lo), "n"((4 * kb + ib + 1) * 1024));
          MP_FENCE();
        }
#define MP_GH_SLOT(pr, AP, BP)                                                   \
    GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP, BP, GH[ib], 0, 0, 0);   \
    gh_task(kb, 6 * ib + pr);                                                    \
    MP_FENCE()
        MP_GH_SLOT(0, a_lo, b_hi);
        MP_GH_SLOT(1, a_hi, b_lo);
        MP_GH_SLOT(2, a_mid, b_mid);
        MP_GH_SLOT(3, a_mid, b_hi);
        MP_GH_SLOT(4, a_hi, b_mid);
        MP_GH_SLOT(5, a_hi, b_hi);
#undef MP_GH_SLOT
        if (4 * kb + ib + 1 < 32) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n_hi), "+v"(n_mid), "+v"(n_lo));
          a_hi = n_hi;
          a_mid = n_mid;
          a_lo = n_lo;
          MP_FENCE();
        }
      }
    }
    // ---- contraction over the 128 hidden features: 64 in this lane's registers, 64 in the other half's ----------------
    float part = 0.0f;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) part += GH[ib][r] * P[ib][r];
    part += __shfl_xor(part, 32, 64);
    if (hh == 0 && valid) {
      if (a.accumulate) a.g_d[e_mine] += part;
      else a.g_d[e_mine] = part;
    }
  }
}
#undef MP_PIN
#undef MP_FENCE

// Image for the kernel above: W1 rows exactly as mp_cfconv_pack_f32 stores them (row k at [4c + blk] = W1[k][32 blk + c],
// bias as row B), then the three bf16 operand images of W2 (see W2_PIECE_FLOATS).
__global__ void cfconv_bwd_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int B,
                                       const float* __restrict__ W2, float* __restrict__ packed) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PACKED_BWD_FLOATS; i += stride) {
    float v = 0.0f;
    if (i < MAX_KROWS * F) {
      const int k = i / F, col = (i % F) / 4 + 32 * (i % 4);
      if (k < B) v = W1[k * F + col];
      else if (k == B && b1) v = b1[col];
    } else if (i >= MAX_KROWS * F + W2_IMG_FLOATS) {
      // bf16 pieces of W1 | b1 in the layout of csrc/mp_cfconv.hip (W1B_FLOATS): 16 B per (piece, k block kb, hidden
      // block ib, lane): element i2 = piece(W1ext[k][32 ib + c]), slot s = 8 kb + i2, k = s + nk hh for s < nk
      const int t = i - (MAX_KROWS * F + W2_IMG_FLOATS);
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
    } else {   // one float slot = two consecutive bf16 elements (2 q, 2 q + 1) of an entry
      const int j = i - MAX_KROWS * F;
      const int piece = j / W2_PIECE_FLOATS, t = j % W2_PIECE_FLOATS;
      const int q = t & 3, ln = (t >> 2) & 63, ib = (t >> 8) & 3, kb = t >> 10;
      const int hl = ln & 31, kh = ln >> 5;
      unsigned bits[2];
      for (int e = 0; e < 2; ++e) {
        const float x = W2[(32 * ib + hl) * F + 16 * kb + 8 * kh + 2 * q + e];
        const __bf16 p0 = static_cast<__bf16>(x);
        const float r1 = x - static_cast<float>(p0);
        const __bf16 p1 = static_cast<__bf16>(r1);
        const float r2 = r1 - static_cast<float>(p1);
        const __bf16 p2 = static_cast<__bf16>(r2);
        const __bf16 pick = piece == 0 ? p0 : (piece == 1 ? p1 : p2);
        bits[e] = static_cast<unsigned>(__builtin_bit_cast(unsigned short, pick));
      }
      v = __uint_as_float(bits[0] | (bits[1] << 16));
    }
    packed[i] = v;
  }
}

}  // namespace

extern "C" {

int mp_cfconv_bwd_packed_floats(void) { return PACKED_BWD_FLOATS; }

int mp_cfconv_bwd_pack_f32(const float* W1, const float* b1, int B, const float* W2, float* packed, mpStream_t stream) {
  MP_REQUIRE(B >= 1 && B <= MAX_KROWS - 2, "mp_cfconv_bwd_pack_f32: basis size B=%d must be in 1..%d", B, MAX_KROWS - 2);
  MP_REQUIRE(W1 && W2 && packed, "mp_cfconv_bwd_pack_f32: null pointer");
  cfconv_bwd_pack_kernel<<<64, 256, 0, mp::as_stream(stream)>>>(W1, b1, B, W2, packed);
  return mp::check_launch("mp_cfconv_bwd_pack_f32");
}

int mp_cfconv_gauss_dist_grad_f32(const float* x, const float* g_out, int64_t N, const float* dist, int bins,
                                  float distance, float sigma, float offset, const float* packed_bwd, const int32_t* recv,
                                  const int32_t* send, int64_t M, int accumulate, float* g_d, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && N >= 0, "mp_cfconv_gauss_dist_grad_f32: bad sizes");
  MP_REQUIRE(bins >= 1 && bins <= MAX_KROWS - 2 && sigma != 0.0f, "mp_cfconv_gauss_dist_grad_f32: bad basis arguments");
  if (M == 0 || N == 0) return MP_OK;
  MP_REQUIRE(x && g_out && dist && packed_bwd && recv && send && g_d, "mp_cfconv_gauss_dist_grad_f32: null pointer");
  MP_REQUIRE(M < (int64_t{1} << 31), "mp_cfconv_gauss_dist_grad_f32: M must fit int32");
  CfconvBwdArgs a{};
  a.x = x; a.g_out = g_out; a.dist = dist; a.packed = packed_bwd; a.recv = recv; a.send = send; a.g_d = g_d;
  a.accumulate = accumulate; a.M = M; a.N = N; a.B = bins;
  a.g_distance = distance;
  a.g_gamma = static_cast<float>(1.0 / static_cast<double>(sigma) / static_cast<double>(sigma) / 2.0);
  a.g_offset = offset;
  a.ntiles = static_cast<int>((M + TE - 1) / TE);
  const size_t lds = sizeof(float) * PACKED_BWD_FLOATS;
  const bool g1b = ((bins + 2) >> 1) <= G1B_MAX_NK;
  static std::mutex mu;                      // dynamic-LDS opt-in: per device (and build), guarded
  static unsigned long long done_mask[2] = {0, 0};
  {
    int dev = 0;
    MP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 64 || !((done_mask[g1b] >> dev) & 1ull)) {
      const void* fn = g1b ? reinterpret_cast<const void*>(&cfconv_dist_grad_kernel<true>)
                           : reinterpret_cast<const void*>(&cfconv_dist_grad_kernel<false>);
      MP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      if (dev < 64) done_mask[g1b] |= 1ull << dev;
    }
  }
  int grid = (a.ntiles + WAVES - 1) / WAVES;
  if (grid > 256) grid = 256;
  if (g1b) cfconv_dist_grad_kernel<true><<<grid, WAVES * 64, lds, mp::as_stream(stream)>>>(a);
  else cfconv_dist_grad_kernel<false><<<grid, WAVES * 64, lds, mp::as_stream(stream)>>>(a);
  return mp::check_launch("mp_cfconv_gauss_dist_grad_f32");
}

}  // extern "C"
