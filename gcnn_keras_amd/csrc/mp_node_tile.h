// Building blocks of the node-side chain kernels (csrc/mp_schnet_node.hip forward, csrc/mp_schnet_bwd.hip reverse):
// a 16-node activation tile in LDS, a wave's 32-column slice of a weight matrix held in registers, chained
// v_mfma_f32_16x16x4_f32 GEMMs.  Header-only; everything is internal to the including translation unit.
#ifndef MP_NODE_TILE_H
#define MP_NODE_TILE_H
#include "mp_common.h"

namespace {

using floatx4 = __attribute__((ext_vector_type(4))) float;

constexpr int F = 128;
constexpr int X_LD = 130;  // padded row stride of the LDS activation tiles: (2*node + k) mod 32 is conflict-free

__device__ __forceinline__ float ssp_exact(float x) { return mp_softplus(x) - 0.6931471805599453f; }
// v_exp_f32 / v_log_f32 form of the shifted softplus in six VALU instructions (same as csrc/mp_cfconv.hip;
// |delta| < 2e-7 vs ssp_exact): max(x,0) + ln2 * log2((1 + 2^(-|x| log2 e)) / 2), the max as an integer max on the bits.
__device__ __forceinline__ float ssp_fast(float x) {
  const float t = __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f);
  const float l = __builtin_amdgcn_logf(__builtin_fmaf(t, 0.5f, 0.5f));
  const int xi = __float_as_int(x);
  return __builtin_fmaf(l, 0.6931471805599453f, __int_as_float(xi > 0 ? xi : 0));
}
template <bool FAST>
__device__ __forceinline__ float ssp(float x) { return FAST ? ssp_fast(x) : ssp_exact(x); }

// Slice of W (K x U, row-major) for output columns col0 + 16*cb + (lane&15), k = 4*s + (lane>>4).
template <int K, int NCB>
__device__ __forceinline__ void load_wslice(const float* __restrict__ W, int U, int col0, int lane,
                                            float (&wr)[NCB][K / 4]) {
  const int g = lane >> 4, cc = lane & 15;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int s = 0; s < K / 4; ++s) wr[cb][s] = W[(4 * s + g) * U + col0 + 16 * cb + cc];
}

// The same slice from a pre-packed image (mp_schnet_node_pack_f32): the lane's registers of four consecutive k-steps are
// one float4, a wave instruction reads 1 KB contiguous - 16-B loads instead of 4-B loads at a 64-B stride (the weight
// load is ~a quarter of a node kernel's time at QM9 batch sizes, where every workgroup serves a single tile).
template <int K, int NCB>
__device__ __forceinline__ void load_wslice_packed(const float* __restrict__ P, int wave, int lane,
                                                   float (&wr)[NCB][K / 4]) {
  const float4* p4 = reinterpret_cast<const float4*>(P);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int q = 0; q < K / 16; ++q) {
      const float4 v = p4[((wave * NCB + cb) * (K / 16) + q) * 64 + lane];
      wr[cb][4 * q + 0] = v.x; wr[cb][4 * q + 1] = v.y; wr[cb][4 * q + 2] = v.z; wr[cb][4 * q + 3] = v.w;
    }
}
template <int K, int NCB, bool PACKED>
__device__ __forceinline__ void load_w(const float* __restrict__ W, int U, int wave, int lane,
                                       float (&wr)[NCB][K / 4]) {
  if constexpr (PACKED) load_wslice_packed<K, NCB>(W, wave, lane, wr);
  else load_wslice<K, NCB>(W, U, wave * (U / 4), lane, wr);
}

// acc[rb][cb] += Xs(16*RB x K) @ Wslice ; A operand from LDS: lane supplies Xs[node = 16 rb + (lane&15)][k = 4s + (lane>>4)];
// every weight register feeds RB MFMAs.
template <int K, int NCB, int RB>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ Xs, int lane, const float (&wr)[NCB][K / 4],
                                          floatx4 (&acc)[RB][NCB]) {
  const float* xp = Xs + (lane & 15) * X_LD + (lane >> 4);
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    float av[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) av[rb] = xp[rb * 16 * X_LD + 4 * s];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb], wr[cb][s], acc[rb][cb], 0, 0, 0);
  }
}

// ---- the same GEMMs on the bf16 matrix pipe as an exact FP32 emulation (csrc/mp_cfconv.hip: every operand split into
//      three bf16 pieces, the six leading cross products on v_mfma_f32_16x16x32_bf16, FP32 accumulate) ----------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using floatx2n = __attribute__((ext_vector_type(2))) float;

// a wave's slice of W as B operands: [column block][k block of 32][piece], 16 B each (lane: column lane & 15, k group
// lane >> 4 holds k = 32 kb + 8 (lane >> 4) + 0..7), from the image of mp_schnet_node_pack_bf16_f32
template <int K, int NCB>
struct WSliceBf {
  bf16x8 p[NCB][K / 32][3];
};
template <int K, int NCB>
__device__ __forceinline__ void load_w_bf(const float* __restrict__ P, int wave, int lane, WSliceBf<K, NCB>& w) {
  const uint4* p4 = reinterpret_cast<const uint4*>(P);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int kb = 0; kb < K / 32; ++kb)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        w.p[cb][kb][pc] = __builtin_bit_cast(bf16x8, p4[(((wave * NCB + cb) * (K / 32) + kb) * 3 + pc) * 64 + lane]);
}

// acc[rb][cb] += Xs(16*RB x K) @ Wslice.  A operand: lane (node lane & 15, k group lane >> 4) reads its 8 values of the
// k block from the FP32 LDS tile and splits them in registers (value pairs: v_cvt_pk_bf16_f32, widen, packed subtract).
template <int K, int NCB, int RB>
__device__ __forceinline__ void gemm_tile_bf(const float* __restrict__ Xs, int lane, const WSliceBf<K, NCB>& w,
                                             floatx4 (&acc)[RB][NCB]) {
  const float* xp = Xs + (lane & 15) * X_LD + 8 * (lane >> 4);
#pragma unroll
  for (int kb = 0; kb < K / 32; ++kb) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      uint4 hi, mid, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const floatx2n x = *reinterpret_cast<const floatx2n*>(xp + rb * 16 * X_LD + 32 * kb + 2 * j);   // 8-B aligned
        const unsigned u0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        const floatx2n r1 = x - floatx2n{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u)};
        const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const floatx2n r2 = r1 - floatx2n{__uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u)};
        const unsigned u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        (j == 0 ? hi.x : j == 1 ? hi.y : j == 2 ? hi.z : hi.w) = u0;
        (j == 0 ? mid.x : j == 1 ? mid.y : j == 2 ? mid.z : mid.w) = u1;
        (j == 0 ? lo.x : j == 1 ? lo.y : j == 2 ? lo.z : lo.w) = u2;
      }
      const bf16x8 a_hi = __builtin_bit_cast(bf16x8, hi), a_mid = __builtin_bit_cast(bf16x8, mid),
                   a_lo = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {   // smallest products first
        floatx4 c = acc[rb][cb];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, w.p[cb][kb][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, w.p[cb][kb][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mid, w.p[cb][kb][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mid, w.p[cb][kb][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, w.p[cb][kb][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, w.p[cb][kb][0], c, 0, 0, 0);
        acc[rb][cb] = c;
      }
    }
  }
}

// ---- the activation tile split ONCE, by whoever produces it ---------------------------------------------------------
// gemm_tile_bf above lets every consuming wave split the whole 16 x K FP32 tile for itself: four waves = four times the
// same ~14 vector instructions per value pair, more issue cycles than the MFMAs they feed.  Here the producer (the
// staging threads for the first GEMM, the previous GEMM's epilogue for the others) writes the three bf16 pieces of every
// element into three LDS planes, and a consumer's A operand is one 16-B read per piece and k block.
//   plane p, row r, column k  ->  P[(p * rows + r) * XP_LD + k]   (bf16 units)
// XP_LD = 144 elements = 288 B = 18 slots of 16 B: the 16 lanes of every ds_read_b128 lane group (MI355X_MICROARCH.md,
// LDS) - rows 0-3 and 12-15 of k group g, rows 4-11 of k group g + 1 - fall on slots (2 row + g) mod 16, all different.
constexpr int XP_LD = 144;
// Row r of a tile is stored at row r ^ ((r >> 2) & 1): the reads above stay conflict-free (within every 16-lane group
// the rows of either k group stay distinct mod 8), and the epilogues' ds_write_b32 do too - the four quarter-waves of an
// MFMA accumulator hold rows 4 q + j, whose unpermuted offsets (72 dwords per row) put rows j and j + 4 on the same banks.
__device__ __forceinline__ int xp_row(int r) { return r ^ ((r >> 2) & 1); }

// the split of gemm_tile_bf (same instructions, same rounding): x = hi + mid + lo piecewise in bf16
__device__ __forceinline__ void split3_pair(floatx2n x, unsigned& u0, unsigned& u1, unsigned& u2) {
  u0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const floatx2n r1 = x - floatx2n{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u)};
  u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const floatx2n r2 = r1 - floatx2n{__uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u)};
  u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// four consecutive k of one row (a staging thread's float4) -> 8 B per plane
template <int ROWS>
__device__ __forceinline__ void put_split4(unsigned short* __restrict__ P, int row, int k, float4 v) {
  unsigned a0, a1, a2, b0, b1, b2;
  split3_pair(floatx2n{v.x, v.y}, a0, a1, a2);
  split3_pair(floatx2n{v.z, v.w}, b0, b1, b2);
  unsigned short* d = P + xp_row(row) * XP_LD + k;
  *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
  *reinterpret_cast<uint2*>(d + ROWS * XP_LD) = make_uint2(a1, b1);
  *reinterpret_cast<uint2*>(d + 2 * ROWS * XP_LD) = make_uint2(a2, b2);
}

// two elements of one COLUMN in rows row_a / row_b (an MFMA accumulator holds four rows of one column per lane)
template <int ROWS>
__device__ __forceinline__ void put_split_rows(unsigned short* __restrict__ P, int row_a, int row_b, int k, float va,
                                               float vb) {
  unsigned u0, u1, u2;
  split3_pair(floatx2n{va, vb}, u0, u1, u2);
  unsigned short* da = P + xp_row(row_a) * XP_LD + k;
  unsigned short* db = P + xp_row(row_b) * XP_LD + k;
  da[0] = static_cast<unsigned short>(u0); db[0] = static_cast<unsigned short>(u0 >> 16);
  da[ROWS * XP_LD] = static_cast<unsigned short>(u1); db[ROWS * XP_LD] = static_cast<unsigned short>(u1 >> 16);
  da[2 * ROWS * XP_LD] = static_cast<unsigned short>(u2); db[2 * ROWS * XP_LD] = static_cast<unsigned short>(u2 >> 16);
}

// columns k (even), k + 1 of one row -> one dword per plane
template <int ROWS>
__device__ __forceinline__ void put_split_cols(unsigned short* __restrict__ P, int row, int k, float left, float right) {
  unsigned u0, u1, u2;
  split3_pair(floatx2n{left, right}, u0, u1, u2);
  unsigned short* d = P + xp_row(row) * XP_LD + k;
  *reinterpret_cast<unsigned*>(d) = u0;
  *reinterpret_cast<unsigned*>(d + ROWS * XP_LD) = u1;
  *reinterpret_cast<unsigned*>(d + 2 * ROWS * XP_LD) = u2;
}

// An MFMA accumulator (rows row0 .. row0 + 3 of column `col` in v[0..3]; lane ^ 1 holds column col ^ 1) handed on as bf16
// pieces: neighbouring lanes swap two values (one DPP move each), so that the even lane writes rows row0, row0 + 1 and the
// odd lane rows row0 + 2, row0 + 3 - as dwords of two columns (six ds_write_b32 instead of twelve 2-byte stores).
template <int ROWS>
__device__ __forceinline__ void put_split_acc(unsigned short* __restrict__ P, int lane, int row0, int col,
                                              const float (&v)[4]) {
  const bool odd = lane & 1;
  const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];          // what the neighbour writes
  const float p0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  const float p1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xf, 0xf, true));
  const int row = row0 + (odd ? 2 : 0), k = col & ~1;
  put_split_cols<ROWS>(P, row, k, odd ? p0 : v[0], odd ? v[2] : p0);
  put_split_cols<ROWS>(P, row + 1, k, odd ? p1 : v[1], odd ? v[3] : p1);
}

// acc[rb][cb] += tile(16*RB x K, planes P) @ Wslice: the six products of gemm_tile_bf per k block, as TWO dependent chains
// per accumulator - the three small products (relative size 2^-16) in a sum of their own, added once at the end - so that
// a wave has 4 RB independent MFMA chains in flight instead of 2 RB (two did not cover the latency of a dependent
// v_mfma_f32_16x16x32_bf16: 26 cycles per instruction measured, 16 is the issue rate).
template <int K, int NCB, int RB, bool PREFETCH = true>
__device__ __forceinline__ void gemm_tile_pre(const unsigned short* __restrict__ P, int lane, const WSliceBf<K, NCB>& w,
                                              floatx4 (&acc)[RB][NCB]) {
  constexpr int ROWS = 16 * RB;
  const unsigned short* xp = P + xp_row(lane & 15) * XP_LD + 8 * (lane >> 4);
  floatx4 small[RB][NCB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) small[rb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
  // PREFETCH: every A piece of the tile requested up front (K / 32 * 3 * RB reads of 16 B: 48 registers at K = 128) - read
  // block by block the compiler waits for each block's reads right in front of its MFMAs, one exposed LDS latency per k
  // block.  Builds without the registers for it (the LAST chain: four weight slices) read block by block.
  constexpr int NPF = PREFETCH ? K / 32 : 1;
  uint4 ap[NPF][RB][3];
  if constexpr (PREFETCH) {
#pragma unroll
    for (int kb = 0; kb < K / 32; ++kb)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          ap[kb][rb][pc] = *reinterpret_cast<const uint4*>(xp + rb * 16 * XP_LD + 32 * kb + pc * ROWS * XP_LD);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int kb = 0; kb < K / 32; ++kb) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      if constexpr (!PREFETCH) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          ap[0][rb][pc] = *reinterpret_cast<const uint4*>(xp + rb * 16 * XP_LD + 32 * kb + pc * ROWS * XP_LD);
      }
      const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ap[PREFETCH ? kb : 0][rb][0]);
      const bf16x8 a_mid = __builtin_bit_cast(bf16x8, ap[PREFETCH ? kb : 0][rb][1]);
      const bf16x8 a_lo = __builtin_bit_cast(bf16x8, ap[PREFETCH ? kb : 0][rb][2]);
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          // small: lo x hi', hi x lo', mid x mid' ; large: mid x hi', hi x mid', hi x hi'
          const bf16x8 as = j == 0 ? a_lo : (j == 1 ? a_hi : a_mid);
          const bf16x8 al = j == 0 ? a_mid : a_hi;
          small[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as, w.p[cb][kb][j == 0 ? 0 : (j == 1 ? 2 : 1)],
                                                                  small[rb][cb], 0, 0, 0);
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w.p[cb][kb][j == 2 ? 0 : (j == 0 ? 0 : 1)],
                                                                acc[rb][cb], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] += small[rb][cb];
}

// shifted softplus and its derivative sigmoid(x) from the same exponential: e = exp(-|x|), sigmoid = (x >= 0 ? 1 : e) / (1 + e)
template <bool FAST>
__device__ __forceinline__ float ssp_with_grad(float x, float& grad) {
  const float e = FAST ? __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f) : expf(-fabsf(x));
  grad = (x >= 0.0f ? 1.0f : e) / (1.0f + e);
  return ssp<FAST>(x);
}

}  // namespace
#endif  // MP_NODE_TILE_H
