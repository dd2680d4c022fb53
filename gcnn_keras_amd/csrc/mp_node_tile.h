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

// shifted softplus and its derivative sigmoid(x) from the same exponential: e = exp(-|x|), sigmoid = (x >= 0 ? 1 : e) / (1 + e)
template <bool FAST>
__device__ __forceinline__ float ssp_with_grad(float x, float& grad) {
  const float e = FAST ? __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f) : expf(-fabsf(x));
  grad = (x >= 0.0f ? 1.0f : e) / (1.0f + e);
  return ssp<FAST>(x);
}

}  // namespace
#endif  // MP_NODE_TILE_H
