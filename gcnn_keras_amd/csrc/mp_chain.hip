// One or two chained Keras Dense layers on 16-row tiles with the weights in registers (kgcnn/layers/modules.py:74-87
// applied back to back, e.g. PAiNNconv's Dense(units, act) -> Dense(3 units) of kgcnn/layers/conv/painn_conv.py:60-62,
// PAiNNUpdate's of painn_conv.py:187-189, and their reverse forms against the transposed kernels).
//
// Why next to csrc/mp_dense.hip: at molecular batch sizes (N = 1-4 k rows, K and U = 128-384) the LDS-tiled GEMM there is
// latency bound - every launch pays a cold first tile, per-k-tile barriers and an epilogue round trip: 11-17 us per GEMM
// measured inside the PaiNN pipeline, 60 % of its energy + force pass.  Here a 512-thread workgroup owns a 16-row tile:
// its 8 waves each hold a 1/8 column slice of BOTH weight matrices in registers (requested together with the input
// tile: one memory round trip), the GEMMs run as v_mfma_f32_16x16x4_f32 chains from an LDS activation tile, and the
// 128-wide intermediate never leaves the CU.  FP32 in, FP32 accumulate (k-ordered fma chains: the 1e-5 budget holds).
//
//   stage 1:  m = x W1 + b1 ;  [save_pre <- m] ;  m = act1(m)      or   m = (x W1) * act1'(grad_pre)   (reverse pass)
//   stage 2:  out = m W2 + b2 + addend                              (absent: out = stage 1 + addend, any U1)
//
// W1 / W2 are mp_chain_pack_f32 images: the k-major register order of one wave's column slice, 16-B loads.
#include <mutex>

#include "mp_common.h"

namespace {

using floatx4 = __attribute__((ext_vector_type(4))) float;

constexpr int WAVES = 8;
constexpr int MID = 128;       // width between the two stages
constexpr int MID_LD = MID + 2;

struct ChainArgs {
  int64_t R;
  int ntiles;
  const float* x;          // (R, K1)
  const float* W1;         // packed (K1, U1)
  const float* b1;         // (U1) or null
  int act1;
  float alpha1;
  float* save_pre;         // (R, U1) or null
  const float* grad_pre;   // (R, U1) or null
  const float* W2;         // packed (128, U2) or null
  const float* b2;         // (U2) or null
  const float* addend;     // (R, U_out) or null; may alias out
  float* out;              // (R, U_out)
};

// slice of a packed image: float4 = the lane's registers of four consecutive k-steps
template <int K, int NCB>
__device__ __forceinline__ void load_slice(const float* __restrict__ P, int wave, int lane, float (&wr)[NCB][K / 4]) {
  const float4* p4 = reinterpret_cast<const float4*>(P);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int q = 0; q < K / 16; ++q) {
      const float4 v = p4[((wave * NCB + cb) * (K / 16) + q) * 64 + lane];
      wr[cb][4 * q + 0] = v.x; wr[cb][4 * q + 1] = v.y; wr[cb][4 * q + 2] = v.z; wr[cb][4 * q + 3] = v.w;
    }
}

// acc[cb] += Xs (16 x K, row stride LD) @ slice ; A operand: lane supplies Xs[lane & 15][4 s + (lane >> 4)]
template <int K, int LD, int NCB>
__device__ __forceinline__ void gemm16(const float* __restrict__ Xs, int lane, const float (&wr)[NCB][K / 4],
                                       floatx4 (&acc)[NCB]) {
  const float* xp = Xs + (lane & 15) * LD + (lane >> 4);
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    const float av = xp[4 * s];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[cb][s], acc[cb], 0, 0, 0);
  }
}

// NCB2 == 0: single stage with U1 = 128 * NCB1 output columns; otherwise NCB1 == 1 (U1 = 128) and U2 = 128 * NCB2.
template <int K1, int NCB1, int NCB2>
__global__ __launch_bounds__(512, 1) void dense_chain_kernel(ChainArgs a) {
  constexpr int LD1 = K1 + 2;   // bank = (2 row + k) mod 32: the 16 rows x 2 k a half-wave fetches hit 32 different banks
  constexpr bool TWO = NCB2 > 0;
  constexpr int U1 = 128 * NCB1;
  constexpr int UO = TWO ? 128 * NCB2 : U1;
  constexpr int NCBO = TWO ? NCB2 : NCB1;
  __shared__ float Xa[16 * LD1];
  __shared__ float Xb[TWO ? 16 * MID_LD : 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nblocks = gridDim.x;

  constexpr int NV = (16 * K1 / 4) / 512;   // float4 per thread and tile
  float4 stg[NV];
  auto stage_load = [&](int t) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + j * 512;
      const int r = i / (K1 / 4), k4 = i % (K1 / 4);
      const int64_t row = static_cast<int64_t>(t) * 16 + r;
      stg[j] = (t < a.ntiles && row < a.R) ? reinterpret_cast<const float4*>(a.x + row * K1)[k4]
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + j * 512;
      float* d = Xa + (i / (K1 / 4)) * LD1 + 4 * (i % (K1 / 4));
      d[0] = stg[j].x; d[1] = stg[j].y; d[2] = stg[j].z; d[3] = stg[j].w;
    }
  };
  stage_load(blockIdx.x);   // first: the tile must not queue behind the weight loads

  float w1[NCB1][K1 / 4];
  float w2[TWO ? NCB2 : 1][TWO ? MID / 4 : 1];
  load_slice<K1, NCB1>(a.W1, wave, lane, w1);
  if constexpr (TWO) load_slice<MID, NCB2>(a.W2, wave, lane, w2);
  float bias1[NCB1], bias2[TWO ? NCB2 : 1];
#pragma unroll
  for (int cb = 0; cb < NCB1; ++cb) bias1[cb] = a.b1 ? a.b1[wave * (U1 / WAVES) + 16 * cb + (lane & 15)] : 0.0f;
  if constexpr (TWO) {
#pragma unroll
    for (int cb = 0; cb < NCB2; ++cb) bias2[cb] = a.b2 ? a.b2[wave * (UO / WAVES) + 16 * cb + (lane & 15)] : 0.0f;
  }

  for (int tile = blockIdx.x; tile < a.ntiles; tile += nblocks) {
    const int64_t row0 = static_cast<int64_t>(tile) * 16;
    stage_store();
    __syncthreads();
    stage_load(tile + nblocks);
    // epilogue operands of this thread's elements, in flight during the GEMMs
    float gp[NCB1][4], ad[NCBO][4];
    if (a.grad_pre) {
#pragma unroll
      for (int cb = 0; cb < NCB1; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + 4 * (lane >> 4) + r;
          gp[cb][r] = row < a.R ? a.grad_pre[row * U1 + wave * (U1 / WAVES) + 16 * cb + (lane & 15)] : 0.0f;
        }
    }
    if (a.addend) {
#pragma unroll
      for (int cb = 0; cb < NCBO; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + 4 * (lane >> 4) + r;
          ad[cb][r] = row < a.R ? a.addend[row * UO + wave * (UO / WAVES) + 16 * cb + (lane & 15)] : 0.0f;
        }
    }
    floatx4 acc1[NCB1];
#pragma unroll
    for (int cb = 0; cb < NCB1; ++cb) acc1[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    gemm16<K1, LD1, NCB1>(Xa, lane, w1, acc1);
#pragma unroll
    for (int cb = 0; cb < NCB1; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 4 * (lane >> 4) + r;
        const int64_t row = row0 + lrow;
        const int col = wave * (U1 / WAVES) + 16 * cb + (lane & 15);
        float v = acc1[cb][r] + bias1[cb];
        if (a.save_pre && row < a.R) a.save_pre[row * U1 + col] = v;
        if (a.grad_pre) v *= mp_act_grad(a.act1, a.alpha1, gp[cb][r]);
        else v = mp_apply_act(a.act1, a.alpha1, v);
        if constexpr (TWO) {
          Xb[lrow * MID_LD + col] = v;
        } else {
          if (a.addend) v += ad[cb][r];
          if (row < a.R) a.out[row * UO + col] = v;
        }
      }
    if constexpr (TWO) {
      __syncthreads();
      floatx4 acc2[NCB2];
#pragma unroll
      for (int cb = 0; cb < NCB2; ++cb) acc2[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
      gemm16<MID, MID_LD, NCB2>(Xb, lane, w2, acc2);
#pragma unroll
      for (int cb = 0; cb < NCB2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + 4 * (lane >> 4) + r;
          const int col = wave * (UO / WAVES) + 16 * cb + (lane & 15);
          float v = acc2[cb][r] + bias2[cb];
          if (a.addend) v += ad[cb][r];
          if (row < a.R) a.out[row * UO + col] = v;
        }
    }
    __syncthreads();   // Xa / Xb are reused by the next tile
  }
}

// ---------------------------------------------------------------------------------------------------- PAiNNUpdate, fused
// PAiNNUpdate.call (kgcnn/layers/conv/painn_conv.py:201-214) around its two Dense layers, one launch per block:
//
//   pre    c = [z | sqrt(relu(sum_k v_v[k]^2))]  (EuclideanNorm, LazyConcatenate),  prod = sum_k v_u[k] v_v[k]  (ScalarProduct)
//   chain  a = act(c Wd + bd) Wa + ba            (the <256, 1, 3> chain above)
//   post   z2 = z + (prod a_sv + a_ss) ,  v2[k] = v[k] + a_vv v_u[k]      with a = [a_vv | a_sv | a_ss]   (+ PAiNN.py:131-132)
//
// with uv (3N, 2F) = v [Wu | Wv] from the chain launch before it.  The two element-wise kernels this replaces
// (csrc/mp_painn_fused.hip: painn_update_pre / painn_update_post, ~5 us each at 64 graphs - launch floor - six per
// forward) become the prologue and the epilogue of the chain: thread t of the 512 owns node t / 32 of the tile and the
// four features 4 (t % 32) ..; what the epilogue needs again (z, prod, v_u, v) waits in LDS, the 16 x 384 output tile
// is exchanged through LDS (a node's a_vv / a_sv / a_ss columns belong to different waves).  Same arithmetic, same
// order as the separate kernels.  c / prod / a reach HBM only when the caller asks (the reverse pass reads them).
struct UpdateArgs {
  int64_t N;
  int ntiles;
  const float* zp;        // (N, F)
  const float* vp;        // (N, 3, F)
  const float* uv;        // (3N, 2F)
  const float* W1;        // packed (256, 128)
  const float* b1;
  int act1;
  float alpha1;
  float* save_pre;        // (N, 128) or null
  const float* W2;        // packed (128, 384)
  const float* b2;
  float* c_out;           // (N, 2F) or null
  float* prod_out;        // (N, F) or null
  float* a_out;           // (N, 3F) or null
  float* z2;              // (N, F)
  float* v2;              // (N, 3, F)
};

constexpr int UPD_K1 = 256, UPD_LD1 = UPD_K1 + 2, UPD_UO = 384, UPD_AT_LD = UPD_UO + 4, UPD_F = 128;
constexpr int UPD_LDS_FLOATS = 16 * UPD_LD1 + 16 * MID_LD + 16 * UPD_AT_LD + 8 * 16 * UPD_F;

__global__ __launch_bounds__(512, 1) void painn_update_chain_kernel(UpdateArgs a) {
  extern __shared__ __align__(16) float upd_lds[];
  float* Xa = upd_lds;                    // [16][LD1]   c tile
  float* Xb = Xa + 16 * UPD_LD1;          // [16][MID_LD] hidden tile
  float* At = Xb + 16 * MID_LD;           // [16][AT_LD] output tile a
  float* St = At + 16 * UPD_AT_LD;        // [8][16][F]  z, prod, v_u[0..2], v[0..2] of the tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nblocks = gridDim.x;
  const int er = tid >> 5, ef = 4 * (tid & 31);   // element-wise part: node er of the tile, features ef .. ef + 3

  // (named registers, no arrays: a loop-carried float4 array captured by the lambda stays on the stack - and a kernel
  //  with a scratch segment pays for it at every launch)
  float4 zq, vu0, vu1, vu2, vv0, vv1, vv2, vq0, vq1, vq2;
  auto tile_load = [&](int t) {
    const int64_t row = static_cast<int64_t>(t) * 16 + er;
    const bool in = t < a.ntiles && row < a.N;
    const int64_t rs = in ? row : 0;    // (clamped: the loads are unconditional, the results of padding rows unused)
    zq = *reinterpret_cast<const float4*>(a.zp + rs * UPD_F + ef);
    vu0 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 0) * 2) * UPD_F + ef);
    vv0 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 0) * 2 + 1) * UPD_F + ef);
    vu1 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 1) * 2) * UPD_F + ef);
    vv1 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 1) * 2 + 1) * UPD_F + ef);
    vu2 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 2) * 2) * UPD_F + ef);
    vv2 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 2) * 2 + 1) * UPD_F + ef);
    vq0 = *reinterpret_cast<const float4*>(a.vp + (rs * 3 + 0) * UPD_F + ef);
    vq1 = *reinterpret_cast<const float4*>(a.vp + (rs * 3 + 1) * UPD_F + ef);
    vq2 = *reinterpret_cast<const float4*>(a.vp + (rs * 3 + 2) * UPD_F + ef);
  };
  tile_load(blockIdx.x);   // first: the tile must not queue behind the weight loads

  float w1[1][UPD_K1 / 4];
  float w2[3][MID / 4];
  load_slice<UPD_K1, 1>(a.W1, wave, lane, w1);
  load_slice<MID, 3>(a.W2, wave, lane, w2);
  float bias1[1], bias2[3];
  bias1[0] = a.b1 ? a.b1[wave * (128 / WAVES) + (lane & 15)] : 0.0f;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) bias2[cb] = a.b2 ? a.b2[wave * (UPD_UO / WAVES) + 16 * cb + (lane & 15)] : 0.0f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += nblocks) {
    const int64_t row0 = static_cast<int64_t>(tile) * 16;
    const int64_t erow = row0 + er;
    // ---- pre (painn_update_pre_kernel's arithmetic, k order) ----
    const float4 vu[3] = {vu0, vu1, vu2}, vv[3] = {vv0, vv1, vv2}, vq[3] = {vq0, vq1, vq2};
    float pr[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pr[0] += vu[k].x * vv[k].x; sq[0] += vv[k].x * vv[k].x;
      pr[1] += vu[k].y * vv[k].y; sq[1] += vv[k].y * vv[k].y;
      pr[2] += vu[k].z * vv[k].z; sq[2] += vv[k].z * vv[k].z;
      pr[3] += vu[k].w * vv[k].w; sq[3] += vv[k].w * vv[k].w;
    }
    const float4 prod4 = make_float4(pr[0], pr[1], pr[2], pr[3]);
    const float4 nrm4 = make_float4(sqrtf(fmaxf(sq[0], 0.0f)), sqrtf(fmaxf(sq[1], 0.0f)), sqrtf(fmaxf(sq[2], 0.0f)),
                                    sqrtf(fmaxf(sq[3], 0.0f)));
    {
      float* d = Xa + er * UPD_LD1 + ef;           // (LD1 = 258: 8-B aligned rows - scalar stores)
      d[0] = zq.x; d[1] = zq.y; d[2] = zq.z; d[3] = zq.w;
      d[UPD_F + 0] = nrm4.x; d[UPD_F + 1] = nrm4.y; d[UPD_F + 2] = nrm4.z; d[UPD_F + 3] = nrm4.w;
      float4* st = reinterpret_cast<float4*>(St + er * UPD_F + ef);
      st[0] = zq;
      st[(16 * UPD_F / 4) * 1] = prod4;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        st[(16 * UPD_F / 4) * (2 + k)] = vu[k];
        st[(16 * UPD_F / 4) * (5 + k)] = vq[k];
      }
      if (erow < a.N) {
        if (a.c_out) {
          *reinterpret_cast<float4*>(a.c_out + erow * 2 * UPD_F + ef) = zq;
          *reinterpret_cast<float4*>(a.c_out + erow * 2 * UPD_F + UPD_F + ef) = nrm4;
        }
        if (a.prod_out) *reinterpret_cast<float4*>(a.prod_out + erow * UPD_F + ef) = prod4;
      }
    }
    __syncthreads();
    // ---- chain: hidden = act(c W1 + b1), a = hidden W2 + b2 ----
    floatx4 acc1[1];
    acc1[0] = floatx4{0.f, 0.f, 0.f, 0.f};
    gemm16<UPD_K1, UPD_LD1, 1>(Xa, lane, w1, acc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = 4 * (lane >> 4) + r;
      const int64_t row = row0 + lrow;
      const int col = wave * (128 / WAVES) + (lane & 15);
      float v = acc1[0][r] + bias1[0];
      if (a.save_pre && row < a.N) a.save_pre[row * 128 + col] = v;
      Xb[lrow * MID_LD + col] = mp_apply_act(a.act1, a.alpha1, v);
    }
    __syncthreads();
    floatx4 acc2[3];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) acc2[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    gemm16<MID, MID_LD, 3>(Xb, lane, w2, acc2);
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 4 * (lane >> 4) + r;
        const int64_t row = row0 + lrow;
        const int col = wave * (UPD_UO / WAVES) + 16 * cb + (lane & 15);
        const float v = acc2[cb][r] + bias2[cb];
        At[lrow * UPD_AT_LD + col] = v;
        if (a.a_out && row < a.N) a.a_out[row * UPD_UO + col] = v;
      }
    __syncthreads();
    // ---- post (painn_update_post_kernel's arithmetic) ----
    if (erow < a.N) {
      const float4 a_vv = *reinterpret_cast<const float4*>(At + er * UPD_AT_LD + ef);
      const float4 a_sv = *reinterpret_cast<const float4*>(At + er * UPD_AT_LD + UPD_F + ef);
      const float4 a_ss = *reinterpret_cast<const float4*>(At + er * UPD_AT_LD + 2 * UPD_F + ef);
      const float4* st = reinterpret_cast<const float4*>(St + er * UPD_F + ef);
      const float4 z0 = st[0], p0 = st[(16 * UPD_F / 4) * 1];
      *reinterpret_cast<float4*>(a.z2 + erow * UPD_F + ef) =
          make_float4(z0.x + (p0.x * a_sv.x + a_ss.x), z0.y + (p0.y * a_sv.y + a_ss.y), z0.z + (p0.z * a_sv.z + a_ss.z),
                      z0.w + (p0.w * a_sv.w + a_ss.w));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 u = st[(16 * UPD_F / 4) * (2 + k)], v0 = st[(16 * UPD_F / 4) * (5 + k)];
        *reinterpret_cast<float4*>(a.v2 + (erow * 3 + k) * UPD_F + ef) =
            make_float4(v0.x + a_vv.x * u.x, v0.y + a_vv.y * u.y, v0.z + a_vv.z * u.z, v0.w + a_vv.w * u.w);
      }
    }
    // next tile's rows: requested only now - 40 registers held across the GEMMs (next to 160 of weights) would spill,
    // and a kernel with a scratch segment pays for it at every launch; at molecular batch sizes a workgroup has one tile
    tile_load(tile + nblocks);
    __syncthreads();   // the LDS tiles are reused by the next tile
  }
}

// Reverse of the fused PAiNNUpdate above, one launch per block (replaces painn_update_post_bwd / painn_update_pre_bwd of
// csrc/mp_painn_fused.hip around the <384, 1, 2> reverse chain):
//   prologue  g_a = [sum_k g_v2[k] v_u[k] | g_z2 prod | g_z2] ,  g_prod = g_z2 a_sv
//   chain     g_c = ((g_a Wa^T) * act'(h2)) Wd^T                                   (N, 2F)
//   epilogue  g_z = g_z2 + g_c[:, :F] ,  g_vu[k] = g_v2[k] a_vv + g_prod v_v[k] ,  g_vv[k] = g_prod v_u[k] + g_c[:, F:] v_v[k] / ||v_v||
// Same arithmetic, same order as the separate kernels; everything the epilogue needs again waits in LDS (13 x 8 KB).
struct UpdateBwdArgs {
  int64_t N;
  int ntiles;
  const float* gz2;       // (N, F)
  const float* gv2;       // (N, 3, F)
  const float* uv;        // (3N, 2F)
  const float* prod;      // (N, F)
  const float* a;         // (N, 3F)
  const float* c;         // (N, 2F)  (the norm half is read)
  const float* W1;        // packed (384, 128) = Wa^T
  int act1;
  float alpha1;
  const float* grad_pre;  // (N, 128) h2
  const float* W2;        // packed (128, 256) = Wd^T
  float* g_zp;            // (N, F)
  float* g_uv;            // (3N, 2F)
};

constexpr int UB_K1 = 384, UB_LD1 = UB_K1 + 2, UB_UO = 256, UB_GT_LD = UB_UO + 4;
constexpr int UB_NST = 13;
constexpr int UB_LDS_FLOATS = 16 * UB_LD1 + 16 * MID_LD + 16 * UB_GT_LD + UB_NST * 16 * UPD_F;

__global__ __launch_bounds__(512, 1) void painn_update_bwd_chain_kernel(UpdateBwdArgs a) {
  extern __shared__ __align__(16) float upd_lds[];
  float* Xa = upd_lds;                    // [16][LD1]    g_a tile
  float* Xb = Xa + 16 * UB_LD1;           // [16][MID_LD]
  float* Gt = Xb + 16 * MID_LD;           // [16][GT_LD]  g_c tile
  float* St = Gt + 16 * UB_GT_LD;         // [13][16][F]: g_z2, g_v2[0..2], v_u[0..2], v_v[0..2], norm, a_vv, g_prod
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nblocks = gridDim.x;
  const int er = tid >> 5, ef = 4 * (tid & 31);
  constexpr int SL = 16 * UPD_F / 4;      // float4 per stash slice

  float4 gz, gv0, gv1, gv2, vu0, vu1, vu2, vv0, vv1, vv2, pq, asv, avv, nq;
  auto tile_load = [&](int t) {
    const int64_t row = static_cast<int64_t>(t) * 16 + er;
    const bool in = t < a.ntiles && row < a.N;
    const int64_t rs = in ? row : 0;
    gz = *reinterpret_cast<const float4*>(a.gz2 + rs * UPD_F + ef);
    if (a.gv2) {   // wave-uniform; null = no gradient reaches v'' (the last block: the readout sees z only)
      gv0 = *reinterpret_cast<const float4*>(a.gv2 + (rs * 3 + 0) * UPD_F + ef);
      gv1 = *reinterpret_cast<const float4*>(a.gv2 + (rs * 3 + 1) * UPD_F + ef);
      gv2 = *reinterpret_cast<const float4*>(a.gv2 + (rs * 3 + 2) * UPD_F + ef);
    } else {
      gv0 = gv1 = gv2 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    vu0 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 0) * 2) * UPD_F + ef);
    vv0 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 0) * 2 + 1) * UPD_F + ef);
    vu1 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 1) * 2) * UPD_F + ef);
    vv1 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 1) * 2 + 1) * UPD_F + ef);
    vu2 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 2) * 2) * UPD_F + ef);
    vv2 = *reinterpret_cast<const float4*>(a.uv + ((rs * 3 + 2) * 2 + 1) * UPD_F + ef);
    pq = *reinterpret_cast<const float4*>(a.prod + rs * UPD_F + ef);
    avv = *reinterpret_cast<const float4*>(a.a + rs * 3 * UPD_F + ef);
    asv = *reinterpret_cast<const float4*>(a.a + rs * 3 * UPD_F + UPD_F + ef);
    nq = *reinterpret_cast<const float4*>(a.c + rs * 2 * UPD_F + UPD_F + ef);
  };
  tile_load(blockIdx.x);

  float w1[1][UB_K1 / 4];
  float w2[2][MID / 4];
  load_slice<UB_K1, 1>(a.W1, wave, lane, w1);
  load_slice<MID, 2>(a.W2, wave, lane, w2);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += nblocks) {
    const int64_t row0 = static_cast<int64_t>(tile) * 16;
    const int64_t erow = row0 + er;
    // ---- prologue (painn_update_post_bwd_kernel's arithmetic) ----
    const float4 gavv = make_float4((0.0f + gv0.x * vu0.x + gv1.x * vu1.x) + gv2.x * vu2.x,
                                    (0.0f + gv0.y * vu0.y + gv1.y * vu1.y) + gv2.y * vu2.y,
                                    (0.0f + gv0.z * vu0.z + gv1.z * vu1.z) + gv2.z * vu2.z,
                                    (0.0f + gv0.w * vu0.w + gv1.w * vu1.w) + gv2.w * vu2.w);
    const float4 gasv = make_float4(gz.x * pq.x, gz.y * pq.y, gz.z * pq.z, gz.w * pq.w);
    const float4 gprod = make_float4(gz.x * asv.x, gz.y * asv.y, gz.z * asv.z, gz.w * asv.w);
    {
      float* d = Xa + er * UB_LD1 + ef;
      d[0] = gavv.x; d[1] = gavv.y; d[2] = gavv.z; d[3] = gavv.w;
      d[UPD_F + 0] = gasv.x; d[UPD_F + 1] = gasv.y; d[UPD_F + 2] = gasv.z; d[UPD_F + 3] = gasv.w;
      d[2 * UPD_F + 0] = gz.x; d[2 * UPD_F + 1] = gz.y; d[2 * UPD_F + 2] = gz.z; d[2 * UPD_F + 3] = gz.w;
      float4* st = reinterpret_cast<float4*>(St + er * UPD_F + ef);
      st[SL * 0] = gz;
      st[SL * 1] = gv0; st[SL * 2] = gv1; st[SL * 3] = gv2;
      st[SL * 4] = vu0; st[SL * 5] = vu1; st[SL * 6] = vu2;
      st[SL * 7] = vv0; st[SL * 8] = vv1; st[SL * 9] = vv2;
      st[SL * 10] = nq; st[SL * 11] = avv; st[SL * 12] = gprod;
    }
    __syncthreads();
    // ---- chain: m = (g_a W1) * act'(h2), g_c = m W2 ----
    float gp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * (lane >> 4) + r;
      gp[r] = row < a.N ? a.grad_pre[row * 128 + wave * (128 / WAVES) + (lane & 15)] : 0.0f;
    }
    floatx4 acc1[1];
    acc1[0] = floatx4{0.f, 0.f, 0.f, 0.f};
    gemm16<UB_K1, UB_LD1, 1>(Xa, lane, w1, acc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = 4 * (lane >> 4) + r;
      const int col = wave * (128 / WAVES) + (lane & 15);
      Xb[lrow * MID_LD + col] = acc1[0][r] * mp_act_grad(a.act1, a.alpha1, gp[r]);
    }
    __syncthreads();
    floatx4 acc2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc2[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    gemm16<MID, MID_LD, 2>(Xb, lane, w2, acc2);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 4 * (lane >> 4) + r;
        const int col = wave * (UB_UO / WAVES) + 16 * cb + (lane & 15);
        Gt[lrow * UB_GT_LD + col] = acc2[cb][r];
      }
    __syncthreads();
    // ---- epilogue (painn_update_pre_bwd_kernel's arithmetic) ----
    if (erow < a.N) {
      const float4 gc0 = *reinterpret_cast<const float4*>(Gt + er * UB_GT_LD + ef);
      const float4 gc1 = *reinterpret_cast<const float4*>(Gt + er * UB_GT_LD + UPD_F + ef);
      const float4* st = reinterpret_cast<const float4*>(St + er * UPD_F + ef);
      const float4 z0 = st[SL * 0], nrm = st[SL * 10], a_vv = st[SL * 11], gpd = st[SL * 12];
      *reinterpret_cast<float4*>(a.g_zp + erow * UPD_F + ef) = make_float4(z0.x + gc0.x, z0.y + gc0.y, z0.z + gc0.z, z0.w + gc0.w);
      const float4 inv = make_float4(nrm.x > 0.0f ? gc1.x / nrm.x : 0.0f, nrm.y > 0.0f ? gc1.y / nrm.y : 0.0f,
                                     nrm.z > 0.0f ? gc1.z / nrm.z : 0.0f, nrm.w > 0.0f ? gc1.w / nrm.w : 0.0f);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 g2 = st[SL * (1 + k)], u = st[SL * (4 + k)], v = st[SL * (7 + k)];
        *reinterpret_cast<float4*>(a.g_uv + ((erow * 3 + k) * 2) * UPD_F + ef) =
            make_float4(g2.x * a_vv.x + gpd.x * v.x, g2.y * a_vv.y + gpd.y * v.y, g2.z * a_vv.z + gpd.z * v.z,
                        g2.w * a_vv.w + gpd.w * v.w);
        *reinterpret_cast<float4*>(a.g_uv + ((erow * 3 + k) * 2 + 1) * UPD_F + ef) =
            make_float4(gpd.x * u.x + inv.x * v.x, gpd.y * u.y + inv.y * v.y, gpd.z * u.z + inv.z * v.z,
                        gpd.w * u.w + inv.w * v.w);
      }
    }
    tile_load(tile + nblocks);
    __syncthreads();
  }
}

// image element i = ((((w * ncb + cb) * (K/16) + q) * 64 + lane) * 4 + j  <-  W[4 (4q + j) + (lane >> 4)][w (U/8) + 16 cb + (lane & 15)]
__global__ void chain_pack_kernel(const float* __restrict__ W, int K, int U, float* __restrict__ packed) {
  const int ncb = U / (16 * WAVES), total = K * U;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63;
    int rest = i >> 8;
    const int q = rest % (K / 16);
    rest /= (K / 16);
    const int cb = rest % ncb, w = rest / ncb;
    const int k = 4 * (4 * q + j) + (lane >> 4);
    const int col = w * (U / WAVES) + 16 * cb + (lane & 15);
    packed[i] = W[k * U + col];
  }
}

template <int K1, int NCB1, int NCB2>
int launch_chain(ChainArgs a, hipStream_t s) {
  a.ntiles = static_cast<int>((a.R + 15) / 16);
  const int grid = a.ntiles < 256 ? a.ntiles : 256;   // one 512-thread workgroup per CU, persistent over the tiles
  dense_chain_kernel<K1, NCB1, NCB2><<<grid, 512, 0, s>>>(a);
  return mp::check_launch("mp_dense_chain_f32");
}

// register budget: the weight slices of both stages (K1/4 * NCB1 + 32 * NCB2 registers per lane) must leave room for the
// accumulators and the prefetched tile within 256 - builds beyond 160 spill and are not instantiated
constexpr bool chain_fits(int k1, int ncb1, int ncb2) { return k1 / 4 * ncb1 + 32 * ncb2 <= 160; }

template <int K1, int NCB1, int NCB2>
int launch_if_built(const ChainArgs& a, hipStream_t s) {
  if constexpr (chain_fits(K1, NCB1, NCB2)) {
    return launch_chain<K1, NCB1, NCB2>(a, s);
  } else {
    mp::set_error("mp_dense_chain_f32: shape not built");
    return MP_EINVAL;
  }
}

template <int K1>
int dispatch_chain(const ChainArgs& a, int ncb1, int ncb2, hipStream_t s) {
  if (ncb2 == 0) {
    switch (ncb1) {
      case 1: return launch_if_built<K1, 1, 0>(a, s);
      case 2: return launch_if_built<K1, 2, 0>(a, s);
      default: return launch_if_built<K1, 3, 0>(a, s);
    }
  }
  switch (ncb2) {
    case 1: return launch_if_built<K1, 1, 1>(a, s);
    case 2: return launch_if_built<K1, 1, 2>(a, s);
    default: return launch_if_built<K1, 1, 3>(a, s);
  }
}

}  // namespace

extern "C" {

int mp_chain_supported(int K1, int U1, int U2) {
  const bool k_ok = K1 == 128 || K1 == 256 || K1 == 384;
  const auto u_ok = [](int u) { return u == 128 || u == 256 || u == 384; };
  if (!k_ok || !u_ok(U1) || (U2 != 0 && (!u_ok(U2) || U1 != 128))) return 0;
  return chain_fits(K1, U1 / 128, U2 / 128) ? 1 : 0;
}

int mp_chain_pack_f32(const float* W, int K, int U, float* packed, mpStream_t stream) {
  MP_REQUIRE(W && packed, "mp_chain_pack_f32: null pointer");
  MP_REQUIRE(K >= 16 && K % 16 == 0 && U >= 128 && U % 128 == 0, "mp_chain_pack_f32: K %% 16 == 0 and U %% 128 == 0 "
             "required (got %d x %d)", K, U);
  chain_pack_kernel<<<64, 256, 0, mp::as_stream(stream)>>>(W, K, U, packed);
  return mp::check_launch("mp_chain_pack_f32");
}

int mp_dense_chain_f32(const float* x, int64_t R, int K1, const float* W1_packed, const float* b1, int U1, int act1,
                       float alpha1, float* save_pre, const float* grad_pre, const float* W2_packed, const float* b2,
                       int U2, const float* addend, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0, "mp_dense_chain_f32: bad sizes");
  MP_REQUIRE(mp_chain_supported(K1, U1, W2_packed ? U2 : 0), "mp_dense_chain_f32: built for K1 in {128,256,384}, a "
             "128-wide intermediate and U in {128,256,384} (got %d -> %d -> %d)", K1, U1, W2_packed ? U2 : 0);
  MP_REQUIRE(act1 >= MP_ACT_LINEAR && act1 <= MP_ACT_LAST, "mp_dense_chain_f32: unknown activation %d", act1);
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && W1_packed && out, "mp_dense_chain_f32: null pointer");
  MP_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0, "mp_dense_chain_f32: x must be 16-byte aligned");
  ChainArgs a{};
  a.R = R; a.x = x; a.W1 = W1_packed; a.b1 = b1; a.act1 = act1; a.alpha1 = alpha1; a.save_pre = save_pre;
  a.grad_pre = grad_pre; a.W2 = W2_packed; a.b2 = b2; a.addend = addend; a.out = out;
  const int ncb1 = U1 / 128, ncb2 = W2_packed ? U2 / 128 : 0;
  hipStream_t s = mp::as_stream(stream);
  if (K1 == 128) return dispatch_chain<128>(a, ncb1, ncb2, s);
  if (K1 == 256) return dispatch_chain<256>(a, ncb1, ncb2, s);
  return dispatch_chain<384>(a, ncb1, ncb2, s);
}

int mp_painn_update_fused_f32(const float* z, const float* v, const float* uv, int64_t N, const float* W1_packed,
                              const float* b1, int act1, float alpha1, float* save_pre, const float* W2_packed,
                              const float* b2, float* c_out, float* prod_out, float* a_out, float* z2, float* v2,
                              mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_painn_update_fused_f32: bad size");
  MP_REQUIRE(act1 >= MP_ACT_LINEAR && act1 <= MP_ACT_LAST, "mp_painn_update_fused_f32: unknown activation %d", act1);
  if (N == 0) return MP_OK;
  MP_REQUIRE(z && v && uv && W1_packed && W2_packed && z2 && v2, "mp_painn_update_fused_f32: null pointer");
  UpdateArgs a{};
  a.N = N; a.ntiles = static_cast<int>((N + 15) / 16);
  a.zp = z; a.vp = v; a.uv = uv; a.W1 = W1_packed; a.b1 = b1; a.act1 = act1; a.alpha1 = alpha1; a.save_pre = save_pre;
  a.W2 = W2_packed; a.b2 = b2; a.c_out = c_out; a.prod_out = prod_out; a.a_out = a_out; a.z2 = z2; a.v2 = v2;
  const size_t lds = sizeof(float) * UPD_LDS_FLOATS;
  static std::mutex mu;                      // dynamic-LDS opt-in: per device, guarded
  static unsigned long long done_mask = 0;
  {
    int dev = 0;
    MP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 64 || !((done_mask >> dev) & 1ull)) {
      MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&painn_update_chain_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      if (dev < 64) done_mask |= 1ull << dev;
    }
  }
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  painn_update_chain_kernel<<<grid, 512, lds, mp::as_stream(stream)>>>(a);
  return mp::check_launch("mp_painn_update_fused_f32");
}

int mp_painn_update_fused_bwd_f32(const float* g_z2, const float* g_v2, const float* uv, const float* prod, const float* a,
                                  const float* c, int64_t N, const float* W1T_packed, int act1, float alpha1,
                                  const float* grad_pre, const float* W2T_packed, float* g_z, float* g_uv,
                                  mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_painn_update_fused_bwd_f32: bad size");
  MP_REQUIRE(act1 >= MP_ACT_LINEAR && act1 <= MP_ACT_LAST, "mp_painn_update_fused_bwd_f32: unknown activation %d", act1);
  if (N == 0) return MP_OK;
  MP_REQUIRE(g_z2 && uv && prod && a && c && W1T_packed && grad_pre && W2T_packed && g_z && g_uv,
             "mp_painn_update_fused_bwd_f32: null pointer");
  UpdateBwdArgs q{};
  q.N = N; q.ntiles = static_cast<int>((N + 15) / 16);
  q.gz2 = g_z2; q.gv2 = g_v2; q.uv = uv; q.prod = prod; q.a = a; q.c = c; q.W1 = W1T_packed; q.act1 = act1;
  q.alpha1 = alpha1; q.grad_pre = grad_pre; q.W2 = W2T_packed; q.g_zp = g_z; q.g_uv = g_uv;
  const size_t lds = sizeof(float) * UB_LDS_FLOATS;
  static std::mutex mu;                      // dynamic-LDS opt-in: per device, guarded
  static unsigned long long done_mask = 0;
  {
    int dev = 0;
    MP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 64 || !((done_mask >> dev) & 1ull)) {
      MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&painn_update_bwd_chain_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      if (dev < 64) done_mask |= 1ull << dev;
    }
  }
  const int grid = q.ntiles < 256 ? q.ntiles : 256;
  painn_update_bwd_chain_kernel<<<grid, 512, lds, mp::as_stream(stream)>>>(q);
  return mp::check_launch("mp_painn_update_fused_bwd_f32");
}

}  // extern "C"
