// Fused PaiNN message block (kgcnn/layers/conv/painn_conv.py:97-115), edge side:
//
//     w   = Dense(3F, linear)(rbf)   [* envelope if cutoff]        :100-102     (M,B) -> (M,3F)
//     sw  = gather_out(s) * w ; sw1, sw2, sw3 = split(sw)           :99,103-104
//     ds  = segsum(sw1)                                              :105         (N,F)
//     dv  = segsum(sw2[:,None,:] * gather_out(v) + sw3[:,None,:] * r_ij[:,:,None])   :106-113   (N,3,F)
//
// in one kernel: the (M,3F) filter, the (M,3F) gathered scalars, the (M,3,F) gathered vectors and the two (M,3,F)
// products of the reference never exist in HBM.  The per-edge filter is a K = B = 20 contraction - far too thin for
// MFMA tiles - so it runs on the VALU with the lane's six weight columns (3 parts x 2 features) held in registers
// (6B VGPRs); the kernel is bound by the gathered rows (s: 3F floats, v: 3F floats per edge, served from L2 / Infinity
// Cache), not by HBM or the matrix pipe.
//
// One wave owns one receiving node (receiver-parallel over the CSR, like csrc/mp_segment.hip): lane l holds features
// l and l + 64, walks the node's edges in edge order (deterministic; the order tf.math.segment_sum uses after the
// stable sort) and writes ds / dv rows once - nodes without edges get zeros (has_unconnected pad).  Everything that
// is the same for all lanes of the wave (edge id, sender id, the edge's rbf row, r_ij, envelope) is fetched with
// wave-uniform (scalar) loads.
#include "mp_common.h"

namespace {

constexpr int F = 128;

struct PainnArgs {
  const float* s;      // (N, 3F)  phi(dense1(z)), node side
  const float* v;      // (N, 3, F) equivariant features
  const float* rbf;    // (M, B)
  const float* env;    // (M) or null
  const float* rij;    // (M, 3)
  const float* Ww;     // (B, 3F)
  const float* bw;     // (3F) or null
  const int32_t* ptr;  // (N+1) CSR over receivers
  const int32_t* perm; // (M) or null
  const int32_t* send; // (M) original edge order
  float* ds;           // (N, F)
  float* dv;           // (N, 3, F)
  int64_t N, M;
  int B;
};

template <int BT>  // BT > 0: basis size fixed at compile time (weights in registers fully unrolled)
__global__ __launch_bounds__(256) void painn_message_kernel(PainnArgs a) {
  constexpr int MAXB = BT > 0 ? BT : 32;
  const int lane = threadIdx.x & 63;
  const int B = BT > 0 ? BT : a.B;

  // the lane's weight columns: part p (0..2), feature f0 = lane / f1 = lane + 64
  float w0[3][MAXB], w1[3][MAXB], b0[3], b1[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
      w0[p][k] = k < B ? a.Ww[k * 3 * F + p * F + lane] : 0.0f;
      w1[p][k] = k < B ? a.Ww[k * 3 * F + p * F + 64 + lane] : 0.0f;
    }
    b0[p] = a.bw ? a.bw[p * F + lane] : 0.0f;
    b1[p] = a.bw ? a.bw[p * F + 64 + lane] : 0.0f;
  }

  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t n0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; n0 < a.N; n0 += nwaves) {
    const int n = __builtin_amdgcn_readfirstlane(static_cast<int>(n0));
    int e_lo = a.ptr[n], e_hi = a.ptr[n + 1];
    e_lo = e_lo < 0 ? 0 : (e_lo > a.M ? static_cast<int>(a.M) : e_lo);
    e_hi = e_hi < e_lo ? e_lo : (e_hi > a.M ? static_cast<int>(a.M) : e_hi);
    float ds0 = 0.0f, ds1 = 0.0f;
    float dv0[3] = {0.0f, 0.0f, 0.0f}, dv1[3] = {0.0f, 0.0f, 0.0f};
    for (int e = e_lo; e < e_hi; ++e) {
      const int r = a.perm ? a.perm[e] : e;           // wave-uniform
      int j = a.send[r];
      j = j < 0 ? 0 : (j >= a.N ? static_cast<int>(a.N) - 1 : j);
      const float* srow = a.s + static_cast<int64_t>(j) * 3 * F + lane;
      const float* vrow = a.v + static_cast<int64_t>(j) * 3 * F + lane;
      // gathered sender rows: six coalesced 256-B reads for s, six for v
      float sj0[3], sj1[3], vj0[3], vj1[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        sj0[p] = srow[p * F];
        sj1[p] = srow[p * F + 64];
        vj0[p] = vrow[p * F];
        vj1[p] = vrow[p * F + 64];
      }
      // per-edge filter w = rbf @ Ww + b (Dense), k-ordered; rbf row is wave-uniform
      const float* rb = a.rbf + static_cast<int64_t>(r) * B;
      float f0[3] = {0.0f, 0.0f, 0.0f}, f1[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < MAXB; ++k) {
        if (k < B) {
          const float x = rb[k];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            f0[p] = fmaf(x, w0[p][k], f0[p]);
            f1[p] = fmaf(x, w1[p][k], f1[p]);
          }
        }
      }
      const float envv = a.env ? a.env[r] : 1.0f;
      float sw0[3], sw1[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float wv0 = f0[p] + b0[p], wv1 = f1[p] + b1[p];
        if (a.env) { wv0 *= envv; wv1 *= envv; }   // lay_mult_cutoff([w, envelope])
        sw0[p] = sj0[p] * wv0;                       // lay_mult([s, w])
        sw1[p] = sj1[p] * wv1;
      }
      ds0 += sw0[0];
      ds1 += sw1[0];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float rk = a.rij[static_cast<int64_t>(r) * 3 + k];
        dv0[k] += sw0[1] * vj0[k] + sw0[2] * rk;     // (sw2 * vj) + (sw3 * r_ij)
        dv1[k] += sw1[1] * vj1[k] + sw1[2] * rk;
      }
    }
    a.ds[static_cast<int64_t>(n) * F + lane] = ds0;
    a.ds[static_cast<int64_t>(n) * F + 64 + lane] = ds1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      a.dv[(static_cast<int64_t>(n) * 3 + k) * F + lane] = dv0[k];
      a.dv[(static_cast<int64_t>(n) * 3 + k) * F + 64 + lane] = dv1[k];
    }
  }
}

}  // namespace

extern "C" {

int mp_painn_message_fused_f32(const float* s, const float* v, int64_t N, const float* rbf, int B, const float* env,
                               const float* rij, const float* Ww, const float* bw, const int32_t* ptr,
                               const int32_t* perm, const int32_t* send, int64_t M, float* ds, float* dv,
                               mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && B >= 1 && B <= 32, "mp_painn_message_fused_f32: bad sizes (B must be 1..32)");
  if (N == 0) return MP_OK;
  MP_REQUIRE(s && v && Ww && ptr && ds && dv && (M == 0 || (rbf && rij && send)),
             "mp_painn_message_fused_f32: null pointer");
  MP_REQUIRE(M < (int64_t{1} << 31) && N < (int64_t{1} << 31), "mp_painn_message_fused_f32: sizes must fit int32");
  PainnArgs a{};
  a.s = s; a.v = v; a.rbf = rbf; a.env = env; a.rij = rij; a.Ww = Ww; a.bw = bw;
  a.ptr = ptr; a.perm = perm; a.send = send; a.ds = ds; a.dv = dv; a.N = N; a.M = M; a.B = B;
  int64_t blocks = mp::ceil_div(N, 4);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = mp::as_stream(stream);
  if (B == 20) painn_message_kernel<20><<<static_cast<unsigned>(blocks), 256, 0, st>>>(a);
  else painn_message_kernel<0><<<static_cast<unsigned>(blocks), 256, 0, st>>>(a);
  return mp::check_launch("mp_painn_message_fused_f32");
}

}  // extern "C"
