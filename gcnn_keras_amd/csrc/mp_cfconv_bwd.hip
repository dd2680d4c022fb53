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
// FP32 MFMA (v_mfma_f32_32x32x2_f32) as in the forward.  LDS per workgroup: W1 image (34 rows, the forward's packing), W2
// re-packed for the A operand (W2p[j][4c+ib] = W2[32 ib + c][j]: one ds_read_b128 feeds four MFMAs), and one 32 x 128
// tile of v per wave (row stride 129: the B-operand reads of a half wave hit 32 different banks).
#include <mutex>

#include "mp_common.h"

namespace {

using floatx16 = __attribute__((ext_vector_type(16))) float;

constexpr int F = 128;
constexpr int TE = 32;
constexpr int MAX_KROWS = 34;   // W1 rows in LDS: B inputs + 1 bias row, padded to even (B <= 32)
constexpr int V_LD = 129;
constexpr int WAVES = 4;
constexpr int PACKED_BWD_FLOATS = MAX_KROWS * F + F * F;

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

__global__ __launch_bounds__(WAVES * 64, 1) void cfconv_dist_grad_kernel(CfconvBwdArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* W1s = lds;                        // [MAX_KROWS][F] packed like the forward
  float* W2s = lds + MAX_KROWS * F;        // [F][F]: W2s[j][4c + ib] = W2[32 ib + c][j]
  float* Vs = W2s + F * F;                 // [WAVES][TE][V_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31;
  const int hh = lane >> 5;
  const int B = a.B;
  const int nk = (B + 2) >> 1;             // k pairs of the basis GEMMs (B inputs + bias row, padded to even)
  {
    const float4* src = reinterpret_cast<const float4*>(a.packed);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < PACKED_BWD_FLOATS / 4; i += WAVES * 64) dst[i] = src[i];
  }
  __syncthreads();
  float* Vw = Vs + wave * TE * V_LD;
  const float* w1_lane = W1s + (nk * hh) * F + 4 * c;   // + s*F : rows s (low half) / nk + s (high half)
  const float* w2_lane = W2s + hh * F + 4 * c;          // + 2s*F: row j = 2s + hh
  const float fbins = static_cast<float>(B);

  // XCD-aware block order (mp_common.h): consecutive edge tiles share one XCD's L2
  for (int tile = static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x)) * WAVES + wave; tile < a.ntiles;
       tile += gridDim.x * WAVES) {
    const int64_t e0 = static_cast<int64_t>(tile) * TE;
    const int64_t e_mine = e0 + c;                      // lane c (both halves) <-> edge c of the tile
    const bool valid = e_mine < a.M;
    const int64_t ec = valid ? e_mine : a.M - 1;
    const int my_recv = a.recv[ec];
    const int my_send = a.send[ec];
    const float d = a.dist[ec];

    // ---- v tile into LDS: v[e][k] = g_out[recv(e)][k] * x[send(e)][k]; a half wave reads one whole 512-B row -------------
#pragma unroll 4
    for (int it = 0; it < TE / 2; ++it) {
      const int r = 2 * it + hh;
      int i = __shfl(my_recv, r, 64), j = __shfl(my_send, r, 64);
      i = i < 0 ? 0 : (i >= a.N ? static_cast<int>(a.N) - 1 : i);
      j = j < 0 ? 0 : (j >= a.N ? static_cast<int>(a.N) - 1 : j);
      const float4 g = *reinterpret_cast<const float4*>(a.g_out + static_cast<int64_t>(i) * F + 4 * c);
      const float4 xv = *reinterpret_cast<const float4*>(a.x + static_cast<int64_t>(j) * F + 4 * c);
      float* dst = Vw + r * V_LD + 4 * c;
      dst[0] = g.x * xv.x;
      dst[1] = g.y * xv.y;
      dst[2] = g.z * xv.z;
      dst[3] = g.w * xv.w;
    }

    // ---- P = pre1^T (with bias row) and Z = (g'(d) W1)^T: B operands are this lane's half of its edge's basis row ----
    floatx16 P[4], Z[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        P[ib][r] = 0.0f;
        Z[ib][r] = 0.0f;
      }
    for (int s = 0; s < nk; ++s) {
      const int k = s + nk * hh;
      const float mu = static_cast<float>(k) / fbins * a.g_distance;
      const float v = (d - a.g_offset) - mu;
      const float gv = expf((v * v) * (a.g_gamma * -1.0f));
      const float rb = k < B ? gv : (k == B ? 1.0f : 0.0f);
      const float rbd = k < B ? gv * (-2.0f * a.g_gamma * v) : 0.0f;   // d/dd exp(-gamma v^2)
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
    // q = ssp'(pre1) * z  (ssp' = softplus' = sigmoid), kept in P
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) P[ib][r] = Z[ib][r] / (1.0f + expf(-P[ib][r]));

    // ---- GH[h][e] = sum_j W2[h][j] v[e][j]: A from the W2 image (row j = 2s + hh), B from this wave's v tile -----------
    floatx16 GH[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) GH[ib][r] = 0.0f;
    const float* v_lane = Vw + c * V_LD + hh;           // + 2s
#pragma unroll 8
    for (int s = 0; s < F / 2; ++s) {
      const float4 av = *reinterpret_cast<const float4*>(w2_lane + 2 * s * F);
      const float bv = v_lane[2 * s];
      GH[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv, GH[0], 0, 0, 0);
      GH[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv, GH[1], 0, 0, 0);
      GH[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv, GH[2], 0, 0, 0);
      GH[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv, GH[3], 0, 0, 0);
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

// Image for the kernel above: W1 rows exactly as mp_cfconv_pack_f32 stores them (row k at [4c + blk] = W1[k][32 blk + c],
// bias as row B), then W2p[j][4c + ib] = W2[32 ib + c][j].
__global__ void cfconv_bwd_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int B,
                                       const float* __restrict__ W2, float* __restrict__ packed) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PACKED_BWD_FLOATS; i += stride) {
    float v = 0.0f;
    if (i < MAX_KROWS * F) {
      const int k = i / F, col = (i % F) / 4 + 32 * (i % 4);
      if (k < B) v = W1[k * F + col];
      else if (k == B && b1) v = b1[col];
    } else {
      const int t = i - MAX_KROWS * F;
      const int j = t / F, cc = (t % F) / 4, ib = t % 4;
      v = W2[(32 * ib + cc) * F + j];
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
  const size_t lds = sizeof(float) * (MAX_KROWS * F + F * F + WAVES * TE * V_LD);
  static std::mutex mu;                      // dynamic-LDS opt-in: per device, guarded
  static unsigned long long done_mask = 0;
  {
    int dev = 0;
    MP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 64 || !((done_mask >> dev) & 1ull)) {
      MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_dist_grad_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      if (dev < 64) done_mask |= 1ull << dev;
    }
  }
  int grid = (a.ntiles + WAVES - 1) / WAVES;
  if (grid > 256) grid = 256;
  cfconv_dist_grad_kernel<<<grid, WAVES * 64, lds, mp::as_stream(stream)>>>(a);
  return mp::check_launch("mp_cfconv_gauss_dist_grad_f32");
}

}  // extern "C"
