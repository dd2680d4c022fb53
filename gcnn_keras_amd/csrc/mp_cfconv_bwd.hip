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
// P and Z (K = B + 1) run on v_mfma_f32_32x32x2_f32.  GH (K = 128, 70 % of the matrix work) runs on the bf16 matrix pipe
// as the exact FP32 emulation of the forward kernel (csrc/mp_cfconv.hip: three bf16 pieces per operand, six products,
// FP32 accumulate): W2's pieces come pre-split from the image (A operand, one ds_read_b128 per piece), v is built in
// REGISTERS - lane (edge e, k half) loads the 8 features of g_out[recv(e)] and x[send(e)] that are its k slots of the
// current k block, multiplies and splits them - so the (32 x 128) v tile of the FP32 build (66 KB of LDS per workgroup,
// a store and a strided read per element) is gone.  LDS per workgroup: W1 image (34 rows, the forward's packing) + the
// three W2 images = 113 KB.
#include <mutex>

#include "mp_common.h"

namespace {

using floatx16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int F = 128;
constexpr int TE = 32;
constexpr int MAX_KROWS = 34;   // W1 rows in LDS: B inputs + 1 bias row, padded to even (B <= 32)
constexpr int WAVES = 4;
// W2 as A operand of v_mfma_f32_32x32x16_bf16: 16 B per (piece, k block kb, row block ib, lane): element i =
// piece(W2[32 ib + (lane & 31)][16 kb + 8 (lane >> 5) + i])
constexpr int W2_PIECE_FLOATS = F * F / 2;
constexpr int W2_IMG_FLOATS = 3 * W2_PIECE_FLOATS;
constexpr int PACKED_BWD_FLOATS = MAX_KROWS * F + W2_IMG_FLOATS;

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
  float* W2s = lds + MAX_KROWS * F;        // three bf16 operand images of W2

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31;
  const int hh = lane >> 5;
  const int B = a.B;
  const int nk = (B + 2) >> 1;             // k pairs of the basis GEMMs (B inputs + bias row, padded to even)
  // the 113 KB image by LDS-DMA (1 KB per wave instruction, no VGPRs, all requests in flight at once; the first version
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

    const int i_node = my_recv < 0 ? 0 : (my_recv >= a.N ? static_cast<int>(a.N) - 1 : my_recv);
    const int j_node = my_send < 0 ? 0 : (my_send >= a.N ? static_cast<int>(a.N) - 1 : my_send);
    // this lane's k slots of k block kb: features 16 kb + 8 hh + (0..7) of its edge's two rows (32 B each)
    const float4* g_row = reinterpret_cast<const float4*>(a.g_out + static_cast<int64_t>(i_node) * F + 8 * hh);
    const float4* x_row = reinterpret_cast<const float4*>(a.x + static_cast<int64_t>(j_node) * F + 8 * hh);
    float4 gq[2][2], xq[2][2];           // [buffer][first / second four features]
    gq[0][0] = g_row[0]; gq[0][1] = g_row[1];
    xq[0][0] = x_row[0]; xq[0][1] = x_row[1];

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
      const float gv = __builtin_amdgcn_exp2f((v * v) * (a.g_gamma * -1.4426950408889634f));   // v_exp_f32
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
      for (int r = 0; r < 16; ++r)   // sigmoid on v_exp_f32 / v_rcp_f32 (1 ulp each)
        P[ib][r] = Z[ib][r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(P[ib][r] * -1.4426950408889634f));

    // ---- GH[h][e] = sum_j W2[h][j] v[e][j] on the bf16 pipe: A = W2 pieces from LDS (read one step ahead, asm-ordered
    //      against the MFMAs as in the forward kernel), B = this lane's eight v values of the k block, split in registers ----
    floatx16 GH[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) GH[ib][r] = 0.0f;
    bf16x8 a_hi, a_mid, a_lo;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a_hi), "=&v"(a_mid), "=&v"(a_lo) : "v"(w2_addr), "v"(w2_addr_mid), "v"(w2_addr_lo));
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb + 1 < 8) {   // next k block's rows, in flight during this block's MFMAs
        gq[(kb + 1) & 1][0] = g_row[4 * (kb + 1)];
        gq[(kb + 1) & 1][1] = g_row[4 * (kb + 1) + 1];
        xq[(kb + 1) & 1][0] = x_row[4 * (kb + 1)];
        xq[(kb + 1) & 1][1] = x_row[4 * (kb + 1) + 1];
      }
      const float4 g0 = gq[kb & 1][0], g1 = gq[kb & 1][1], x0 = xq[kb & 1][0], x1 = xq[kb & 1][1];
      const float vv[8] = {g0.x * x0.x, g0.y * x0.y, g0.z * x0.z, g0.w * x0.w,
                           g1.x * x1.x, g1.y * x1.y, g1.z * x1.z, g1.w * x1.w};
      bf16x8 b_hi, b_mid, b_lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xval = valid ? vv[i] : 0.0f;
        const __bf16 p0 = static_cast<__bf16>(xval);
        const float r1 = xval - static_cast<float>(p0);
        const __bf16 p1 = static_cast<__bf16>(r1);
        const float r2 = r1 - static_cast<float>(p1);
        b_hi[i] = p0;
        b_mid[i] = p1;
        b_lo[i] = static_cast<__bf16>(r2);
      }
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        bf16x8 n_hi, n_mid, n_lo;
        if (4 * kb + ib + 1 < 32) {
          asm volatile("ds_read_b128 %0, %3 offset:%6\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %5 offset:%6"
                       : "=&v"(n_hi), "=&v"(n_mid), "=&v"(n_lo)
                       : "v"(w2_addr), "v"(w2_addr_mid), "v"(w2_addr_lo), "n"((4 * kb + ib + 1) * 1024), "a"(GH[ib]));
          asm volatile("" : "+a"(GH[ib]));
        }
        GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, GH[ib], 0, 0, 0);
        GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, GH[ib], 0, 0, 0);
        GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_mid, GH[ib], 0, 0, 0);
        GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, GH[ib], 0, 0, 0);
        GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, GH[ib], 0, 0, 0);
        GH[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, GH[ib], 0, 0, 0);
        if (4 * kb + ib + 1 < 32) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n_hi), "+v"(n_mid), "+v"(n_lo), "+a"(GH[ib]));
          a_hi = n_hi;
          a_mid = n_mid;
          a_lo = n_lo;
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
  const size_t lds = sizeof(float) * (MAX_KROWS * F + W2_IMG_FLOATS);
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
