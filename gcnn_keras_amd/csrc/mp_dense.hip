// Keras Dense on flat ragged values (kgcnn/layers/modules.py:74-87): out = act(x @ W + b).
//
// FP32-in / FP32-accumulate MFMA (v_mfma_f32_32x32x2_f32): exact f32 products, k-ordered fma chain, which keeps
// the 1e-5 relative budget of BASELINE.json with room to spare (no xf32/TF32 on gfx950).  Generic in R, K, U
// (SchNet (M,20)x(20,128), PaiNN (N,3,128)x(128,384), GCN (2708,1433)x(1433,64), heads with U = 1 or 7).
// 64x64 output tile per 256-thread workgroup = 2x2 waves of one 32x32 accumulator each, BK = 64, operands staged
// through LDS (A padded to 65 floats per row so the 32 rows a half-wave reads for one k hit 32 different banks,
// W rows read contiguously), next tile's global loads (16-B where the shape allows) issued before the MFMA loop of the
// current one.
// The fused SchNet kernels (mp_cfconv.hip / mp_schnet_node.hip) are the throughput path; this kernel serves
// the layer-by-layer API and the GEMMs that are not worth fusing.
#include "mp_common.h"

namespace {

using floatx16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 64, BN = 64, BK = 64;
constexpr int A_LD = BK + 1;

// Prologue / epilogue options of mp_dense_ex_f32 (the fused PaiNN pipeline and its reverse pass): IN_MODE 1 applies an
// activation to x while the tile is staged (x holds a saved pre-activation: Dense(act) -> Dense becomes one launch per
// GEMM with only the pre-activation kept), IN_MODE 2 multiplies x by act'(in_pre) (the reverse pass through an
// activation, fused into the transposed-weight GEMM that follows it); `addend` (nullable, may alias `out`) is added to
// the result after the output activation (residual adds / gradient accumulation).  The epilogue forms - `out_pre` keeps
// the pre-activation next to the activated output, `grad_pre` multiplies the result by act'(saved pre-activation) - do
// the same jobs at the PRODUCING GEMM, once per element, and are what the PaiNN pipeline uses: the prologue forms redo
// the transcendental in every column-block workgroup and in front of its MFMAs (measured 21 vs 15 us per GEMM there).
struct DenseExtra {
  int in_act;
  float in_alpha;
  const float* in_pre;   // (R, K) for IN_MODE 2
  const float* addend;   // (R, U) or null
  float* out_pre;        // (R, U) or null: also store the pre-activation x W + b (kept for the reverse pass)
  const float* grad_pre; // (R, U) or null: multiply the result by in_act'(grad_pre) (reverse pass through an activation
                         // fused into the epilogue of the GEMM that produces the upstream gradient)
  float* partial;        // split-K: (gridDim.z, R, U) raw partial products, reduced by dense_splitk_reduce_kernel
  int64_t kchunk;        // split-K: k range of one z-slice (a multiple of BK)
};

// VEC: K % 4 == 0, U % 4 == 0 and 16-B aligned operands: both tiles are fetched with 16-B loads (a quarter of the load
// instructions of the scalar path, which stays for odd shapes such as GCN's 1433 input features or U = 1 heads).
// The prologue transform is applied when the prefetched registers are written to LDS - i.e. AFTER the MFMA loop that
// covers their latency - not where they are loaded (measured on the PaiNN chains: 21 -> ~9 us per (1344,128)x(128,384)
// GEMM; applied at the load, the activation's first use of the data stalls the wave in front of its MFMAs).
template <int IN_MODE, bool VEC>
__global__ __launch_bounds__(256) void dense_mfma_kernel(const float* __restrict__ x, int64_t R, int64_t Kfull,
                                                         const float* __restrict__ W, const float* __restrict__ b,
                                                         int64_t U, int act, float alpha, float* __restrict__ out,
                                                         DenseExtra ex) {
  const int64_t xld = Kfull;   // row stride of x (and of in_pre)
  const int64_t K = (ex.partial && (static_cast<int64_t>(blockIdx.z) + 1) * ex.kchunk < Kfull)
                        ? (static_cast<int64_t>(blockIdx.z) + 1) * ex.kchunk : Kfull;   // end of this workgroup's k range
  __shared__ __align__(16) float As[BM * A_LD];
  __shared__ __align__(16) float Bs[BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * BM;
  const int64_t col0 = static_cast<int64_t>(blockIdx.y) * BN;

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

  constexpr int NV = (BM * BK) / (256 * 4);   // float4 per thread and tile (A and B tiles have the same size)
  constexpr int NS = (BM * BK) / 256;         // floats per thread and tile, scalar path
  float4 ra4[VEC ? NV : 1], rb4[VEC ? NV : 1], rp4[(VEC && IN_MODE == 2) ? NV : 1];
  float ra[VEC ? 1 : NS], rb[VEC ? 1 : NS], rp[(!VEC && IN_MODE == 2) ? NS : 1];
  // Every load of a tile is UNCONDITIONAL, from an address clamped into the operand (row R-1 / the slice's last k / the
  // last column group); what lies outside is zeroed when the registers are written to LDS (store_tile).  With the loads
  // behind `ok ? load : 0` hipcc emits one branch per load and, between the fifth and sixth of a tile's eight loads, an
  // s_waitcnt vmcnt(0): the prefetch of the next k tile then costs a full memory round trip in FRONT of the MFMA loop that
  // was meant to cover it (seen in the ISA; every k tile of every layer-path Dense paid it).
  const int64_t r_last = R - 1;
  auto load_tile = [&](int64_t k0) {
    if constexpr (VEC) {
      const int64_t k_last = K - 4, c_last = U - 4;    // VEC: K, U multiples of 4
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        const int ar = idx / (BK / 4), ac = (idx % (BK / 4)) * 4;
        int64_t gr = row0 + ar, gk = k0 + ac;
        gr = gr < R ? gr : r_last;
        gk = gk < K ? gk : k_last;
        ra4[i] = *reinterpret_cast<const float4*>(x + gr * xld + gk);
        if constexpr (IN_MODE == 2) rp4[i] = *reinterpret_cast<const float4*>(ex.in_pre + gr * xld + gk);
        const int br = idx / (BN / 4), bc = (idx % (BN / 4)) * 4;
        int64_t gk2 = k0 + br, gc = col0 + bc;
        gk2 = gk2 < K ? gk2 : K - 1;
        gc = gc < U ? gc : c_last;
        rb4[i] = *reinterpret_cast<const float4*>(W + gk2 * U + gc);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = tid + i * 256;
        const int ar = idx / BK, ac = idx % BK;
        int64_t gr = row0 + ar, gk = k0 + ac;
        gr = gr < R ? gr : r_last;
        gk = gk < K ? gk : K - 1;
        ra[i] = x[gr * xld + gk];
        if constexpr (IN_MODE == 2) rp[i] = ex.in_pre[gr * xld + gk];
        const int br = idx / BN, bc = idx % BN;
        int64_t gk2 = k0 + br, gc = col0 + bc;
        gk2 = gk2 < K ? gk2 : K - 1;
        gc = gc < U ? gc : U - 1;
        rb[i] = W[gk2 * U + gc];
      }
    }
  };
  // the staged value: act(x) / x * act'(pre); padding is exactly 0 (act(0) need not be 0, act'(0) * 0 is)
  auto prologue = [&](float xv, float pv, bool ok) -> float {
    if constexpr (IN_MODE == 1) return ok ? mp_apply_act(ex.in_act, ex.in_alpha, xv) : 0.0f;
    if constexpr (IN_MODE == 2) return ok ? xv * mp_act_grad(ex.in_act, ex.in_alpha, pv) : 0.0f;
    return ok ? xv : 0.0f;
  };
  auto store_tile = [&](int64_t k0) {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        const int ar = idx / (BK / 4), ac = (idx % (BK / 4)) * 4;
        const bool ok = (row0 + ar) < R && (k0 + ac) < K;
        float* d = As + ar * A_LD + ac;
        const float4 pv = (IN_MODE == 2) ? rp4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        d[0] = prologue(ra4[i].x, pv.x, ok);
        d[1] = prologue(ra4[i].y, pv.y, ok);
        d[2] = prologue(ra4[i].z, pv.z, ok);
        d[3] = prologue(ra4[i].w, pv.w, ok);
        const bool okb = (k0 + idx / (BN / 4)) < K && (col0 + (idx % (BN / 4)) * 4) < U;
        *reinterpret_cast<float4*>(Bs + idx * 4) = okb ? rb4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = tid + i * 256;
        const bool ok = (row0 + idx / BK) < R && (k0 + idx % BK) < K;
        As[(idx / BK) * A_LD + (idx % BK)] = prologue(ra[i], (IN_MODE == 2) ? rp[i] : 0.0f, ok);
        const bool okb = (k0 + idx / BN) < K && (col0 + idx % BN) < U;
        Bs[idx] = okb ? rb[i] : 0.0f;
      }
    }
  };

  // split-K (few output tiles, long K: GCN's (2708,1433)x(1433,64) has 43 tiles for 256 CUs): blockIdx.z owns the k range
  // [kbeg, K) - the aliased K below is that slice's end - and writes raw partial sums
  const int64_t kbeg = ex.partial ? static_cast<int64_t>(blockIdx.z) * ex.kchunk : 0;
  const int64_t ktiles = (K - kbeg + BK - 1) / BK;
  load_tile(kbeg);
  for (int64_t t = 0; t < ktiles; ++t) {
    store_tile(kbeg + t * BK);
    __syncthreads();
    if (t + 1 < ktiles) load_tile(kbeg + (t + 1) * BK);
    const int64_t left = K - kbeg - t * BK;
    const int steps = left >= BK ? BK / 2 : static_cast<int>((left + 1) / 2);   // k pairs that hold data
    const float* a_ptr = As + (wr * 32 + (lane & 31)) * A_LD + (lane >> 5);
    const float* b_ptr = Bs + (lane >> 5) * BN + wc * 32 + (lane & 31);
    if (steps == BK / 2) {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[kk * 2], b_ptr[kk * 2 * BN], acc, 0, 0, 0);
    } else {   // last, partial k tile (zero padded up to an even k)
      for (int kk = 0; kk < steps; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[kk * 2], b_ptr[kk * 2 * BN], acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int64_t col = col0 + wc * 32 + (lane & 31);
  if (ex.partial) {
    if (col < U) {
      float* dst = ex.partial + static_cast<int64_t>(blockIdx.z) * R * U;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < R) dst[row * U + col] = acc[r];
      }
    }
    return;
  }
  if (col < U) {
    const float bias = b ? b[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < R) {
        const float pre = acc[r] + bias;
        if (ex.out_pre) ex.out_pre[row * U + col] = pre;
        float v = mp_apply_act(act, alpha, pre);
        if (ex.grad_pre) v *= mp_act_grad(ex.in_act, ex.in_alpha, ex.grad_pre[row * U + col]);
        if (ex.addend) v += ex.addend[row * U + col];
        out[row * U + col] = v;
      }
    }
  }
}

// out = act(sum_z partial[z] + b): the slices are added in z order (fixed), so the result does not depend on scheduling
__global__ void dense_splitk_reduce_kernel(const float* __restrict__ partial, int splits, int64_t R, int64_t U,
                                           const float* __restrict__ b, int act, float alpha, float* __restrict__ out) {
  const int64_t total = R * U;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    float s = 0.0f;
    for (int z = 0; z < splits; ++z) s += partial[z * total + t];
    out[t] = mp_apply_act(act, alpha, s + (b ? b[t % U] : 0.0f));
  }
}

template <int IN_MODE>
void launch_dense(dim3 grid, hipStream_t s, const float* x, int64_t R, int64_t K, const float* W, const float* b,
                  int64_t U, int act, float alpha, float* out, const DenseExtra& ex) {
  const bool vec = (K % 4 == 0) && (U % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(W) % 16 == 0) &&
                   (ex.in_pre == nullptr || reinterpret_cast<uintptr_t>(ex.in_pre) % 16 == 0);
  if (vec) dense_mfma_kernel<IN_MODE, true><<<grid, 256, 0, s>>>(x, R, K, W, b, U, act, alpha, out, ex);
  else dense_mfma_kernel<IN_MODE, false><<<grid, 256, 0, s>>>(x, R, K, W, b, U, act, alpha, out, ex);
}

__global__ void activation_kernel(int act, float alpha, const float* __restrict__ x, int64_t n,
                                  float* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = mp_apply_act(act, alpha, x[i]);
}

// Row softmax (Keras "softmax" on the last axis; GCN node classifier, kgcnn/training/hyper/hyper_cora_lu.py:144).
// One 64-lane wave per row, lanes stride the row, wave-wide max / sum by xor shuffles.
__global__ void softmax_rows_kernel(const float* __restrict__ x, int64_t R, int64_t C, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t r = wave_global; r < R; r += nwaves) {
    const float* row = x + r * C;
    float mx = -INFINITY;
    for (int64_t c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.0f;
    for (int64_t c = lane; c < C; c += 64) sum += expf(row[c] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    for (int64_t c = lane; c < C; c += 64) out[r * C + c] = expf(row[c] - mx) / sum;
  }
}

// Keras LayerNormalization over the last axis (kgcnn/layers/norm.py:60-63, 94-105 on the values of a ragged tensor):
// mean and biased variance of each row (tf.nn.moments: variance as the mean squared difference from the mean), then
// x * inv + (beta - mean * inv) with inv = rsqrt(var + eps) * gamma (the tf.nn.batch_normalization form Keras uses).
// One wave per row; the row is read twice (L2-resident), reduced with wave shuffles.
__global__ void layer_norm_rows_kernel(const float* __restrict__ x, int64_t R, int64_t C,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const float inv_c = 1.0f / static_cast<float>(C);
  for (int64_t r = wave_global; r < R; r += nwaves) {
    const float* row = x + r * C;
    float sum = 0.0f;
    for (int64_t c = lane; c < C; c += 64) sum += row[c];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * inv_c;
    float sq = 0.0f;
    for (int64_t c = lane; c < C; c += 64) {
      const float d = row[c] - mean;
      sq += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq * inv_c + eps);
    for (int64_t c = lane; c < C; c += 64) {
      const float inv = gamma ? rstd * gamma[c] : rstd;
      const float shift = (beta ? beta[c] : 0.0f) - mean * inv;
      out[r * C + c] = row[c] * inv + shift;
    }
  }
}

}  // namespace

extern "C" {

int mp_layer_norm_f32(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float epsilon,
                      float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && C >= 1, "mp_layer_norm_f32: bad sizes");
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && out, "mp_layer_norm_f32: null pointer");
  layer_norm_rows_kernel<<<mp::grid_for(R * 64), 256, 0, mp::as_stream(stream)>>>(x, R, C, gamma, beta, epsilon, out);
  return mp::check_launch("mp_layer_norm_f32");
}

int mp_dense_f32(const float* x, int64_t R, int64_t K, const float* W, const float* b, int64_t U, int act,
                 float act_alpha, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && K >= 1 && U >= 1, "mp_dense_f32: bad sizes R=%lld K=%lld U=%lld", (long long)R, (long long)K,
             (long long)U);
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST, "mp_dense_f32: unknown activation %d", act);
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && W && out, "mp_dense_f32: null pointer");
  const int64_t gy = mp::ceil_div(R, BM), gx = mp::ceil_div(U, BN);
  MP_REQUIRE(gx <= 65535 && gy < (int64_t{1} << 31), "mp_dense_f32: grid too large");
  dim3 grid(static_cast<unsigned>(gy), static_cast<unsigned>(gx));
  launch_dense<0>(grid, mp::as_stream(stream), x, R, K, W, b, U, act, act_alpha, out,
                  DenseExtra{0, 0.0f, nullptr, nullptr, nullptr, nullptr, nullptr, 0});
  return mp::check_launch("mp_dense_f32");
}

int mp_dense_ex_f32(const float* x, int64_t R, int64_t K, const float* W, const float* b, int64_t U, int act,
                    float act_alpha, int in_mode, int in_act, float in_alpha, const float* in_pre, const float* addend,
                    float* out_pre, const float* grad_pre, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && K >= 1 && U >= 1, "mp_dense_ex_f32: bad sizes R=%lld K=%lld U=%lld", (long long)R, (long long)K,
             (long long)U);
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST && in_act >= MP_ACT_LINEAR && in_act <= MP_ACT_LAST,
             "mp_dense_ex_f32: unknown activation");
  MP_REQUIRE(in_mode >= 0 && in_mode <= 2 && (in_mode != 2 || in_pre != nullptr), "mp_dense_ex_f32: bad prologue mode");
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && W && out, "mp_dense_ex_f32: null pointer");
  const int64_t gy = mp::ceil_div(R, BM), gx = mp::ceil_div(U, BN);
  MP_REQUIRE(gx <= 65535 && gy < (int64_t{1} << 31), "mp_dense_ex_f32: grid too large");
  dim3 grid(static_cast<unsigned>(gy), static_cast<unsigned>(gx));
  const DenseExtra ex{in_act, in_alpha, in_pre, addend, out_pre, grad_pre, nullptr, 0};
  hipStream_t s = mp::as_stream(stream);
  if (in_mode == 0) launch_dense<0>(grid, s, x, R, K, W, b, U, act, act_alpha, out, ex);
  else if (in_mode == 1) launch_dense<1>(grid, s, x, R, K, W, b, U, act, act_alpha, out, ex);
  else launch_dense<2>(grid, s, x, R, K, W, b, U, act, act_alpha, out, ex);
  return mp::check_launch("mp_dense_ex_f32");
}

int mp_dense_splitk_workspace_bytes(int64_t R, int64_t U, int splits, size_t* bytes_out_host) {
  MP_REQUIRE(R >= 0 && U >= 1 && splits >= 1 && bytes_out_host, "mp_dense_splitk_workspace_bytes: bad arguments");
  *bytes_out_host = sizeof(float) * static_cast<size_t>(R) * static_cast<size_t>(U) * static_cast<size_t>(splits);
  return MP_OK;
}

int mp_dense_splitk_f32(const float* x, int64_t R, int64_t K, const float* W, const float* b, int64_t U, int act,
                        float act_alpha, int splits, void* ws, size_t ws_bytes, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && K >= 1 && U >= 1 && splits >= 1 && splits <= 64, "mp_dense_splitk_f32: bad sizes");
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST, "mp_dense_splitk_f32: unknown activation %d", act);
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && W && out && ws, "mp_dense_splitk_f32: null pointer");
  // k range of a slice: a multiple of the k tile; trailing slices that would be empty are dropped
  int64_t kchunk = mp::ceil_div(mp::ceil_div(K, splits), BK) * BK;
  const int used = static_cast<int>(mp::ceil_div(K, kchunk));
  MP_REQUIRE(ws_bytes >= sizeof(float) * static_cast<size_t>(R) * static_cast<size_t>(U) * static_cast<size_t>(used),
             "mp_dense_splitk_f32: workspace too small");
  const int64_t gy = mp::ceil_div(R, BM), gx = mp::ceil_div(U, BN);
  MP_REQUIRE(gx <= 65535 && gy < (int64_t{1} << 31), "mp_dense_splitk_f32: grid too large");
  dim3 grid(static_cast<unsigned>(gy), static_cast<unsigned>(gx), static_cast<unsigned>(used));
  hipStream_t s = mp::as_stream(stream);
  launch_dense<0>(grid, s, x, R, K, W, nullptr, U, 0, 0.0f, out,
                  DenseExtra{0, 0.0f, nullptr, nullptr, nullptr, nullptr, static_cast<float*>(ws), kchunk});
  int rc = mp::check_launch("mp_dense_splitk_f32");
  if (rc != MP_OK) return rc;
  dense_splitk_reduce_kernel<<<mp::grid_for(R * U), 256, 0, s>>>(static_cast<const float*>(ws), used, R, U, b, act,
                                                                 act_alpha, out);
  return mp::check_launch("mp_dense_splitk_f32 (reduce)");
}

int mp_activation_f32(int act, float act_alpha, const float* x, int64_t n, float* out, mpStream_t stream) {
  MP_REQUIRE(n >= 0, "mp_activation_f32: bad size");
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST, "mp_activation_f32: unknown activation %d", act);
  if (n == 0) return MP_OK;
  MP_REQUIRE(x && out, "mp_activation_f32: null pointer");
  activation_kernel<<<mp::grid_for(n), 256, 0, mp::as_stream(stream)>>>(act, act_alpha, x, n, out);
  return mp::check_launch("mp_activation_f32");
}

int mp_softmax_rows_f32(const float* x, int64_t R, int64_t C, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && C >= 1, "mp_softmax_rows_f32: bad sizes");
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && out, "mp_softmax_rows_f32: null pointer");
  softmax_rows_kernel<<<mp::grid_for(R * 64), 256, 0, mp::as_stream(stream)>>>(x, R, C, out);
  return mp::check_launch("mp_softmax_rows_f32");
}

}  // extern "C"
