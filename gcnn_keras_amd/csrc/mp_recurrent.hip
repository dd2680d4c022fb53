// Recurrent cells and the edge-network matvec of the NMPN / Set2Set family, as their reference layers use them:
//  * PoolingSet2Set (kgcnn/layers/pool/set2set.py:172-199) calls a STATELESS Keras LSTM on a length-1 sequence in every
//    iteration, i.e. one LSTM step from the zero state: the input GEMM runs on the matrix cores (mp_dense_f32), this file
//    holds the gate arithmetic;
//  * GRUUpdate (kgcnn/layers/conv/mpnn_conv.py:111-210) is one Keras GRUCell step (reset_after = True) on the flat node
//    values: two GEMMs + the combine below;
//  * MatMulMessages (mpnn_conv.py:69-108) is a per-edge matrix-vector product with the (M,F,F) matrices the edge network
//    produced: memory bound on the matrices (16 KB per edge at F = 64), read once with 16-B loads.
#include "mp_common.h"

namespace {

// Keras LSTM gate order along the 4U axis: input, forget, cell candidate, output.
__global__ void lstm_zero_state_kernel(const float* __restrict__ z, int64_t R, int64_t U, int act, int rec_act,
                                       float* __restrict__ out) {
  const int64_t total = R * U;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / U, u = t % U;
    const float* zr = z + r * 4 * U;
    const float i = mp_apply_act(rec_act, 0.0f, zr[u]);
    const float c = i * mp_apply_act(act, 0.0f, zr[2 * U + u]);   // c = f * c0 + i * c~ with c0 = 0
    const float o = mp_apply_act(rec_act, 0.0f, zr[3 * U + u]);
    out[t] = o * mp_apply_act(act, 0.0f, c);
  }
}

// Keras GRUCell, reset_after = True; gate order along the 3U axis: update z, reset r, candidate h.
__global__ void gru_combine_kernel(const float* __restrict__ mx, const float* __restrict__ mh,
                                   const float* __restrict__ h, int64_t R, int64_t U, int act, int rec_act,
                                   float* __restrict__ out) {
  const int64_t total = R * U;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / U, u = t % U;
    const float* x = mx + r * 3 * U;
    const float* y = mh + r * 3 * U;
    const float zg = mp_apply_act(rec_act, 0.0f, x[u] + y[u]);
    const float rg = mp_apply_act(rec_act, 0.0f, x[U + u] + y[U + u]);
    const float hh = mp_apply_act(act, 0.0f, x[2 * U + u] + rg * y[2 * U + u]);
    out[t] = zg * h[t] + (1.0f - zg) * hh;
  }
}

// out[m][r] = sum_c mat[m][r][c] vec[m][c].  LPR = C / 4 lanes share a row (one float4 each), so a wave instruction reads
// 64 / LPR whole rows = 1 KB contiguous; the LPR partial dot products are reduced with log2(LPR) xor shuffles.
template <int LPR>
__global__ __launch_bounds__(256) void batched_matvec_kernel(const float* __restrict__ mat, const float* __restrict__ vec,
                                                             int64_t M, int64_t Ro, float* __restrict__ out) {
  constexpr int C = 4 * LPR;
  constexpr int RPI = 64 / LPR;   // rows per wave instruction
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t m = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; m < M; m += nwaves) {
    const float4 v = *reinterpret_cast<const float4*>(vec + m * C + 4 * c4);
    const float* base = mat + m * Ro * C;
    for (int64_t r0 = 0; r0 < Ro; r0 += RPI) {
      const int64_t r = r0 + sub;
      float acc = 0.0f;
      if (r < Ro) {
        const float4 a = *reinterpret_cast<const float4*>(base + r * C + 4 * c4);
        acc = a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
      }
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (c4 == 0 && r < Ro) out[m * Ro + r] = acc;
    }
  }
}

__global__ void batched_matvec_generic_kernel(const float* __restrict__ mat, const float* __restrict__ vec, int64_t M,
                                              int64_t Ro, int64_t C, float* __restrict__ out) {
  const int64_t total = M * Ro;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t m = t / Ro;
    const float* a = mat + t * C;
    const float* v = vec + m * C;
    float acc = 0.0f;
    for (int64_t c = 0; c < C; ++c) acc += a[c] * v[c];
    out[t] = acc;
  }
}

}  // namespace

extern "C" {

int mp_lstm_zero_state_f32(const float* z, int64_t R, int64_t U, int act, int rec_act, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && U >= 1, "mp_lstm_zero_state_f32: bad sizes");
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST && rec_act >= MP_ACT_LINEAR && rec_act <= MP_ACT_LAST,
             "mp_lstm_zero_state_f32: unknown activation");
  if (R == 0) return MP_OK;
  MP_REQUIRE(z && out, "mp_lstm_zero_state_f32: null pointer");
  lstm_zero_state_kernel<<<mp::grid_for(R * U), 256, 0, mp::as_stream(stream)>>>(z, R, U, act, rec_act, out);
  return mp::check_launch("mp_lstm_zero_state_f32");
}

int mp_gru_combine_f32(const float* mx, const float* mh, const float* h, int64_t R, int64_t U, int act, int rec_act,
                       float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && U >= 1, "mp_gru_combine_f32: bad sizes");
  MP_REQUIRE(act >= MP_ACT_LINEAR && act <= MP_ACT_LAST && rec_act >= MP_ACT_LINEAR && rec_act <= MP_ACT_LAST,
             "mp_gru_combine_f32: unknown activation");
  if (R == 0) return MP_OK;
  MP_REQUIRE(mx && mh && h && out, "mp_gru_combine_f32: null pointer");
  gru_combine_kernel<<<mp::grid_for(R * U), 256, 0, mp::as_stream(stream)>>>(mx, mh, h, R, U, act, rec_act, out);
  return mp::check_launch("mp_gru_combine_f32");
}

int mp_batched_matvec_f32(const float* mat, const float* vec, int64_t M, int64_t Ro, int64_t C, float* out,
                          mpStream_t stream) {
  MP_REQUIRE(M >= 0 && Ro >= 1 && C >= 1, "mp_batched_matvec_f32: bad sizes");
  if (M == 0) return MP_OK;
  MP_REQUIRE(mat && vec && out, "mp_batched_matvec_f32: null pointer");
  hipStream_t s = mp::as_stream(stream);
  const bool aligned = (reinterpret_cast<uintptr_t>(mat) % 16 == 0) && (reinterpret_cast<uintptr_t>(vec) % 16 == 0);
  const unsigned grid = mp::grid_for(M * 64);
  if (aligned && C == 256) batched_matvec_kernel<64><<<grid, 256, 0, s>>>(mat, vec, M, Ro, out);
  else if (aligned && C == 128) batched_matvec_kernel<32><<<grid, 256, 0, s>>>(mat, vec, M, Ro, out);
  else if (aligned && C == 64) batched_matvec_kernel<16><<<grid, 256, 0, s>>>(mat, vec, M, Ro, out);
  else if (aligned && C == 32) batched_matvec_kernel<8><<<grid, 256, 0, s>>>(mat, vec, M, Ro, out);
  else if (aligned && C == 16) batched_matvec_kernel<4><<<grid, 256, 0, s>>>(mat, vec, M, Ro, out);
  else batched_matvec_generic_kernel<<<mp::grid_for(M * Ro), 256, 0, s>>>(mat, vec, M, Ro, C, out);
  return mp::check_launch("mp_batched_matvec_f32");
}

}  // extern "C"
