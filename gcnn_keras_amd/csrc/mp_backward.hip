// Backward helpers for EnergyForceModel (kgcnn/model/force.py:159-186): forces are -dE/dx, i.e. one reverse pass through
// the same layers.  The heavy twins are the forward kernels themselves (gather-backward = segment-sum over the CSR of
// the gathered column, segment-sum-backward = gather by the receiver ids, dense-backward = dense with the transposed
// kernel); this file holds the elementwise derivatives that have no forward counterpart.
#include "mp_common.h"

namespace {

__device__ __forceinline__ float act_grad(int act, float alpha, float x) { return mp_act_grad(act, alpha, x); }

__global__ void activation_grad_kernel(int act, float alpha, const float* __restrict__ pre,
                                       const float* __restrict__ gy, int64_t n, float* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = gy[i] * act_grad(act, alpha, pre[i]);
}

// out (R, D2) = sum over D1 (axis 1)  or  out (R, D1) = sum over D2 (axis 2) of x (R, D1, D2)
__global__ void sum_axis_kernel(const float* __restrict__ x, int64_t R, int64_t D1, int64_t D2, int axis,
                                float* __restrict__ out) {
  const int64_t total = axis == 1 ? R * D2 : R * D1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    float s = 0.0f;
    if (axis == 1) {
      const int64_t j = t % D2, r = t / D2;
      for (int64_t i = 0; i < D1; ++i) s += x[(r * D1 + i) * D2 + j];
    } else {
      const int64_t i = t % D1, r = t / D1;
      for (int64_t j = 0; j < D2; ++j) s += x[(r * D1 + i) * D2 + j];
    }
    out[t] = s;
  }
}

// gradient of EuclideanNorm (geom.py:181-193) on an (R, D, C) view: f(s), s = sum_d x^2
__global__ void euclidean_norm_grad_kernel(const float* __restrict__ x, const float* __restrict__ gy, int64_t R,
                                           int64_t D, int64_t C, int flags, float* __restrict__ gx) {
  const int64_t total = R * C;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const bool invert = flags & 1, add_eps = flags & 2, square_norm = flags & 8;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % C, r = t / C;
    float s = 0.0f;
    for (int64_t d = 0; d < D; ++d) {
      const float v = x[(r * D + d) * C + c];
      s += v * v;
    }
    if (add_eps) s += 1e-7f;
    float dfds;  // d f / d s
    if (s <= 0.0f) dfds = 0.0f;  // TF yields inf/NaN at the cusp; the engine returns a zero sub-gradient there
    else if (!square_norm && !invert) dfds = 0.5f / sqrtf(s);
    else if (!square_norm && invert) dfds = -0.5f / (s * sqrtf(s));
    else if (square_norm && !invert) dfds = 1.0f;
    else dfds = -1.0f / (s * s);
    const float g = gy[t] * dfds * 2.0f;
    for (int64_t d = 0; d < D; ++d) gx[(r * D + d) * C + c] = g * x[(r * D + d) * C + c];
  }
}

__device__ __forceinline__ float ipow(float x, int n) {
  float r = 1.0f;
  for (int i = 0; i < n; ++i) r *= x;
  return r;
}

// d/dd of BesselBasisLayer (geom.py:772-785): sum_k gy[e,k] * (env'(x) sin(f x) + env(x) f cos(f x)) / cutoff
__global__ void bessel_grad_kernel(const float* __restrict__ d, int64_t M, const float* __restrict__ freq,
                                   int num_radial, float inv_cutoff, int p, float a, float b, float c,
                                   const float* __restrict__ gy, float* __restrict__ gd) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < M; e += stride) {
    const float xs = d[e] * inv_cutoff;
    float acc = 0.0f;
    if (xs < 1.0f && xs > 0.0f) {
      const float xp2 = ipow(xs, p - 2);
      const float xp1 = xp2 * xs;
      const float env = 1.0f / xs + a * xp1 + b * (xp1 * xs) + c * (xp1 * xs * xs);
      const float denv = -1.0f / (xs * xs) + a * (p - 1) * xp2 + b * p * xp1 + c * (p + 1) * (xp1 * xs);
      for (int k = 0; k < num_radial; ++k) {
        const float f = freq[k];
        acc += gy[e * num_radial + k] * (denv * sinf(f * xs) + env * f * cosf(f * xs));
      }
    }
    gd[e] = acc * inv_cutoff;
  }
}

// d/dd of GaussBasisLayer (geom.py:567-571)
__global__ void gauss_grad_kernel(const float* __restrict__ d, int64_t M, int bins, float distance, float gamma,
                                  float offset, const float* __restrict__ gy, float* __restrict__ gd) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const float fbins = static_cast<float>(bins);
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < M; e += stride) {
    float acc = 0.0f;
    for (int k = 0; k < bins; ++k) {
      const float mu = static_cast<float>(k) / fbins * distance;
      const float v = (d[e] - offset) - mu;
      acc += gy[e * bins + k] * expf(-gamma * v * v) * (-2.0f * gamma * v);
    }
    gd[e] = acc;
  }
}

// d/dd of CosCutOffEnvelope (geom.py:831-837)
__global__ void cos_cutoff_grad_kernel(const float* __restrict__ d, int64_t n, float cutoff,
                                       const float* __restrict__ gy, float* __restrict__ gd) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const float scale = 3.14159265358979323846f / cutoff;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = d[i];
    gd[i] = (v > -cutoff && v < cutoff) ? gy[i] * (-0.5f * scale * sinf(v * scale)) : 0.0f;
  }
}

}  // namespace

extern "C" {

int mp_activation_grad_f32(int act, float act_alpha, const float* pre, const float* gy, int64_t n, float* out,
                           mpStream_t stream) {
  MP_REQUIRE(n >= 0 && act >= MP_ACT_LINEAR && act <= MP_ACT_LAST, "mp_activation_grad_f32: bad arguments");
  if (n == 0) return MP_OK;
  MP_REQUIRE(pre && gy && out, "mp_activation_grad_f32: null pointer");
  activation_grad_kernel<<<mp::grid_for(n), 256, 0, mp::as_stream(stream)>>>(act, act_alpha, pre, gy, n, out);
  return mp::check_launch("mp_activation_grad_f32");
}

int mp_sum_axis_f32(const float* x, int64_t R, int64_t D1, int64_t D2, int axis, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && D1 >= 1 && D2 >= 1 && (axis == 1 || axis == 2), "mp_sum_axis_f32: bad arguments");
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && out, "mp_sum_axis_f32: null pointer");
  sum_axis_kernel<<<mp::grid_for(axis == 1 ? R * D2 : R * D1), 256, 0, mp::as_stream(stream)>>>(x, R, D1, D2, axis,
                                                                                              out);
  return mp::check_launch("mp_sum_axis_f32");
}

int mp_euclidean_norm_grad_f32(const float* x, const float* gy, int64_t R, int64_t D, int64_t C, int flags, float* gx,
                               mpStream_t stream) {
  MP_REQUIRE(R >= 0 && D >= 1 && C >= 1, "mp_euclidean_norm_grad_f32: bad sizes");
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && gy && gx, "mp_euclidean_norm_grad_f32: null pointer");
  euclidean_norm_grad_kernel<<<mp::grid_for(R * C), 256, 0, mp::as_stream(stream)>>>(x, gy, R, D, C, flags, gx);
  return mp::check_launch("mp_euclidean_norm_grad_f32");
}

int mp_bessel_basis_grad_f32(const float* d, int64_t M, const float* frequencies, int num_radial, float cutoff,
                             int envelope_exponent, const float* gy, float* gd, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && num_radial >= 1 && cutoff != 0.0f && envelope_exponent >= 1, "mp_bessel_basis_grad_f32: bad arguments");
  if (M == 0) return MP_OK;
  MP_REQUIRE(d && frequencies && gy && gd, "mp_bessel_basis_grad_f32: null pointer");
  const int p = envelope_exponent + 1;
  const float a = static_cast<float>(-(p + 1) * (p + 2) / 2.0);
  const float b = static_cast<float>(p * (p + 2));
  const float c = static_cast<float>(-p * (p + 1) / 2.0);
  bessel_grad_kernel<<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(d, M, frequencies, num_radial, 1.0f / cutoff,
                                                                          p, a, b, c, gy, gd);
  return mp::check_launch("mp_bessel_basis_grad_f32");
}

int mp_gauss_basis_grad_f32(const float* d, int64_t M, int bins, float distance, float sigma, float offset,
                            const float* gy, float* gd, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && bins >= 1 && sigma != 0.0f, "mp_gauss_basis_grad_f32: bad arguments");
  if (M == 0) return MP_OK;
  MP_REQUIRE(d && gy && gd, "mp_gauss_basis_grad_f32: null pointer");
  const float gamma = static_cast<float>(1.0 / static_cast<double>(sigma) / static_cast<double>(sigma) / 2.0);
  gauss_grad_kernel<<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(d, M, bins, distance, gamma, offset, gy, gd);
  return mp::check_launch("mp_gauss_basis_grad_f32");
}

int mp_cos_cutoff_grad_f32(const float* d, int64_t n, float cutoff, const float* gy, float* gd, mpStream_t stream) {
  MP_REQUIRE(n >= 0 && cutoff > 0.0f, "mp_cos_cutoff_grad_f32: bad arguments");
  if (n == 0) return MP_OK;
  MP_REQUIRE(d && gy && gd, "mp_cos_cutoff_grad_f32: null pointer");
  cos_cutoff_grad_kernel<<<mp::grid_for(n), 256, 0, mp::as_stream(stream)>>>(d, n, cutoff, gy, gd);
  return mp::check_launch("mp_cos_cutoff_grad_f32");
}

}  // extern "C"
