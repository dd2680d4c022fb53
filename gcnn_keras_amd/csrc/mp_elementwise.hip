// Elementwise / geometry kernels on flat ragged values: the Lazy* layers (kgcnn/layers/modules.py:187-416),
// the geometry pre-step (kgcnn/layers/geom.py lines cited per function) and ChangeTensorType
// (kgcnn/layers/casting.py:79-84).  All HBM-bound streaming kernels: grid-stride, coalesced, one pass.
#include "mp_common.h"

namespace {

__global__ void binary_kernel(int op, const float* __restrict__ a, int64_t sa0, int64_t sa1, int64_t sa2,
                              const float* __restrict__ b, int64_t sb0, int64_t sb1, int64_t sb2, int64_t R,
                              int64_t D1, int64_t D2, float* __restrict__ out) {
  const int64_t total = R * D1 * D2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t j = t % D2;
    const int64_t ri = t / D2;
    const int64_t i = ri % D1;
    const int64_t r = ri / D1;
    const float va = a[r * sa0 + i * sa1 + j * sa2];
    const float vb = b[r * sb0 + i * sb1 + j * sb2];
    out[t] = op == MP_ADD ? va + vb : (op == MP_SUB ? va - vb : va * vb);
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int64_t src_ld, int64_t src_off,
                                 float* __restrict__ dst, int64_t dst_ld, int64_t dst_off, int64_t R, int64_t C) {
  const int64_t total = R * C;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % C, r = t / C;
    dst[r * dst_ld + dst_off + c] = src[r * src_ld + src_off + c];
  }
}

// kgcnn/layers/geom.py:181-193: out = tf.nn.relu(reduce_sum(square(x))) [+eps] ; sqrt ; optional divide_no_nan(1, .)
__global__ void euclidean_norm_kernel(const float* __restrict__ x, int64_t R, int64_t D, int64_t C, int flags,
                                      float* __restrict__ out) {
  const int64_t total = R * C;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const bool invert = flags & 1, add_eps = flags & 2, no_nan = flags & 4, square_norm = flags & 8;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % C, r = t / C;
    float s = 0.0f;
    for (int64_t d = 0; d < D; ++d) {
      const float v = x[(r * D + d) * C + c];
      s += v * v;
    }
    s = fmaxf(s, 0.0f);
    if (add_eps) s += 1e-7f;  // ks.backend.epsilon()
    if (!square_norm) s = sqrtf(s);
    if (invert) s = (no_nan && s == 0.0f) ? 0.0f : 1.0f / s;
    out[t] = s;
  }
}

__global__ void scalar_product_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t R, int64_t D,
                                      int64_t C, float* __restrict__ out) {
  const int64_t total = R * C;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t c = t % C, r = t / C;
    float s = 0.0f;
    for (int64_t d = 0; d < D; ++d) s += a[(r * D + d) * C + c] * b[(r * D + d) * C + c];
    out[t] = s;
  }
}

// kgcnn/layers/geom.py:567-571: gbs = range(bins)/bins*distance ; exp(square(d - offset - gbs) * (-gamma))
__global__ void gauss_basis_kernel(const float* __restrict__ d, int64_t M, int bins, float distance, float gamma,
                                   float offset, float* __restrict__ out) {
  const int64_t total = M * bins;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const float fbins = static_cast<float>(bins);
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int k = static_cast<int>(t % bins);
    const int64_t e = t / bins;
    const float mu = static_cast<float>(k) / fbins * distance;
    const float v = (d[e] - offset) - mu;
    out[t] = expf((v * v) * (gamma * -1.0f));
  }
}

__device__ __forceinline__ float ipow(float x, int n) {
  float r = 1.0f;
  for (int i = 0; i < n; ++i) r *= x;
  return r;
}

// kgcnn/layers/geom.py:772-785: d_scaled = d * (1/cutoff); env = 1/x + a x^(p-1) + b x^p + c x^(p+1), 0 for x >= 1
__global__ void bessel_basis_kernel(const float* __restrict__ d, int64_t M, const float* __restrict__ freq,
                                    int num_radial, float inv_cutoff, int p, float a, float b, float c,
                                    float* __restrict__ out) {
  const int64_t total = M * num_radial;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int k = static_cast<int>(t % num_radial);
    const int64_t e = t / num_radial;
    const float xs = d[e] * inv_cutoff;
    const float xp1 = ipow(xs, p - 1);
    const float env = 1.0f / xs + a * xp1 + b * (xp1 * xs) + c * (xp1 * xs * xs);
    const float cut = xs < 1.0f ? env : 0.0f;
    out[t] = cut * sinf(freq[k] * xs);
  }
}

// kgcnn/layers/geom.py:831-837
__global__ void cos_cutoff_kernel(const float* __restrict__ d, int64_t n, float cutoff, float* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const float scale = 3.14159265358979323846f / cutoff;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = fminf(fmaxf(d[i], -cutoff), cutoff);
    out[i] = (cosf(v * scale) + 1.0f) * 0.5f;
  }
}

// NodePosition -> LazySubtract -> EuclideanNorm(keepdims) [-> EdgeDirectionNormalized] in one pass.
__global__ void edge_geometry_kernel(const float* __restrict__ xyz, int64_t N, const int32_t* __restrict__ recv,
                                     const int32_t* __restrict__ send, int64_t M, float* __restrict__ dist,
                                     float* __restrict__ dir) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < M; e += stride) {
    int64_t i = recv[e], j = send[e];
    i = i < 0 ? 0 : (i >= N ? N - 1 : i);
    j = j < 0 ? 0 : (j >= N ? N - 1 : j);
    const float dx = xyz[i * 3 + 0] - xyz[j * 3 + 0];
    const float dy = xyz[i * 3 + 1] - xyz[j * 3 + 1];
    const float dz = xyz[i * 3 + 2] - xyz[j * 3 + 2];
    const float s = sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 0.0f));
    if (dist) dist[e] = s;
    if (dir) {
      const float inv = s == 0.0f ? 0.0f : 1.0f / s;
      dir[e * 3 + 0] = dx * inv;
      dir[e * 3 + 1] = dy * inv;
      dir[e * 3 + 2] = dz * inv;
    }
  }
}

__device__ __forceinline__ int64_t owner_of(const int64_t* __restrict__ splits, int64_t G, int64_t e) {
  int64_t lo = 0, hi = G;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (splits[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void ragged_to_padded_kernel(const float* __restrict__ values, const int64_t* __restrict__ splits,
                                        int64_t G, int64_t Nmax, int64_t row_elems, float* __restrict__ padded,
                                        float* __restrict__ mask) {
  const int64_t total = G * Nmax * row_elems;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t f = t % row_elems;
    const int64_t gn = t / row_elems;
    const int64_t n = gn % Nmax, g = gn / Nmax;
    const int64_t len = splits[g + 1] - splits[g];
    const bool valid = n < len;
    padded[t] = valid ? values[(splits[g] + n) * row_elems + f] : 0.0f;
    if (mask) mask[t] = valid ? 1.0f : 0.0f;
  }
}

}  // namespace

extern "C" {

int mp_binary_f32(int op, const float* a, const int64_t* sa, const float* b, const int64_t* sb, int64_t R, int64_t D1,
                  int64_t D2, float* out, mpStream_t stream) {
  MP_REQUIRE(op >= MP_ADD && op <= MP_MUL, "mp_binary_f32: unknown op %d", op);
  MP_REQUIRE(R >= 0 && D1 >= 1 && D2 >= 1 && sa && sb, "mp_binary_f32: bad arguments");
  if (R == 0) return MP_OK;
  MP_REQUIRE(a && b && out, "mp_binary_f32: null pointer");
  binary_kernel<<<mp::grid_for(R * D1 * D2), 256, 0, mp::as_stream(stream)>>>(op, a, sa[0], sa[1], sa[2], b, sb[0],
                                                                             sb[1], sb[2], R, D1, D2, out);
  return mp::check_launch("mp_binary_f32");
}

int mp_copy_cols_f32(const float* src, int64_t src_ld, int64_t src_off, float* dst, int64_t dst_ld, int64_t dst_off,
                     int64_t R, int64_t C, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && C >= 0 && src_off >= 0 && dst_off >= 0 && src_off + C <= src_ld && dst_off + C <= dst_ld,
             "mp_copy_cols_f32: column block outside the row");
  if (R == 0 || C == 0) return MP_OK;
  MP_REQUIRE(src && dst, "mp_copy_cols_f32: null pointer");
  copy_cols_kernel<<<mp::grid_for(R * C), 256, 0, mp::as_stream(stream)>>>(src, src_ld, src_off, dst, dst_ld, dst_off,
                                                                          R, C);
  return mp::check_launch("mp_copy_cols_f32");
}

int mp_euclidean_norm_f32(const float* x, int64_t R, int64_t D, int64_t C, int flags, float* out, mpStream_t stream) {
  MP_REQUIRE(R >= 0 && D >= 1 && C >= 1, "mp_euclidean_norm_f32: bad sizes");
  if (R == 0) return MP_OK;
  MP_REQUIRE(x && out, "mp_euclidean_norm_f32: null pointer");
  euclidean_norm_kernel<<<mp::grid_for(R * C), 256, 0, mp::as_stream(stream)>>>(x, R, D, C, flags, out);
  return mp::check_launch("mp_euclidean_norm_f32");
}

int mp_scalar_product_f32(const float* a, const float* b, int64_t R, int64_t D, int64_t C, float* out,
                          mpStream_t stream) {
  MP_REQUIRE(R >= 0 && D >= 1 && C >= 1, "mp_scalar_product_f32: bad sizes");
  if (R == 0) return MP_OK;
  MP_REQUIRE(a && b && out, "mp_scalar_product_f32: null pointer");
  scalar_product_kernel<<<mp::grid_for(R * C), 256, 0, mp::as_stream(stream)>>>(a, b, R, D, C, out);
  return mp::check_launch("mp_scalar_product_f32");
}

int mp_gauss_basis_f32(const float* d, int64_t M, int bins, float distance, float sigma, float offset, float* out,
                       mpStream_t stream) {
  MP_REQUIRE(M >= 0 && bins >= 1 && sigma != 0.0f, "mp_gauss_basis_f32: bad arguments");
  if (M == 0) return MP_OK;
  MP_REQUIRE(d && out, "mp_gauss_basis_f32: null pointer");
  const float gamma = static_cast<float>(1.0 / static_cast<double>(sigma) / static_cast<double>(sigma) / 2.0);
  gauss_basis_kernel<<<mp::grid_for(M * bins), 256, 0, mp::as_stream(stream)>>>(d, M, bins, distance, gamma, offset,
                                                                               out);
  return mp::check_launch("mp_gauss_basis_f32");
}

int mp_bessel_basis_f32(const float* d, int64_t M, const float* frequencies, int num_radial, float cutoff,
                        int envelope_exponent, float* out, mpStream_t stream) {
  MP_REQUIRE(M >= 0 && num_radial >= 1 && cutoff != 0.0f && envelope_exponent >= 0, "mp_bessel_basis_f32: bad arguments");
  if (M == 0) return MP_OK;
  MP_REQUIRE(d && frequencies && out, "mp_bessel_basis_f32: null pointer");
  const int p = envelope_exponent + 1;
  const float a = static_cast<float>(-(p + 1) * (p + 2) / 2.0);
  const float b = static_cast<float>(p * (p + 2));
  const float c = static_cast<float>(-p * (p + 1) / 2.0);
  bessel_basis_kernel<<<mp::grid_for(M * num_radial), 256, 0, mp::as_stream(stream)>>>(
      d, M, frequencies, num_radial, 1.0f / cutoff, p, a, b, c, out);
  return mp::check_launch("mp_bessel_basis_f32");
}

int mp_cos_cutoff_f32(const float* d, int64_t n, float cutoff, float* out, mpStream_t stream) {
  MP_REQUIRE(n >= 0 && cutoff > 0.0f, "mp_cos_cutoff_f32: bad arguments");
  if (n == 0) return MP_OK;
  MP_REQUIRE(d && out, "mp_cos_cutoff_f32: null pointer");
  cos_cutoff_kernel<<<mp::grid_for(n), 256, 0, mp::as_stream(stream)>>>(d, n, cutoff, out);
  return mp::check_launch("mp_cos_cutoff_f32");
}

int mp_edge_geometry_f32(const float* xyz, int64_t N, const int32_t* recv, const int32_t* send, int64_t M, float* dist,
                         float* dir, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0, "mp_edge_geometry_f32: bad sizes");
  if (M == 0) return MP_OK;
  MP_REQUIRE(xyz && recv && send && N > 0 && (dist || dir), "mp_edge_geometry_f32: null pointer / no nodes");
  edge_geometry_kernel<<<mp::grid_for(M), 256, 0, mp::as_stream(stream)>>>(xyz, N, recv, send, M, dist, dir);
  return mp::check_launch("mp_edge_geometry_f32");
}

int mp_ragged_to_padded_f32(const float* values, const int64_t* row_splits, int64_t G, int64_t Nmax, int64_t row_elems,
                            float* padded, float* mask, mpStream_t stream) {
  MP_REQUIRE(G >= 0 && Nmax >= 0 && row_elems >= 1, "mp_ragged_to_padded_f32: bad sizes");
  if (G == 0 || Nmax == 0) return MP_OK;
  MP_REQUIRE(row_splits && padded, "mp_ragged_to_padded_f32: null pointer");
  ragged_to_padded_kernel<<<mp::grid_for(G * Nmax * row_elems), 256, 0, mp::as_stream(stream)>>>(
      values, row_splits, G, Nmax, row_elems, padded, mask);
  return mp::check_launch("mp_ragged_to_padded_f32");
}

}  // extern "C"
