// Numerics probe for gfx950: a 32 x 128 x 128 FP32 GEMM tile computed (1) with v_mfma_f32_32x32x2_f32 and (2) as FP32
// emulated on the bf16 matrix pipe - every operand split exactly into three bf16 pieces (8 + 8 + 8 mantissa bits), the six
// leading cross products on v_mfma_f32_32x32x16_bf16 with FP32 accumulation - both against a float64 host reference.
// Build: hipcc -O3 --offload-arch=gfx950 bf16x3_probe.hip -o bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using floatx16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = static_cast<__bf16>(x);
  const float r1 = x - static_cast<float>(hi);
  mid = static_cast<__bf16>(r1);
  const float r2 = r1 - static_cast<float>(mid);
  lo = static_cast<__bf16>(r2);
}

// A (32, 128) row-major, B (128, 128) row-major, C (32, 128).  One wave.
__global__ void gemm_f32(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
  for (int jb = 0; jb < 4; ++jb) {
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k = 0; k < 128; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c * 128 + k + hh], B[(k + hh) * 128 + 32 * jb + c], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 128 + 32 * jb + c] = acc[r];
  }
}

template <int NPROD>
__global__ void gemm_bf16x3(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
  for (int jb = 0; jb < 4; ++jb) {
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < 128; k0 += 16) {
      bf16x8 a[3], b[3];
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + 8 * hh + i;
        __bf16 h, m, l;
        split3(A[c * 128 + k], h, m, l);
        a[0][i] = h; a[1][i] = m; a[2][i] = l;
        split3(B[k * 128 + 32 * jb + c], h, m, l);
        b[0][i] = h; b[1][i] = m; b[2][i] = l;
      }
      // smallest terms first
      if (NPROD >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      }
      if (NPROD >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 128 + 32 * jb + c] = acc[r];
  }
}

int main() {
  std::vector<float> A(32 * 128), B(128 * 128), C(32 * 128);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX) * 2.0f - 0.3f;     // softplus-like: mostly positive
  for (auto& v : B) v = ((rand() / (float)RAND_MAX) * 2.0f - 1.0f) * 0.15f;
  std::vector<double> ref(32 * 128, 0.0);
  double scale = 0.0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 128; ++j) {
      double s = 0.0;
      for (int k = 0; k < 128; ++k) s += (double)A[i * 128 + k] * (double)B[k * 128 + j];
      ref[i * 128 + j] = s;
      scale = fmax(scale, fabs(s));
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0, rms = 0.0;
    for (size_t i = 0; i < C.size(); ++i) {
      const double e = fabs((double)C[i] - ref[i]);
      worst = fmax(worst, e);
      rms += e * e;
    }
    printf("%-40s max |err| / max |C| = %.3e   rms / max |C| = %.3e\n", name, worst / scale, sqrt(rms / C.size()) / scale);
  };
  hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC); report("v_mfma_f32_32x32x2_f32");
  hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC); report("bf16 x 3 pieces, 6 products");
  hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC); report("bf16 x 2 pieces, 3 products");
  hipLaunchKernelGGL(gemm_bf16x3<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC); report("plain bf16");
  return 0;
}
