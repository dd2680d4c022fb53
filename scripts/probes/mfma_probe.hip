// Micro-probe for gfx950: which work of a wave overlaps with its own (or a sibling wave's) v_mfma_f32_32x32x2_f32 stream?
// Prints shader-clock ticks per MFMA for a chain of MFMAs with N filler instructions of one kind after each MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-result mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using floatx16 = __attribute__((ext_vector_type(16))) float;

enum Kind { K_NONE, K_FMA, K_IADD, K_CNDMASK, K_EXP, K_SALU, K_LDS, K_MOV, K_PKFMA };

template <int KIND, int NFILL>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, float a0, float b0, int i0) {
  __shared__ float lds[2048];
  lds[threadIdx.x] = a0; lds[threadIdx.x + 512] = b0;
  __syncthreads();
  floatx16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float v[8]; int iv[8]; int sv = i0;
  for (int i = 0; i < 8; ++i) { v[i] = a0 + i + threadIdx.x; iv[i] = i0 + i + threadIdx.x; }
  float a = a0 + threadIdx.x, b = b0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 32; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[k & 1]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NFILL; ++j) {
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j % 8]) : "v"(b));
        if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<float2*>(&v[2 * (j % 4)])) : "v"(*reinterpret_cast<float2*>(&v[0])));
        if (KIND == K_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[j % 8]) : "v"(i0));
        if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[j % 8]) : "v"(i0));
        if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j % 8]));
        if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sv));
        if (KIND == K_LDS) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j % 8]) : "v"(static_cast<int>(threadIdx.x * 4)));
        if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[j % 8]) : "v"(i0));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i] + iv[i];
  s += sv;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int NFILL>
void run(const char* name, float* out, unsigned long long* cyc) {
  for (int threads : {256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((probe<KIND, NFILL>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f, 2.0f, 3);
      hipDeviceSynchronize();
    }
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s fill=%2d waves/SIMD=%d : %7.1f ticks per MFMA (per wave)\n", name, NFILL, threads / 256,
           (double)h / (32 * 32));
  }
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 1024); hipMalloc(&cyc, 8);
  run<K_NONE, 0>("none", out, cyc);
  run<K_FMA, 8>("v_fma_f32", out, cyc);
  run<K_PKFMA, 8>("v_pk_fma_f32", out, cyc);
  run<K_IADD, 8>("v_add_u32", out, cyc);
  run<K_CNDMASK, 8>("v_cndmask_b32", out, cyc);
  run<K_MOV, 8>("v_mov_b32", out, cyc);
  run<K_EXP, 4>("v_exp_f32", out, cyc);
  run<K_SALU, 8>("s_add_u32", out, cyc);
  run<K_LDS, 4>("ds_read_b32", out, cyc);
  run<K_FMA, 16>("v_fma_f32", out, cyc);
  run<K_IADD, 16>("v_add_u32", out, cyc);
  run<K_SALU, 16>("s_add_u32", out, cyc);
  return 0;
}
