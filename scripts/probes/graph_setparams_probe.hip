// Host cost of (a) eight hipLaunchKernel calls, (b) hipGraphLaunch of a captured 8-kernel graph, (c) eight
// hipGraphExecKernelNodeSetParams + hipGraphLaunch - the three ways to issue a never-seen batch's forward.
// Build: hipcc -O2 --offload-arch=gfx950 graph_setparams_probe.hip -o graph_setparams_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Args { float* p; int n; long long pad[30]; };   // ~256 B of kernel arguments, like the engine's descriptors
__global__ void k(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0 && a.n < 0) a.p[0] = 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  float* d; CK(hipMalloc(&d, 1024));
  hipStream_t s; CK(hipStreamCreate(&s));
  Args a{d, 1, {}};
  const int N = 2000;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto t0, auto t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, a);
  CK(hipStreamSynchronize(s));
  auto t0 = now();
  for (int i = 0; i < N; ++i) for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(k, dim3(200 + j), dim3(256), 0, s, a);
  auto t1 = now();
  CK(hipStreamSynchronize(s));
  printf("8 x hipLaunchKernel:                         %.2f us per group (host issue)\n", us(t0, t1) / N);
  // graph with 8 kernel nodes in a chain
  hipGraph_t g; CK(hipGraphCreate(&g, 0));
  std::vector<hipGraphNode_t> nodes(8);
  void* kargs[1] = {&a};
  for (int j = 0; j < 8; ++j) {
    hipKernelNodeParams p{};
    p.func = reinterpret_cast<void*>(k); p.gridDim = dim3(200 + j); p.blockDim = dim3(256); p.sharedMemBytes = 0;
    p.kernelParams = kargs; p.extra = nullptr;
    CK(hipGraphAddKernelNode(&nodes[j], g, j ? &nodes[j - 1] : nullptr, j ? 1 : 0, &p));
  }
  const int E = 4;
  hipGraphExec_t ex[E];
  for (int e = 0; e < E; ++e) CK(hipGraphInstantiate(&ex[e], g, nullptr, nullptr, 0));
  for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ex[i % E], s));
  CK(hipStreamSynchronize(s));
  t0 = now();
  for (int i = 0; i < N; ++i) CK(hipGraphLaunch(ex[i % E], s));
  t1 = now();
  CK(hipStreamSynchronize(s));
  printf("hipGraphLaunch (8 kernels, 4 execs round robin): %.2f us per launch (host issue)\n", us(t0, t1) / N);
  t0 = now();
  for (int i = 0; i < N; ++i) {
    Args b{d, i & 7, {}};
    void* kb[1] = {&b};
    for (int j = 0; j < 8; ++j) {
      hipKernelNodeParams p{};
      p.func = reinterpret_cast<void*>(k); p.gridDim = dim3(180 + (i & 31) + j); p.blockDim = dim3(256);
      p.kernelParams = kb; p.extra = nullptr;
      CK(hipGraphExecKernelNodeSetParams(ex[i % E], nodes[j], &p));
    }
    CK(hipGraphLaunch(ex[i % E], s));
  }
  t1 = now();
  CK(hipStreamSynchronize(s));
  printf("8 x ExecKernelNodeSetParams + hipGraphLaunch:  %.2f us per group (host issue)\n", us(t0, t1) / N);
  return 0;
}
