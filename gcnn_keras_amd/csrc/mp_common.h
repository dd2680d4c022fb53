// Shared host-side helpers of libmpengine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mpengine.h"

namespace mp {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(mpStream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return MP_EHIP;
  }
  return MP_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, never more than the work.
inline unsigned grid_for(int64_t work_items, int block = 256) {
  int64_t blocks = ceil_div(work_items, block);
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return static_cast<unsigned>(blocks);
}

}  // namespace mp

#define MP_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      mp::set_error(__VA_ARGS__);    \
      return MP_EINVAL;              \
    }                                \
  } while (0)

#define MP_HIP(call)                                                          \
  do {                                                                        \
    hipError_t _e = (call);                                                   \
    if (_e != hipSuccess) {                                                   \
      mp::set_error("%s failed: %s", #call, hipGetErrorString(_e));           \
      return MP_EHIP;                                                         \
    }                                                                         \
  } while (0)

// Activation functors shared by the dense epilogue and the standalone activation kernel.
// shifted_softplus restates kgcnn/ops/activ.py:15 with TF's thresholded softplus
// (x > -thr -> x ; x < thr -> exp(x) ; else log1p(exp(x)), thr = log(eps_f32) + 2).
// XCD-aware block order.  The hardware deals workgroups to the 8 XCDs round-robin (workgroup b runs on XCD b % 8), and each
// XCD has its own L2: a kernel whose blocks walk nodes / edges in order touches every molecule from every XCD, so all
// eight L2s fetch the whole working set.  Mapping block b to logical block `mp_xcd_block(b, grid)` gives XCD k one
// contiguous range of logical blocks (of size grid/8, +1 for the first grid%8 XCDs - exactly the blocks it is dealt),
// so each L2 holds one eighth of the batch.  A bijection on [0, grid).
__device__ __forceinline__ unsigned mp_xcd_block(unsigned b, unsigned grid) {
  const unsigned k = b & 7u, i = b >> 3;
  const unsigned q = grid >> 3, r = grid & 7u;
  return k * q + (k < r ? k : r) + i;
}

// OR a thread's MP_FLAG_* bits into the batch's flag word: reduced across the wave first (three ballots), published by
// one lane, and only if the word does not hold the bits yet - an unsorted column raises its bit in nearly every thread,
// and 2.5 M same-address atomics cost more than the rest of the index pass (measured 80 of 111 us).
__device__ __forceinline__ void mp_publish_flags(int32_t* flags, int local_flags) {
  int wave_flags = 0;
#pragma unroll
  for (int bit = 1; bit <= 4; bit <<= 1)
    if (__ballot((local_flags & bit) != 0) != 0ull) wave_flags |= bit;
  if (wave_flags != 0 && (threadIdx.x & 63) == 0) {
    if ((__atomic_load_n(flags, __ATOMIC_RELAXED) & wave_flags) != wave_flags) atomicOr(flags, wave_flags);
  }
}

__device__ __forceinline__ float mp_softplus(float x) {
  const float thr = -13.942385f;  // logf(1.1920929e-07f) + 2
  float ex = expf(x);
  float mid = log1pf(ex);
  return x > -thr ? x : (x < thr ? ex : mid);
}

__device__ __forceinline__ float mp_apply_act(int act, float alpha, float v) {
  switch (act) {
    case MP_ACT_RELU: return fmaxf(v, 0.0f);
    case MP_ACT_SHIFTED_SOFTPLUS: return mp_softplus(v) - 0.6931471805599453f;
    case MP_ACT_SOFTPLUS: return mp_softplus(v);
    case MP_ACT_SWISH: return v * (1.0f / (1.0f + expf(-v)));
    case MP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case MP_ACT_TANH: return tanhf(v);
    case MP_ACT_LEAKY_RELU: return v >= 0.0f ? v : alpha * v;
    case MP_ACT_SOFTPLUS2: return fmaxf(v, 0.0f) + logf(0.5f * expf(-fabsf(v)) + 0.5f);
    case MP_ACT_SELU: return 1.05070098f * (v > 0.0f ? v : 1.67326324f * (expf(v) - 1.0f));
    default: return v;
  }
}

// d act / d pre-activation (used by the elementwise derivative kernel and by the Dense prologue of the reverse pass)
__device__ __forceinline__ float mp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float mp_act_grad(int act, float alpha, float x) {
  switch (act) {
    case MP_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    case MP_ACT_SHIFTED_SOFTPLUS:
    case MP_ACT_SOFTPLUS2:
    case MP_ACT_SOFTPLUS: return mp_sigmoid(x);
    case MP_ACT_SWISH: { const float s = mp_sigmoid(x); return s + x * s * (1.0f - s); }
    case MP_ACT_SIGMOID: { const float s = mp_sigmoid(x); return s * (1.0f - s); }
    case MP_ACT_TANH: { const float t = tanhf(x); return 1.0f - t * t; }
    case MP_ACT_LEAKY_RELU: return x >= 0.0f ? 1.0f : alpha;
    case MP_ACT_SELU: return 1.05070098f * (x > 0.0f ? 1.0f : 1.67326324f * expf(x));
    default: return 1.0f;
  }
}
