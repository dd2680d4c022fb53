// Graph readout in one launch: PoolingNodes(sum) + MLP([H, 1]) (kgcnn/literature/PAiNN.py:146-147 with the default
// output_mlp {"units": [128, 1], "activation": ["swish", "linear"]}; kgcnn/layers/pooling.py:215-218, mlp.py), and -
// for the energy + force pass - the reverse of that readout in the same launch: dE_g/dx_n = W0 (W1 * act'(pre_g)) for
// every node n of graph g (kgcnn/model/force.py:159-177 obtains it from the tape).
//
// Why not pool + two Dense launches: on G = 64 rows each of the three is a latency-bound launch of its own (5.7 + 20.9 +
// 19.7 us measured in the PaiNN pipeline, 18 % of its forward).  One wave per graph: the lanes hold the pooled row (K/64
// registers), the first layer's matrix sits in LDS - swizzled, element (k, c) at k H + (c ^ (k & 31)), so both the
// forward's row reads (k fixed, c = lane) and the reverse's column reads (c fixed, k = lane) are bank-conflict free -
// and the pooled / gradient vectors are broadcast with v_readlane.  Sequential node order (tf.math.segment_sum's).
#include "mp_common.h"

namespace {

__device__ __forceinline__ float rl(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int NK, int NH>   // K = 64 NK input features, H = 64 NH hidden units
__global__ __launch_bounds__(256) void pool_mlp2_kernel(const float* __restrict__ x, const int64_t* __restrict__ splits,
                                                        int64_t G, const float* __restrict__ W0,
                                                        const float* __restrict__ b0, int act0, float alpha0,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        float* __restrict__ out, float* __restrict__ g_x) {
  constexpr int K = 64 * NK, H = 64 * NH;
  __shared__ float Ws[K * H];
  // 16-B global loads (all of a thread's requests in flight at once), swizzled scalar LDS stores
  constexpr int NV4 = K * H / 4 / 256;
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  // the wave's first graph: row range requested before the weight matrix, its first 16 rows right behind it - the
  // chain row_splits -> rows then runs beside the staging of W0 instead of behind it
  constexpr bool PREFETCH = NK * NH != 4;   // K = H = 128: every register is taken (a spill = a scratch segment per launch)
  constexpr int PRE = 16;
  int64_t lo0 = 0, hi0 = 0;
  if (wave_global < G) {
    lo0 = splits[wave_global];
    hi0 = splits[wave_global + 1];
  }
  float4 stage[NV4];
#pragma unroll
  for (int j = 0; j < NV4; ++j) stage[j] = reinterpret_cast<const float4*>(W0)[threadIdx.x + j * 256];
  float v0[PREFETCH ? PRE : 1][NK];
#pragma unroll
  for (int u = 0; u < (PREFETCH ? PRE : 1); ++u)
#pragma unroll
    for (int q = 0; q < NK; ++q) v0[u][q] = 0.0f;
  if (PREFETCH && hi0 > lo0) {   // wave-uniform
#pragma unroll
    for (int u = 0; u < PRE; ++u)
#pragma unroll
      for (int q = 0; q < NK; ++q) v0[u][q] = x[(lo0 + u < hi0 ? lo0 + u : hi0 - 1) * K + 64 * q + lane];
  }
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    const int i = 4 * (threadIdx.x + j * 256);
    const int k = i / H, c = i % H;            // c % 4 == 0: the four columns stay inside one aligned group of 32
    float* row = Ws + k * H;
    row[(c + 0) ^ (k & 31)] = stage[j].x;
    row[(c + 1) ^ (k & 31)] = stage[j].y;
    row[(c + 2) ^ (k & 31)] = stage[j].z;
    row[(c + 3) ^ (k & 31)] = stage[j].w;
  }
  float b0v[NH], w1v[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    b0v[h] = b0 ? b0[64 * h + lane] : 0.0f;
    w1v[h] = W1[64 * h + lane];
  }
  const float b1v = b1 ? b1[0] : 0.0f;
  __syncthreads();
  for (int64_t g = wave_global; g < G; g += nwaves) {
    const bool first = g == wave_global;
    const int64_t lo = first ? lo0 : splits[g], hi = first ? hi0 : splits[g + 1];
    float pooled[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) pooled[q] = 0.0f;
    for (int64_t base = lo; base < hi; base += 8) {   // eight rows in flight, added in node order
      float v[8][NK];
      if (PREFETCH && first && base == lo) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < NK; ++q) v[u][q] = v0[PREFETCH ? u : 0][q];
      } else if (PREFETCH && first && base == lo + 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < NK; ++q) v[u][q] = v0[PREFETCH ? 8 + u : 0][q];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < NK; ++q) v[u][q] = (base + u < hi) ? x[(base + u) * K + 64 * q + lane] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < NK; ++q) pooled[q] += (base + u < hi) ? v[u][q] : 0.0f;
    }
    float y[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) y[h] = 0.0f;
#pragma unroll
    for (int q = 0; q < NK; ++q)
      for (int kk = 0; kk < 64; ++kk) {
        const int k = 64 * q + kk;
        const float pk = rl(pooled[q], kk);
#pragma unroll
        for (int h = 0; h < NH; ++h) y[h] = fmaf(pk, Ws[k * H + ((64 * h + lane) ^ (k & 31))], y[h]);
      }
    float o = 0.0f, cv[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float pre = y[h] + b0v[h];
      o += mp_apply_act(act0, alpha0, pre) * w1v[h];
      cv[h] = w1v[h] * mp_act_grad(act0, alpha0, pre);   // dE / d pre-activation of this lane's hidden units
    }
    for (int off = 32; off > 0; off >>= 1) o += __shfl_xor(o, off, 64);
    if (lane == 0) out[g] = o + b1v;
    if (g_x) {
      float gp[NK];
#pragma unroll
      for (int q = 0; q < NK; ++q) gp[q] = 0.0f;
#pragma unroll
      for (int h = 0; h < NH; ++h)
        for (int cc = 0; cc < 64; ++cc) {
          const int c = 64 * h + cc;
          const float cvc = rl(cv[h], cc);
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            const int k = 64 * q + lane;
            gp[q] = fmaf(Ws[k * H + (c ^ (k & 31))], cvc, gp[q]);
          }
        }
      for (int64_t n = lo; n < hi; ++n)
#pragma unroll
        for (int q = 0; q < NK; ++q) g_x[n * K + 64 * q + lane] = gp[q];
    }
  }
}

}  // namespace

extern "C" {

int mp_pool_mlp2_f32(const float* x, const int64_t* node_splits, int64_t G, int K, const float* W0, const float* b0, int H,
                     int act0, float alpha0, const float* W1, const float* b1, float* out, float* g_x, mpStream_t stream) {
  MP_REQUIRE(G >= 0, "mp_pool_mlp2_f32: bad sizes");
  MP_REQUIRE((K == 64 || K == 128) && (H == 64 || H == 128), "mp_pool_mlp2_f32: built for K, H in {64, 128} (got %d, %d)",
             K, H);
  MP_REQUIRE(act0 >= MP_ACT_LINEAR && act0 <= MP_ACT_LAST, "mp_pool_mlp2_f32: unknown activation %d", act0);
  if (G == 0) return MP_OK;
  MP_REQUIRE(x && node_splits && W0 && W1 && out, "mp_pool_mlp2_f32: null pointer");
  const unsigned grid = static_cast<unsigned>(mp::ceil_div(G, 4) < 1024 ? mp::ceil_div(G, 4) : 1024);
  hipStream_t s = mp::as_stream(stream);
#define MP_POOL_MLP2(NK, NH) \
  pool_mlp2_kernel<NK, NH><<<grid, 256, 0, s>>>(x, node_splits, G, W0, b0, act0, alpha0, W1, b1, out, g_x)
  if (K == 64 && H == 64) MP_POOL_MLP2(1, 1);
  else if (K == 64) MP_POOL_MLP2(1, 2);
  else if (H == 64) MP_POOL_MLP2(2, 1);
  else MP_POOL_MLP2(2, 2);
#undef MP_POOL_MLP2
  return mp::check_launch("mp_pool_mlp2_f32");
}

}  // extern "C"
