// Fused PaiNN pipeline (kgcnn/literature/PAiNN.py:100-155) and its reverse pass for forces (kgcnn/model/force.py:159-186).
//
// The forward of one PaiNN block (kgcnn/layers/conv/painn_conv.py:97-115 PAiNNconv, :201-214 PAiNNUpdate) runs as
//
//   GEMM  h1 = z W1 + b1                       GEMM  s = swish(h1) Wphi + bphi          (mp_dense_ex_f32, prologue swish)
//   EDGE  z' = z + sum_e s_j w_e |1 ,  v' = v + sum_e (s_j w_e)|2 (x) v_j + (s_j w_e)|3 (x) r_ij      (painn_message_kernel)
//   GEMM  [v_u | v_v] = v' [Wu | Wv]            NODE  c = [z' | ||v_v||],  prod = <v_u, v_v>          (painn_update_pre)
//   GEMM  h2 = c Wd + bd                        GEMM  a = swish(h2) Wa + ba
//   NODE  z'' = z' + prod a_sv + a_ss ,  v'' = v' + a_vv (x) v_u                                       (painn_update_post)
//
// i.e. five GEMMs on the FP32 matrix cores and three memory-bound kernels per block instead of the ~30 primitive launches
// of the layer path; none of the (M,3F) / (M,3,F) edge tensors of the reference exists.  The reverse pass mirrors it
// kernel for kernel (transposed-weight GEMMs with the activation derivative fused as a prologue; the edge kernel runs
// SENDER-parallel over the CSR of column 1, because the message's inputs s_j, v_j live at the sender: their gradients
// are then register accumulations of one wave, no atomics, fixed order).  Distances enter through the radial basis and
// the unit vectors only, so the edge kernel reduces dE/d(rbf_e) to ONE scalar per edge with the basis derivative
// rbf'(d_e) prepared by stage 0:  dE/dd_e = sum_f g_w[f] (rbf'_e Ww)[f];  a last node-parallel kernel turns
// (dE/dd_e, dE/dr_ij) into dE/dx over both CSRs.
//
// All kernels are HBM/L2-bound streaming or gather kernels except the GEMMs; the per-edge filter (K = B = 20) is far too
// thin for MFMA tiles and runs on the VALU as packed FP32 FMAs with the lane's weight columns in registers.
#include <cstdlib>
#include <mutex>

#include "mp_common.h"
#include "mp_edge_prepare.h"

namespace {

constexpr int F = 128;
using floatx2 = __attribute__((ext_vector_type(2))) float;

// acc.lo += x.lo * w, acc.hi += x.hi * w with ONE weight register for both halves: v_pk_fma_f32 whose weight operand is a
// register pair (w_k, w_k+1) and op_sel picks the same half for both result lanes - x is a scalar pair (two v_readlane
// results).  The compiler's own packing of this pattern duplicates every weight into a (w, w) pair, and the doubled
// register count costs these latency-bound kernels the occupancy they live on (measured: slower).  Same fused arithmetic.
__device__ __forceinline__ void pk_fma_wlo(floatx2& acc, floatx2 x, floatx2 wpair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(x), "v"(wpair));
}
__device__ __forceinline__ void pk_fma_whi(floatx2& acc, floatx2 x, floatx2 wpair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(x), "v"(wpair));
}

__device__ __forceinline__ float ipow(float x, int n) {
  float r = 1.0f;
  for (int i = 0; i < n; ++i) r *= x;
  return r;
}

// v_readlane_b32 of a float register (lane index wave-uniform)
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Wave-wide sum on the VALU's DPP path, result wave-uniform: two quad permutes, row_half_mirror, row_mirror (after which
// every lane of a 16-lane row holds the row's sum) and four v_readlane for the rows.  The xor-shuffle form above goes
// through the LDS crossbar (ds_bpermute): six dependent round trips per value - the message reverse kernel reduces four
// values per edge, 96 ds_bpermute per 4-edge chunk before, none now.  Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// LDS reads of DIFFERENT kinds (4-B, two-address, 16-B) in flight together are not safe behind a partial
// `s_waitcnt lgkmcnt(N)` on this part: the consumer has been seen with the old register contents (the address the read was
// issued with) in the lanes written last (48-63) - single edge contributions missing, different from run to run.  hipcc
// assumes issue-order retirement: it consumes each value behind a partial wait of its own, merges 4-B reads into two-address
// ones and copies loaded registers early, whatever `s_waitcnt` builtins, scheduling barriers or "+v" operand ties the source
// puts after the loads.  So the mixed-width groups of the tile kernels are ONE asm statement each: the reads and a full
// `lgkmcnt(0)` inside, results as early-clobber outputs - nothing can be consumed, merged or copied in between.  Uniform
// groups (all 4-B, all 16-B) stay with the compiler.
using floatx4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using floatx16 = __attribute__((ext_vector_type(16))) float;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)p));
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The tail of an MFMA burst is allowed to finish (64 idle cycles after the last issue; one 32x32x16 MFMA takes 32) before
// the first LDS read that follows it: without this the reverse tile kernel lost single reads in lanes 48-63 (the register
// kept what it held before - an operand piece of the burst), 1 call in 30 to every call depending on the build; with it
// 500 of 500 calls are bit-identical (scripts/stress_painn_bwd.py).  The accumulators pass through the statement, so
// neither the MFMAs nor the reads move across it.
__device__ __forceinline__ void mfma_drained(floatx16 (&acc)[3]) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
}
// one edge of a group: three rows F floats apart at `rows` (+ one value at `one`, when ONE) + a 16-B entry at `entry`
struct EdgeReads {
  float a[3], b[3], one;
  floatx4 e;
};
// forward: s_j (3 parts) at `sa`, v_j (3 components) at `va`, {r_ij, envelope} at `pa`, for two edges
__device__ __forceinline__ void lds_group_fwd(unsigned sa0, unsigned va0, unsigned pa0, unsigned sa1, unsigned va1,
                                              unsigned pa1, EdgeReads& x, EdgeReads& y) {
  asm volatile(
      "ds_read_b32 %0, %14\n\tds_read_b32 %1, %14 offset:512\n\tds_read_b32 %2, %14 offset:1024\n\t"
      "ds_read_b32 %3, %15\n\tds_read_b32 %4, %15 offset:512\n\tds_read_b32 %5, %15 offset:1024\n\t"
      "ds_read_b128 %6, %16\n\t"
      "ds_read_b32 %7, %17\n\tds_read_b32 %8, %17 offset:512\n\tds_read_b32 %9, %17 offset:1024\n\t"
      "ds_read_b32 %10, %18\n\tds_read_b32 %11, %18 offset:512\n\tds_read_b32 %12, %18 offset:1024\n\t"
      "ds_read_b128 %13, %19\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(x.a[0]), "=&v"(x.a[1]), "=&v"(x.a[2]), "=&v"(x.b[0]), "=&v"(x.b[1]), "=&v"(x.b[2]), "=&v"(x.e),
        "=&v"(y.a[0]), "=&v"(y.a[1]), "=&v"(y.a[2]), "=&v"(y.b[0]), "=&v"(y.b[1]), "=&v"(y.b[2]), "=&v"(y.e)
      : "v"(sa0), "v"(va0), "v"(pa0), "v"(sa1), "v"(va1), "v"(pa1)
      : "memory");
}
// reverse: g_z at `za`, g_dv (3 components) at `ga`, {r_ij, envelope} at `pa` (and envelope' at `ea`, when ENV), two edges
template <bool ENV>
__device__ __forceinline__ void lds_group_bwd(unsigned za0, unsigned ga0, unsigned pa0, unsigned ea0, unsigned za1,
                                              unsigned ga1, unsigned pa1, unsigned ea1, EdgeReads& x, EdgeReads& y) {
  if constexpr (ENV) {
    asm volatile(
        "ds_read_b32 %0, %12\n\tds_read_b32 %1, %13\n\tds_read_b32 %2, %13 offset:512\n\tds_read_b32 %3, %13 offset:1024\n\t"
        "ds_read_b128 %4, %14\n\tds_read_b32 %5, %15\n\t"
        "ds_read_b32 %6, %16\n\tds_read_b32 %7, %17\n\tds_read_b32 %8, %17 offset:512\n\tds_read_b32 %9, %17 offset:1024\n\t"
        "ds_read_b128 %10, %18\n\tds_read_b32 %11, %19\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x.one), "=&v"(x.b[0]), "=&v"(x.b[1]), "=&v"(x.b[2]), "=&v"(x.e), "=&v"(x.a[0]), "=&v"(y.one), "=&v"(y.b[0]),
          "=&v"(y.b[1]), "=&v"(y.b[2]), "=&v"(y.e), "=&v"(y.a[0])
        : "v"(za0), "v"(ga0), "v"(pa0), "v"(ea0), "v"(za1), "v"(ga1), "v"(pa1), "v"(ea1)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b32 %0, %10\n\tds_read_b32 %1, %11\n\tds_read_b32 %2, %11 offset:512\n\tds_read_b32 %3, %11 offset:1024\n\t"
        "ds_read_b128 %4, %12\n\t"
        "ds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %14 offset:512\n\tds_read_b32 %8, %14 offset:1024\n\t"
        "ds_read_b128 %9, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x.one), "=&v"(x.b[0]), "=&v"(x.b[1]), "=&v"(x.b[2]), "=&v"(x.e), "=&v"(y.one), "=&v"(y.b[0]), "=&v"(y.b[1]),
          "=&v"(y.b[2]), "=&v"(y.e)
        : "v"(za0), "v"(ga0), "v"(pa0), "v"(za1), "v"(ga1), "v"(pa1)
        : "memory");
    x.a[0] = 0.0f;
    y.a[0] = 0.0f;
  }
}
static_assert(F == 128, "the asm read groups address rows 512 B apart");

// x[l] + x[l ^ 32] in every lane (lower half's value first)
__device__ __forceinline__ float sum_halves(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}

// Four wave-wide sums at once (results wave-uniform): gfx950's v_permlane32_swap / v_permlane16_swap fold two registers
// per instruction - (a, b) and (c, d) over the wave halves, then the two results over the row pairs - leaving row 0 with
// a's, row 1 with c's, row 2 with b's, row 3 with d's per-lane-of-row partials; four DPP steps finish the rows and four
// v_readlane hand the totals out: 14 vector instructions instead of 4 x 11.  Fixed order: deterministic.
__device__ __forceinline__ void wave_sum4_uniform(float& a, float& b, float& c, float& d) {
  auto fold32 = [](float x, float y) {   // lanes 0-31: x[l] + x[l + 32] ; lanes 32-63: y[l - 32] + y[l]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  };
  const float p = fold32(a, b), q = fold32(c, d);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // rows: a | c | b | d
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  a = readlane_f(v, 0);
  c = readlane_f(v, 16);
  b = readlane_f(v, 32);
  d = readlane_f(v, 48);
}

// ---------------------------------------------------------------------------------------------------- stage 0
// Edge continuation of the index pass: EdgeDirectionNormalized (geom.py:331-378, divide_no_nan).
struct PainnEdgeExtra {
  static constexpr bool active = true;
  float* rij;         // (M,3)
  __device__ __forceinline__ void operator()(int64_t e, float dx, float dy, float dz, float s) const {
    const float inv = s == 0.0f ? 0.0f : 1.0f / s;
    rij[e * 3 + 0] = dx * inv;
    rij[e * 3 + 1] = dy * inv;
    rij[e * 3 + 2] = dz * inv;
  }
};

// BesselBasisLayer (geom.py:772-785) with the arithmetic of bessel_basis_kernel (csrc/mp_elementwise.hip), its derivative
// as in bessel_grad_kernel (csrc/mp_backward.hip), CosCutOffEnvelope (geom.py:831-837) and its derivative: one thread per
// (edge, basis function) - the sin / cos evaluations are the cost, and M x B threads fill the chip where M threads with a
// serial loop over B do not (measured 12-17 us for the one-thread-per-edge form at 20 k edges).
struct PainnBasisArgs {
  const float* dist;  // (M)
  int64_t M;
  const float* freq;  // (B)
  int B;
  float inv_cutoff;
  int p;
  float a, b, c;
  float* rbf;         // (M,B)
  float* rbfd;        // (M,B) or null
  float cos_cutoff;   // <= 0: no envelope
  float* env;         // (M) or null
  float* envd;        // (M) or null
};

__global__ __launch_bounds__(256) void painn_basis_kernel(PainnBasisArgs q) {
  const int64_t total = q.M * q.B;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / q.B;
    const int k = static_cast<int>(t % q.B);
    const float s = q.dist[e];
    const float xs = s * q.inv_cutoff;
    const float xp1 = ipow(xs, q.p - 1);
    const float envp = 1.0f / xs + q.a * xp1 + q.b * (xp1 * xs) + q.c * (xp1 * xs * xs);
    const float cut = xs < 1.0f ? envp : 0.0f;
    const float f = q.freq[k];
    const float sn = sinf(f * xs);
    q.rbf[t] = cut * sn;
    if (q.rbfd != nullptr) {
      float dv = 0.0f;
      if (xs < 1.0f && xs > 0.0f) {
        const float xp2 = ipow(xs, q.p - 2);
        const float xq1 = xp2 * xs;
        const float env_in = 1.0f / xs + q.a * xq1 + q.b * (xq1 * xs) + q.c * (xq1 * xs * xs);
        const float denv = -1.0f / (xs * xs) + q.a * (q.p - 1) * xp2 + q.b * q.p * xq1 + q.c * (q.p + 1) * (xq1 * xs);
        dv = (denv * sn + env_in * f * cosf(f * xs)) * q.inv_cutoff;
      }
      q.rbfd[t] = dv;
    }
    if (k == 0 && q.env != nullptr) {
      const float scale = 3.14159265358979323846f / q.cos_cutoff;
      const float v = fminf(fmaxf(s, -q.cos_cutoff), q.cos_cutoff);
      q.env[e] = (cosf(v * scale) + 1.0f) * 0.5f;
      if (q.envd != nullptr) q.envd[e] = (s > -q.cos_cutoff && s < q.cos_cutoff) ? -0.5f * scale * sinf(s * scale) : 0.0f;
    }
  }
}

struct PainnNodeInit {
  const void* numbers;   // (N) node numbers, float32 or int64 (Keras Embedding casts to int32)
  int numbers_i64;
  const float* emb;      // (vocab, F)
  int vocab;
  int64_t N;
  float v_init;          // EquivariantInitialize constant (zeros / eps / ones / const)
  float* z0;             // (N, F)
  float* v0;             // (N, 3, F)
};

// Workgroups [0, node_blocks) initialise the node state, the rest run the edge pass.
template <bool LDS_SPLITS>
__global__ __launch_bounds__(256) void painn_stage0_kernel(PainnNodeInit ni, mp_prep::EdgePrepArgs p, PainnEdgeExtra ex,
                                                           int node_blocks) {
  if (static_cast<int>(blockIdx.x) < node_blocks) {
    const int64_t total = ni.N * F;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < total;
         t += static_cast<int64_t>(node_blocks) * 256) {
      const int64_t n = t / F;
      const int f = static_cast<int>(t % F);
      const int zi = ni.numbers_i64 ? static_cast<int>(static_cast<const int64_t*>(ni.numbers)[n])
                                    : static_cast<int>(static_cast<const float*>(ni.numbers)[n]);
      ni.z0[t] = (zi >= 0 && zi < ni.vocab) ? ni.emb[static_cast<int64_t>(zi) * F + f] : 0.0f;
      ni.v0[(n * 3 + 0) * F + f] = ni.v_init;
      ni.v0[(n * 3 + 1) * F + f] = ni.v_init;
      ni.v0[(n * 3 + 2) * F + f] = ni.v_init;
    }
  } else {
    mp_prep::edge_prepare_body<LDS_SPLITS, true, PainnEdgeExtra>(p, static_cast<int64_t>(blockIdx.x) - node_blocks,
                                                                  static_cast<int64_t>(gridDim.x) - node_blocks, ex);
  }
}

// ---------------------------------------------------------------------------------------------------- message, forward
struct PainnMsgArgs {
  const float* s;       // (N, 3F)  phi(swish(dense1 z)), node side
  const float* v;       // (N, 3, F) equivariant features (block input)
  const float* rbf;     // (M, B)
  const float* env;     // (M) or null
  const float* rij;     // (M, 3)
  const float* Ww;      // (B, 3F)
  const float* bw;      // (3F) or null
  const int32_t* ptr;   // (N+1) CSR over receivers
  const int32_t* perm;  // (M) or null
  const int32_t* send;  // (M) original edge order
  const float* z_in;    // (N, F) or null: when given the outputs are the residual sums z_in + ds, v + dv
  float* ds;            // (N, F)
  float* dv;            // (N, 3, F)
  int64_t N, M;
  int B;
};

// TWO waves per receiving node, one per half of the 128 features (lane = one feature): receiver-parallel over the CSR,
// sequential in edge order = the order tf.math.segment_sum uses after the stable sort, so the result is deterministic.
// At MD17 / QM9 batch sizes the kernel is latency bound (1344 nodes x ~15 edges), so it is built to keep loads in flight:
//  * the edge ids and sender ids of up to 64 edges of the node are fetched with ONE coalesced load each into a VGPR
//    (lane = edge) and handed out with v_readlane - no dependent scalar-load chain perm -> send -> row per edge;
//  * eight senders' rows (s: 3 values, v: 3 values per lane) are requested before the first is used;
//  * half the features per wave halves the registers (60 filter weights) and doubles the waves per node.
// Wave-uniform data (rbf row, r_ij, envelope) is fetched with scalar loads.
template <int BT>
__global__ __launch_bounds__(256) void painn_message_kernel(PainnMsgArgs a) {
  constexpr int MAXB = BT > 0 ? BT : 32;
  constexpr int EC = 8;
  const int lane = threadIdx.x & 63;
  const int B = BT > 0 ? BT : a.B;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  // XCD-aware: consecutive nodes (one molecule's rows of s / v and its edges) stay in one XCD's L2
  const int64_t wave0 = (static_cast<int64_t>(mp_xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x) >> 6;
  const int half = __builtin_amdgcn_readfirstlane(static_cast<int>(wave0 & 1));   // nwaves is even: fixed per wave
  const int f0 = half * 64 + lane;                                                 // this lane's feature
  floatx2 wp[3][MAXB / 2];   // (w[p][2 q], w[p][2 q + 1]): the weight operand of the packed FMAs below
  float bias[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int q = 0; q < MAXB / 2; ++q) {
      wp[p][q].x = 2 * q < B ? a.Ww[(2 * q) * 3 * F + p * F + f0] : 0.0f;
      wp[p][q].y = 2 * q + 1 < B ? a.Ww[(2 * q + 1) * 3 * F + p * F + f0] : 0.0f;
    }
    bias[p] = a.bw ? a.bw[p * F + f0] : 0.0f;
  }
  for (int64_t wv = wave0; wv < 2 * a.N; wv += nwaves) {
    const int n = __builtin_amdgcn_readfirstlane(static_cast<int>(wv >> 1));
    int e_lo = a.ptr[n], e_hi = a.ptr[n + 1];
    e_lo = e_lo < 0 ? 0 : (e_lo > a.M ? static_cast<int>(a.M) : e_lo);
    e_hi = e_hi < e_lo ? e_lo : (e_hi > a.M ? static_cast<int>(a.M) : e_hi);
    float ds = 0.0f, dv[3] = {0.0f, 0.0f, 0.0f};
    for (int base = e_lo; base < e_hi; base += 64) {
      const int cnt = e_hi - base < 64 ? e_hi - base : 64;
      int my_r = 0, my_j = 0;
      if (lane < cnt) {
        my_r = a.perm ? a.perm[base + lane] : base + lane;
        my_j = a.send[my_r];
        my_j = my_j < 0 ? 0 : (my_j >= a.N ? static_cast<int>(a.N) - 1 : my_j);
      }
      for (int u0 = 0; u0 < cnt; u0 += EC) {
        float sj[EC][3], vj[EC][3], meta[EC];
#pragma unroll
        for (int u = 0; u < EC; ++u) {
          const int slot = u0 + u < cnt ? u0 + u : cnt - 1;   // the tail repeats the last edge (loads only)
          const int j = __builtin_amdgcn_readlane(my_j, slot);
          const int r = __builtin_amdgcn_readlane(my_r, slot);
          const float* srow = a.s + static_cast<int64_t>(j) * 3 * F + f0;
          const float* vrow = a.v + static_cast<int64_t>(j) * 3 * F + f0;
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            sj[u][p] = srow[p * F];
            vj[u][p] = vrow[p * F];
          }
          // the edge's wave-uniform data in ONE vector load: lanes 0..B-1 its rbf row, B..B+2 r_ij, B+3 the envelope;
          // handed out with v_readlane below (per-edge scalar loads would each wait for a round trip in front of the FMAs)
          const float* mp_ = lane < B ? a.rbf + static_cast<int64_t>(r) * B + lane
                                      : (lane < B + 3 || !a.env) ? a.rij + static_cast<int64_t>(r) * 3 + (lane < B + 3 ? lane - B : 0)
                                                                 : a.env + r;
          meta[u] = lane < B + 4 ? *mp_ : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < EC; u += 2) {
          if (u0 + u < cnt) {   // wave-uniform
            // the filters of TWO edges as register pairs (f_u[p], f_u+1[p]): one packed FMA per basis function, part and
            // edge pair with the scalar pair (rbf_k of edge u, of edge u + 1) - pk_fma_wlo / whi, same fused arithmetic, k
            // ascending; the tail repeats the last edge (loads) and drops the second half below
            floatx2 ff[3] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
            for (int q = 0; q < MAXB / 2; ++q) {
              if (2 * q < B) {
                const floatx2 x0 = {readlane_f(meta[u], 2 * q), readlane_f(meta[u + 1], 2 * q)};
                const floatx2 x1 = {readlane_f(meta[u], 2 * q + 1), readlane_f(meta[u + 1], 2 * q + 1)};
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                  pk_fma_wlo(ff[p], x0, wp[p][q]);
                  pk_fma_whi(ff[p], x1, wp[p][q]);
                }
              }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if (u0 + u + h < cnt) {   // wave-uniform
                const float envv = a.env ? readlane_f(meta[u + h], B + 3) : 1.0f;
                float sw[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                  float wv_ = (h ? ff[p].y : ff[p].x) + bias[p];   // Dense: x W + b
                  if (a.env) wv_ *= envv;                          // lay_mult_cutoff([w, envelope])
                  sw[p] = sj[u + h][p] * wv_;                      // lay_mult([s, w])
                }
                ds += sw[0];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                  const float rk = readlane_f(meta[u + h], B + k);
                  dv[k] += sw[1] * vj[u + h][k] + sw[2] * rk;   // (sw2 * v_j) + (sw3 * r_ij)
                }
              }
            }
          }
        }
      }
    }
    if (a.z_in) {   // residual adds of PAiNN.py:126-127 fused: z + ds, v + dv
      ds += a.z_in[static_cast<int64_t>(n) * F + f0];
#pragma unroll
      for (int k = 0; k < 3; ++k) dv[k] += a.v[(static_cast<int64_t>(n) * 3 + k) * F + f0];
    }
    a.ds[static_cast<int64_t>(n) * F + f0] = ds;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.dv[(static_cast<int64_t>(n) * 3 + k) * F + f0] = dv[k];
  }
}

// ---------------------------------------------------------------------------------------------------- message, MFMA
// The same message step with the per-edge filter  w_e = rbf_e Ww + bw  (K = B + 1 <= 32 -> 3F = 384 columns, 15 kflop per
// edge: 85 % of the step's arithmetic) on the matrix pipe instead of 60 FMAs per lane and edge on the VALU.  FP32-exact
// as in csrc/mp_cfconv.hip: both operands are split into three bf16 pieces (8 + 8 + 8 mantissa bits, every difference exact
// in FP32) and the six leading cross products run on v_mfma_f32_32x32x16_bf16 with FP32 accumulation (the dropped products
// are below 2^-24 of |a||b|, the rounding of one FP32 product).
//
// Work split: a workgroup of four waves serves a PAIR of receiving nodes, wave q the feature quarter 32 q .. 32 q + 31 of
// all three filter parts.  One MFMA tile = 32 edge rows x 32 features: the accumulator leaves rows {0-3, 8-11, 16-19,
// 24-27} in lanes 0-31 and rows {4-7, 12-15, ...} in lanes 32-63, sixteen rows per lane - so lane half h is given the
// edges of receiver 2 pair + h (row m of the A operand = edge (m & 3) + 4 (m >> 3) of the half (m >> 2) & 1), and each half
// walks ITS receiver's edges in register order = edge order (the order of tf.math.segment_sum after the stable sort:
// deterministic, no atomics, no cross-lane step at all).  A lane = one feature of one receiver: the sender rows s_j, v_j
// arrive as 128-B pieces (32 lanes x 4 B), eight edges in flight per lane.  Rows beyond a receiver's edge count carry a
// zero A row (bias slot included), so their filter - and message - is exactly 0; their loads re-read a valid edge.
// The weights Ww | bw are split into the wave's 18 B-operand registers x 4 once per wave (workgroups are persistent over
// pairs); the basis rows are split per tile.

__device__ __forceinline__ void split3_into(float x, bf16x8& hi, bf16x8& mid, bf16x8& lo, int i) {
  const __bf16 p0 = static_cast<__bf16>(x);
  const float r1 = x - static_cast<float>(p0);
  const __bf16 p1 = static_cast<__bf16>(r1);
  const float r2 = r1 - static_cast<float>(p1);
  hi[i] = p0;
  mid[i] = p1;
  lo[i] = static_cast<__bf16>(r2);
}

// acc += A B as the six leading products of the three-piece split, smallest first
__device__ __forceinline__ floatx16 mfma_bf16x3(const bf16x8 (&qa)[3], const bf16x8 (&qb)[3], floatx16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[2], qb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0], qb[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1], qb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1], qb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0], qb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0], qb[0], acc, 0, 0, 0);
  return acc;
}

// B operands of the filter GEMM for this lane: column f0 of part p, k = 16 ks + 8 (lane >> 5) + i; row B is the bias.
// Every load is unconditional from a clamped address and masked afterwards (a conditional load costs a branch and a
// full wait per load in hipcc's output).
template <int BT>
__device__ __forceinline__ void painn_filter_operands(const float* __restrict__ Ww, const float* __restrict__ bw, int B,
                                                      int f0, int hh, bf16x8 (&wb)[3][2][3]) {
  const float* bwp = bw != nullptr ? bw : Ww;       // any readable address; masked by has_b
  const float has_b = bw != nullptr ? 1.0f : 0.0f;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const float bias = bwp[p * F + f0] * has_b;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * ks + 8 * hh + i;
        if (BT > 0 && 16 * ks + i > BT) {             // beyond the bias slot for both lane halves: compile-time zero
          split3_into(0.0f, wb[p][ks][0], wb[p][ks][1], wb[p][ks][2], i);
          continue;
        }
        const int kc = k < B ? k : B - 1;
        const float w = Ww[static_cast<int64_t>(kc) * 3 * F + p * F + f0];
        // masks, not selects: a select lets the compiler sink the load into a branch
        const float x = w * (k < B ? 1.0f : 0.0f) + bias * (k == B ? 1.0f : 0.0f);
        split3_into(x, wb[p][ks][0], wb[p][ks][1], wb[p][ks][2], i);
      }
  }
}

// A operand of one basis tile for this lane: row = edge `er` (valid or not), k = 16 ks + 8 hh + i; slot k == B holds 1.
template <int BT>
__device__ __forceinline__ void painn_basis_load(const float* __restrict__ rbf, int B, int64_t er, int hh, float (&ra)[2][8]) {
  if constexpr (BT == 20) {   // 80-B rows: 16-B pieces
    const float4* rp = reinterpret_cast<const float4*>(rbf + er * 20);
    const float4 t0 = rp[2 * hh], t1 = rp[2 * hh + 1], t2 = rp[4];
    ra[0][0] = t0.x; ra[0][1] = t0.y; ra[0][2] = t0.z; ra[0][3] = t0.w;
    ra[0][4] = t1.x; ra[0][5] = t1.y; ra[0][6] = t1.z; ra[0][7] = t1.w;
    ra[1][0] = t2.x; ra[1][1] = t2.y; ra[1][2] = t2.z; ra[1][3] = t2.w;
    ra[1][4] = 1.0f; ra[1][5] = 0.0f; ra[1][6] = 0.0f; ra[1][7] = 0.0f;
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * ks + 8 * hh + i;
        const float x = rbf[er * B + (k < B ? k : B - 1)];
        ra[ks][i] = k < B ? x : (k == B ? 1.0f : 0.0f);
      }
  }
}
template <int BT>
__device__ __forceinline__ void painn_basis_split(const float (&ra)[2][8], bool valid, int hh, bf16x8 (&qa)[2][3]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bool on = valid;
      if (BT == 20 && ks == 1) on = valid && hh == 0;   // k = 24..31 of the upper lane half: zero
      split3_into(on ? ra[ks][i] : 0.0f, qa[ks][0], qa[ks][1], qa[ks][2], i);
    }
}

template <int BT, bool PERM, bool ENV>
__global__ __launch_bounds__(256, 2) void painn_message_mfma_kernel(PainnMsgArgs a) {
  constexpr int EB = 4;     // sender rows in flight per lane: 4 x 10 loads, well inside the 6-bit vmcnt (63)
  const int lane = threadIdx.x & 63;
  const int fq = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int hh = lane >> 5, c = lane & 31;
  const int B = BT > 0 ? BT : a.B;
  const int f0 = 32 * fq + c;
  const int rm = (c & 3) + 4 * (c >> 3), hm = (c >> 2) & 1;   // A row c = edge rm of lane half hm
  const int M = static_cast<int>(a.M), N = static_cast<int>(a.N);
  const int npairs = (N + 1) >> 1;
  int pr = static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x));
  // first pair's edge range requested before the weights, so both fly together
  int n = 2 * pr + hh;
  int nc = n < N ? n : N - 1;
  int e_lo = a.ptr[nc], e_hi = a.ptr[nc + 1];
  bf16x8 wb[3][2][3];
  painn_filter_operands<BT>(a.Ww, a.bw, B, f0, hh, wb);
  for (; pr < npairs; pr += static_cast<int>(gridDim.x)) {
    const bool has = n < N;
    e_lo = e_lo < 0 ? 0 : (e_lo > M ? M : e_lo);
    e_hi = e_hi < e_lo ? e_lo : (e_hi > M ? M : e_hi);
    const int cnt = has ? e_hi - e_lo : 0;
    const int lo0 = __builtin_amdgcn_readlane(e_lo, 0), lo1 = __builtin_amdgcn_readlane(e_lo, 32);
    const int cnt0 = __builtin_amdgcn_readlane(cnt, 0), cnt1 = __builtin_amdgcn_readlane(cnt, 32);
    const int maxc = cnt0 > cnt1 ? cnt0 : cnt1;
    const int n_this = n;
    // next pair's edge range: in flight under this pair's work
    n = 2 * (pr + static_cast<int>(gridDim.x)) + hh;
    nc = n < N ? n : N - 1;
    const int nx_lo = a.ptr[nc], nx_hi = a.ptr[nc + 1];
    float ds = 0.0f, dv[3] = {0.0f, 0.0f, 0.0f};
    for (int cb = 0; cb < maxc; cb += 16) {
      // ---- phase A: sender ids of this lane half's 16 edges (register order; past the end: the receiver's last edge,
      //      whose filter row is 0) and the basis tile's row c - all requested together
      int jr[16], er[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int slot = e_lo + (cb + r < cnt ? cb + r : (cnt > 0 ? cnt - 1 : 0));
        slot = slot < M ? slot : M - 1;
        er[r] = PERM ? a.perm[slot] : slot;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) jr[r] = a.send[er[r]];
      const int lo_m = hm ? lo1 : lo0, cnt_m = hm ? cnt1 : cnt0;
      const bool rv = cb + rm < cnt_m;
      int slot_m = lo_m + (rv ? cb + rm : 0);
      slot_m = slot_m < M ? slot_m : M - 1;
      const int64_t er_m = PERM ? a.perm[slot_m] : slot_m;
      float ra[2][8];
      painn_basis_load<BT>(a.rbf, B, er_m, hh, ra);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) jr[r] = jr[r] < 0 ? 0 : (jr[r] >= N ? N - 1 : jr[r]);
      // ---- phase B: the first batch of sender rows goes out before the matrix work, so that it flies under it
      float sj[EB][3], vj[EB][3], rr[EB][3], ev[EB];
      auto request = [&](int bt) {
#pragma unroll
        for (int u = 0; u < EB; ++u) {

          const float* srow = a.s + static_cast<int64_t>(jr[bt + u]) * 3 * F + f0;
          const float* vrow = a.v + static_cast<int64_t>(jr[bt + u]) * 3 * F + f0;
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            sj[u][p] = srow[p * F];
            vj[u][p] = vrow[p * F];
            rr[u][p] = a.rij[static_cast<int64_t>(er[bt + u]) * 3 + p];
          }
          ev[u] = ENV ? a.env[er[bt + u]] : 1.0f;
        }
      };
      request(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase C: filter tile on the matrix pipe
      bf16x8 qa[2][3];
      painn_basis_split<BT>(ra, rv, hh, qa);
      floatx16 acc[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
        acc[p] = mfma_bf16x3(qa[0], wb[p][0], acc[p]);
        acc[p] = mfma_bf16x3(qa[1], wb[p][1], acc[p]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase D: messages in edge order
#pragma unroll
      for (int bt = 0; bt < 16; bt += EB) {
        if (bt > 0) {
          if (bt >= maxc - cb) break;   // wave-uniform: neither half has edges in this batch
          request(bt);
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          float sw[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            float wv_ = acc[p][bt + u];                   // Dense: x W + b (bias = k slot B)
            if (ENV) wv_ *= ev[u];                         // lay_mult_cutoff([w, envelope])
            sw[p] = sj[u][p] * wv_;                        // lay_mult([s, w])
          }
          ds += sw[0];
#pragma unroll
          for (int k = 0; k < 3; ++k) dv[k] += sw[1] * vj[u][k] + sw[2] * rr[u][k];   // (sw2 * v_j) + (sw3 * r_ij)
        }
      }
    }
    if (has) {
      if (a.z_in) {   // residual adds of PAiNN.py:126-127 fused: z + ds, v + dv
        ds += a.z_in[static_cast<int64_t>(n_this) * F + f0];
#pragma unroll
        for (int k = 0; k < 3; ++k) dv[k] += a.v[(static_cast<int64_t>(n_this) * 3 + k) * F + f0];
      }
      a.ds[static_cast<int64_t>(n_this) * F + f0] = ds;
#pragma unroll
      for (int k = 0; k < 3; ++k) a.dv[(static_cast<int64_t>(n_this) * 3 + k) * F + f0] = dv[k];
    }
    e_lo = nx_lo;
    e_hi = nx_hi;
  }
}

template <int BT>
void launch_message_mfma(const PainnMsgArgs& a, unsigned blocks, hipStream_t st) {
  if (a.perm != nullptr) {
    if (a.env != nullptr) painn_message_mfma_kernel<BT, true, true><<<blocks, 256, 0, st>>>(a);
    else painn_message_mfma_kernel<BT, true, false><<<blocks, 256, 0, st>>>(a);
  } else {
    if (a.env != nullptr) painn_message_mfma_kernel<BT, false, true><<<blocks, 256, 0, st>>>(a);
    else painn_message_mfma_kernel<BT, false, false><<<blocks, 256, 0, st>>>(a);
  }
}

// ---------------------------------------------------------------------------------------------------- message, tiles
// The production form of the matrix-pipe message step: node TILES staged in LDS.  Measured on the kernel above (config 3):
// what bounds the message step is not arithmetic but the vector-memory pipe - every edge pulls its sender's s_j and v_j
// rows (3 KB) through the CU's L1, 62 MB per launch, and every wave re-reads the filter weights - with the phases of all
// waves in step (all load, then all compute).  Batched molecular graphs have no edge between graphs, so the senders of a
// range of receivers lie inside the receivers' own graph: a workgroup takes a tile = a few consecutive receivers of ONE
// graph and stages
//   * the s and v rows of the whole graph (1.5 KB + 1.5 KB per node, contiguous in memory) by LDS-DMA
//     (global_load_lds_dwordx4: 1 KB per wave instruction, no registers, each byte crosses L1 once per tile), and
//   * the tile's edge data - basis rows, unit vectors, sender ids, envelope: contiguous too, the edge list being
//     receiver-sorted - through registers;
// after ONE barrier the tile runs from LDS: filter on the matrix pipe exactly as above (A rows from the LDS basis rows,
// B operands = the wave's slice of a pre-split bf16 image of Ww | bw, mp_painn_filter_pack_f32, 18 x 16 B per lane), the
// sender rows by ds_read_b32 (lane = feature: conflict-free).  The tile table (receivers, graph node range, edge range per
// tile) is built once per bound batch by the host (gcnn_keras_amd/fused_painn.py).  Unsorted edge lists (perm) and graphs
// whose rows do not fit LDS keep the gather kernels above.
struct PainnTileArgs {
  const float* s;        // (N, 3F)
  const float* v;        // (N, 3, F)
  const float* rbf;      // (M, B)
  const float* env;      // (M) or null
  const float* rij;      // (M, 3)
  const void* wimg;      // mp_painn_filter_pack_f32 image
  const int32_t* ptr;    // (N+1) CSR over receivers (edges in receiver order: no perm)
  const int32_t* send;   // (M)
  const int32_t* tiles;  // (T, 8): r_lo, r_hi, s_lo, s_hi, e_lo, e_hi, -, -
  const float* z_in;     // (N, F) or null
  float* ds;             // (N, F)
  float* dv;             // (N, 3, F)
  int64_t N, M;
  int B, ntiles, max_rows, max_edges;
};

constexpr int PAINN_FILTER_IMAGE_BYTES = 4 * 3 * 2 * 3 * 64 * 16;   // [quarter][part][k block][piece][lane] x 16 B

__global__ __launch_bounds__(256) void painn_filter_pack_kernel(const float* __restrict__ Ww, const float* __restrict__ bw,
                                                                int B, bf16x8* __restrict__ image) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;     // one 16-B entry per thread
  if (t >= 4 * 3 * 2 * 3 * 64) return;
  const int lane = t & 63, pc = (t >> 6) % 3, ks = (t / 192) & 1, p = (t / 384) % 3, fq = t / 1152;
  const int hh = lane >> 5, col = p * F + 32 * fq + (lane & 31);
  bf16x8 piece[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = 16 * ks + 8 * hh + i;
    const float x = k < B ? Ww[static_cast<int64_t>(k) * 3 * F + col] : ((k == B && bw != nullptr) ? bw[col] : 0.0f);
    split3_into(x, piece[0], piece[1], piece[2], i);
  }
  image[t] = piece[pc];
}

// LDS regions of a tile, each padded to whole DMA instructions (1 KB for 16-B pieces, 256 B for dword pieces)
__host__ __device__ __forceinline__ int painn_pad(int floats, int unit) { return ((floats + unit - 1) / unit) * unit; }
__host__ __device__ __forceinline__ int painn_tile_lds_floats(int max_rows, int max_edges, int B, bool env) {
  return 2 * painn_pad(max_rows * 3 * F, 256) + painn_pad(max_edges * B, 256) + painn_pad(max_edges * 3, 64) +
         painn_pad(max_edges, 64) * (env ? 2 : 1) + painn_pad(max_edges * 4, 64);
}

template <int BT, bool ENV>
__global__ __launch_bounds__(256, 2) void painn_message_tile_kernel(PainnTileArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63;
  const int fq = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int hh = lane >> 5, c = lane & 31;
  const int B = BT > 0 ? BT : a.B;
  const int f0 = 32 * fq + c;
  const int rm = (c & 3) + 4 * (c >> 3), hm = (c >> 2) & 1;   // A row c = edge rm of lane half hm
  const int node_part = painn_pad(a.max_rows * 3 * F, 256);
  float* const Ss = lds;                                      // [rows][3F]
  float* const Vs = Ss + node_part;                           // [rows][3][F]
  float* const Rb = Vs + node_part;                           // [edges][B]
  float* const Rj = Rb + painn_pad(a.max_edges * B, 256);     // [edges][3]
  int* const Sd = reinterpret_cast<int*>(Rj + painn_pad(a.max_edges * 3, 64));   // [edges] sender ids (global)
  float* const Ev = reinterpret_cast<float*>(Sd + painn_pad(a.max_edges, 64));   // [edges] envelope
  float4* const P4 = reinterpret_cast<float4*>(Ev + (ENV ? painn_pad(a.max_edges, 64) : 0));   // [edges] {r_ij, envelope}
  // B operands: this wave's slice of the pre-split image
  bf16x8 wb[3][2][3];
  {
    const bf16x8* img = reinterpret_cast<const bf16x8*>(a.wimg) + static_cast<size_t>(fq) * (3 * 2 * 3 * 64) + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wb[p][ks][pc] = img[((p * 2 + ks) * 3 + pc) * 64];
  }
  for (int tile = static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x)); tile < a.ntiles; tile += static_cast<int>(gridDim.x)) {
    const int4 d0 = *reinterpret_cast<const int4*>(a.tiles + static_cast<int64_t>(tile) * 8);
    const int2 d1 = *reinterpret_cast<const int2*>(a.tiles + static_cast<int64_t>(tile) * 8 + 4);
    const int N = static_cast<int>(a.N), M = static_cast<int>(a.M);
    int r_lo = __builtin_amdgcn_readfirstlane(d0.x), r_hi = __builtin_amdgcn_readfirstlane(d0.y);
    int s_lo = __builtin_amdgcn_readfirstlane(d0.z), s_hi = __builtin_amdgcn_readfirstlane(d0.w);
    int e_lo = __builtin_amdgcn_readfirstlane(d1.x), e_hi = __builtin_amdgcn_readfirstlane(d1.y);
    // a malformed table must not fault: everything is clamped into the arrays and into the LDS regions
    r_lo = r_lo < 0 ? 0 : (r_lo > N ? N : r_lo);
    r_hi = r_hi < r_lo ? r_lo : (r_hi > N ? N : r_hi);
    if (r_hi - r_lo > 62) r_hi = r_lo + 62;
    s_lo = s_lo < 0 ? 0 : (s_lo > N - 1 ? N - 1 : s_lo);
    s_hi = s_hi <= s_lo ? s_lo + 1 : (s_hi > N ? N : s_hi);
    if (s_hi - s_lo > a.max_rows) s_hi = s_lo + a.max_rows;
    e_lo = e_lo < 0 ? 0 : (e_lo > M ? M : e_lo);
    e_hi = e_hi < e_lo ? e_lo : (e_hi > M ? M : e_hi);
    if (e_hi - e_lo > a.max_edges) e_hi = e_lo + a.max_edges;
    const int rows = s_hi - s_lo, ne = e_hi - e_lo;
    if (tile != static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x))) __syncthreads();   // the previous tile's readers are done
    // ---- node rows by LDS-DMA: two contiguous blocks of rows * 1536 B, 1-KB chunks dealt to the four waves; a chunk's
    //      lanes past the block re-read its last 16 B (the LDS regions are padded to whole chunks)
    {
      const int nfl = rows * 3 * F;                           // floats per block
      const int chunks = (nfl + 255) / 256;
      const float* srcS = a.s + static_cast<int64_t>(s_lo) * 3 * F;
      const float* srcV = a.v + static_cast<int64_t>(s_lo) * 3 * F;
      for (int ch = fq; ch < chunks; ch += 4) {
        int off = ch * 256 + lane * 4;
        off = off < nfl - 4 ? off : nfl - 4;
        __builtin_amdgcn_global_load_lds(srcS + off, (__attribute__((address_space(3))) void*)(Ss + ch * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(srcV + off, (__attribute__((address_space(3))) void*)(Vs + ch * 256), 16, 0, 0);
      }
    }
    // ---- edge data, contiguous blocks too (the edge list is receiver-sorted): basis rows in 16-B pieces where the rows
    //      allow it, unit vectors / sender ids / envelope as dword DMA (256 B per wave instruction).  Nothing here waits: a
    //      rolled load -> ds_write loop costs one memory round trip per iteration.
    {
      const int nfl = ne * B;
      if ((B & 3) == 0) {
        const float* src = a.rbf + static_cast<int64_t>(e_lo) * B;
        for (int ch = fq; ch * 256 < nfl; ch += 4) {
          int off = ch * 256 + lane * 4;
          off = off < nfl - 4 ? off : nfl - 4;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void*)(Rb + ch * 256), 16, 0, 0);
        }
      } else {
        const float* src = a.rbf + static_cast<int64_t>(e_lo) * B;
        for (int ch = fq; ch * 64 < nfl; ch += 4) {
          int off = ch * 64 + lane;
          off = off < nfl ? off : nfl - 1;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void*)(Rb + ch * 64), 4, 0, 0);
        }
      }
      const float* srcr = a.rij + static_cast<int64_t>(e_lo) * 3;
      for (int ch = fq; ch * 64 < ne * 3; ch += 4) {
        int off = ch * 64 + lane;
        off = off < ne * 3 ? off : ne * 3 - 1;
        __builtin_amdgcn_global_load_lds(srcr + off, (__attribute__((address_space(3))) void*)(Rj + ch * 64), 4, 0, 0);
      }
      for (int ch = fq; ch * 64 < ne; ch += 4) {
        int off = ch * 64 + lane;
        off = off < ne ? off : ne - 1;
        __builtin_amdgcn_global_load_lds(a.send + e_lo + off, (__attribute__((address_space(3))) void*)(Sd + ch * 64), 4, 0, 0);
        if (ENV)
          __builtin_amdgcn_global_load_lds(a.env + e_lo + off, (__attribute__((address_space(3))) void*)(Ev + ch * 64), 4, 0, 0);
      }
    }
    // edge ranges of the tile's receivers: lane i holds ptr[r_lo + i]
    int ptrv = a.ptr[(r_lo + lane) <= r_hi ? r_lo + lane : r_hi];
    ptrv = ptrv < e_lo ? e_lo : (ptrv > e_hi ? e_hi : ptrv);
    __builtin_amdgcn_s_waitcnt(0);      // DMA + stores issued by this wave have landed
    __syncthreads();
    // sender ids -> float offsets of the senders' staged rows (clamped into the staged range), once per tile
    // and the edge's unit vector + envelope as ONE 16-B entry (one LDS read per edge instead of four)
    for (int i = threadIdx.x; i < ne; i += 256) {
      int j = Sd[i] - s_lo;
      j = j < 0 ? 0 : (j >= rows ? rows - 1 : j);
      Sd[i] = j * 3 * F;
      P4[i] = make_float4(Rj[3 * i], Rj[3 * i + 1], Rj[3 * i + 2], ENV ? Ev[i] : 1.0f);
    }
    __syncthreads();
    const int npairs = (r_hi - r_lo + 1) >> 1;
    for (int pr = 0; pr < npairs; ++pr) {
      const int lo0 = __builtin_amdgcn_readlane(ptrv, 2 * pr), lo1 = __builtin_amdgcn_readlane(ptrv, 2 * pr + 1);
      const bool two = r_lo + 2 * pr + 1 < r_hi;               // wave-uniform
      const int hi1 = two ? __builtin_amdgcn_readlane(ptrv, 2 * pr + 2) : lo1;
      const int cnt0 = lo1 - lo0, cnt1 = hi1 - lo1;
      const int maxc = cnt0 > cnt1 ? cnt0 : cnt1;
      const int n = r_lo + 2 * pr + hh;
      const bool has = hh == 0 || two;
      const int own_lo = (hh ? lo1 : lo0) - e_lo, cnt = hh ? cnt1 : cnt0;    // local edge offsets
      const int lo_m = (hm ? lo1 : lo0) - e_lo, cnt_m = hm ? cnt1 : cnt0;
      float ds = 0.0f, dv[3] = {0.0f, 0.0f, 0.0f};
      for (int cb = 0; cb < maxc; cb += 16) {
        // A operand: row c of the tile = edge rm of receiver hm
        const bool rv = cb + rm < cnt_m;
        const int le_m = rv ? lo_m + cb + rm : 0;
        float ra[2][8];
        if constexpr (BT == 20) {
          const float4* rp = reinterpret_cast<const float4*>(Rb + le_m * 20);
          const float4 t0 = rp[2 * hh], t1 = rp[2 * hh + 1], t2 = rp[4];
          ra[0][0] = t0.x; ra[0][1] = t0.y; ra[0][2] = t0.z; ra[0][3] = t0.w;
          ra[0][4] = t1.x; ra[0][5] = t1.y; ra[0][6] = t1.z; ra[0][7] = t1.w;
          ra[1][0] = t2.x; ra[1][1] = t2.y; ra[1][2] = t2.z; ra[1][3] = t2.w;
          ra[1][4] = 1.0f; ra[1][5] = 0.0f; ra[1][6] = 0.0f; ra[1][7] = 0.0f;
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int k = 16 * ks + 8 * hh + i;
              const float x = Rb[le_m * B + (k < B ? k : B - 1)];
              ra[ks][i] = k < B ? x : (k == B ? 1.0f : 0.0f);
            }
        }
        bf16x8 qa[2][3];
        painn_basis_split<BT>(ra, rv, hh, qa);
        floatx16 acc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
          acc[p] = mfma_bf16x3(qa[0], wb[p][0], acc[p]);
          acc[p] = mfma_bf16x3(qa[1], wb[p][1], acc[p]);
        }
        mfma_drained(acc);
        // messages in edge order.  LDS reads are issued in groups of two edges - the step's eight sender offsets first, then
        // per group 2 x (3 + 3 + 1) = 14 read instructions - and each group is complete before its arithmetic: one asm
        // statement with a full lgkmcnt(0) inside (lds_group_fwd; the note at its definition).  With the step's 72 reads
        // issued up front behind partial waits, registers were consumed before their data had arrived (wrong v' components
        // in lanes 48-63, varying from run to run); read-next-to-use on the other hand exposes one LDS round trip per value.  Rows past the
        // receiver's edge count have a zero filter; their reads repeat edge 0 of the tile.
#pragma unroll
        for (int bt = 0; bt < 16; bt += 8) {
          if (bt < maxc - cb) {     // wave-uniform
            int le[8], jb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              le[u] = cb + bt + u < cnt ? own_lo + cb + bt + u : 0;
              jb[u] = Sd[le[u]];                                 // float offset of the sender's staged rows (made above)
            }
            lds_wait_all();                                       // eight 4-B reads: uniform
            const unsigned ss_base = lds_addr(Ss) + 4u * f0, vs_base = lds_addr(Vs) + 4u * f0, p4_base = lds_addr(P4);
#pragma unroll
            for (int g2 = 0; g2 < 8; g2 += 2) {
              EdgeReads rd[2];
              lds_group_fwd(ss_base + 4u * jb[g2], vs_base + 4u * jb[g2], p4_base + 16u * le[g2], ss_base + 4u * jb[g2 + 1],
                            vs_base + 4u * jb[g2 + 1], p4_base + 16u * le[g2 + 1], rd[0], rd[1]);
              float sj[2][3], vj[2][3];
              float4 re[2];
#pragma unroll
              for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                  sj[u][p] = rd[u].a[p];
                  vj[u][p] = rd[u].b[p];
                }
                re[u] = make_float4(rd[u].e[0], rd[u].e[1], rd[u].e[2], rd[u].e[3]);
              }
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const float rr[3] = {re[u].x, re[u].y, re[u].z};
                float sw[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                  float wv_ = acc[p][bt + g2 + u];                // Dense: x W + b (bias = k slot B)
                  if (ENV) wv_ *= re[u].w;                        // lay_mult_cutoff([w, envelope])
                  sw[p] = sj[u][p] * wv_;                         // lay_mult([s, w])
                }
                ds += sw[0];
#pragma unroll
                for (int k = 0; k < 3; ++k) dv[k] += sw[1] * vj[u][k] + sw[2] * rr[k];   // (sw2 * v_j) + (sw3 * r_ij)
              }
            }
          }
        }
      }
      if (has) {
        if (a.z_in) {   // residual adds of PAiNN.py:126-127 fused: z + ds, v + dv (the node's own v row is staged)
          ds += a.z_in[static_cast<int64_t>(n) * F + f0];
          const int own = n - s_lo;
          const int ob = (own < 0 ? 0 : (own >= rows ? rows - 1 : own)) * 3 * F + f0;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            dv[k] += Vs[ob + k * F];
        }
        a.ds[static_cast<int64_t>(n) * F + f0] = ds;
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dv[(static_cast<int64_t>(n) * 3 + k) * F + f0] = dv[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- message, backward
struct PainnMsgBwdArgs {
  const float* s;        // (N, 3F)   saved forward s
  const float* v;        // (N, 3, F) saved block-input v
  const float* rbf;      // (M, B)
  const float* rbfd;     // (M, B)   d rbf / d d
  const float* env;      // (M) or null
  const float* envd;     // (M) or null
  const float* rij;      // (M, 3)
  const float* Ww;       // (B, 3F)
  const float* bw;       // (3F) or null
  const int32_t* ptr1;   // (N+1) CSR over SENDERS
  const int32_t* perm1;  // (M) sender-sorted position -> original edge, or null
  const int32_t* recv;   // (M) original edge order
  const float* g_ds;     // (N, F)    upstream gradient of ds (= dE/dz')
  const float* g_dv;     // (N, 3, F) upstream gradient of dv (= dE/dv')
  float* g_s;            // (N, 3F)   out
  float* g_v;            // (N, 3, F) out = g_dv (residual path) + message path; null: not needed (first block)
  float* g_d;            // (2, M)    dE/dd_e per feature half: written (accumulate = 0) or added to
  float* g_rij;          // (2, M, 3) dE/dr_ij per feature half
  int accumulate;
  int64_t N, M;
  int B;
};

// Two waves per SENDING node j (one per feature half, lane = one feature): s_j and v_j are the wave's own values, the
// upstream gradients are gathered from the receivers of j's edges (edge / receiver ids of up to 64 edges held lane-wise,
// four receivers' rows in flight), the gradients w.r.t. s_j and v_j accumulate in registers in edge order
// (deterministic) and are written once.  Per edge the wave also reduces four scalars (dE/dd_e, dE/dr_ij) over its 64
// features; the two halves of a node write separate slices g_d[half], g_rij[half] (each edge has exactly one sender, so
// within a slice the per-edge accumulators are plain read-modify-writes); the geometry kernel adds the slices.
template <int BT>
__global__ __launch_bounds__(256) void painn_message_bwd_kernel(PainnMsgBwdArgs a) {
  constexpr int MAXB = BT > 0 ? BT : 32;
  constexpr int EC = 4;
  const int lane = threadIdx.x & 63;
  const int B = BT > 0 ? BT : a.B;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  // XCD-aware: consecutive nodes (one molecule's rows of s / v and its edges) stay in one XCD's L2
  const int64_t wave0 = (static_cast<int64_t>(mp_xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x) >> 6;
  const int half = __builtin_amdgcn_readfirstlane(static_cast<int>(wave0 & 1));
  const int f0 = half * 64 + lane;
  float* const g_d = a.g_d + static_cast<int64_t>(half) * a.M;
  float* const g_rij = a.g_rij + static_cast<int64_t>(half) * a.M * 3;
  floatx2 wp[3][MAXB / 2];   // (w[p][2 q], w[p][2 q + 1]): the weight operand of the packed FMAs below
  float bias[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int q = 0; q < MAXB / 2; ++q) {
      wp[p][q].x = 2 * q < B ? a.Ww[(2 * q) * 3 * F + p * F + f0] : 0.0f;
      wp[p][q].y = 2 * q + 1 < B ? a.Ww[(2 * q + 1) * 3 * F + p * F + f0] : 0.0f;
    }
    bias[p] = a.bw ? a.bw[p * F + f0] : 0.0f;
  }
  for (int64_t wv = wave0; wv < 2 * a.N; wv += nwaves) {
    const int j = __builtin_amdgcn_readfirstlane(static_cast<int>(wv >> 1));
    int e_lo = a.ptr1[j], e_hi = a.ptr1[j + 1];
    e_lo = e_lo < 0 ? 0 : (e_lo > a.M ? static_cast<int>(a.M) : e_lo);
    e_hi = e_hi < e_lo ? e_lo : (e_hi > a.M ? static_cast<int>(a.M) : e_hi);
    float sj[3], vj[3], gs[3] = {0.0f, 0.0f, 0.0f}, gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      sj[p] = a.s[static_cast<int64_t>(j) * 3 * F + p * F + f0];
      vj[p] = a.v[static_cast<int64_t>(j) * 3 * F + p * F + f0];
    }
    for (int base = e_lo; base < e_hi; base += 64) {
      const int cnt = e_hi - base < 64 ? e_hi - base : 64;
      int my_r = 0, my_i = 0;
      if (lane < cnt) {
        my_r = a.perm1 ? a.perm1[base + lane] : base + lane;
        my_i = a.recv[my_r];
        my_i = my_i < 0 ? 0 : (my_i >= a.N ? static_cast<int>(a.N) - 1 : my_i);
      }
      for (int u0 = 0; u0 < cnt; u0 += EC) {
        float gz[EC], gdv[EC][3], meta[EC], metad[EC];
#pragma unroll
        for (int u = 0; u < EC; ++u) {
          const int slot = u0 + u < cnt ? u0 + u : cnt - 1;
          const int i = __builtin_amdgcn_readlane(my_i, slot);
          const int r = __builtin_amdgcn_readlane(my_r, slot);
          gz[u] = a.g_ds[static_cast<int64_t>(i) * F + f0];
#pragma unroll
          for (int k = 0; k < 3; ++k) gdv[u][k] = a.g_dv[(static_cast<int64_t>(i) * 3 + k) * F + f0];
          // wave-uniform edge data as two vector loads (see the forward kernel): rbf | r_ij | env and rbf' | env'
          const float* mp_ = lane < B ? a.rbf + static_cast<int64_t>(r) * B + lane
                                      : (lane < B + 3 || !a.env) ? a.rij + static_cast<int64_t>(r) * 3 + (lane < B + 3 ? lane - B : 0)
                                                                 : a.env + r;
          meta[u] = lane < B + 4 ? *mp_ : 0.0f;
          const float* md_ = (lane < B || !a.envd) ? a.rbfd + static_cast<int64_t>(r) * B + (lane < B ? lane : 0) : a.envd + r;
          metad[u] = lane < B + 1 ? *md_ : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < EC; ++u) {
          if (u0 + u < cnt) {   // wave-uniform
            const int r = __builtin_amdgcn_readlane(my_r, u0 + u);
            // (f[p], fd[p]) as one register pair: three packed FMAs per basis function instead of six (pk_fma_wlo / whi);
            // k ascending as before.  For odd B the last pair's second weight is 0 (its scalar is a finite lane of meta).
            floatx2 ff[3] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
            for (int q = 0; q < MAXB / 2; ++q) {
              if (2 * q < B) {
                const floatx2 x0 = {readlane_f(meta[u], 2 * q), readlane_f(metad[u], 2 * q)};
                const floatx2 x1 = {readlane_f(meta[u], 2 * q + 1), readlane_f(metad[u], 2 * q + 1)};
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                  pk_fma_wlo(ff[p], x0, wp[p][q]);
                  pk_fma_whi(ff[p], x1, wp[p][q]);
                }
              }
            }
            const float f[3] = {ff[0].x, ff[1].x, ff[2].x}, fd[3] = {ff[0].y, ff[1].y, ff[2].y};
            float rk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) rk[k] = readlane_f(meta[u], B + k);
            // filter and its derivative w.r.t. the distance (the envelope is a second factor: product rule)
            float wf[3], wd[3];
            const float envv = a.env ? readlane_f(meta[u], B + 3) : 1.0f;
            const float envdv = (a.env && a.envd) ? readlane_f(metad[u], B) : 0.0f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              const float wx = f[p] + bias[p];
              wf[p] = a.env ? wx * envv : wx;
              wd[p] = a.env ? fd[p] * envv + wx * envdv : fd[p];
            }
            // upstream gradients of the three parts of sw = s_j * w
            float gsw[3];
            gsw[0] = gz[u];
            gsw[1] = gdv[u][0] * vj[0] + gdv[u][1] * vj[1] + gdv[u][2] * vj[2];
            gsw[2] = gdv[u][0] * rk[0] + gdv[u][1] * rk[1] + gdv[u][2] * rk[2];
            float gd = 0.0f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              gs[p] += gsw[p] * wf[p];
              gd += gsw[p] * sj[p] * wd[p];
            }
            const float sw2 = sj[1] * wf[1], sw3 = sj[2] * wf[2];
            float gr[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              gv[k] += gdv[u][k] * sw2;
              gr[k] = gdv[u][k] * sw3;
            }
            wave_sum4_uniform(gd, gr[0], gr[1], gr[2]);
            if (lane == 0) {
              if (a.accumulate) {
                g_d[r] += gd;
#pragma unroll
                for (int k = 0; k < 3; ++k) g_rij[static_cast<int64_t>(r) * 3 + k] += gr[k];
              } else {
                g_d[r] = gd;
#pragma unroll
                for (int k = 0; k < 3; ++k) g_rij[static_cast<int64_t>(r) * 3 + k] = gr[k];
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) a.g_s[static_cast<int64_t>(j) * 3 * F + p * F + f0] = gs[p];
    if (a.g_v) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int64_t at = (static_cast<int64_t>(j) * 3 + k) * F + f0;
        a.g_v[at] = a.g_dv[at] + gv[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- reverse, tile form
// The reverse message step on SENDER tiles: a workgroup serves a few consecutive sending nodes of one graph, wave q the
// feature quarter 32 q .. 32 q + 31.  Staged in LDS after ONE barrier: the upstream gradients g_ds / g_dv of the graph's
// whole node range (the receivers of a tile's edges lie in its own graph) by contiguous LDS-DMA, the basis rows rbf_e and
// rbf'_e of the tile's edges by GATHER LDS-DMA (per-lane source address through perm1: sender order is a permutation of
// the stored edge order), receiver row / unit vector / envelope per edge by one thread per edge.
//
// Per sender and step of up to 16 edges one 32 x 32 filter tile per part on the matrix pipe (bf16x3, the forward kernel's
// pre-split weight image): A row m = [rbf_e | 1] (m even) or [rbf'_e | 0] (m odd) of the edge in slot (m & 3) / 2 +
// 2 (m >> 3) of lane half (m >> 2) & 1, so accumulator registers (2 u, 2 u + 1) of a lane hold the filter and its
// derivative w.r.t. the distance for the lane's own feature and the edge `2 u + half` of the step: the two lane halves
// walk the even and the odd edges of the step in register order.  dE/ds_j, dE/dv_j accumulate in registers (even / odd
// partial sums added once per sender: fixed order, deterministic), the four per-edge scalars (dE/dd_e, dE/dr_ij) of the
// step's slots are reduced over the 32 features of a lane half by a TRANSPOSING reduction (32 values per lane in, one
// total per lane out: 16 + 8 + 4 + 2 + 1 exchange-adds instead of 32 x 5), parked per wave in LDS and summed over the four
// waves in fixed order at the end of the tile.
struct PainnBwdTileArgs {
  const float* s;        // (N, 3F)   saved forward s
  const float* v;        // (N, 3, F) saved block-input v
  const float* rbf;      // (M, B)
  const float* rbfd;     // (M, B)
  const float* env;      // (M) or null
  const float* envd;     // (M) or null
  const float* rij;      // (M, 3)
  const void* wimg;      // packed filter image (mp_painn_filter_pack_f32)
  const int32_t* tiles;  // (T, 8): j_lo, j_hi, s_lo, s_hi, e_lo, e_hi (sender-order positions), -, -
  const int32_t* ptr1;   // (N+1) CSR over senders
  const int32_t* perm1;  // (M) sender-order position -> stored edge, or null
  const int32_t* recv;   // (M) stored edge order
  const float* g_ds;     // (N, F)
  const float* g_dv;     // (N, 3, F)
  float* g_s;            // (N, 3F)
  float* g_v;            // (N, 3, F) or null
  float* g_d;            // (M)    written (accumulate = 0) or added to
  float* g_rij;          // (M, 3)
  int accumulate;
  int64_t N, M;
  int B, ntiles, max_rows, max_senders, max_edges;
};

// k groups of 8 per A row held as bf16 pieces: k = 0 .. B (bias slot) -> ceil((B + 1) / 8)
__host__ __device__ __forceinline__ int painn_kgroups(int B) { return (B + 8) >> 3; }
__host__ __device__ __forceinline__ int painn_bwd_tile_lds_floats(int max_rows, int max_senders, int max_edges, int B,
                                                                  bool env) {
  return painn_pad(max_rows * F, 256) + painn_pad(max_rows * 3 * F, 256) + 2 * painn_pad(max_senders * 3 * F, 256) +
         painn_pad((2 * max_edges * painn_kgroups(B) * 3 + 3) * 4, 64) + painn_pad(max_edges * 4, 64) +
         (env ? painn_pad(max_edges, 64) : 0) + 2 * painn_pad(max_edges, 64) + painn_pad(4 * max_edges * 4, 64);
}

// x[l] + x[l ^ 16] for the lanes of even 16-lane rows, y[l] + y[l ^ 16] for the odd rows
__device__ __forceinline__ float fold16(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lanes with `bit` clear: x[l] + x[partner]; lanes with it set: y[l] + y[partner] (partner = the DPP pattern's source)
template <int CTRL>
__device__ __forceinline__ float fold_dpp(bool bit, float x, float y) {
  const float send = bit ? x : y, keep = bit ? y : x;
  return keep + dpp_mov<CTRL>(send);
}
__device__ __forceinline__ float fold4(bool bit, float x, float y) {   // partner l ^ 4: row_shl:4 for banks 0, 2; row_shr:4 for 1, 3
  const float send = bit ? x : y, keep = bit ? y : x;
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x104, 0xf, 0x5, false);
  t = __builtin_amdgcn_update_dpp(t, __float_as_int(send), 0x114, 0xf, 0xA, false);
  return keep + __int_as_float(t);
}
__device__ __forceinline__ float xor4_sum(float x) {                   // x[l] + x[l ^ 4]
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0x5, false);
  t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x114, 0xf, 0xA, false);
  return x + __int_as_float(t);
}
// NV = 8, 16 or 32 values per lane -> every lane c of a 32-lane half holds the half's total of value
// v(c) = bit4(c) + 2 bit3(c) + 4 bit2(c) [+ 8 bit1(c) [+ 16 bit0(c)]]: each exchange-add halves the number of live values
// (NV - 1 of them instead of 5 NV); lane bits beyond log2(NV) are finished by plain butterflies.
template <int NV>
__device__ __forceinline__ float transpose_sum(const float (&val)[NV], int c) {
  float w2[NV / 2], w4[NV / 4], w8[NV / 8];
#pragma unroll
  for (int i = 0; i < NV / 2; ++i) w2[i] = fold16(val[2 * i], val[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < NV / 4; ++i) w4[i] = fold_dpp<0x128>((c & 8) != 0, w2[2 * i], w2[2 * i + 1]);   // row_ror:8
#pragma unroll
  for (int i = 0; i < NV / 8; ++i) w8[i] = fold4((c & 4) != 0, w4[2 * i], w4[2 * i + 1]);
  if constexpr (NV == 8) {
    float t = w8[0];
    t += dpp_mov<0x4E>(t);                                               // quad_perm [2,3,0,1]
    return t + dpp_mov<0xB1>(t);                                         // quad_perm [1,0,3,2]
  } else {
    float w16[NV / 16];
#pragma unroll
    for (int i = 0; i < NV / 16; ++i) w16[i] = fold_dpp<0x4E>((c & 2) != 0, w8[2 * i], w8[2 * i + 1]);
    if constexpr (NV == 16) return w16[0] + dpp_mov<0xB1>(w16[0]);
    else return fold_dpp<0xB1>((c & 1) != 0, w16[0], w16[1]);
  }
}

// The messages of G groups of two slots (lane half hh: edge 2 u + hh of the step) from the filter tile `acc`, their
// per-edge scalars reduced over the lane half and parked in `part` (this wave's slab, indexed by the tile's edge).
template <int G, bool ENV>
__device__ __forceinline__ void painn_bwd_slots(const floatx16 (&acc)[3], const float* Gz, const float* Gv, const float4* P4,
                                                const float* Ed, const int* Sd, float4* part, int first_edge, int nh, int hh,
                                                int c, int f0, const float (&sj)[3], const float (&vj)[3], float (&gs)[3],
                                                float (&gv)[3]) {
  constexpr int NS = G == 3 ? 8 : 2 * G;      // slots reduced (three groups use the four-group reduction, zeros on top)
  float red[4 * NS];
  int le[2 * G], ir[2 * G];
#pragma unroll
  for (int u = 0; u < 2 * G; ++u) {
    le[u] = 2 * u + hh < nh ? first_edge + 2 * u + hh : 0;   // rows past the step: zero filter; reads repeat edge 0
    ir[u] = Sd[le[u]];
  }
  lds_wait_all();                                             // 4-B reads only: uniform
  const unsigned gz_base = lds_addr(Gz) + 4u * f0, gv_base = lds_addr(Gv) + 4u * f0, p4_base = lds_addr(P4),
                 ed_base = ENV ? lds_addr(Ed) : 0u;
#pragma unroll
  for (int g2 = 0; g2 < 2 * G; g2 += 2) {
    // the group's reads (4-B and 16-B) and their full wait as one statement
    EdgeReads rd[2];
    // (24-bit multiplies: the 64-bit multiply-add hipcc picks for `12 * x + base` is a multi-pass instruction whose result
    // an LDS instruction issued right behind it has been seen to read too early in lanes 48-63 - a wrong FIRST read of a
    // freshly computed address, different from run to run.  No v_mad_u64_u32 / v_mul_hi feeds an LDS address in this kernel.)
    lds_group_bwd<ENV>(gz_base + (static_cast<unsigned>(ir[g2]) << 2), gv_base + __umul24(ir[g2], 12u), p4_base + 16u * le[g2],
                       ed_base + 4u * le[g2], gz_base + (static_cast<unsigned>(ir[g2 + 1]) << 2),
                       gv_base + __umul24(ir[g2 + 1], 12u), p4_base + 16u * le[g2 + 1], ed_base + 4u * le[g2 + 1], rd[0], rd[1]);
    float gz[2], gdv[2][3], ed[2];
    float4 re[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      gz[u] = rd[u].one;
#pragma unroll
      for (int k = 0; k < 3; ++k) gdv[u][k] = rd[u].b[k];
      re[u] = make_float4(rd[u].e[0], rd[u].e[1], rd[u].e[2], rd[u].e[3]);
      ed[u] = rd[u].a[0];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int su = g2 + u;
      const float rk[3] = {re[u].x, re[u].y, re[u].z};
      float wf[3], wd[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const float wx = acc[p][2 * su], fd = acc[p][2 * su + 1];   // x W + b (bias = k slot B) ; x' W
        wf[p] = ENV ? wx * re[u].w : wx;
        wd[p] = ENV ? fd * re[u].w + wx * ed[u] : fd;
      }
      float gsw[3];
      gsw[0] = gz[u];
      gsw[1] = gdv[u][0] * vj[0] + gdv[u][1] * vj[1] + gdv[u][2] * vj[2];
      gsw[2] = gdv[u][0] * rk[0] + gdv[u][1] * rk[1] + gdv[u][2] * rk[2];
      float gd = 0.0f;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        gs[p] += gsw[p] * wf[p];
        gd += gsw[p] * sj[p] * wd[p];
      }
      const float sw2 = sj[1] * wf[1], sw3 = sj[2] * wf[2];
      red[4 * su] = gd;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gv[k] += gdv[u][k] * sw2;
        red[4 * su + 1 + k] = gdv[u][k] * sw3;
      }
    }
  }
#pragma unroll
  for (int i = 8 * G; i < 4 * NS; ++i) red[i] = 0.0f;
  // lane c keeps component k_red of slot u_red; lanes whose low bits are not part of the slot index hold copies
  const float tot = transpose_sum<4 * NS>(red, c);
  const int k_red = ((c >> 4) & 1) + 2 * ((c >> 3) & 1);
  int u_red = (c >> 2) & 1;
  bool mine = true;
  if constexpr (NS >= 4) u_red += 2 * ((c >> 1) & 1); else mine = (c & 2) == 0;
  if constexpr (NS >= 8) u_red += 4 * (c & 1); else mine = mine && (c & 1) == 0;
  if (mine && 2 * u_red + hh < nh) reinterpret_cast<float*>(part + first_edge + 2 * u_red + hh)[k_red] = tot;
}

template <int BT, bool ENV>
__global__ __launch_bounds__(256, 2) void painn_message_bwd_tile_kernel(PainnBwdTileArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63;
  const int fq = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int hh = lane >> 5, c = lane & 31;
  const int B = BT > 0 ? BT : a.B;
  const int NG = BT == 20 ? 3 : painn_kgroups(B);
  const int f0 = 32 * fq + c;
  // A row c: slot, lane half and kind (0: basis row with the bias slot, 1: its derivative) of the edge it carries
  const int uA = ((c & 3) >> 1) + 2 * (c >> 3), hA = (c >> 2) & 1, kind = c & 1;
  float* const Gz = lds;                                                       // [rows][F]
  float* const Gv = Gz + painn_pad(a.max_rows * F, 256);                       // [rows][3][F]
  float* const Sj = Gv + painn_pad(a.max_rows * 3 * F, 256);                   // [senders][3F]    saved s of the tile's senders
  float* const Vj = Sj + painn_pad(a.max_senders * 3 * F, 256);                // [senders][3][F]  block-input v
  bf16x8* const Pc = reinterpret_cast<bf16x8*>(Vj + painn_pad(a.max_senders * 3 * F, 256));   // [edges][2][NG][3] + 3 zero
  const int zunit = 2 * a.max_edges * NG * 3;
  float4* const P4 = reinterpret_cast<float4*>(reinterpret_cast<float*>(Pc) + painn_pad((zunit + 3) * 4, 64));   // {r_ij, envelope}
  float* const Ed = reinterpret_cast<float*>(P4) + painn_pad(a.max_edges * 4, 64);      // [edges] envelope' (ENV)
  int* const Sd = reinterpret_cast<int*>(Ed + (ENV ? painn_pad(a.max_edges, 64) : 0));  // [edges] receiver row offset (floats of Gz)
  int* const Rr = Sd + painn_pad(a.max_edges, 64);                             // [edges] stored edge id
  float4* const Part = reinterpret_cast<float4*>(Rr + painn_pad(a.max_edges, 64));      // [4][edges] per-wave {g_d, g_rij}
  if (threadIdx.x < 12) reinterpret_cast<float*>(Pc + zunit)[threadIdx.x] = 0.0f;
  bf16x8 wb[3][2][3];
  {
    const bf16x8* img = reinterpret_cast<const bf16x8*>(a.wimg) + static_cast<size_t>(fq) * (3 * 2 * 3 * 64) + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wb[p][ks][pc] = img[((p * 2 + ks) * 3 + pc) * 64];
  }
  const int N = static_cast<int>(a.N), M = static_cast<int>(a.M);
  const int first = static_cast<int>(mp_xcd_block(blockIdx.x, gridDim.x));
  for (int tile = first; tile < a.ntiles; tile += static_cast<int>(gridDim.x)) {
    const int4 d0 = *reinterpret_cast<const int4*>(a.tiles + static_cast<int64_t>(tile) * 8);
    const int2 d1 = *reinterpret_cast<const int2*>(a.tiles + static_cast<int64_t>(tile) * 8 + 4);
    int j_lo = __builtin_amdgcn_readfirstlane(d0.x), j_hi = __builtin_amdgcn_readfirstlane(d0.y);
    int s_lo = __builtin_amdgcn_readfirstlane(d0.z), s_hi = __builtin_amdgcn_readfirstlane(d0.w);
    int e_lo = __builtin_amdgcn_readfirstlane(d1.x), e_hi = __builtin_amdgcn_readfirstlane(d1.y);
    // a malformed table must not fault: everything is clamped into the arrays and into the LDS regions
    j_lo = j_lo < 0 ? 0 : (j_lo > N ? N : j_lo);
    j_hi = j_hi < j_lo ? j_lo : (j_hi > N ? N : j_hi);
    if (j_hi - j_lo > 62) j_hi = j_lo + 62;
    if (j_hi - j_lo > a.max_senders) j_hi = j_lo + a.max_senders;
    s_lo = s_lo < 0 ? 0 : (s_lo > N - 1 ? N - 1 : s_lo);
    s_hi = s_hi <= s_lo ? s_lo + 1 : (s_hi > N ? N : s_hi);
    if (s_hi - s_lo > a.max_rows) s_hi = s_lo + a.max_rows;
    e_lo = e_lo < 0 ? 0 : (e_lo > M ? M : e_lo);
    e_hi = e_hi < e_lo ? e_lo : (e_hi > M ? M : e_hi);
    if (e_hi - e_lo > a.max_edges) e_hi = e_lo + a.max_edges;
    const int rows = s_hi - s_lo, ne = e_hi - e_lo, nj = j_hi - j_lo;
    if (tile != first) __syncthreads();   // the previous tile's readers are done
    // ---- the senders' own rows of s and v (contiguous), then the upstream gradients of the graph's nodes: 1-KB chunks
    //      dealt to the four waves
    if (nj > 0) {
      const int nfl = nj * 3 * F;
      const float* srcS = a.s + static_cast<int64_t>(j_lo) * 3 * F;
      const float* srcV = a.v + static_cast<int64_t>(j_lo) * 3 * F;
      for (int ch = fq; ch * 256 < nfl; ch += 4) {
        int off = ch * 256 + lane * 4;
        off = off < nfl - 4 ? off : nfl - 4;
        __builtin_amdgcn_global_load_lds(srcS + off, (__attribute__((address_space(3))) void*)(Sj + ch * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(srcV + off, (__attribute__((address_space(3))) void*)(Vj + ch * 256), 16, 0, 0);
      }
    }
    {
      const int nz = rows * F, nv = rows * 3 * F;
      const float* srcZ = a.g_ds + static_cast<int64_t>(s_lo) * F;
      const float* srcV = a.g_dv + static_cast<int64_t>(s_lo) * 3 * F;
      for (int ch = fq; ch * 256 < nv; ch += 4) {
        int off = ch * 256 + lane * 4;
        int offv = off < nv - 4 ? off : nv - 4;
        __builtin_amdgcn_global_load_lds(srcV + offv, (__attribute__((address_space(3))) void*)(Gv + ch * 256), 16, 0, 0);
        if (ch * 256 < nz) {
          int offz = off < nz - 4 ? off : nz - 4;
          __builtin_amdgcn_global_load_lds(srcZ + offz, (__attribute__((address_space(3))) void*)(Gz + ch * 256), 16, 0, 0);
        }
      }
    }
    // ---- A operands of the tile's edges, split ONCE for the four waves: task = (edge, kind, k group of 8) gathers its eight
    //      values through perm1 and leaves the three bf16 pieces as 16-B units [edge][kind][group][piece]; two tasks per
    //      thread and pass, their loads issued together (index arithmetic by shifts and 24-bit multiplies only)
    {
      const int ntask = 2 * ne * 4;                           // task = 4 row + k group (group 3 idles when NG = 3)
      for (int t0 = threadIdx.x; t0 < ntask; t0 += 512) {
        float x[2][8];
        int tt[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int t = t0 + 256 * h;
          t = t < ntask ? t : ntask - 1;                      // a repeated task rewrites the same values
          const int row = t >> 2, e = row >> 1, kd = row & 1;
          int g = t & 3;
          g = g < NG ? g : NG - 1;                            // the idle group repeats the last one's values
          tt[h] = __umul24(row, 3 * NG) + 3 * g;              // first 16-B unit of the task
          int r = a.perm1 ? a.perm1[e_lo + e] : e_lo + e;
          r = r < 0 ? 0 : (r >= M ? M - 1 : r);
          const float* tab = (kd ? a.rbfd : a.rbf) + static_cast<int64_t>(r) * B;
          if constexpr (BT == 20) {   // 80-B rows: 16-B pieces, unconditional from clamped offsets, masked afterwards
            const float4 q0 = *reinterpret_cast<const float4*>(tab + (g < 2 ? 8 * g : 16));
            const float4 q1 = *reinterpret_cast<const float4*>(tab + (g < 2 ? 8 * g + 4 : 16));
            const float m = g < 2 ? 1.0f : 0.0f;
            x[h][0] = q0.x; x[h][1] = q0.y; x[h][2] = q0.z; x[h][3] = q0.w;
            x[h][4] = g < 2 ? q1.x : (kd ? 0.0f : 1.0f);       // k = 20: the bias slot
            x[h][5] = q1.y * m; x[h][6] = q1.z * m; x[h][7] = q1.w * m;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int k = 8 * g + i;
              const float y = tab[k < B ? k : B - 1];
              x[h][i] = k < B ? y : ((k == B && !kd) ? 1.0f : 0.0f);
            }
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 pc[3];
#pragma unroll
          for (int i = 0; i < 8; ++i) split3_into(x[h][i], pc[0], pc[1], pc[2], i);
#pragma unroll
          for (int q = 0; q < 3; ++q) Pc[tt[h] + q] = pc[q];
        }
      }
    }
    // ---- per edge: stored id, receiver row inside the tile, unit vector + envelope as one 16-B entry
    for (int i = threadIdx.x; i < ne; i += 256) {
      int r = a.perm1 ? a.perm1[e_lo + i] : e_lo + i;
      r = r < 0 ? 0 : (r >= M ? M - 1 : r);
      int ir = a.recv[r] - s_lo;
      ir = ir < 0 ? 0 : (ir >= rows ? rows - 1 : ir);
      Rr[i] = r;
      Sd[i] = ir * F;
      P4[i] = make_float4(a.rij[static_cast<int64_t>(r) * 3], a.rij[static_cast<int64_t>(r) * 3 + 1],
                          a.rij[static_cast<int64_t>(r) * 3 + 2], ENV ? a.env[r] : 1.0f);
      if (ENV) Ed[i] = a.envd ? a.envd[r] : 0.0f;
    }
    // edge ranges of the tile's senders: lane i holds ptr1[j_lo + i]
    int ptrv = a.ptr1[(j_lo + lane) <= j_hi ? j_lo + lane : j_hi];
    ptrv = ptrv < e_lo ? e_lo : (ptrv > e_hi ? e_hi : ptrv);
    __builtin_amdgcn_s_waitcnt(0);      // DMA + stores issued by this wave have landed
    __syncthreads();
    float4* const part = Part + fq * a.max_edges;
    for (int jj = 0; jj < nj; ++jj) {
      const int j = j_lo + jj;
      const int lo = __builtin_amdgcn_readlane(ptrv, jj) - e_lo;
      const int cnt = __builtin_amdgcn_readlane(ptrv, jj + 1) - e_lo - lo;
      float sj[3], vj[3], gs[3] = {0.0f, 0.0f, 0.0f}, gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        sj[p] = Sj[jj * 3 * F + p * F + f0];
        vj[p] = Vj[jj * 3 * F + p * F + f0];
      }
      lds_wait_all();                                         // six 4-B reads, complete before the 16-B operand reads
      for (int cb = 0; cb < cnt; cb += 16) {
        const int nh = cnt - cb < 16 ? cnt - cb : 16;         // edges of this step
        // A operand: row c = kind `kind` of the edge in slot uA of lane half hA; rows past the step and the k groups
        // beyond the bias slot read the zero units
        const bool rv = 2 * uA + hA < nh;
        const int rowu = __umul24(2 * (lo + cb + 2 * uA + hA) + kind, 3 * NG);
        const int u0 = rv ? rowu + 3 * hh : zunit;
        const int u1 = (rv && 2 + hh < NG) ? rowu + 6 + 3 * hh : zunit;
        bf16x8 qa[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          qa[0][q] = Pc[u0 + q];
          qa[1][q] = Pc[u1 + q];
        }
        lds_wait_all();                                       // six 16-B reads: uniform
        // the six leading products of the three-piece split, smallest first, for both k blocks and the three parts in turn
        // (36 MFMAs of 32 cycles: at two waves per SIMD this phase runs at the matrix pipe's rate)
        floatx16 acc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
        {
          constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int p = 0; p < 3; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks][PA[pr]], wb[p][ks][PB[pr]], acc[p], 0, 0, 0);
        }
        mfma_drained(acc);
        // the lane's slots: edge 2 u + hh of the step, in groups of two; a code path per group count, so that no register
        // state of a skipped group has to be merged
        const int first_edge = lo + cb;
        const int ng = (nh + 3) >> 2;
        if (ng >= 4) painn_bwd_slots<4, ENV>(acc, Gz, Gv, P4, Ed, Sd, part, first_edge, nh, hh, c, f0, sj, vj, gs, gv);
        else if (ng == 3) painn_bwd_slots<3, ENV>(acc, Gz, Gv, P4, Ed, Sd, part, first_edge, nh, hh, c, f0, sj, vj, gs, gv);
        else if (ng == 2) painn_bwd_slots<2, ENV>(acc, Gz, Gv, P4, Ed, Sd, part, first_edge, nh, hh, c, f0, sj, vj, gs, gv);
        else painn_bwd_slots<1, ENV>(acc, Gz, Gv, P4, Ed, Sd, part, first_edge, nh, hh, c, f0, sj, vj, gs, gv);
      }
      // even + odd edges of the sender
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        gs[p] = sum_halves(gs[p]);
        gv[p] = sum_halves(gv[p]);
      }
      if (hh == 0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a.g_s[static_cast<int64_t>(j) * 3 * F + p * F + f0] = gs[p];
        if (a.g_v) {     // the sender's own upstream row is among the staged ones
          int jr = j - s_lo;
          jr = jr < 0 ? 0 : (jr >= rows ? rows - 1 : jr);
          float own[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) own[k] = Gv[jr * 3 * F + k * F + f0];
          lds_wait_all();
#pragma unroll
          for (int k = 0; k < 3; ++k) a.g_v[(static_cast<int64_t>(j) * 3 + k) * F + f0] = own[k] + gv[k];
        }
      }
    }
    __syncthreads();     // every wave's per-edge partials are parked
    for (int e = threadIdx.x; e < ne; e += 256) {   // one thread per edge: the four waves' slabs in fixed order
      const int r = Rr[e];
      lds_wait_all();                                         // the 4-B read is complete before the 16-B ones are issued
      const float4 p0 = Part[e], p1 = Part[a.max_edges + e], p2 = Part[2 * a.max_edges + e], p3 = Part[3 * a.max_edges + e];
      lds_wait_all();
      float t[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                    ((p0.w + p1.w) + p2.w) + p3.w};
      float* dr = a.g_rij + static_cast<int64_t>(r) * 3;
      if (a.accumulate) {
        const float o0 = a.g_d[r], o1 = dr[0], o2 = dr[1], o3 = dr[2];
        t[0] += o0; t[1] += o1; t[2] += o2; t[3] += o3;
      }
      a.g_d[r] = t[0];
      dr[0] = t[1]; dr[1] = t[2]; dr[2] = t[3];
    }
  }
}

// ---------------------------------------------------------------------------------------------------- update, node side
// uv (3N, 2F): row (n,k) = [v_u | v_v] of component k (one GEMM with the concatenated kernels [Wu | Wv]).
// pre:  c (N, 2F) = [z' | sqrt(relu(sum_k v_v^2))]  (EuclideanNorm(axis=2), geom.py:181-193; LazyConcatenate),
//       prod (N, F) = sum_k v_u v_v                   (ScalarProduct(axis=2), geom.py:261)
__global__ void painn_update_pre_kernel(const float* __restrict__ zp, const float* __restrict__ uv, int64_t N,
                                        float* __restrict__ c, float* __restrict__ prod) {
  const int64_t total = N * F;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t n = t / F;
    const int f = static_cast<int>(t % F);
    float pr = 0.0f, sq = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float vu = uv[((n * 3 + k) * 2) * F + f];
      const float vv = uv[((n * 3 + k) * 2 + 1) * F + f];
      pr += vu * vv;
      sq += vv * vv;
    }
    c[n * 2 * F + f] = zp[t];
    c[n * 2 * F + F + f] = sqrtf(fmaxf(sq, 0.0f));
    prod[t] = pr;
  }
}

// post: z'' = z' + prod a_sv + a_ss ; v''[k] = v'[k] + a_vv v_u[k]     (painn_conv.py:208-213 + PAiNN.py:131-132)
__global__ void painn_update_post_kernel(const float* __restrict__ zp, const float* __restrict__ vp,
                                         const float* __restrict__ uv, const float* __restrict__ prod,
                                         const float* __restrict__ a, int64_t N, float* __restrict__ z2,
                                         float* __restrict__ v2) {
  const int64_t total = N * F;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t n = t / F;
    const int f = static_cast<int>(t % F);
    const float a_vv = a[n * 3 * F + f], a_sv = a[n * 3 * F + F + f], a_ss = a[n * 3 * F + 2 * F + f];
    z2[t] = zp[t] + (prod[t] * a_sv + a_ss);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float vu = uv[((n * 3 + k) * 2) * F + f];
      v2[(n * 3 + k) * F + f] = vp[(n * 3 + k) * F + f] + a_vv * vu;
    }
  }
}

// reverse of post: g_a (N,3F) = [sum_k g_v2[k] v_u[k] | g_z2 prod | g_z2],  g_prod (N,F) = g_z2 a_sv
__global__ void painn_update_post_bwd_kernel(const float* __restrict__ gz2, const float* __restrict__ gv2,
                                             const float* __restrict__ uv, const float* __restrict__ prod,
                                             const float* __restrict__ a, int64_t N, float* __restrict__ g_a,
                                             float* __restrict__ g_prod) {
  const int64_t total = N * F;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t n = t / F;
    const int f = static_cast<int>(t % F);
    float gavv = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) gavv += gv2[(n * 3 + k) * F + f] * uv[((n * 3 + k) * 2) * F + f];
    const float g = gz2[t];
    g_a[n * 3 * F + f] = gavv;
    g_a[n * 3 * F + F + f] = g * prod[t];
    g_a[n * 3 * F + 2 * F + f] = g;
    g_prod[t] = g * a[n * 3 * F + F + f];
  }
}

// reverse of pre (+ the v_u part of post): with g_c (N,2F) = dE/dc,
//   g_zp = g_z2 + g_c[:, :F]
//   g_vu[k] = g_v2[k] a_vv + g_prod v_v[k] ;  g_vv[k] = g_prod v_u[k] + g_c[:, F:] v_v[k] / ||v_v||   (0 at the cusp)
__global__ void painn_update_pre_bwd_kernel(const float* __restrict__ gz2, const float* __restrict__ gv2,
                                            const float* __restrict__ uv, const float* __restrict__ c,
                                            const float* __restrict__ a, const float* __restrict__ g_prod,
                                            const float* __restrict__ g_c, int64_t N, float* __restrict__ g_zp,
                                            float* __restrict__ g_uv) {
  const int64_t total = N * F;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t n = t / F;
    const int f = static_cast<int>(t % F);
    const float nrm = c[n * 2 * F + F + f];
    const float gn = g_c[n * 2 * F + F + f];
    const float gp = g_prod[t];
    const float a_vv = a[n * 3 * F + f];
    const float inv = nrm > 0.0f ? gn / nrm : 0.0f;   // d sqrt(s) = v_v / sqrt(s); zero sub-gradient at s = 0
    g_zp[t] = gz2[t] + g_c[n * 2 * F + f];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float vu = uv[((n * 3 + k) * 2) * F + f];
      const float vv = uv[((n * 3 + k) * 2 + 1) * F + f];
      g_uv[((n * 3 + k) * 2) * F + f] = gv2[(n * 3 + k) * F + f] * a_vv + gp * vv;
      g_uv[((n * 3 + k) * 2 + 1) * F + f] = gp * vu + inv * vv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- geometry, backward
// dE/dx_n = sum_{e: recv = n} t_e - sum_{e: send = n} t_e,  t_e = g_d r_ij + (g_r - (g_r . r_ij) r_ij) / d
// (d = |x_i - x_j|, r_ij = (x_i - x_j) / d with divide_no_nan: no contribution at d = 0); g_d / g_rij arrive as `slices`
// partial buffers (the two feature halves of the message reverse kernel) that are added here.  One wave per node, lanes
// stride the node's receiver-side and sender-side edge lists, fixed-shape wave reduction: deterministic.  scale = -1
// returns the physical force directly.
__global__ __launch_bounds__(256) void painn_geometry_bwd_kernel(const float* __restrict__ g_d,
                                                                 const float* __restrict__ g_rij, int slices,
                                                                 const float* __restrict__ rij,
                                                                 const float* __restrict__ d,
                                                                 const int32_t* __restrict__ ptr0,
                                                                 const int32_t* __restrict__ perm0,
                                                                 const int32_t* __restrict__ ptr1,
                                                                 const int32_t* __restrict__ perm1, int64_t N, int64_t M,
                                                                 float scale, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t n = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; n < N; n += nwaves) {
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int side = 0; side < 2; ++side) {
      const int32_t* ptr = side == 0 ? ptr0 : ptr1;
      const int32_t* perm = side == 0 ? perm0 : perm1;
      const float sign = side == 0 ? 1.0f : -1.0f;
      int lo = ptr[n], hi = ptr[n + 1];
      lo = lo < 0 ? 0 : lo;
      hi = hi > M ? static_cast<int>(M) : hi;
      for (int e = lo + lane; e < hi; e += 64) {
        const int64_t r = perm ? perm[e] : e;
        const float dist = d[r];
        if (dist == 0.0f) continue;
        const float r0 = rij[r * 3], r1 = rij[r * 3 + 1], r2 = rij[r * 3 + 2];
        float gd = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
        for (int sl = 0; sl < slices; ++sl) {
          gd += g_d[sl * M + r];
          q0 += g_rij[(sl * M + r) * 3];
          q1 += g_rij[(sl * M + r) * 3 + 1];
          q2 += g_rij[(sl * M + r) * 3 + 2];
        }
        const float dot = q0 * r0 + q1 * r1 + q2 * r2;
        const float inv = 1.0f / dist;
        acc[0] += sign * (gd * r0 + (q0 - dot * r0) * inv);
        acc[1] += sign * (gd * r1 + (q1 - dot * r1) * inv);
        acc[2] += sign * (gd * r2 + (q2 - dot * r2) * inv);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
      gx[n * 3 + 0] = scale * acc[0];
      gx[n * 3 + 1] = scale * acc[1];
      gx[n * 3 + 2] = scale * acc[2];
    }
  }
}

// MPENGINE_PAINN_MFMA_GATHER=1 routes mp_painn_message_f32 to painn_message_mfma_kernel (filter on the matrix pipe, sender
// rows gathered from global memory): measured 19.9 us against 16.2 us for the VALU build at config 3 - what the MFMAs save
// in vector issue this form loses to its serial phases at two waves per SIMD - so it is an experiment, not the default.  The
// default matrix-pipe route is the LDS tile kernel behind mp_painn_message_tiles_f32.
bool painn_mfma_gather() {
  static const bool v = [] {
    const char* e = getenv("MPENGINE_PAINN_MFMA_GATHER");
    return e != nullptr && e[0] == '1';
  }();
  return v;
}

}  // namespace

extern "C" {

int mp_painn_stage0_f32(const void* numbers, int numbers_i64, int64_t N, const float* emb, int vocab, float v_init,
                        float* z0, float* v0,
                        const int64_t* idx, int64_t M, const int64_t* node_splits, const int64_t* edge_splits, int64_t G,
                        const float* xyz, const float* frequencies, int num_radial, float bessel_cutoff,
                        int envelope_exponent, float cos_cutoff, int32_t* recv, int32_t* send, int32_t* flags, float* dist,
                        float* rij, float* rbf, float* rbfd, float* env, float* envd, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && G >= 0 && vocab >= 1, "mp_painn_stage0_f32: bad sizes");
  MP_REQUIRE(num_radial >= 1 && num_radial <= 32 && bessel_cutoff != 0.0f && envelope_exponent >= 1,
             "mp_painn_stage0_f32: bad Bessel basis arguments");
  MP_REQUIRE(N < (int64_t{1} << 31) && M < (int64_t{1} << 31), "mp_painn_stage0_f32: N, M must fit int32");
  if (N == 0) return MP_OK;
  MP_REQUIRE(numbers && emb && z0 && v0 && node_splits && edge_splits && flags, "mp_painn_stage0_f32: null pointer");
  MP_REQUIRE(M == 0 || (idx && xyz && frequencies && recv && send && dist && rij && rbf),
             "mp_painn_stage0_f32: null edge pointer");
  MP_REQUIRE(cos_cutoff <= 0.0f || env != nullptr, "mp_painn_stage0_f32: envelope requested without a buffer");
  PainnNodeInit ni{numbers, numbers_i64, emb, vocab, N, v_init, z0, v0};
  mp_prep::EdgePrepArgs p{idx, M, node_splits, edge_splits, G, N, xyz, recv, send, dist, flags};
  PainnEdgeExtra ex{rij};
  const int node_blocks = static_cast<int>(mp::grid_for(N * F));
  const int edge_blocks = M > 0 ? static_cast<int>(mp::grid_for(M)) : 0;
  hipStream_t s = mp::as_stream(stream);
  if (G <= mp_prep::PREP_LDS_GRAPHS)
    painn_stage0_kernel<true><<<node_blocks + edge_blocks, 256, 0, s>>>(ni, p, ex, node_blocks);
  else
    painn_stage0_kernel<false><<<node_blocks + edge_blocks, 256, 0, s>>>(ni, p, ex, node_blocks);
  if (M > 0) {
    const int pe = envelope_exponent + 1;
    PainnBasisArgs q{};
    q.dist = dist; q.M = M; q.freq = frequencies; q.B = num_radial; q.inv_cutoff = 1.0f / bessel_cutoff; q.p = pe;
    q.a = static_cast<float>(-(pe + 1) * (pe + 2) / 2.0);
    q.b = static_cast<float>(pe * (pe + 2));
    q.c = static_cast<float>(-pe * (pe + 1) / 2.0);
    q.rbf = rbf; q.rbfd = rbfd; q.cos_cutoff = cos_cutoff;
    q.env = cos_cutoff > 0.0f ? env : nullptr;
    q.envd = cos_cutoff > 0.0f ? envd : nullptr;
    const int rc = mp::check_launch("mp_painn_stage0_f32");
    if (rc != MP_OK) return rc;
    painn_basis_kernel<<<mp::grid_for(M * num_radial), 256, 0, s>>>(q);
  }
  return mp::check_launch("mp_painn_stage0_f32");
}

int mp_painn_message_f32(const float* s, const float* v, int64_t N, const float* rbf, int B, const float* env,
                         const float* rij, const float* Ww, const float* bw, const int32_t* ptr, const int32_t* perm,
                         const int32_t* send, int64_t M, const float* z_in, float* ds, float* dv, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && B >= 1 && B <= 32, "mp_painn_message_f32: bad sizes (B must be 1..32)");
  if (N == 0) return MP_OK;
  MP_REQUIRE(s && v && Ww && ptr && ds && dv && (M == 0 || (rbf && rij && send)), "mp_painn_message_f32: null pointer");
  MP_REQUIRE(M < (int64_t{1} << 31) && N < (int64_t{1} << 31), "mp_painn_message_f32: sizes must fit int32");
  MP_REQUIRE(dv != v, "mp_painn_message_f32: dv must not alias v (other waves still gather v)");
  PainnMsgArgs a{s, v, rbf, env, rij, Ww, bw, ptr, perm, send, z_in, ds, dv, N, M, B};
  hipStream_t st = mp::as_stream(stream);
  if (B == 20 && M > 0 && painn_mfma_gather()) {   // opt-in experiment (see painn_mfma_gather), 20-function basis build only
    int64_t blocks = (N + 1) / 2;                    // a workgroup per receiver pair, persistent beyond two per CU
    if (blocks > 512) blocks = 512;
    launch_message_mfma<20>(a, static_cast<unsigned>(blocks), st);   // (the generic-basis build of this form spills)
    return mp::check_launch("mp_painn_message_f32");
  }
  int64_t blocks = mp::ceil_div(2 * N, 4);   // two waves (feature halves) per node, four waves per workgroup
  if (blocks > 4096) blocks = 4096;
  if (B == 20) painn_message_kernel<20><<<static_cast<unsigned>(blocks), 256, 0, st>>>(a);
  else painn_message_kernel<0><<<static_cast<unsigned>(blocks), 256, 0, st>>>(a);
  return mp::check_launch("mp_painn_message_f32");
}

int mp_painn_filter_pack_f32(const float* Ww, const float* bw, int B, void* image, mpStream_t stream) {
  MP_REQUIRE(B >= 1 && B <= 31, "mp_painn_filter_pack_f32: the basis size must be 1..31 (K = B + 1 bias slot <= 32)");
  MP_REQUIRE(Ww && image, "mp_painn_filter_pack_f32: null pointer");
  painn_filter_pack_kernel<<<(PAINN_FILTER_IMAGE_BYTES / 16 + 255) / 256, 256, 0, mp::as_stream(stream)>>>(
      Ww, bw, B, static_cast<bf16x8*>(image));
  return mp::check_launch("mp_painn_filter_pack_f32");
}

int mp_painn_message_tiles_lds_bytes(int max_rows, int max_edges, int B, int with_env, size_t* out) {
  MP_REQUIRE(max_rows >= 1 && max_edges >= 0 && B >= 1 && B <= 31 && out, "mp_painn_message_tiles_lds_bytes: bad arguments");
  MP_REQUIRE(max_rows <= 4096 && max_edges <= (1 << 20), "mp_painn_message_tiles_lds_bytes: tile far beyond LDS");
  *out = static_cast<size_t>(painn_tile_lds_floats(max_rows, max_edges, B, with_env != 0)) * 4;
  return MP_OK;
}

int mp_painn_message_tiles_f32(const float* s, const float* v, int64_t N, const float* rbf, int B, const float* env,
                               const float* rij, const void* wimage, const int32_t* ptr, const int32_t* send, int64_t M,
                               const int32_t* tiles, int ntiles, int max_rows, int max_edges, const float* z_in, float* ds,
                               float* dv, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && B >= 1 && B <= 31 && ntiles >= 0, "mp_painn_message_tiles_f32: bad sizes (B must be 1..31)");
  if (N == 0 || ntiles == 0) return MP_OK;
  MP_REQUIRE(s && v && wimage && ptr && tiles && ds && dv && (M == 0 || (rbf && rij && send)),
             "mp_painn_message_tiles_f32: null pointer");
  MP_REQUIRE(M < (int64_t{1} << 31) && N < (int64_t{1} << 31), "mp_painn_message_tiles_f32: sizes must fit int32");
  MP_REQUIRE(dv != v, "mp_painn_message_tiles_f32: dv must not alias v");
  size_t lds = 0;
  const int rc = mp_painn_message_tiles_lds_bytes(max_rows, max_edges, B, env != nullptr, &lds);
  if (rc != MP_OK) return rc;
  MP_REQUIRE(lds <= 160 * 1024, "mp_painn_message_tiles_f32: a tile of %d node rows and %d edges needs %zu B of LDS (> 160 KB)",
             max_rows, max_edges, lds);
  PainnTileArgs a{s, v, rbf, env, rij, wimage, ptr, send, tiles, z_in, ds, dv, N, M, B, ntiles, max_rows, max_edges};
  hipStream_t st = mp::as_stream(stream);
  const unsigned blocks = static_cast<unsigned>(ntiles < 2048 ? ntiles : 2048);
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and per kernel: opted in once per (instantiation, device)
  auto launch = [&](auto kernel) -> int {
    static std::mutex mu;
    static unsigned long long done = 0;       // one static pair per instantiation of this lambda's call operator
    int dev = 0;
    MP_HIP(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(mu);
      if (dev >= 64 || !((done >> dev) & 1ull)) {
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        if (dev < 64) done |= 1ull << dev;
      }
    }
    kernel<<<blocks, 256, lds, st>>>(a);
    return MP_OK;
  };
  int lr;
  if (B == 20) lr = env ? launch(painn_message_tile_kernel<20, true>) : launch(painn_message_tile_kernel<20, false>);
  else lr = env ? launch(painn_message_tile_kernel<0, true>) : launch(painn_message_tile_kernel<0, false>);
  if (lr != MP_OK) return lr;
  return mp::check_launch("mp_painn_message_tiles_f32");
}

int mp_painn_message_bwd_tiles_lds_bytes(int max_rows, int max_senders, int max_edges, int B, int with_env, size_t* out) {
  MP_REQUIRE(max_rows >= 1 && max_senders >= 1 && max_senders <= 62 && max_edges >= 0 && B >= 1 && B <= 31 && out,
             "mp_painn_message_bwd_tiles_lds_bytes: bad arguments");
  MP_REQUIRE(max_rows <= 4096 && max_edges <= (1 << 20), "mp_painn_message_bwd_tiles_lds_bytes: tile far beyond LDS");
  *out = static_cast<size_t>(painn_bwd_tile_lds_floats(max_rows, max_senders, max_edges, B, with_env != 0)) * 4;
  return MP_OK;
}

int mp_painn_message_bwd_tiles_f32(const float* s, const float* v, int64_t N, const float* rbf, const float* rbfd, int B,
                                   const float* env, const float* envd, const float* rij, const void* wimage,
                                   const int32_t* ptr1, const int32_t* perm1, const int32_t* recv, int64_t M,
                                   const int32_t* tiles, int ntiles, int max_rows, int max_senders, int max_edges,
                                   const float* g_ds, const float* g_dv, float* g_s, float* g_v, float* g_d, float* g_rij,
                                   int accumulate, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && B >= 1 && B <= 31 && ntiles >= 0,
             "mp_painn_message_bwd_tiles_f32: bad sizes (B must be 1..31)");
  if (N == 0 || ntiles == 0) return MP_OK;
  MP_REQUIRE(s && v && wimage && ptr1 && tiles && g_ds && g_dv && g_s && (M == 0 || (rbf && rbfd && rij && recv && g_d && g_rij)),
             "mp_painn_message_bwd_tiles_f32: null pointer");
  MP_REQUIRE(M < (int64_t{1} << 31) && N < (int64_t{1} << 31), "mp_painn_message_bwd_tiles_f32: sizes must fit int32");
  MP_REQUIRE(g_v != g_dv, "mp_painn_message_bwd_tiles_f32: g_v must not alias g_dv (other tiles still stage g_dv)");
  size_t lds = 0;
  const int rc = mp_painn_message_bwd_tiles_lds_bytes(max_rows, max_senders, max_edges, B, env != nullptr, &lds);
  if (rc != MP_OK) return rc;
  MP_REQUIRE(lds <= 160 * 1024,
             "mp_painn_message_bwd_tiles_f32: a tile of %d node rows, %d senders and %d edges needs %zu B of LDS (> 160 KB)",
             max_rows, max_senders, max_edges, lds);
  PainnBwdTileArgs a{s,    v,    rbf,  rbfd, env, envd, rij,   wimage,     tiles, ptr1, perm1, recv, g_ds,      g_dv,
                     g_s,  g_v,  g_d,  g_rij, accumulate, N,   M,   B,     ntiles, max_rows, max_senders, max_edges};
  hipStream_t st = mp::as_stream(stream);
  const unsigned blocks = static_cast<unsigned>(ntiles < 2048 ? ntiles : 2048);
  auto launch = [&](auto kernel) -> int {
    static std::mutex mu;
    static unsigned long long done = 0;       // one static pair per instantiation of this lambda's call operator
    int dev = 0;
    MP_HIP(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(mu);
      if (dev >= 64 || !((done >> dev) & 1ull)) {
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        if (dev < 64) done |= 1ull << dev;
      }
    }
    kernel<<<blocks, 256, lds, st>>>(a);
    return MP_OK;
  };
  int lr;
  if (B == 20) lr = env ? launch(painn_message_bwd_tile_kernel<20, true>) : launch(painn_message_bwd_tile_kernel<20, false>);
  else lr = env ? launch(painn_message_bwd_tile_kernel<0, true>) : launch(painn_message_bwd_tile_kernel<0, false>);
  if (lr != MP_OK) return lr;
  return mp::check_launch("mp_painn_message_bwd_tiles_f32");
}

int mp_painn_message_bwd_f32(const float* s, const float* v, int64_t N, const float* rbf, const float* rbfd, int B,
                             const float* env, const float* envd, const float* rij, const float* Ww, const float* bw,
                             const int32_t* ptr1, const int32_t* perm1, const int32_t* recv, int64_t M, const float* g_ds,
                             const float* g_dv, float* g_s, float* g_v, float* g_d, float* g_rij, int accumulate,
                             mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && B >= 1 && B <= 32, "mp_painn_message_bwd_f32: bad sizes (B must be 1..32)");
  if (N == 0) return MP_OK;
  MP_REQUIRE(s && v && Ww && ptr1 && g_ds && g_dv && g_s && (M == 0 || (rbf && rbfd && rij && recv && g_d && g_rij)),
             "mp_painn_message_bwd_f32: null pointer");
  MP_REQUIRE(M < (int64_t{1} << 31) && N < (int64_t{1} << 31), "mp_painn_message_bwd_f32: sizes must fit int32");
  MP_REQUIRE(g_v != g_dv, "mp_painn_message_bwd_f32: g_v must not alias g_dv (other waves still gather g_dv)");
  PainnMsgBwdArgs a{s, v, rbf, rbfd, env, envd, rij, Ww, bw, ptr1, perm1, recv, g_ds, g_dv, g_s, g_v, g_d, g_rij,
                    accumulate, N, M, B};
  int64_t blocks = mp::ceil_div(2 * N, 4);   // two waves (feature halves) per node
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = mp::as_stream(stream);
  if (B == 20) painn_message_bwd_kernel<20><<<static_cast<unsigned>(blocks), 256, 0, st>>>(a);
  else painn_message_bwd_kernel<0><<<static_cast<unsigned>(blocks), 256, 0, st>>>(a);
  return mp::check_launch("mp_painn_message_bwd_f32");
}

int mp_painn_update_pre_f32(const float* zp, const float* uv, int64_t N, float* c, float* prod, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_painn_update_pre_f32: bad size");
  if (N == 0) return MP_OK;
  MP_REQUIRE(zp && uv && c && prod, "mp_painn_update_pre_f32: null pointer");
  painn_update_pre_kernel<<<mp::grid_for(N * F), 256, 0, mp::as_stream(stream)>>>(zp, uv, N, c, prod);
  return mp::check_launch("mp_painn_update_pre_f32");
}

int mp_painn_update_post_f32(const float* zp, const float* vp, const float* uv, const float* prod, const float* a,
                             int64_t N, float* z2, float* v2, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_painn_update_post_f32: bad size");
  if (N == 0) return MP_OK;
  MP_REQUIRE(zp && vp && uv && prod && a && z2 && v2, "mp_painn_update_post_f32: null pointer");
  painn_update_post_kernel<<<mp::grid_for(N * F), 256, 0, mp::as_stream(stream)>>>(zp, vp, uv, prod, a, N, z2, v2);
  return mp::check_launch("mp_painn_update_post_f32");
}

int mp_painn_update_post_bwd_f32(const float* g_z2, const float* g_v2, const float* uv, const float* prod,
                                 const float* a, int64_t N, float* g_a, float* g_prod, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_painn_update_post_bwd_f32: bad size");
  if (N == 0) return MP_OK;
  MP_REQUIRE(g_z2 && g_v2 && uv && prod && a && g_a && g_prod, "mp_painn_update_post_bwd_f32: null pointer");
  painn_update_post_bwd_kernel<<<mp::grid_for(N * F), 256, 0, mp::as_stream(stream)>>>(g_z2, g_v2, uv, prod, a, N, g_a,
                                                                                       g_prod);
  return mp::check_launch("mp_painn_update_post_bwd_f32");
}

int mp_painn_update_pre_bwd_f32(const float* g_z2, const float* g_v2, const float* uv, const float* c, const float* a,
                                const float* g_prod, const float* g_c, int64_t N, float* g_zp, float* g_uv,
                                mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_painn_update_pre_bwd_f32: bad size");
  if (N == 0) return MP_OK;
  MP_REQUIRE(g_z2 && g_v2 && uv && c && a && g_prod && g_c && g_zp && g_uv, "mp_painn_update_pre_bwd_f32: null pointer");
  painn_update_pre_bwd_kernel<<<mp::grid_for(N * F), 256, 0, mp::as_stream(stream)>>>(g_z2, g_v2, uv, c, a, g_prod, g_c,
                                                                                      N, g_zp, g_uv);
  return mp::check_launch("mp_painn_update_pre_bwd_f32");
}

int mp_edge_geometry_bwd_f32(const float* g_d, const float* g_rij, int slices, const float* rij, const float* dist,
                             const int32_t* ptr0, const int32_t* perm0, const int32_t* ptr1, const int32_t* perm1,
                             int64_t N, int64_t M, float scale, float* g_xyz, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && slices >= 1, "mp_edge_geometry_bwd_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(ptr0 && ptr1 && g_xyz && (M == 0 || (g_d && g_rij && rij && dist)), "mp_edge_geometry_bwd_f32: null pointer");
  painn_geometry_bwd_kernel<<<mp::grid_for(N * 64), 256, 0, mp::as_stream(stream)>>>(g_d, g_rij, slices, rij, dist, ptr0,
                                                                                     perm0, ptr1, perm1, N, M, scale,
                                                                                     g_xyz);
  return mp::check_launch("mp_edge_geometry_bwd_f32");
}

}  // extern "C"
