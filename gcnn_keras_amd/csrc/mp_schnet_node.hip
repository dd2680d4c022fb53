// Node-side chains of the SchNet forward (kgcnn/literature/Schnet.py:110-133, schnet_conv.py:159-165), fused per
// 16-node tile so that no intermediate (N,F) activation makes a round trip through HBM:
//
//   IN   : n = Dense(F, linear)(Embedding(Z))            Schnet.py:110-111,125      x = Dense_nobias(n)   conv:160
//   MID  : y = Dense(lin)(Dense(ssp)(agg)) ; n += y      schnet_conv.py:162-164     x = Dense_nobias(n)   (next block)
//   LAST : y = ...; n += y ; h = MLP([F, 64], ssp)(n)    Schnet.py:129  (last_mlp)  -> (N, 64)
//   readout: PoolingNodes(sum) + MLP([64, 1])            Schnet.py:133-135
//
// These chains are latency bound at QM9 batch sizes (2301 nodes = 144 tiles of 16): four waves of a workgroup
// cooperate on one 16-node tile, each wave producing a 32-column slice of every GEMM with v_mfma_f32_16x16x4_f32 (two
// independent 16x16 accumulators per wave cover its 40-cycle dependent latency).  The wave keeps its slice of
// every weight matrix in REGISTERS for the life of the persistent workgroup (64 VGPRs per 128x32 slice, loaded
// once with coalesced 64-B reads), so the only LDS traffic is the 16x128 activation tile handed from one GEMM to
// the next.  The MID / LAST kernels also re-zero the aggregation rows they consumed, so the next cfconv launch
// needs no memset.
#include <type_traits>

#include "mp_common.h"
#include "mp_edge_prepare.h"
#include "mp_node_tile.h"

// Diagnostic build (make diag -> libmpengine_diag.so, scripts/probe_node_diag.py): cycle stamps around the phases of the
// node-update tile loop.  Compiles to nothing in the product library.
#ifdef MP_NODE_DIAG
__device__ unsigned long long g_node_diag[8];
#define MP_NSTAMP(i)                                                      \
  {                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                    \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();        \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   \
    __builtin_amdgcn_sched_barrier(0);                                    \
    dsum[i] += t_now - t_prev;                                            \
    t_prev = t_now;                                                       \
  }
#else
#define MP_NSTAMP(i)
#endif

namespace {

struct NodeArgs {
  int64_t N;
  int ntiles;
  int keep_agg;          // do not re-zero the consumed aggregation rows (layer API: the rows belong to the caller)
  float* n_out;          // UPD: updated node state (n itself is read only)
  // IN
  const void* numbers;   // (N) node numbers: float32 (kgcnn/literature/Schnet.py:26) or int64 (the fork's force scripts)
  int numbers_i64;
  const float* emb;      // (vocab, E)
  int vocab;
  const float* W0;       // (E, F)
  const float* b0;       // (F)
  // MID / LAST
  float* agg;            // (N, F) aggregated messages, re-zeroed on exit
  const float* W2;       // (F, F) interaction dense2 (ssp)
  const float* b2;
  const float* W3;       // (F, F) interaction dense3 (linear)
  const float* b3;
  // all
  float* n;              // (N, F) node state (IN: written; MID/LAST: updated in place)
  const float* Wx;       // (F, F) next block's dense1 (no bias)          [IN, MID]
  float* x;              // (N, F) next block's sender features           [IN, MID]
  // LAST
  const float* Wl0;      // (F, F) last_mlp[0] (ssp)
  const float* bl0;
  const float* Wl1;      // (F, 64) last_mlp[1] (ssp)
  const float* bl1;
  float* h;              // (N, 64)
  // SAVE builds (energy + force pass): d act / d pre-activation of every ssp on the chain, kept for the reverse pass
  float* save_d2;        // (N, F)  sigmoid(agg W2 + b2)          [MID, LAST]
  float* save_dl0;       // (N, F)  sigmoid(n Wl0 + bl0)          [LAST]
  float* save_dl1;       // (N, 64) sigmoid(u Wl1 + bl1)          [LAST]
};

// UPD = MID without the next block's Dense_nobias, out of place: SchNetInteraction.call's node side on its own
enum NodeMode { NODE_IN = 0, NODE_MID = 1, NODE_LAST = 2, NODE_UPD = 3 };

// Epilogue helper: visit the wave's output elements.  C layout of 16x16x4: col = lane&15, row = 4*(lane>>4) + r.
#define MP_FOR_OUT(cb, r, row, col, body)                             \
  _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) {                 \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) {              \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                 \
        const int row = 16 * rb + 4 * (lane >> 4) + r;                \
        const int col = wave * (16 * NCB) + 16 * cb + (lane & 15);    \
        body                                                          \
      }                                                               \
    }                                                                 \
  }

// `block` / `nblocks`: this workgroup's position among the workgroups running the node chain (the stage-0 kernel runs
// edge preparation on the remaining workgroups of the same launch).
// BF: the GEMMs on the bf16 matrix pipe as an exact FP32 emulation (mp_node_tile.h: gemm_tile_bf), weights as
// mp_schnet_node_pack_bf16_f32 images (three bf16 pieces per element: 1.5x the bytes, 2.67x the matrix rate).
template <int MODE, int E, int RB, bool FAST, bool PACKED, bool SAVE = false, bool BF = false>
__device__ __forceinline__ void schnet_node_body(const NodeArgs& a, int block, int nblocks) {
  // activation tiles: FP32 rows (FP32 matrix instructions) or three bf16-piece planes written by the producer (BF builds:
  // mp_node_tile.h, "the activation tile split once")
  // BF builds run EIGHT waves on 16-column slices (NCB = 1): half the weight registers per wave (144 for the MID chain -
  // no slice parked in AGPRs and fetched back in front of its MFMA), two waves per SIMD to hide each other's LDS and
  // epilogue latencies; the pre-split tile is what makes the second wave per SIMD cheap (it reads pieces, it does not
  // split the tile again).  The weight images are the same: slice (wave w of 4, column block cb) = slice (wave 2 w + cb of 8).
  constexpr int NW = BF ? 8 : 4;
  constexpr int NCB = BF ? 1 : 2;
  constexpr int NT = 64 * NW;
  constexpr int TNR = 16 * RB;
  constexpr int TILE_BYTES = BF ? 3 * TNR * XP_LD * 2 : TNR * X_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char tile_a[TILE_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char tile_b[TILE_BYTES];
  float* const Xa = reinterpret_cast<float*>(tile_a);
  float* const Xb = reinterpret_cast<float*>(tile_b);
  unsigned short* const Pa = reinterpret_cast<unsigned short*>(tile_a);
  unsigned short* const Pb = reinterpret_cast<unsigned short*>(tile_b);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- input-tile staging through registers.  In the 16-node (latency-bound) build the first tile's rows are
  //      requested BEFORE the weight slices (vmcnt retires in order: the tile must not queue behind 192 weight loads)
  //      and the next tile's rows are requested while the current tile computes. ------------------------------------
  constexpr int TN = 16 * RB;
  constexpr bool EARLY_STAGE = RB == 1;
  // larger tiles: the next tile's rows are requested before the LAST big GEMM of the current tile (GEMM 3; GEMM 2 for the
  // input chain), into the registers that held the residual rows until the epilogue before it - no extra pressure
  constexpr bool LATE_STAGE = !EARLY_STAGE;
  constexpr int SV_IN = (TN * E) / NT;       // floats per thread (NODE_IN: embedding rows)
  constexpr int SV = (TN * F / 4) / NT;      // float4 per thread (aggregation rows)
  float stg_in[MODE == NODE_IN ? SV_IN : 1];
  unsigned stg_ok = 0u;   // NODE_IN: bit j = element j of stg_in is a real embedding element (else 0)
  float4 stg[MODE == NODE_IN ? 1 : SV];
  auto stage_load = [&](int t) {
    const int64_t n0 = static_cast<int64_t>(t) * TN;
    if constexpr (MODE == NODE_IN) {
      // two unconditional phases - every node number of the thread's elements, then every embedding element - so that the
      // chain is two round trips; written as one guarded (number -> element) pair per element the compiler waits for each
      // load before it issues the next: 2 * SV_IN dependent round trips (seen in the ISA, ~3 of stage 0's 5.2 us)
      const bool live = t < a.ntiles;
      const int64_t n_last = a.N > 0 ? a.N - 1 : 0;
      int z[SV_IN];
      if (a.numbers_i64) {
#pragma unroll
        for (int j = 0; j < SV_IN; ++j) {
          const int64_t node = n0 + (tid + j * NT) / E;
          z[j] = static_cast<int>(static_cast<const int64_t*>(a.numbers)[(live && node < a.N) ? node : n_last]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < SV_IN; ++j) {
          const int64_t node = n0 + (tid + j * NT) / E;
          // Keras Embedding casts its input to int32
          z[j] = static_cast<int>(static_cast<const float*>(a.numbers)[(live && node < a.N) ? node : n_last]);
        }
      }
#pragma unroll
      for (int j = 0; j < SV_IN; ++j) {
        const int zc = z[j] < 0 ? 0 : (z[j] >= a.vocab ? a.vocab - 1 : z[j]);
        stg_in[j] = a.emb[static_cast<int64_t>(zc) * E + (tid + j * NT) % E];
      }
      // validity is applied when the registers are written to LDS (stage_store): a select here would make the wave wait
      // for the embedding elements before it requests its weight slices
      stg_ok = 0u;
#pragma unroll
      for (int j = 0; j < SV_IN; ++j) {
        const int64_t node = n0 + (tid + j * NT) / E;
        if (live && node < a.N && z[j] >= 0 && z[j] < a.vocab) stg_ok |= 1u << j;
      }
    } else {
#pragma unroll
      for (int j = 0; j < SV; ++j) {
        const int i = tid + j * NT;
        const int r = i / (F / 4), k4 = i % (F / 4);
        const int64_t node = n0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < a.ntiles && node < a.N) {
          float4* p = reinterpret_cast<float4*>(a.agg + node * F) + k4;
          v = *p;
          if (!a.keep_agg) *p = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        stg[j] = v;
      }
    }
  };
  auto stage_store = [&]() {
    if constexpr (MODE == NODE_IN && BF) {
      static_assert(MODE != NODE_IN || !BF || SV_IN % 2 == 0, "element pairs");
#pragma unroll
      for (int j = 0; j + 1 < SV_IN; j += 2) {   // elements j, j + 1 of a thread: one column, rows NT / E apart
        const int i = tid + j * NT;
        put_split_rows<TNR>(Pa, i / E, (i + NT) / E, i % E, ((stg_ok >> j) & 1u) ? stg_in[j] : 0.0f,
                            ((stg_ok >> (j + 1)) & 1u) ? stg_in[j + 1] : 0.0f);
      }
    } else if constexpr (MODE == NODE_IN) {
#pragma unroll
      for (int j = 0; j < SV_IN; ++j) {
        const int i = tid + j * NT;
        Xa[(i / E) * X_LD + (i % E)] = ((stg_ok >> j) & 1u) ? stg_in[j] : 0.0f;
      }
    } else if constexpr (BF) {
#pragma unroll
      for (int j = 0; j < SV; ++j) {
        const int i = tid + j * NT;
        put_split4<TNR>(Pa, i / (F / 4), 4 * (i % (F / 4)), stg[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < SV; ++j) {
        const int i = tid + j * NT;
        float* d = Xa + (i / (F / 4)) * X_LD + 4 * (i % (F / 4));
        d[0] = stg[j].x; d[1] = stg[j].y; d[2] = stg[j].z; d[3] = stg[j].w;
      }
    }
  };
  // BF builds: an epilogue hands its output tile on as bf16 pieces (put_split_acc)
  float ov[BF ? RB : 1][NCB][4];
  auto put_tile = [&](unsigned short* P) {
    if constexpr (BF) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          put_split_acc<TNR>(P, lane, 16 * rb + 4 * (lane >> 4), wave * (16 * NCB) + 16 * cb + (lane & 15), ov[rb][cb]);
        }
    }
  };
  stage_load(block);

  // ---- weight slices -> registers (once per persistent workgroup) -------------------------------------------
  constexpr int K1 = MODE == NODE_IN ? E : F;
  constexpr int K3 = (MODE == NODE_IN || MODE == NODE_UPD) ? (BF ? 32 : 4) : F;   // (unused slices: smallest legal size)
  constexpr int K4 = MODE == NODE_LAST ? F : (BF ? 32 : 4);
  std::conditional_t<BF, WSliceBf<K1, NCB>, float[NCB][K1 / 4]> w_first;    // IN: W0 ; MID/LAST: W2
  std::conditional_t<BF, WSliceBf<F, NCB>, float[NCB][F / 4]> w_second;     // IN: Wx ; MID/LAST: W3
  std::conditional_t<BF, WSliceBf<K3, NCB>, float[NCB][K3 / 4]> w_third;    // MID: Wx ; LAST: Wl0
  std::conditional_t<BF, WSliceBf<K4, 1>, float[1][K4 / 4]> w_fourth;       // LAST: Wl1 (16 columns per wave, waves 0-3)
  float bias_first[NCB], bias_second[NCB], bias_third[NCB], bias_fourth;
  const bool has_fourth = wave < 4;   // (wave-uniform) the 64 columns of GEMM 4 are four 16-column slices
#define MP_LOAD_W(KK, NCB, PTR, UU, DST)                                          \
  if constexpr (BF) load_w_bf<KK, NCB>(PTR, wave, lane, DST);                     \
  else load_w<KK, NCB, PACKED>(PTR, UU, wave, lane, DST)
#define MP_GEMM(KK, NCB, XS, WW, ACC)                                             \
  if constexpr (BF) gemm_tile_pre<KK, NCB, RB, MODE != NODE_LAST>(reinterpret_cast<const unsigned short*>(XS), lane, WW, ACC); \
  else gemm_tile<KK, NCB, RB>(XS, lane, WW, ACC)
  if constexpr (MODE == NODE_IN) {
    MP_LOAD_W(E, NCB, a.W0, F, w_first);
    MP_LOAD_W(F, NCB, a.Wx, F, w_second);
  } else {
    MP_LOAD_W(F, NCB, a.W2, F, w_first);
    MP_LOAD_W(F, NCB, a.W3, F, w_second);
    if constexpr (MODE == NODE_MID) { MP_LOAD_W(F, NCB, a.Wx, F, w_third); }
    if constexpr (MODE == NODE_LAST) {
      MP_LOAD_W(F, NCB, a.Wl0, F, w_third);
      if (has_fourth) { MP_LOAD_W(F, 1, a.Wl1, 64, w_fourth); }
    }
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int col = wave * (16 * NCB) + 16 * cb + (lane & 15);
    if constexpr (MODE == NODE_IN) {
      bias_first[cb] = a.b0 ? a.b0[col] : 0.0f;
      bias_second[cb] = 0.0f;
      bias_third[cb] = 0.0f;
    } else {
      bias_first[cb] = a.b2 ? a.b2[col] : 0.0f;
      bias_second[cb] = a.b3 ? a.b3[col] : 0.0f;
      bias_third[cb] = (MODE == NODE_LAST && a.bl0) ? a.bl0[col] : 0.0f;
    }
  }
  bias_fourth = (MODE == NODE_LAST && a.bl1 && has_fourth) ? a.bl1[wave * 16 + (lane & 15)] : 0.0f;

#ifdef MP_NODE_DIAG
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int tile = block; tile < a.ntiles; tile += nblocks) {
    const int64_t node0 = static_cast<int64_t>(tile) * TN;
    MP_NSTAMP(7)

    // ---- stage the input tile into Xa (coalesced; the consumed aggregation rows were re-zeroed by stage_load) --------
    stage_store();   // the rows were requested during the previous tile (or before the loop)
    __syncthreads();
    MP_NSTAMP(0)

    if constexpr (EARLY_STAGE) stage_load(tile + nblocks);  // next tile, in flight during the GEMMs
    floatx4 acc[RB][NCB];
    // residual rows of n: requested well before the epilogue that adds them - never inside it, where every load would
    // have to wait behind the previous element's store to the same array (measured at 225 k nodes: 24 k of a tile's 48 k
    // cycles).  The 16-node build asks at tile start (hidden under GEMM 1 and 2), the larger tiles right before GEMM 2
    // so that the values are not live during GEMM 1.
    constexpr bool N_EARLY = RB == 1;
    float n_res[RB][NCB][4];
    if constexpr (MODE != NODE_IN && N_EARLY) {
      MP_FOR_OUT(cb, r, row, col, {
        n_res[rb][cb][r] = (node0 + row < a.N) ? a.n[(node0 + row) * F + col] : 0.0f;
      })
    }
    // ---- GEMM 1: IN: n = emb @ W0 + b0 ; MID/LAST: t = ssp(agg @ W2 + b2) ---------------------------------------
#define MP_ZERO_ACC                                              \
  _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)              \
      _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    MP_ZERO_ACC
    MP_GEMM(K1, NCB, Xa, w_first, acc);
    MP_NSTAMP(1)
    MP_FOR_OUT(cb, r, row, col, {
      float v = acc[rb][cb][r] + bias_first[cb];
      if constexpr (MODE != NODE_IN && SAVE) {
        float dv;
        v = ssp_with_grad<FAST>(v, dv);
        if (node0 + row < a.N) a.save_d2[(node0 + row) * F + col] = dv;
      } else if constexpr (MODE != NODE_IN) {
        v = ssp<FAST>(v);
      }
      if constexpr (BF) ov[rb][cb][r] = v;
      else Xb[row * X_LD + col] = v;
      if constexpr (MODE == NODE_IN) {
        if (node0 + row < a.N) a.n[(node0 + row) * F + col] = v;
      }
    })
    put_tile(Pb);
    __syncthreads();
    MP_NSTAMP(2)

    // ---- GEMM 2: IN: x = n @ Wx ; MID/LAST: n += t @ W3 + b3 ----------------------------------------------------
    MP_ZERO_ACC
    if constexpr (MODE == NODE_IN && LATE_STAGE) stage_load(tile + nblocks);
    if constexpr (MODE != NODE_IN && !N_EARLY) {
      MP_FOR_OUT(cb, r, row, col, {
        n_res[rb][cb][r] = (node0 + row < a.N) ? a.n[(node0 + row) * F + col] : 0.0f;
      })
    }
    MP_GEMM(F, NCB, Xb, w_second, acc);
    MP_NSTAMP(3)
    if constexpr (MODE == NODE_IN) {
      MP_FOR_OUT(cb, r, row, col, {
        if (node0 + row < a.N) a.x[(node0 + row) * F + col] = acc[rb][cb][r];
      })
    } else {
      MP_FOR_OUT(cb, r, row, col, {
        const bool ok = node0 + row < a.N;
        const float y = acc[rb][cb][r] + bias_second[cb];
        const float nn = n_res[rb][cb][r] + y;  // LazyAdd([node, x])
        if (ok && MODE == NODE_MID) a.n[(node0 + row) * F + col] = nn;
        if (ok && MODE == NODE_UPD) a.n_out[(node0 + row) * F + col] = nn;
        if constexpr (MODE != NODE_UPD) {
          if constexpr (BF) ov[rb][cb][r] = nn;
          else Xa[row * X_LD + col] = nn;
        }
      })
      if constexpr (MODE != NODE_UPD) {
      put_tile(Pa);
      __syncthreads();
      MP_NSTAMP(4)

      // ---- GEMM 3: MID: x = n @ Wx ; LAST: u = ssp(n @ Wl0 + bl0) ----------------------------------------------
      MP_ZERO_ACC
      if constexpr (LATE_STAGE) stage_load(tile + nblocks);
      MP_GEMM(K3, NCB, Xa, w_third, acc);
      MP_NSTAMP(5)
      if constexpr (MODE == NODE_MID) {
        MP_FOR_OUT(cb, r, row, col, {
          if (node0 + row < a.N) a.x[(node0 + row) * F + col] = acc[rb][cb][r];
        })
      } else {
        MP_FOR_OUT(cb, r, row, col, {
          float v = acc[rb][cb][r] + bias_third[cb];
          if constexpr (SAVE) {
            float dv;
            v = ssp_with_grad<FAST>(v, dv);
            if (node0 + row < a.N) a.save_dl0[(node0 + row) * F + col] = dv;
          } else {
            v = ssp<FAST>(v);
          }
          if constexpr (BF) ov[rb][cb][r] = v;
          else Xb[row * X_LD + col] = v;
        })
        put_tile(Pb);
        __syncthreads();
        // ---- GEMM 4 (LAST): h = ssp(u @ Wl1 + bl1), 64 output columns = 16 per wave ------------------------------
        if (has_fourth) {
        floatx4 acc4[RB][1];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc4[rb][0] = floatx4{0.f, 0.f, 0.f, 0.f};
        MP_GEMM(K4, 1, Xb, w_fourth, acc4);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rb + 4 * (lane >> 4) + r;
            const int col = wave * 16 + (lane & 15);
            if (node0 + row < a.N) {
              float v = acc4[rb][0][r] + bias_fourth;
              if constexpr (SAVE) {
                float dv;
                v = ssp_with_grad<FAST>(v, dv);
                a.save_dl1[(node0 + row) * 64 + col] = dv;
              } else {
                v = ssp<FAST>(v);
              }
              a.h[(node0 + row) * 64 + col] = v;
            }
          }
        }
      }
      }
    }
    __syncthreads();  // Xa / Xb are reused by the next tile
    MP_NSTAMP(6)
  }
#ifdef MP_NODE_DIAG
  if (tid == 0 && MODE == NODE_MID) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_node_diag[i], dsum[i]);
  }
#endif
#undef MP_LOAD_W
#undef MP_GEMM
}

template <int MODE, int E, int RB, bool FAST, bool PACKED, bool SAVE, bool BF = false>
// (the SAVE builds of the energy + force pass need a few registers more than 256, the bf16-piece builds hold 1.5x the
// weight registers: one workgroup per CU instead of a scratch segment, which every launch of the kernel would pay for)
__global__ __launch_bounds__(BF ? 512 : 256, (MODE != NODE_LAST && !SAVE && !BF) ? 2 : 1) void schnet_node_kernel(NodeArgs a) {
  schnet_node_body<MODE, E, RB, FAST, PACKED, SAVE, BF>(a, blockIdx.x, gridDim.x);
}

// Stage 0 of the fused forward: the node-input chain (Embedding -> Dense -> Dense_nobias) and the edge preparation
// (index shift, receiver/sender split, flags, distance) are independent, and at QM9 batch sizes each alone fills less
// than half of the chip for ~5 us - so one launch runs both, on disjoint workgroups (role by block index).
template <int E, bool FAST, bool LDS_SPLITS, bool PACKED, bool BF = false>
__global__ __launch_bounds__(BF ? 512 : 256) void schnet_stage0_kernel(NodeArgs a, mp_prep::EdgePrepArgs p, int node_blocks) {
  if (static_cast<int>(blockIdx.x) < node_blocks) {
    schnet_node_body<NODE_IN, E, 1, FAST, PACKED, false, BF>(a, blockIdx.x, node_blocks);
  } else {
    mp_prep::edge_prepare_body<LDS_SPLITS>(p, static_cast<int64_t>(blockIdx.x) - node_blocks,
                                           static_cast<int64_t>(gridDim.x) - node_blocks);
  }
}

// Readout: PoolingNodes(sum) over each graph's rows of h (N,64), then MLP([64,1], [ssp, linear]).
// One wave per graph, lane = feature; sequential node order (the order tf.math.segment_sum uses).  The 64x64 weight
// matrix is staged once per workgroup in LDS; the pooled vector is broadcast with v_readlane (scalar operand of the
// fma), so the 64-step dot product touches neither memory nor the LDS crossbar for its left operand.
__global__ __launch_bounds__(256) void schnet_readout_kernel(const float* __restrict__ h,
                                                             const int64_t* __restrict__ splits, int64_t G,
                                                             const float* __restrict__ Wo0,
                                                             const float* __restrict__ bo0,
                                                             const float* __restrict__ Wo1,
                                                             const float* __restrict__ bo1, float* __restrict__ out,
                                                             float* __restrict__ g_pool) {
  __shared__ float Ws[64 * 64];
  // g_pool (nullable, MLP head only): dE_g / d pooled_g (G, 64) = Wo0 (Wo1 * sigmoid(pre)) for the reverse pass; its
  // matrix-vector product reads Wo0 by rows, from a second LDS image padded to 65 floats per row (conflict-free)
  __shared__ float Wr[64 * 65];
  // Wo0 == NULL: linear head - the model ends in last_mlp [.., 64, 1(linear)] + PoolingNodes(sum) without an output MLP
  // (use_output_mlp=False, the fork's force configuration): out[g] = sum_n (h_n . Wo1 + bo1)
  const bool linear_head = Wo0 == nullptr;
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  // The wave's first graph: its row range is requested before the weight matrix and its first 24 rows right behind
  // it, so that the chain row_splits -> rows runs beside the staging of Wo0 instead of behind it (the kernel is a chain
  // of dependent round trips: 4.2 us for 128 graphs).
  int64_t lo0 = 0, hi0 = 0;
  if (wave_global < G) {
    lo0 = splits[wave_global];
    hi0 = splits[wave_global + 1];
  }
  float4 t[4];
  if (!linear_head) {
    // four 16-B loads per thread, all requested before the first is stored (a rolled copy loop waits for every load
    // before it issues the next: four serial round trips to L2 in a 4-us kernel)
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = reinterpret_cast<const float4*>(Wo0)[threadIdx.x + 256 * k];
  }
  constexpr int PRE = 24;   // three rounds of eight: covers most QM9 / MD17 molecules in one round trip
  float v0[PRE];
#pragma unroll
  for (int u = 0; u < PRE; ++u) v0[u] = 0.0f;
  if (hi0 > lo0) {   // wave-uniform
#pragma unroll
    for (int u = 0; u < PRE; ++u) v0[u] = h[(lo0 + u < hi0 ? lo0 + u : hi0 - 1) * 64 + lane];
  }
  if (!linear_head) {
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(Ws)[threadIdx.x + 256 * k] = t[k];
    if (g_pool) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = 4 * (threadIdx.x + 256 * k);   // element (i >> 6, i & 63 .. + 3)
        float* dst = Wr + (i >> 6) * 65 + (i & 63);
        dst[0] = t[k].x; dst[1] = t[k].y; dst[2] = t[k].z; dst[3] = t[k].w;
      }
    }
  }
  const float b0v = (!linear_head && bo0) ? bo0[lane] : 0.0f;
  const float w1v = Wo1[lane];
  const float b1v = bo1 ? bo1[0] : 0.0f;
  __syncthreads();
  for (int64_t g = wave_global; g < G; g += nwaves) {
    float pooled = 0.0f;
    const bool first = g == wave_global;
    const int64_t lo = first ? lo0 : splits[g], hi = first ? hi0 : splits[g + 1];
    // eight independent row loads in flight per step (a dependent one-row-at-a-time loop pays one L2 round trip per
    // node); the adds stay in node order
    for (int64_t base = lo; base < hi; base += 8) {
      float v[8];
      if (first && base == lo) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = v0[u];
      } else if (first && base == lo + 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = v0[8 + u];
      } else if (first && base == lo + 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = v0[16 + u];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (base + u < hi) ? h[(base + u) * 64 + lane] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) pooled += (base + u < hi) ? v[u] : 0.0f;
    }
    float y = pooled;
    if (!linear_head) {
      y = 0.0f;
#pragma unroll
      for (int k = 0; k < 64; ++k) y = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pooled), k)), Ws[k * 64 + lane], y);
      if (g_pool) {
        float dy;
        y = ssp_with_grad<false>(y + b0v, dy);
        const float c = dy * w1v;   // dE / d pre-activation of hidden unit `lane`
        float gp = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k)
          gp = fmaf(Wr[lane * 65 + k], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), k)), gp);
        g_pool[g * 64 + lane] = gp;
      } else {
        y = ssp_exact(y + b0v);
      }
    }
    float o = y * w1v;
    for (int off = 32; off > 0; off >>= 1) o += __shfl_xor(o, off, 64);
    if (lane == 0) out[g] = o + (linear_head ? static_cast<float>(hi - lo) * b1v : b1v);
  }
}

// 16-node tiles for every batch size: the build whose three weight slices, accumulators, staging and residual rows fit
// 250 registers without spills, with the next tile's rows and the residual rows in flight during the GEMMs.  Larger
// tiles (each weight register feeding 2 or 4 MFMAs) were measured slower once the epilogue's load-after-store
// serialisation was gone - 64-node tiles spill the weights, 32-node tiles spill ~25 registers whose reloads queue behind
// the tile prefetch (3371 vs 3328 us per forward at 225 k nodes, 397 vs 380 us at 18 k).  The persistent grid is two
// workgroups per CU.
template <int MODE, int E, bool FAST, bool PACKED, bool SAVE = false, bool BF = false>
int launch_node_impl(NodeArgs a, hipStream_t s, const char* what, int flags = 0) {
  a.ntiles = static_cast<int>((a.N + 15) / 16);
  // bf16-piece builds: one workgroup per CU.  Flag bit 9 ("several launch sequences in flight"): a launch with many tiles
  // runs on HALF the CUs - each workgroup loads its 288 KB of weight slices for twice the tiles and the other CUs serve the
  // other sequences' kernels: +3 % for four launch groups in flight (1115 -> 1151 M edges/s), costs a lone launch
  // latency (a single 128-graph forward 63.5 -> 70 us if it applied there: it does not, 144 tiles stay one per workgroup)
  const int cap = BF ? (((flags & 512) && a.ntiles >= 256) ? 128 : 256) : 512;
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  schnet_node_kernel<MODE, E, 1, FAST, PACKED, SAVE, BF><<<grid, BF ? 512 : 256, 0, s>>>(a);
  return mp::check_launch(what);
}

// Pre-packed image of a Keras kernel W (K, U), U = 64 * NCB: element i = (((w * NCB + cb) * (K/16) + q) * 64 + lane) * 4 + j
// holds W[k = 4 * (4q + j) + (lane >> 4)][col = w * (U/4) + 16 cb + (lane & 15)] - the order load_wslice_packed reads.
__global__ void node_pack_kernel(const float* __restrict__ W, int K, int U, float* __restrict__ packed) {
  const int ncb = U / 64, total = K * U;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63;
    int rest = i >> 8;
    const int q = rest % (K / 16);
    rest /= (K / 16);
    const int cb = rest % ncb, w = rest / ncb;
    const int k = 4 * (4 * q + j) + (lane >> 4);
    const int col = w * (U / 4) + 16 * cb + (lane & 15);
    packed[i] = W[k * U + col];
  }
}

// bf16-piece image of a Keras kernel W (K, U), U = 64 * NCB, for load_w_bf: 16 B per (wave w, column block cb, k block kb,
// piece, lane): element i = piece(W[32 kb + 8 (lane >> 4) + i][w (U/4) + 16 cb + (lane & 15)]); one dword = two elements
__global__ void node_pack_bf16_kernel(const float* __restrict__ W, int K, int U, float* __restrict__ packed) {
  const int ncb = U / 64, total = K * U * 3 / 2;   // dwords
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int d = i & 3, lane = (i >> 2) & 63;
    int rest = i >> 8;
    const int pc = rest % 3;
    rest /= 3;
    const int kb = rest % (K / 32);
    rest /= (K / 32);
    const int cb = rest % ncb, w = rest / ncb;
    const int col = w * (U / 4) + 16 * cb + (lane & 15);
    unsigned bits[2];
    for (int e = 0; e < 2; ++e) {
      const int k = 32 * kb + 8 * (lane >> 4) + 2 * d + e;
      const float x = W[k * U + col];
      const __bf16 p0 = static_cast<__bf16>(x);
      const float r1 = x - static_cast<float>(p0);
      const __bf16 p1 = static_cast<__bf16>(r1);
      const float r2 = r1 - static_cast<float>(p1);
      const __bf16 p2 = static_cast<__bf16>(r2);
      const __bf16 pick = pc == 0 ? p0 : (pc == 1 ? p1 : p2);
      bits[e] = static_cast<unsigned>(__builtin_bit_cast(unsigned short, pick));
    }
    packed[i] = __uint_as_float(bits[0] | (bits[1] << 16));
  }
}

template <int MODE, int E>
int launch_node(const NodeArgs& a, int flags, hipStream_t s, const char* what) {
  if constexpr (MODE == NODE_MID || MODE == NODE_LAST) {   // the energy + force pass: packed images only
    if (a.save_d2)
      return (flags & 1) ? launch_node_impl<MODE, E, true, true, true>(a, s, what, flags)
                         : launch_node_impl<MODE, E, false, true, true>(a, s, what, flags);
  }
  if constexpr (MODE != NODE_UPD) {   // flags bit 6 (with bit 1): mp_schnet_node_pack_bf16_f32 images, bf16-pipe GEMMs
    if ((flags & 66) == 66)
      return (flags & 1) ? launch_node_impl<MODE, E, true, true, false, true>(a, s, what, flags)
                         : launch_node_impl<MODE, E, false, true, false, true>(a, s, what, flags);
  }
  if constexpr (MODE != NODE_UPD) {   // flags bit 1: the weight pointers are mp_schnet_node_pack_f32 images
    if (flags & 2)
      return (flags & 1) ? launch_node_impl<MODE, E, true, true>(a, s, what, flags)
                         : launch_node_impl<MODE, E, false, true>(a, s, what, flags);
  }
  return (flags & 1) ? launch_node_impl<MODE, E, true, false>(a, s, what, flags)
                     : launch_node_impl<MODE, E, false, false>(a, s, what, flags);
}

template <int E>
void launch_stage0(const NodeArgs& a, const mp_prep::EdgePrepArgs& p, int node_blocks, unsigned grid, int flags_arg,
                   hipStream_t s) {
  if ((flags_arg & 66) == 66) {   // bf16-piece weight images
    // (eight-wave node workgroups; the edge-preparation workgroups of the launch are block-size agnostic: half as many)
    // at most 64 of them: every edge workgroup first stages both row-split arrays in LDS (10 KB at 640 graphs), which is
    // most of what a workgroup with 512 edges does - a launch group's 131 k edges on 64 workgroups (four edges per thread)
    // instead of 256: +2 % in flight (scripts/probes/ab_stage0_edge_cap.sh); a lone 128-graph batch has 51 anyway
    unsigned eblocks = (grid - static_cast<unsigned>(node_blocks) + 1) / 2;
    if (eblocks > 64) eblocks = 64;
    const unsigned grid8 = static_cast<unsigned>(node_blocks) + eblocks;
    if (flags_arg & 1) schnet_stage0_kernel<E, true, true, true, true><<<grid8, 512, 0, s>>>(a, p, node_blocks);
    else schnet_stage0_kernel<E, false, true, true, true><<<grid8, 512, 0, s>>>(a, p, node_blocks);
    return;
  }
  switch (flags_arg & 3) {
    case 0: schnet_stage0_kernel<E, false, true, false><<<grid, 256, 0, s>>>(a, p, node_blocks); break;
    case 1: schnet_stage0_kernel<E, true, true, false><<<grid, 256, 0, s>>>(a, p, node_blocks); break;
    case 2: schnet_stage0_kernel<E, false, true, true><<<grid, 256, 0, s>>>(a, p, node_blocks); break;
    default: schnet_stage0_kernel<E, true, true, true><<<grid, 256, 0, s>>>(a, p, node_blocks); break;
  }
}

}  // namespace

extern "C" {

#ifdef MP_NODE_DIAG
int mp_debug_node_diag(unsigned long long* out8_host) {  // diagnostic build only: read and clear the phase sums
  MP_HIP(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_node_diag), 64));
  const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_node_diag), zero, 64));
  return MP_OK;
}
#endif

// flags bit 8 (value 256): `numbers` points to int64 node numbers instead of float32
int mp_schnet_node_in_f32(const float* numbers, int64_t N, const float* emb, int vocab, int emb_dim, const float* W0,
                          const float* b0, const float* Wx, float* n_out, float* x_out, int flags, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && vocab >= 1, "mp_schnet_node_in_f32: bad sizes");
  MP_REQUIRE(emb_dim == 64 || emb_dim == 128, "mp_schnet_node_in_f32: built for embedding width 64 or 128 (got %d)",
             emb_dim);
  if (N == 0) return MP_OK;
  MP_REQUIRE(numbers && emb && W0 && Wx && n_out && x_out, "mp_schnet_node_in_f32: null pointer");
  NodeArgs a{};
  a.N = N;
  a.numbers = numbers; a.numbers_i64 = (flags & 256) ? 1 : 0;
  a.emb = emb; a.vocab = vocab; a.W0 = W0; a.b0 = b0; a.Wx = Wx; a.n = n_out; a.x = x_out;
  if (emb_dim == 128) return launch_node<NODE_IN, 128>(a, flags, mp::as_stream(stream), "mp_schnet_node_in_f32");
  return launch_node<NODE_IN, 64>(a, flags, mp::as_stream(stream), "mp_schnet_node_in_f32");
}

int mp_schnet_stage0_f32(const float* numbers, int64_t N, const float* emb, int vocab, int emb_dim, const float* W0,
                         const float* b0, const float* Wx, float* n_out, float* x_out, const int64_t* idx, int64_t M,
                         const int64_t* node_splits, const int64_t* edge_splits, int64_t G, const float* xyz,
                         int32_t* recv, int32_t* send, float* dist, int32_t* flags, int flags_arg, mpStream_t stream) {
  MP_REQUIRE(N >= 0 && M >= 0 && G >= 0 && vocab >= 1, "mp_schnet_stage0_f32: bad sizes");
  MP_REQUIRE(emb_dim == 64 || emb_dim == 128, "mp_schnet_stage0_f32: built for embedding width 64 or 128 (got %d)",
             emb_dim);
  MP_REQUIRE(N < (int64_t{1} << 31) && M < (int64_t{1} << 31), "mp_schnet_stage0_f32: N, M must fit int32");
  const int64_t tiles16 = (N + 15) / 16;
  if (N == 0 || M == 0 || tiles16 > 1024 || G > mp_prep::PREP_LDS_GRAPHS) {
    // outside the latency-bound regime the two stages run as their own (throughput-shaped) launches
    int rc = mp_edge_prepare_i64_f32(idx, M, node_splits, edge_splits, G, N, xyz, recv, send, dist, flags, stream);
    if (rc != MP_OK) return rc;
    return mp_schnet_node_in_f32(numbers, N, emb, vocab, emb_dim, W0, b0, Wx, n_out, x_out, flags_arg, stream);
  }
  MP_REQUIRE(numbers && emb && W0 && Wx && n_out && x_out && idx && node_splits && edge_splits && recv && send && flags,
             "mp_schnet_stage0_f32: null pointer");
  MP_REQUIRE((dist == nullptr) || (xyz != nullptr), "mp_schnet_stage0_f32: dist requested without coordinates");
  NodeArgs a{};
  a.N = N; a.ntiles = static_cast<int>(tiles16);
  a.numbers = numbers; a.numbers_i64 = (flags_arg & 256) ? 1 : 0;
  a.emb = emb; a.vocab = vocab; a.W0 = W0; a.b0 = b0; a.Wx = Wx; a.n = n_out; a.x = x_out;
  mp_prep::EdgePrepArgs p{idx, M, node_splits, edge_splits, G, N, xyz, recv, send, dist, flags};
  // node chain: persistent over the tiles beyond one workgroup per CU (a workgroup's weight slices - 72 KB as bf16 pieces -
  // are loaded once for its tiles, not once per tile: launch groups of five batches have 720 tiles)
  // eight-wave bf16-piece build: one workgroup per CU; half the CUs with flag bit 9 (launch_node_impl)
  const int node_cap = ((flags_arg & 66) == 66) ? (((flags_arg & 512) && a.ntiles >= 256) ? 128 : 256) : 512;
  const int node_blocks = a.ntiles < node_cap ? a.ntiles : node_cap;
  const int edge_blocks = static_cast<int>(mp::grid_for(M));
  hipStream_t s = mp::as_stream(stream);
  const unsigned grid = static_cast<unsigned>(node_blocks + edge_blocks);
  if (emb_dim == 128) launch_stage0<128>(a, p, node_blocks, grid, flags_arg, s);
  else launch_stage0<64>(a, p, node_blocks, grid, flags_arg, s);
  return mp::check_launch("mp_schnet_stage0_f32");
}

int mp_schnet_node_update_save_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                                   const float* b3, float* n_inout, const float* Wx_next, float* x_out, float* d2_out,
                                   int flags, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_schnet_node_update_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(agg && W2 && W3 && n_inout && Wx_next && x_out, "mp_schnet_node_update_f32: null pointer");
  MP_REQUIRE(d2_out == nullptr || (flags & 2), "mp_schnet_node_update_save_f32: saving needs packed weight images");
  NodeArgs a{};
  a.N = N;
  a.agg = agg; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.n = n_inout; a.Wx = Wx_next; a.x = x_out;
  a.save_d2 = d2_out;
  return launch_node<NODE_MID, 64>(a, flags, mp::as_stream(stream), "mp_schnet_node_update_f32");
}

int mp_schnet_node_update_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                              const float* b3, float* n_inout, const float* Wx_next, float* x_out, int flags,
                              mpStream_t stream) {
  return mp_schnet_node_update_save_f32(agg, N, W2, b2, W3, b3, n_inout, Wx_next, x_out, nullptr, flags, stream);
}

int mp_schnet_node_pack_f32(const float* W, int K, int U, float* packed, mpStream_t stream) {
  MP_REQUIRE(W && packed, "mp_schnet_node_pack_f32: null pointer");
  MP_REQUIRE(K >= 16 && K % 16 == 0 && U >= 64 && U % 64 == 0, "mp_schnet_node_pack_f32: K %% 16 == 0 and U %% 64 == 0 "
             "required (got %d x %d)", K, U);
  node_pack_kernel<<<64, 256, 0, mp::as_stream(stream)>>>(W, K, U, packed);
  return mp::check_launch("mp_schnet_node_pack_f32");
}

int mp_schnet_node_pack_bf16_f32(const float* W, int K, int U, float* packed, mpStream_t stream) {
  MP_REQUIRE(W && packed, "mp_schnet_node_pack_bf16_f32: null pointer");
  MP_REQUIRE(K >= 32 && K % 32 == 0 && U >= 64 && U % 64 == 0, "mp_schnet_node_pack_bf16_f32: K %% 32 == 0 and "
             "U %% 64 == 0 required (got %d x %d)", K, U);
  node_pack_bf16_kernel<<<64, 256, 0, mp::as_stream(stream)>>>(W, K, U, packed);
  return mp::check_launch("mp_schnet_node_pack_bf16_f32");
}

int mp_schnet_node_residual_f32(const float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                                const float* b3, const float* n_in, float* n_out, int flags, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_schnet_node_residual_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(agg && W2 && W3 && n_in && n_out, "mp_schnet_node_residual_f32: null pointer");
  NodeArgs a{};
  a.N = N;
  a.agg = const_cast<float*>(agg); a.keep_agg = 1;
  a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.n = const_cast<float*>(n_in); a.n_out = n_out;
  return launch_node<NODE_UPD, 64>(a, flags, mp::as_stream(stream), "mp_schnet_node_residual_f32");
}

int mp_schnet_node_last_save_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                                 const float* b3, const float* n_in, const float* Wl0, const float* bl0,
                                 const float* Wl1, const float* bl1, float* h_out, float* d2_out, float* dl0_out,
                                 float* dl1_out, int flags, mpStream_t stream) {
  MP_REQUIRE(N >= 0, "mp_schnet_node_last_f32: bad sizes");
  if (N == 0) return MP_OK;
  MP_REQUIRE(agg && W2 && W3 && n_in && Wl0 && Wl1 && h_out, "mp_schnet_node_last_f32: null pointer");
  MP_REQUIRE((d2_out == nullptr) == (dl0_out == nullptr) && (d2_out == nullptr) == (dl1_out == nullptr),
             "mp_schnet_node_last_save_f32: give all three derivative buffers or none");
  MP_REQUIRE(d2_out == nullptr || (flags & 2), "mp_schnet_node_last_save_f32: saving needs packed weight images");
  NodeArgs a{};
  a.N = N;
  a.agg = agg; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.n = const_cast<float*>(n_in);
  a.Wl0 = Wl0; a.bl0 = bl0; a.Wl1 = Wl1; a.bl1 = bl1; a.h = h_out;
  a.save_d2 = d2_out; a.save_dl0 = dl0_out; a.save_dl1 = dl1_out;
  return launch_node<NODE_LAST, 64>(a, flags, mp::as_stream(stream), "mp_schnet_node_last_f32");
}

int mp_schnet_node_last_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3, const float* b3,
                            const float* n_in, const float* Wl0, const float* bl0, const float* Wl1, const float* bl1,
                            float* h_out, int flags, mpStream_t stream) {
  return mp_schnet_node_last_save_f32(agg, N, W2, b2, W3, b3, n_in, Wl0, bl0, Wl1, bl1, h_out, nullptr, nullptr, nullptr,
                                      flags, stream);
}

int mp_schnet_readout_grad_f32(const float* h, const int64_t* node_splits, int64_t G, const float* Wo0,
                               const float* bo0, const float* Wo1, const float* bo1, float* out, float* g_pool_out,
                               mpStream_t stream) {
  MP_REQUIRE(G >= 0, "mp_schnet_readout_f32: bad sizes");
  if (G == 0) return MP_OK;
  MP_REQUIRE(h && node_splits && Wo1 && out, "mp_schnet_readout_f32: null pointer");
  MP_REQUIRE(g_pool_out == nullptr || Wo0 != nullptr, "mp_schnet_readout_grad_f32: g_pool is for the MLP head "
             "(the linear head's dE/dh is Wo1 itself)");
  schnet_readout_kernel<<<static_cast<unsigned>(mp::ceil_div(G, 4) < 1024 ? mp::ceil_div(G, 4) : 1024), 256, 0,
                          mp::as_stream(stream)>>>(h, node_splits, G, Wo0, bo0, Wo1, bo1, out, g_pool_out);
  return mp::check_launch("mp_schnet_readout_f32");
}

int mp_schnet_readout_f32(const float* h, const int64_t* node_splits, int64_t G, const float* Wo0, const float* bo0,
                          const float* Wo1, const float* bo1, float* out, mpStream_t stream) {
  return mp_schnet_readout_grad_f32(h, node_splits, G, Wo0, bo0, Wo1, bo1, out, nullptr, stream);
}

}  // extern "C"
