// Fused GCN forward on 16-node tiles (kgcnn/literature/GCN.py:95-109, kgcnn/layers/conv/gcn_conv.py:85-90):
//
//   input launch :  n  = X W0 + b0                       GCN.py:97   (Dense on the raw node features, K up to thousands)
//                   h0 = n Wg0 + bg0                      gcn_conv.py:86 (lay_dense of the first GCN layer)
//   layer launch :  n  = act( sum_{e: recv(e)=i} w_e h[send(e)] )     gcn_conv.py:87-90 (gather, weighted pool, activation)
//                   h' = n Wg' + bg'                      the NEXT layer's lay_dense, or - after the last layer -
//                   out = MLP(n) [softmax]                GCN.py:107 (GraphMLP on the node embedding)
//
// i.e. 1 + depth launches per forward and no (N,units) intermediate but the h the next layer gathers from; the
// layer-by-layer path issues 14 launches (Dense, aggregate, Dense, ... , three Dense, softmax, cast) of 5-20 us each on
// the Cora-shaped graph of BASELINE config 5, nearly all of it launch / first-tile latency.
//
// One 256-thread workgroup owns 16 consecutive nodes:
//  * input GEMM (HBM / FP32-MFMA bound: 2708 x 1433 x 64): no LDS staging - the A operand of v_mfma_f32_16x16x4_f32 is
//    read straight from X with 16-B loads (rows of 1433 floats are only 4-B aligned: global_load_dwordx4 needs dword
//    alignment only), the k order inside a 16-k block permuted so that one load feeds four MFMAs; W0 rows likewise, the
//    output columns permuted (4j + c) so that one 16-B load feeds four column blocks.  The four waves split K
//    (block-cyclic) and their partial tiles are added in wave order through LDS: deterministic.
//  * aggregate: the tile's edge range is cut into equal contiguous chunks, one per thread group (units/4 lanes = one
//    16-B piece of a row per lane), so a 129-edge hub (Cora's largest receiver) is shared by all groups instead of
//    being one group's nine dependent rounds; every group sums its chunk per receiver in edge order and parks the
//    partial rows in LDS, the partials of a receiver are added in chunk order: deterministic, no atomics.
//  * the Dense layers that follow run on the tile in LDS (A operand) with the weights read from L2 (B operand: at most
//    128 x 128 floats per layer), 16 output columns per wave and step.
#include <type_traits>

#include "mp_common.h"

namespace {

using floatx4 = __attribute__((ext_vector_type(4))) float;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-B load from a 4-B aligned address

constexpr int TR = 16;       // nodes per tile
constexpr int LD = 132;      // LDS row stride of a tile (floats): widths up to 128, (4 row + k) mod 32 conflict-free
constexpr int MAXW = 128;    // largest layer width
// scratch: input mode - the four waves' partial tiles; aggregate mode - at most 16 + groups - 1 partial rows (the chunks
// are contiguous and ordered, so every chunk boundary splits at most one receiver): 47 x 32, 31 x 64 or 23 x 128 floats
constexpr int PART_FLOATS = 4 * TR * LD;

struct GcnLayer {
  const float* W;   // (K, units) Keras kernel
  const float* b;   // (units) or null
  int units;
  int act;
  float alpha;
};

struct GcnArgs {
  int64_t N;
  // input mode
  const float* x;        // (N, K)
  int64_t K;
  const float* W_in;     // (K, UA)
  const float* b_in;
  // aggregate mode
  const float* h;        // (N, UA) rows to gather
  const int32_t* ptr;    // (N+1) CSR over the receiver-sorted edge order
  const int32_t* perm;   // (M) position -> edge, or null when the list is receiver-sorted
  const int32_t* send;   // (M) sender of edge e
  const float* weight;   // (M) scalar edge weights, or null
  int64_t M;
  int agg_act;
  float agg_alpha;
  const int32_t* tile_start;   // (ntiles + 1) first node of every tile, or null: tiles of 16 consecutive nodes
  // chain
  int n_layers;
  GcnLayer layer[3];
  int softmax_last;
  float* out;            // (N, units of the last layer)
};

// One Dense layer on the tile: Tout = act(Tin (16 x K) @ W (K x U) + b); the last layer of the chain writes global rows.
__device__ __forceinline__ void dense_on_tile(const float* __restrict__ Tin, int K, const GcnLayer& L,
                                              float* __restrict__ Tout, float* __restrict__ gout, int64_t row0,
                                              int rows, int wave, int lane) {
  const int U = L.units;
  const int g = lane >> 4, cc = lane & 15;
  const float* ap = Tin + cc * LD + g;
  const int ksteps = (K + 3) >> 2;
  for (int cb = wave; cb * 16 < U; cb += 4) {
    const int col = cb * 16 + cc;
    const bool col_ok = col < U;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* wp = L.W + static_cast<int64_t>(g) * U + col;
    for (int s0 = 0; s0 < ksteps; s0 += 8) {   // eight k steps' operands requested before the first MFMA needs one
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = 4 * (s0 + u) + g;
        av[u] = k < K ? ap[4 * (s0 + u)] : 0.0f;
        bv[u] = (col_ok && k < K) ? wp[static_cast<int64_t>(4 * (s0 + u)) * U] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    const float bias = (col_ok && L.b) ? L.b[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * g + r;
      const float v = mp_apply_act(L.act, L.alpha, acc[r] + bias);
      if (gout) {
        if (col_ok && row < rows) gout[(row0 + row) * U + col] = v;
      } else if (col_ok) {
        Tout[row * LD + col] = v;
      }
    }
  }
}

// MODE 0: input GEMM; MODE 1: aggregate.  UA: width of the tile phase A produces (32, 64 or 128).
template <int MODE, int UA>
__global__ __launch_bounds__(256) void gcn_tile_kernel(GcnArgs a) {
  __shared__ __align__(16) float Ta[TR * LD];
  __shared__ __align__(16) float Tb[TR * LD];
  __shared__ __align__(16) float part[PART_FLOATS];
  __shared__ int seg[TR + 1];
  __shared__ unsigned touched[32];   // per thread group: bit i = the group's chunk holds edges of receiver i
  __shared__ int gbase[32];          // per thread group: first partial row of the group

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // aggregate launches may come with a tile table (at most 16 nodes AND a bounded number of edges per tile: a hub-heavy
  // stretch of nodes is cut into several tiles instead of setting the launch's time alone)
  const int64_t row0 = (MODE == 1 && a.tile_start) ? a.tile_start[blockIdx.x] : static_cast<int64_t>(blockIdx.x) * TR;
  const int64_t row_end = (MODE == 1 && a.tile_start) ? a.tile_start[blockIdx.x + 1] : (row0 + TR < a.N ? row0 + TR : a.N);
  const int rows = row_end - row0 < 0 ? 0 : (row_end - row0 > TR ? TR : static_cast<int>(row_end - row0));

  if constexpr (MODE == 0) {
    // ---- n = X W0 + b0 on the tile ----------------------------------------------------------------------------------
    constexpr int NCG = (UA + 63) / 64;          // groups of 64 output columns (one 16-B W load per lane and group)
    const int g = lane >> 4, cc = lane & 15;
    const int64_t K = a.K;
    const int64_t row = row0 + cc;
    const bool row_ok = row < a.N;
    const float* xrow = a.x + (row_ok ? row : 0) * K;
    floatx4 acc[NCG][4];
#pragma unroll
    for (int q = 0; q < NCG; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[q][c] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int64_t full = K / 16;                 // 16-k blocks that need no bounds checks
    // block kb: lane (cc, g) supplies X[row][16 kb + 4 g + i] for MFMA i, and W[16 kb + 4 g + i][64 q + 4 cc + c]
    auto load_a = [&](int64_t kb, f4u& av) {
      av = *reinterpret_cast<const f4u*>(xrow + 16 * kb + 4 * g);   // rows past N read row 0; their results are dropped
    };
    auto load_b = [&](int64_t kb, f4u (&bv)[NCG][4]) {
      const int64_t k0 = 16 * kb + 4 * g;
#pragma unroll
      for (int q = 0; q < NCG; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = 64 * q + 4 * cc;
          if constexpr (64 * NCG == UA) {
            bv[q][i] = *reinterpret_cast<const f4u*>(a.W_in + (k0 + i) * UA + col);
          } else {   // UA = 32: lanes 8..15 have no column
            bv[q][i] = *reinterpret_cast<const f4u*>(a.W_in + (k0 + i) * UA + (col < UA ? col : 0));
            if (col >= UA) bv[q][i] = f4u{0.f, 0.f, 0.f, 0.f};
          }
        }
    };
    auto mma_block = [&](const f4u& av, const f4u (&bv)[NCG][4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < NCG; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[q][i][c], acc[q][c], 0, 0, 0);
    };
    {
      // wave w owns blocks w, w + 4, ...  The A rows come from HBM (latency 1-2 us against 0.25 us of MFMA work per
      // block): seven blocks of A are in flight ahead of the one being multiplied (4 registers each); the W rows are L2
      // hits and 16 * NCG registers per block: three ahead.
      // Every load of the loop is unconditional (steps past the wave's last block re-read that block, unused): with a
      // load inside a conditional the compiler's wait-count bookkeeping falls back to s_waitcnt vmcnt(0) right behind
      // the prefetch - one full memory round trip per block (seen in the ISA; 15.4 us per launch).  Only the MFMAs of a
      // step are skipped, by a wave-uniform branch.
      const int64_t nsteps = wave < full ? (full - wave + 3) / 4 : 0;
      if (nsteps > 0) {
        auto blk = [&](int64_t step) { return wave + 4 * (step < nsteps ? step : nsteps - 1); };
        f4u av[8], bv[4][NCG][4];
#pragma unroll
        for (int u = 0; u < 7; ++u) load_a(blk(u), av[u]);
#pragma unroll
        for (int u = 0; u < 3; ++u) load_b(blk(u), bv[u]);
        for (int64_t t = 0; t < nsteps; t += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            load_a(blk(t + u + 7), av[(u + 7) & 7]);
            load_b(blk(t + u + 3), bv[(u + 3) & 3]);
            if (t + u < nsteps) mma_block(av[u], bv[u & 3]);
          }
        }
      }
    }
    if (K % 16 != 0 && wave == static_cast<int>(full & 3)) {   // the ragged last block, element-wise bounds
      const int64_t k0 = 16 * full + 4 * g;
      f4u av, bv[NCG][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = k0 + i < K;
        av[i] = (row_ok && ok) ? xrow[k0 + i] : 0.0f;
#pragma unroll
        for (int q = 0; q < NCG; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = 64 * q + 4 * cc + c;
            bv[q][i][c] = (ok && col < UA) ? a.W_in[(k0 + i) * UA + col] : 0.0f;
          }
      }
      mma_block(av, bv);
    }
    // the four waves' partial tiles, added in wave order
    float* mine = part + wave * (TR * LD);
#pragma unroll
    for (int q = 0; q < NCG; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 64 * q + 4 * cc + c;
        if (col < UA) {
#pragma unroll
          for (int r = 0; r < 4; ++r) mine[(4 * g + r) * LD + col] = acc[q][c][r];
        }
      }
    __syncthreads();
    for (int i = tid; i < TR * UA; i += 256) {
      const int r = i / UA, c = i % UA;
      float s = part[r * LD + c];
#pragma unroll
      for (int w = 1; w < 4; ++w) s += part[w * (TR * LD) + r * LD + c];
      Ta[r * LD + c] = s + (a.b_in ? a.b_in[c] : 0.0f);
    }
  } else {
    // ---- n_i = act(sum_e w_e h[send(e)]) for the 16 receivers of the tile ---------------------------------------
    constexpr int LPR = UA / 4;                  // lanes per row
    constexpr int NG = 256 / LPR;                // thread groups
    constexpr int ROWS = TR + 32;                // partial rows (at most 16 + groups - 1)
    constexpr int ECAP = (PART_FLOATS - ROWS * UA) / 3 < 1792 ? (PART_FLOATS - ROWS * UA) / 3 : 1792;   // edges per window
    // the tile's edge data staged in LDS (one coalesced pass): sender, weight, receiver within the tile
    int* s_send = reinterpret_cast<int*>(part + ROWS * UA);
    float* s_w = part + ROWS * UA + ECAP;
    int* s_nid = reinterpret_cast<int*>(part + ROWS * UA + 2 * ECAP);
    const int grp = tid / LPR, gl = tid % LPR;
    if (tid <= TR) {
      const int64_t n = tid < rows ? row0 + tid : row_end;
      int64_t p = a.ptr[n];
      p = p < 0 ? 0 : (p > a.M ? a.M : p);
      seg[tid] = static_cast<int>(p);
    }
    for (int i = tid; i < TR * LPR; i += 256)
      *reinterpret_cast<float4*>(Tb + (i / LPR) * LD + 4 * (i % LPR)) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int t_lo = seg[0], t_hi = seg[TR] < seg[0] ? seg[0] : seg[TR];
    // Windows of at most ECAP edges (one for all but hub-heavy tiles); the sums of a window are added to the running
    // tile in Tb in window order.
    for (int e_lo = t_lo; e_lo < t_hi; e_lo += ECAP) {
      const int e_hi = e_lo + ECAP < t_hi ? e_lo + ECAP : t_hi;
      for (int k = e_lo + tid; k < e_hi; k += 256) {
        const int e = a.perm ? a.perm[k] : k;
        const int j = a.send[e];
        s_send[k - e_lo] = j < 0 ? 0 : (j >= a.N ? static_cast<int>(a.N) - 1 : j);
        s_w[k - e_lo] = a.weight ? a.weight[e] : 1.0f;
        int r = 0;                               // receiver of position k: the last i with seg[i] <= k
#pragma unroll
        for (int step = 8; step > 0; step >>= 1)
          if (seg[r + step] <= k) r += step;
        s_nid[k - e_lo] = r;
      }
      const int per = (e_hi - e_lo + NG - 1) / NG;
      if (tid < 64) {
        // which receivers each group's chunk touches (from the CSR alone) and where its partial rows start
        unsigned m = 0u;
        if (tid < NG) {
          const int lo_q = e_lo + tid * per;
          const int hi_q = lo_q + per < e_hi ? lo_q + per : e_hi;
          for (int i = 0; i < TR; ++i) {
            const int lo = seg[i] > lo_q ? seg[i] : lo_q;
            const int hi = seg[i + 1] < hi_q ? seg[i + 1] : hi_q;
            if (lo < hi) m |= 1u << i;
          }
        }
        int incl = __popc(m);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          const int up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        if (tid < NG) {
          touched[tid] = m;
          gbase[tid] = incl - __popc(m);
        }
      }
      __syncthreads();
      {
        // the group's chunk in rounds of eight rows in flight, across receiver boundaries; a receiver's partial sum is
        // parked when the next receiver starts
        const int c_lo = e_lo + grp * per;
        const int c_hi = c_lo + per < e_hi ? c_lo + per : e_hi;
        int slot = gbase[grp];
        int cur = -1;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        auto round = [&](int e0, auto rows_tag) {
          constexpr int R = decltype(rows_tag)::value;
          int src[R], nid[R];
          float wv[R];
          float4 v[R];
#pragma unroll
          for (int u = 0; u < R; ++u) {
            const int k = (e0 + u < c_hi ? e0 + u : c_hi - 1) - e_lo;
            src[u] = s_send[k];
            wv[u] = s_w[k];
            nid[u] = s_nid[k];
          }
#pragma unroll
          for (int u = 0; u < R; ++u)
            v[u] = *reinterpret_cast<const float4*>(a.h + static_cast<int64_t>(src[u]) * UA + 4 * gl);
#pragma unroll
          for (int u = 0; u < R; ++u) {
            if (e0 + u < c_hi) {
              if (nid[u] != cur) {
                if (cur >= 0) {
                  *reinterpret_cast<float4*>(part + slot * UA + 4 * gl) = s;
                  ++slot;
                }
                cur = nid[u];
                s = make_float4(0.f, 0.f, 0.f, 0.f);
              }
              s.x += wv[u] * v[u].x; s.y += wv[u] * v[u].y; s.z += wv[u] * v[u].z; s.w += wv[u] * v[u].w;
            }
          }
        };
        int e0 = c_lo;
        for (; c_hi - e0 >= 16; e0 += 16) round(e0, std::integral_constant<int, 16>());
        for (; e0 < c_hi; e0 += 8) round(e0, std::integral_constant<int, 8>());
        if (cur >= 0) *reinterpret_cast<float4*>(part + slot * UA + 4 * gl) = s;
      }
      __syncthreads();
      for (int i = tid; i < TR * LPR; i += 256) {
        const int r = i / LPR, c4 = i % LPR;
        float4 s = *reinterpret_cast<const float4*>(Tb + r * LD + 4 * c4);
        for (int q = 0; q < NG; ++q) {
          const unsigned m = touched[q];
          if ((m >> r) & 1u) {
            const int slot = gbase[q] + __popc(m & ((1u << r) - 1u));
            const float4 p = *reinterpret_cast<const float4*>(part + slot * UA + 4 * c4);
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
          }
        }
        *reinterpret_cast<float4*>(Tb + r * LD + 4 * c4) = s;
      }
      __syncthreads();   // the staging area and the partial rows are rewritten by the next window
    }
    for (int i = tid; i < TR * LPR; i += 256) {
      const int r = i / LPR, c4 = i % LPR;
      const float4 s = *reinterpret_cast<const float4*>(Tb + r * LD + 4 * c4);
      float* d = Ta + r * LD + 4 * c4;
      d[0] = mp_apply_act(a.agg_act, a.agg_alpha, s.x);
      d[1] = mp_apply_act(a.agg_act, a.agg_alpha, s.y);
      d[2] = mp_apply_act(a.agg_act, a.agg_alpha, s.z);
      d[3] = mp_apply_act(a.agg_act, a.agg_alpha, s.w);
    }
  }
  __syncthreads();

  // ---- the Dense layers on the tile --------------------------------------------------------------------------------
  float* tin = Ta;
  float* tout = Tb;
  int K = UA;
  for (int l = 0; l < a.n_layers; ++l) {
    const bool last = l + 1 == a.n_layers;
    dense_on_tile(tin, K, a.layer[l], tout, (last && !a.softmax_last) ? a.out : nullptr, row0, rows, wave, lane);
    __syncthreads();
    K = a.layer[l].units;
    float* t = tin; tin = tout; tout = t;
  }
  if (a.softmax_last) {
    // Keras softmax over the last axis: exp(x - max) / sum, one thread per node row
    if (tid < rows) {
      const float* r = tin + tid * LD;
      float mx = -INFINITY;
      for (int c = 0; c < K; ++c) mx = fmaxf(mx, r[c]);
      float sum = 0.0f;
      for (int c = 0; c < K; ++c) sum += expf(r[c] - mx);
      float* o = a.out + (row0 + tid) * K;
      for (int c = 0; c < K; ++c) o[c] = expf(r[c] - mx) / sum;
    }
  } else if (a.n_layers == 0) {
    for (int i = tid; i < TR * UA; i += 256) {
      const int r = i / UA, c = i % UA;
      if (r < rows) a.out[(row0 + r) * UA + c] = tin[r * LD + c];
    }
  }
}

template <int MODE>
int launch_gcn(const GcnArgs& a, int ua, int64_t ntiles, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(ntiles);
  if (ua == 32) gcn_tile_kernel<MODE, 32><<<grid, 256, 0, s>>>(a);
  else if (ua == 64) gcn_tile_kernel<MODE, 64><<<grid, 256, 0, s>>>(a);
  else gcn_tile_kernel<MODE, 128><<<grid, 256, 0, s>>>(a);
  return mp::check_launch("mp_gcn_tile_f32");
}

}  // namespace

extern "C" int mp_gcn_tile_f32(const mp_gcn_tile_desc* d, mpStream_t stream) {
  MP_REQUIRE(d != nullptr, "mp_gcn_tile_f32: null descriptor");
  MP_REQUIRE(d->N >= 0 && d->N < (int64_t{1} << 31) * TR, "mp_gcn_tile_f32: bad N");
  MP_REQUIRE(d->units_in == 32 || d->units_in == 64 || d->units_in == 128,
             "mp_gcn_tile_f32: tile width %d (built for 32, 64, 128)", d->units_in);
  MP_REQUIRE(d->n_layers >= 0 && d->n_layers <= 3, "mp_gcn_tile_f32: 0..3 Dense layers per launch");
  for (int l = 0; l < d->n_layers; ++l) {
    MP_REQUIRE(d->layer[l].W != nullptr && d->layer[l].units >= 1 && d->layer[l].units <= MAXW,
               "mp_gcn_tile_f32: layer %d needs a kernel and 1..128 units", l);
    MP_REQUIRE(d->layer[l].act >= MP_ACT_LINEAR && d->layer[l].act <= MP_ACT_LAST, "mp_gcn_tile_f32: unknown activation");
  }
  MP_REQUIRE(!d->softmax_last || d->n_layers >= 1, "mp_gcn_tile_f32: softmax needs a Dense layer in front");
  if (d->N == 0) return MP_OK;
  MP_REQUIRE(d->out != nullptr, "mp_gcn_tile_f32: null output");
  GcnArgs a{};
  a.N = d->N;
  a.n_layers = d->n_layers;
  for (int l = 0; l < d->n_layers; ++l)
    a.layer[l] = GcnLayer{d->layer[l].W, d->layer[l].b, d->layer[l].units, d->layer[l].act, d->layer[l].alpha};
  a.softmax_last = d->softmax_last;
  a.out = d->out;
  hipStream_t s = mp::as_stream(stream);
  if (d->x != nullptr) {
    MP_REQUIRE(d->K >= 1 && d->W_in != nullptr, "mp_gcn_tile_f32: input mode needs K >= 1 and W_in");
    a.x = d->x; a.K = d->K; a.W_in = d->W_in; a.b_in = d->b_in;
    return launch_gcn<0>(a, d->units_in, mp::ceil_div(a.N, TR), s);
  }
  MP_REQUIRE(d->h != nullptr && d->ptr != nullptr && d->M >= 0 && d->M < (int64_t{1} << 31) && (d->M == 0 || d->send != nullptr),
             "mp_gcn_tile_f32: aggregate mode needs h, ptr, send");
  MP_REQUIRE(d->agg_act >= MP_ACT_LINEAR && d->agg_act <= MP_ACT_LAST, "mp_gcn_tile_f32: unknown activation");
  a.h = d->h; a.ptr = d->ptr; a.perm = d->perm; a.send = d->send; a.weight = d->weight; a.M = d->M;
  a.agg_act = d->agg_act; a.agg_alpha = d->agg_alpha;
  MP_REQUIRE(d->tile_start == nullptr || (d->n_tiles >= 1 && d->n_tiles < (int64_t{1} << 31)),
             "mp_gcn_tile_f32: a tile table needs n_tiles >= 1");
  a.tile_start = d->tile_start;
  return launch_gcn<1>(a, d->units_in, d->tile_start ? d->n_tiles : mp::ceil_div(a.N, TR), s);
}
