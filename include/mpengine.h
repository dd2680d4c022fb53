/*
 * mpengine.h - C ABI of the MI355X-native message-passing engine (libmpengine.so).
 *
 * Drop-in boundary for the ragged scatter-gather hot path of kgcnn (Tacitus523/gcnn_keras):
 * every entry point replaces a short composition of TensorFlow ops inside one reference function,
 * cited as file:line relative to the reference tree.  The reference has no FFI of its own (it is
 * pure Python on TensorFlow); INTEGRATION.md shows the ctypes binding a maintainer adds to
 * kgcnn/layers/{gather,pooling}.py to route through this library.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HIP, gfx950) unless the name ends in _host.  Plain pointers and
 *    sizes only; no framework types.  Values are float32 row-major; API indices / row_splits are int64
 *    exactly as the reference's RaggedTensor delivers them (kgcnn/data/utils.py:129-157).
 *  - Index convention: idx[e] = (i, j), column 0 = receiving node i, column 1 = sending node j
 *    (reference README.md:66); indices are per-graph local ("sample" indexing, kgcnn/layers/base.py:42-43).
 *  - Every call is asynchronous on the caller's stream (mpStream_t == hipStream_t), re-entrant, and keeps no
 *    pointer after it returns.  The caller allocates every input, output and workspace buffer.
 *  - Return: MP_OK or a negative status; mp_last_error() gives a thread-local message.  Host code maps
 *    MP_EINVAL -> ValueError/TypeError, MP_EINDEX -> IndexError (TF InvalidArgumentError), MP_EHIP -> RuntimeError.
 *  - Out-of-range indices never fault: they are clamped and recorded in a caller-provided device flag word
 *    (the analogue of ragged_validate / TF-GPU's silent zero fill); the caller decides when to read the flag.
 */
#ifndef MPENGINE_H
#define MPENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mpStream_t; /* hipStream_t */

enum mp_status { MP_OK = 0, MP_EINVAL = -1, MP_EINDEX = -2, MP_EHIP = -3, MP_ENOTSUP = -4 };

/* kgcnn/ops/segment.py:39-46 names -> op */
enum mp_reduce_op { MP_SUM = 0, MP_MEAN = 1, MP_MAX = 2, MP_MIN = 3 };

/* Activations reachable from SchNet / PaiNN / GCN configs (kgcnn/ops/activ.py:6-15, Keras strings) */
enum mp_activation {
  MP_ACT_LINEAR = 0, MP_ACT_RELU = 1, MP_ACT_SHIFTED_SOFTPLUS = 2, MP_ACT_SOFTPLUS = 3, MP_ACT_SWISH = 4,
  MP_ACT_SIGMOID = 5, MP_ACT_TANH = 6, MP_ACT_LEAKY_RELU = 7,
  MP_ACT_SOFTPLUS2 = 8, /* kgcnn/ops/activ.py:19-29: relu(x) + log(0.5 exp(-|x|) + 0.5) */
  MP_ACT_SELU = 9,      /* Keras "selu" (kgcnn/literature/NMPN.py:41): scale * (x > 0 ? x : alpha * (exp(x) - 1)) */
  MP_ACT_LAST = MP_ACT_SELU
};

enum mp_binary_op { MP_ADD = 0, MP_SUB = 1, MP_MUL = 2 };

/* bits of the device flag word written by mp_index_prepare_i64 */
enum mp_index_flag { MP_FLAG_OOB = 1, MP_FLAG_UNSORTED_COL0 = 2, MP_FLAG_UNSORTED_COL1 = 4 };

/* element types of the host packer (mp_pack_*_host) */
enum mp_dtype { MP_DT_F32 = 0, MP_DT_F64 = 1, MP_DT_I32 = 2, MP_DT_I64 = 3 };

/* ---------------------------------------------------------------- runtime -------------------------------- */
const char* mp_last_error(void);
int mp_version(void);
/* Fills name (<= name_len bytes), CU count and LDS bytes per CU of the current HIP device. */
int mp_device_info(char* name_host, int name_len, int* num_cu_host, int* lds_bytes_host);

/* HIP-graph capture of a launch sequence on `stream` (replaces eager per-op dispatch of Keras/TF). */
int mp_graph_begin(mpStream_t stream);
int mp_graph_end(mpStream_t stream, void** graph_exec_out_host);
int mp_graph_launch(void* graph_exec, mpStream_t stream);
int mp_graph_destroy(void* graph_exec);

/* HIP events on the caller's stream, for timing inside bench.py. */
int mp_event_create(void** event_out_host);
int mp_event_record(void* event, mpStream_t stream);
int mp_event_elapsed_ms(void* start, void* stop, float* ms_out_host); /* synchronises on `stop` */
int mp_event_destroy(void* event);

/* ---------------------------------------------------------------- index ops ------------------------------ */
/* kgcnn/ops/partition.py:97-162 partition_row_indexing (row_splits target, row_splits index):
 * out[e,k] = idx[e,k] + direction * node_splits[graph_of_edge(e)], direction = +1 sample->batch, -1 batch->sample.
 * Bit-exact int64. */
int mp_shift_index_i64(const int64_t* idx, int64_t M, int K, const int64_t* node_splits, const int64_t* edge_splits,
                       int64_t G, int direction, int64_t* out, mpStream_t stream);

/* One pass over the API's (M,K) int64 sample indices that every gather / pooling call of the reference recomputes
 * (partition.py:140-155 + gather.py:228 column pick): writes shifted int32 columns cols[k*M + e], and ORs
 * MP_FLAG_* bits into *flags (caller zeroes it): out-of-range (clamped) and per-column sortedness (K<=2 tracked).
 * N = total node count (node_splits[G]). */
int mp_index_prepare_i64(const int64_t* idx, int64_t M, int K, const int64_t* node_splits, const int64_t* edge_splits,
                         int64_t G, int64_t N, int32_t* cols, int32_t* flags, mpStream_t stream);

/* CSR offsets ptr[0..N] from receiver-sorted segment ids (what tf.math.segment_* derives implicitly). */
int mp_csr_from_sorted_i32(const int32_t* seg, int64_t M, int64_t N, int32_t* ptr, mpStream_t stream);

/* Stable argsort by segment id = tf.argsort(stable=True) of kgcnn/layers/pooling.py:66; ws from mp_sort_workspace_bytes. */
int mp_sort_workspace_bytes(int64_t M, size_t* bytes_out_host);
int mp_sort_segments_i32(const int32_t* seg, int64_t M, int32_t* seg_sorted, int32_t* perm, void* ws, size_t ws_bytes,
                         mpStream_t stream);

/* ---------------------------------------------------------------- gather --------------------------------- */
/* tf.gather(node, idx[:, col], axis=0) of kgcnn/layers/gather.py:83,228 on prepared int32 columns:
 * out[(e*ncols + c)*row_elems + f] = x[cols[colsel[c]*M + e]*row_elems + f]; ncols=2, colsel={0,1} gives the
 * GatherNodes concat layout [x_i || x_j] (gather.py:88-90).  Bit-exact copy. */
int mp_gather_rows_f32(const float* x, int64_t N, int64_t row_elems, const int32_t* cols, int64_t M, int ncols,
                       const int32_t* colsel_host, float* out, mpStream_t stream);

/* Same, straight from the API's int64 sample indices with the shift fused (no prepared columns). */
int mp_gather_rows_i64_f32(const float* x, int64_t N, int64_t row_elems, const int64_t* idx, int64_t M, int K,
                           int col /* -1: all K columns, concat layout */, const int64_t* node_splits,
                           const int64_t* edge_splits, int64_t G, float* out, mpStream_t stream);

/* GatherState, kgcnn/layers/gather.py:363-369: out = repeat(state, row_lengths(splits), axis=0). */
int mp_repeat_rows_f32(const float* state, const int64_t* splits, int64_t G, int64_t row_elems, int64_t N, float* out,
                       mpStream_t stream);

/* Embedding lookup of float node numbers (cast to int32 like Keras Embedding; kgcnn/layers/modules.py:526-528). */
int mp_embedding_f32(const float* table, int64_t vocab, int64_t dim, const float* numbers, int64_t N, float* out,
                     int32_t* flags, mpStream_t stream);

/* ---------------------------------------------------------------- segment reduce ------------------------- */
/* Sorted segment reduce of kgcnn/layers/pooling.py:63-76 + ops/segment.py:39-46 fused with the has_unconnected
 * zero pad: out[n] = op_{e in [ptr[n], ptr[n+1])} (weight[r] *) data[r], r = perm ? perm[e] : e, accumulated
 * sequentially in edge order (the order tf.math.segment_* uses after the stable sort); empty rows give 0 for
 * every op.  weight (M) nullable multiplies BEFORE the reduce (pooling.py:152); normalize_by_weight divides by
 * sum(weight) with divide_no_nan (pooling.py:164-165).  N_out rows are written. */
int mp_segment_reduce_csr_f32(int op, const float* data, int64_t M, int64_t row_elems, const int32_t* ptr,
                              const int32_t* perm, int64_t N_out, const float* weight, int normalize_by_weight,
                              float* out, mpStream_t stream);

/* GCN aggregation, kgcnn/layers/conv/gcn_conv.py:87-89 in one pass: GatherNodesOutgoing + PoolingWeightedLocalEdges +
 * Activation: out[n] = act( op_{e in segment n} weight[e] * x[send[e]] ); the (M,F) gathered rows never exist
 * (20M + 8NF algorithmic bytes = 125 B/edge at config 5).  send in original edge order, ptr/perm as above. */
int mp_gather_segment_reduce_csr_f32(int op, const float* x, int64_t N, int64_t row_elems, const int32_t* send,
                                     int64_t M, const int32_t* ptr, const int32_t* perm, int64_t N_out,
                                     const float* weight, int normalize_by_weight, int act, float act_alpha, float* out,
                                     mpStream_t stream);

/* PoolingEmbedding / PoolingNodes, kgcnn/layers/pooling.py:215-218: per-graph reduce driven by int64 row_splits
 * (value_rowids never materialised).  Writes G rows (the host trims trailing empty graphs like TF does). */
int mp_pool_graph_f32(int op, const float* x, const int64_t* row_splits, int64_t G, int64_t row_elems,
                      const float* weight, float* out, mpStream_t stream);

/* segment_softmax, kgcnn/ops/segment.py:5-24, on CSR segments: out[r] = exp(a[r]-max_seg)/sum_seg exp(...). */
int mp_segment_softmax_csr_f32(const float* a, int64_t M, int64_t row_elems, const int32_t* ptr, const int32_t* perm,
                               int64_t N, float* out, mpStream_t stream);

/* tensor_scatter_nd_{add,max,min}, kgcnn/ops/scatter.py:18-23 as used by RelationalPoolingLocalEdges
 * (pooling.py:658-666): out (N,R,F) must be zero-initialised by the caller; unsorted atomic scatter. */
int mp_scatter_relational_f32(int op, const float* edges, int64_t M, int64_t row_elems, const int32_t* recv,
                              const int32_t* relation, int64_t N, int64_t R, float* out, mpStream_t stream);

/* ---------------------------------------------------------------- dense / elementwise -------------------- */
/* Keras Dense on flat values, kgcnn/layers/modules.py:85: out = act(x @ W + b); x (R,K), W (K,U) row-major
 * ("kernel" layout), b (U) nullable.  FP32 MFMA (v_mfma_f32_32x32x2_f32), k-ordered fma chain. */
int mp_dense_f32(const float* x, int64_t R, int64_t K, const float* W, const float* b, int64_t U, int act,
                 float act_alpha, float* out, mpStream_t stream);

/* Dense with a fused prologue / epilogue, for chains of Dense layers and their reverse pass (PAiNNconv / PAiNNUpdate,
 * kgcnn/layers/conv/painn_conv.py:98-99, 206-207; EnergyForceModel's tape, kgcnn/model/force.py:159-177):
 * in_mode 0: as mp_dense_f32; 1: x := in_act(x) while staging (x is a saved pre-activation); 2: x := x * in_act'(in_pre)
 * (in_pre (R,K): the reverse pass through an activation, fused into the GEMM with the transposed kernel).
 * Epilogue, in this order: out_pre (R,U) nullable receives the pre-activation x W + b; the result act(x W + b) is
 * multiplied by in_act'(grad_pre) if grad_pre (R,U) is given (the activation derivative applied where the upstream
 * gradient is produced); addend (R,U) nullable, may alias out, is added last. */
int mp_dense_ex_f32(const float* x, int64_t R, int64_t K, const float* W, const float* b, int64_t U, int act,
                    float act_alpha, int in_mode, int in_act, float in_alpha, const float* in_pre, const float* addend,
                    float* out_pre, const float* grad_pre, float* out, mpStream_t stream);

/* mp_dense_f32 for few output tiles and a long contraction (GCN's first layer, kgcnn/literature/GCN.py:95:
 * (2708,1433) x (1433,64) = 43 tiles of 64x64 for 256 CUs): the k range is cut into `splits` slices computed by separate
 * workgroups into the workspace (splits * R * U floats), a second kernel adds the slices in order and applies bias and
 * activation - deterministic, no atomics. */
int mp_dense_splitk_workspace_bytes(int64_t R, int64_t U, int splits, size_t* bytes_out_host);
int mp_dense_splitk_f32(const float* x, int64_t R, int64_t K, const float* W, const float* b, int64_t U, int act,
                        float act_alpha, int splits, void* ws, size_t ws_bytes, float* out, mpStream_t stream);

int mp_activation_f32(int act, float act_alpha, const float* x, int64_t n, float* out, mpStream_t stream);
int mp_softmax_rows_f32(const float* x, int64_t R, int64_t C, float* out, mpStream_t stream);
/* GraphLayerNormalization over the last axis of the values (kgcnn/layers/norm.py:8-110 = Keras LayerNormalization):
 * out = (x - mean_row) * rsqrt(var_row + epsilon) * gamma + beta, biased variance; gamma / beta (C) nullable. */
int mp_layer_norm_f32(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float epsilon,
                      float* out, mpStream_t stream);

/* Broadcasting binary op on (R, D1, D2) views: operand strides in elements, 0 = broadcast
 * (LazyAdd/LazySubtract/LazyMultiply of kgcnn/layers/modules.py:187-301, incl. PaiNN's (M,1,F)*(M,3,F)). */
int mp_binary_f32(int op, const float* a, const int64_t* a_strides_host, const float* b, const int64_t* b_strides_host,
                  int64_t R, int64_t D1, int64_t D2, float* out, mpStream_t stream);

/* Strided 2-D column-block copy: dst[r*dst_ld + dst_off + c] = src[r*src_ld + src_off + c], c < C
 * (LazyConcatenate modules.py:305-364 and SplitEmbedding painn_conv.py:329-340 on values). */
int mp_copy_cols_f32(const float* src, int64_t src_ld, int64_t src_off, float* dst, int64_t dst_ld, int64_t dst_off,
                     int64_t R, int64_t C, mpStream_t stream);

/* ---------------------------------------------------------------- geometry ------------------------------- */
/* EuclideanNorm._compute_euclidean_norm, kgcnn/layers/geom.py:181-193 on an (R, D, C) view reducing D:
 * out (R,C) = [1/]sqrt(relu(sum_d x^2) [+eps]). flags: bit0 invert, bit1 add_eps, bit2 no_nan, bit3 square_norm. */
int mp_euclidean_norm_f32(const float* x, int64_t R, int64_t D, int64_t C, int flags, float* out, mpStream_t stream);
/* ScalarProduct, geom.py:261: out (R,C) = sum_d a*b. */
int mp_scalar_product_f32(const float* a, const float* b, int64_t R, int64_t D, int64_t C, float* out,
                          mpStream_t stream);
/* GaussBasisLayer, geom.py:567-571. */
int mp_gauss_basis_f32(const float* d, int64_t M, int bins, float distance, float sigma, float offset, float* out,
                       mpStream_t stream);
/* BesselBasisLayer, geom.py:772-785 (poly envelope, frequencies (num_radial) trainable weight). */
int mp_bessel_basis_f32(const float* d, int64_t M, const float* frequencies, int num_radial, float cutoff,
                        int envelope_exponent, float* out, mpStream_t stream);
/* CosCutOffEnvelope, geom.py:831-837. */
int mp_cos_cutoff_f32(const float* d, int64_t n, float cutoff, float* out, mpStream_t stream);

/* Fused NodePosition -> LazySubtract -> EuclideanNorm (Schnet.py:116-117, PAiNN.py:116-118) on prepared columns:
 * dist (M) = ||x_i - x_j||, dir (M,3) nullable = (x_i-x_j)*divide_no_nan(1, dist). */
int mp_edge_geometry_f32(const float* xyz, int64_t N, const int32_t* recv, const int32_t* send, int64_t M,
                         float* dist, float* dir, mpStream_t stream);

/* ChangeTensorType ragged -> (padded, mask), kgcnn/layers/casting.py:79-84. */
int mp_ragged_to_padded_f32(const float* values, const int64_t* row_splits, int64_t G, int64_t Nmax, int64_t row_elems,
                            float* padded, float* mask, mpStream_t stream);

/* ---------------------------------------------------------------- fused SchNet path --------------------- */
/* mp_index_prepare_i64 for K = 2 fused with NodePosition -> LazySubtract -> EuclideanNorm
 * (kgcnn/literature/Schnet.py:116-117): recv (M), send (M) shifted int32 ids, dist (M) nullable = ||x_i - x_j||,
 * flags as mp_index_prepare_i64 (column 0 sortedness only). */
int mp_edge_prepare_i64_f32(const int64_t* idx, int64_t M, const int64_t* node_splits, const int64_t* edge_splits,
                            int64_t G, int64_t N, const float* xyz, int32_t* recv, int32_t* send, float* dist,
                            int32_t* flags, mpStream_t stream);

/* SchNetCFconv.call, kgcnn/layers/conv/schnet_conv.py:73-79, fused (F = units = 128, cfconv_pool = sum,
 * activation = shifted_softplus): out[i] += sum_{e: recv[e]=i} x[send[e]] * (ssp(rbf[e] W1 + b1) W2 + b2).
 * The filter-MLP weights (Keras layouts W1 (B,128), b1 (128)|NULL, W2 (128,128), b2 (128)|NULL) are packed once
 * per weight update into the kernel's LDS image by mp_cfconv_pack_f32 (mp_cfconv_packed_floats() floats).
 * recv_sorted ascending; perm (nullable) maps sorted position -> original edge (rbf / send are in original
 * order); out (N,128) must be zero on entry.  flags bit0: fast softplus (v_exp/v_log form, |delta| < 2e-7);
 * bit2 / bit3 force the 8-wave / 4-wave workgroup (default: by tile count). */
int mp_cfconv_packed_floats(void);
int mp_cfconv_pack_f32(const float* W1, const float* b1, int B, const float* W2, const float* b2, float* packed,
                       mpStream_t stream);
int mp_cfconv_fused_f32(const float* x, int64_t N, const float* rbf, int B, const float* packed,
                        const int32_t* recv_sorted, const int32_t* send, const int32_t* perm, int64_t M, int flags,
                        float* out_zeroed, mpStream_t stream);
/* Same with GaussBasisLayer (kgcnn/layers/geom.py:567-571) expanded in registers from the edge distance. */
int mp_cfconv_gauss_fused_f32(const float* x, int64_t N, const float* dist, int bins, float distance, float sigma,
                              float offset, const float* packed, const int32_t* recv_sorted, const int32_t* send,
                              const int32_t* perm, int64_t M, int flags, float* out_zeroed, mpStream_t stream);

/* The same two entry points with a caller workspace, which flags bit 5 (value 32) needs: DETERMINISTIC mode.  The partial
 * sums of every tile's first and last segment (the ones that may continue in a neighbouring 32-edge tile) are then not
 * added to `out` with float atomics but parked in the workspace and added per receiver, in edge order, by a second small
 * kernel: bit-identical results from run to run also for receivers with more than 32 incoming edges (the default mode can
 * differ there in the last bit, float atomics arrive in any order).  Costs one extra launch and 1 KB of workspace per tile;
 * without bit 5 the workspace is ignored (NULL allowed). */
int mp_cfconv_det_workspace_bytes(int64_t M, size_t* bytes_out_host);
int mp_cfconv_fused_ws_f32(const float* x, int64_t N, const float* rbf, int B, const float* packed,
                           const int32_t* recv_sorted, const int32_t* send, const int32_t* perm, int64_t M, int flags,
                           float* out_zeroed, void* ws, size_t ws_bytes, mpStream_t stream);
int mp_cfconv_gauss_fused_ws_f32(const float* x, int64_t N, const float* dist, int bins, float distance, float sigma,
                                 float offset, const float* packed, const int32_t* recv_sorted, const int32_t* send,
                                 const int32_t* perm, int64_t M, int flags, float* out_zeroed, void* ws, size_t ws_bytes,
                                 mpStream_t stream);

/* Reverse pass of SchNetCFconv for forces (kgcnn/model/force.py:159-177 through schnet_conv.py:73-79, geom.py:567-571).
 * dE/dx_j is the forward kernel itself with the two index columns swapped (g_out in place of x, the sender-sorted list in
 * place of the receiver-sorted one).  mp_cfconv_gauss_dist_grad_f32 is the other half: per edge (original order)
 *   g_d[e] (+)= sum_h (v_e W2^T)_h * ssp'(g(d_e) W1 + b1)_h * (g'(d_e) W1)_h ,  v_e = g_out[recv[e]] * x[send[e]]
 * with the weights in the image of mp_cfconv_bwd_pack_f32 (mp_cfconv_bwd_packed_floats() floats: W1 | b1 as the forward
 * packs them, W2 re-ordered for the A operand).  Same FP32 MFMA tiles as the forward (three chains per 32-edge tile). */
int mp_cfconv_bwd_packed_floats(void);
int mp_cfconv_bwd_pack_f32(const float* W1, const float* b1, int B, const float* W2, float* packed, mpStream_t stream);
int mp_cfconv_gauss_dist_grad_f32(const float* x, const float* g_out, int64_t N, const float* dist, int bins,
                                  float distance, float sigma, float offset, const float* packed_bwd, const int32_t* recv,
                                  const int32_t* send, int64_t M, int accumulate, float* g_d, mpStream_t stream);

/* Diagnostic build of mp_cfconv_gauss_fused_f32 (20 bins, fast softplus): adds per-phase shader-cycle sums into
 * diag8 (8 x uint64, caller-zeroed): [0] weight staging, [1] tile setup + Gauss basis, [2] GEMM1, [3] softplus +
 * sender-row loads issued, [4] GEMM2, [5] multiply + slab write, [6] slab read, [7] segmented sum + stores/atomics.
 * Shares only: the stamps fence the schedule. */
int mp_cfconv_gauss_diag_f32(const float* x, int64_t N, const float* dist, int bins, float distance, float sigma,
                             float offset, const float* packed, const int32_t* recv_sorted, const int32_t* send,
                             const int32_t* perm, int64_t M, float* out_zeroed, unsigned long long* diag8,
                             mpStream_t stream);

/* PAiNNconv.call edge side, kgcnn/layers/conv/painn_conv.py:99-113, fused (F = units = 128, conv_pool = sum):
 * w = rbf Ww + bw [* env]; sw = s[send] * w; ds[i] = sum sw1; dv[i] = sum (sw2 (x) v[send] + sw3 (x) r_ij) over the edges
 * of receiver i in edge order (CSR ptr / perm as mp_segment_reduce_csr_f32).  s (N,3F) = phi(dense1(z)), v (N,3,F),
 * rbf (M,B<=32), env (M)|NULL, rij (M,3), Ww (B,3F), bw (3F)|NULL; writes ds (N,F), dv (N,3,F) incl. zero rows. */
int mp_painn_message_fused_f32(const float* s, const float* v, int64_t N, const float* rbf, int B, const float* env,
                               const float* rij, const float* Ww, const float* bw, const int32_t* ptr,
                               const int32_t* perm, const int32_t* send, int64_t M, float* ds, float* dv,
                               mpStream_t stream);

/* ---------------------------------------------------------------- recurrent / edge-network pieces -------- */
/* One Keras LSTM step from the zero state, as PoolingSet2Set runs it (kgcnn/layers/pool/set2set.py:190: a stateless LSTM
 * layer called on a length-1 sequence): z (R,4U) = x kernel + bias in Keras gate order [i | f | c | o];
 * out (R,U) = rec(z_o) * act(rec(z_i) * act(z_c))  (h0 = c0 = 0: the forget gate and the recurrent kernel drop out). */
int mp_lstm_zero_state_f32(const float* z, int64_t R, int64_t U, int act, int rec_act, float* out, mpStream_t stream);
/* Keras GRUCell combine, reset_after = True (kgcnn/layers/conv/mpnn_conv.py:193 GRUUpdate): mx (R,3U) = x kernel + b_in,
 * mh (R,3U) = h recurrent_kernel + b_rec, gate order [z | r | h]:
 * zg = rec(mx_z + mh_z), rg = rec(mx_r + mh_r), hh = act(mx_h + rg * mh_h), out = zg * h + (1 - zg) * hh. */
int mp_gru_combine_f32(const float* mx, const float* mh, const float* h, int64_t R, int64_t U, int act, int rec_act,
                       float* out, mpStream_t stream);
/* MatMulMessages, kgcnn/layers/conv/mpnn_conv.py:111 (tf.keras.backend.batch_dot): out[m] = mat[m] (Ro,C) . vec[m] (C). */
int mp_batched_matvec_f32(const float* mat, const float* vec, int64_t M, int64_t Ro, int64_t C, float* out,
                          mpStream_t stream);

/* ---------------------------------------------------------------- fused PaiNN path (+ reverse pass) ------ */
/* Stage 0 of kgcnn/literature/PAiNN.py:100-119 in one launch: OptionalInputEmbedding (z0 (N,128)), EquivariantInitialize
 * with a constant (v0 (N,3,128) = v_init), mp_index_prepare_i64 for K = 2 (recv, send, flag word incl. both sortedness
 * bits), NodePosition -> EdgeDirectionNormalized (rij (M,3), geom.py:331-378) -> NodeDistanceEuclidean (dist (M)) ->
 * BesselBasisLayer (rbf (M,B), geom.py:772-785) and, if rbfd != NULL, d rbf / d dist (M,B) for the reverse pass;
 * cos_cutoff > 0 adds CosCutOffEnvelope (env (M), geom.py:831-837) and its derivative envd (nullable). */
int mp_painn_stage0_f32(const void* numbers /* float32, or int64 if numbers_i64 */, int numbers_i64, int64_t N,
                        const float* emb, int vocab, float v_init, float* z0, float* v0,
                        const int64_t* idx, int64_t M, const int64_t* node_splits, const int64_t* edge_splits, int64_t G,
                        const float* xyz, const float* frequencies, int num_radial, float bessel_cutoff,
                        int envelope_exponent, float cos_cutoff, int32_t* recv, int32_t* send, int32_t* flags, float* dist,
                        float* rij, float* rbf, float* rbfd, float* env, float* envd, mpStream_t stream);
/* mp_painn_message_fused_f32 with the residual adds of PAiNN.py:126-127 fused (z_in != NULL: ds := z_in + ds,
 * dv := v + dv; dv must not alias v), four edges in flight per wave, packed FP32 arithmetic. */
int mp_painn_message_f32(const float* s, const float* v, int64_t N, const float* rbf, int B, const float* env,
                         const float* rij, const float* Ww, const float* bw, const int32_t* ptr, const int32_t* perm,
                         const int32_t* send, int64_t M, const float* z_in, float* ds, float* dv, mpStream_t stream);
/* The same message step with node TILES staged in LDS and the filter Dense (painn_conv.py:100, w = Dense(3F)(rbf)) on the
 * matrix pipe (FP32-exact bf16-piece split, csrc/mp_painn_fused.hip): for receiver-sorted edge lists (no perm) of batched
 * graphs.  tiles (T,8) int32 = {r_lo, r_hi, s_lo, s_hi, e_lo, e_hi, 0, 0}: receivers [r_lo, r_hi) (at most 62) of ONE graph
 * whose nodes are [s_lo, s_hi) (<= max_rows: the s / v rows staged) and whose edges are [e_lo, e_hi) = [ptr[r_lo],
 * ptr[r_hi]) (<= max_edges); every receiver must appear in exactly one tile.  wimage: mp_painn_filter_pack_f32 of Ww | bw
 * (MP_PAINN_FILTER_IMAGE_BYTES, rebuilt when the weights change).  LDS per workgroup: mp_painn_message_tiles_lds_bytes
 * (<= 160 KB, else MP_EINVAL).  Results as mp_painn_message_f32 (same accumulation order per receiver). */
#define MP_PAINN_FILTER_IMAGE_BYTES 73728
int mp_painn_filter_pack_f32(const float* Ww, const float* bw, int B, void* image, mpStream_t stream);
int mp_painn_message_tiles_lds_bytes(int max_rows, int max_edges, int B, int with_env, size_t* out);
int mp_painn_message_tiles_f32(const float* s, const float* v, int64_t N, const float* rbf, int B, const float* env,
                               const float* rij, const void* wimage, const int32_t* ptr, const int32_t* send, int64_t M,
                               const int32_t* tiles, int ntiles, int max_rows, int max_edges, const float* z_in, float* ds,
                               float* dv, mpStream_t stream);
/* Reverse pass of the message block (painn_conv.py:99-113) for forces: sender-parallel over the CSR of column 1
 * (ptr1 / perm1).  Given g_ds (N,F) and g_dv (N,3,F): g_s (N,3F) = dE/ds, g_v (N,3,F) = g_dv + dE/dv through the
 * messages (nullable), and per edge dE/dd (through the filter: rbfd = d rbf / d d, envelope by the product rule) and
 * dE/dr_ij, written (accumulate = 0) or added (accumulate = 1, later blocks) to g_d (2,M), g_rij (2,M,3): one slice per
 * half of the feature axis (two waves serve a sender), summed by mp_edge_geometry_bwd_f32 (slices = 2). */
int mp_painn_message_bwd_f32(const float* s, const float* v, int64_t N, const float* rbf, const float* rbfd, int B,
                             const float* env, const float* envd, const float* rij, const float* Ww, const float* bw,
                             const int32_t* ptr1, const int32_t* perm1, const int32_t* recv, int64_t M, const float* g_ds,
                             const float* g_dv, float* g_s, float* g_v, float* g_d, float* g_rij, int accumulate,
                             mpStream_t stream);
/* The same reverse step on SENDER tiles staged in LDS with the filter AND its distance derivative on the matrix pipe (A rows
 * [rbf_e | 1] and [rbf'_e | 0] against the forward kernel's packed image `wimage`): any edge order (the tile's basis rows are
 * gathered through perm1), batched graphs.  tiles (T,8) int32 = {j_lo, j_hi, s_lo, s_hi, e_lo, e_hi, 0, 0}: senders
 * [j_lo, j_hi) (at most max_senders <= 62: their s / v rows are staged too) of ONE graph whose nodes are [s_lo, s_hi)
 * (<= max_rows: the g_ds / g_dv rows staged) and whose
 * edges are the sender-order positions [e_lo, e_hi) = [ptr1[j_lo], ptr1[j_hi]) (<= max_edges); every sender in exactly one
 * tile.  g_d (M), g_rij (M,3): ONE slice (mp_edge_geometry_bwd_f32 with slices = 1), summed over the feature axis in a
 * fixed order inside the kernel.  LDS per workgroup: mp_painn_message_bwd_tiles_lds_bytes (<= 160 KB, else MP_EINVAL). */
int mp_painn_message_bwd_tiles_lds_bytes(int max_rows, int max_senders, int max_edges, int B, int with_env, size_t* out);
int mp_painn_message_bwd_tiles_f32(const float* s, const float* v, int64_t N, const float* rbf, const float* rbfd, int B,
                                   const float* env, const float* envd, const float* rij, const void* wimage,
                                   const int32_t* ptr1, const int32_t* perm1, const int32_t* recv, int64_t M,
                                   const int32_t* tiles, int ntiles, int max_rows, int max_senders, int max_edges,
                                   const float* g_ds, const float* g_dv, float* g_s, float* g_v, float* g_d, float* g_rij,
                                   int accumulate, mpStream_t stream);
/* PAiNNUpdate.call (painn_conv.py:201-214) around its GEMMs; uv (3N,2F) = v [Wu | Wv] (rows (n,k)):
 * pre:  c (N,2F) = [z | EuclideanNorm_k(v_v)], prod (N,F) = ScalarProduct_k(v_u, v_v);
 * post: z2 = z + prod a_sv + a_ss, v2 = v + a_vv (x) v_u  with a (N,3F) = [a_vv | a_sv | a_ss] (+ PAiNN.py:131-132);
 * post_bwd: g_a (N,3F), g_prod (N,F) from g_z2, g_v2;  pre_bwd: g_z = g_z2 + g_c[:, :F], g_uv (3N,2F). */
int mp_painn_update_pre_f32(const float* z, const float* uv, int64_t N, float* c, float* prod, mpStream_t stream);
/* pre + Dense(act) + Dense + post of one PAiNNUpdate in ONE launch (csrc/mp_chain.hip): the element-wise steps as
 * prologue / epilogue of the two-layer chain (W1 (256,128), W2 (128,384) as mp_chain_pack_f32 images).  save_pre keeps
 * c Wd + bd; c_out / prod_out / a_out (any may be NULL) receive what the reverse pass reads. */
int mp_painn_update_fused_f32(const float* z, const float* v, const float* uv, int64_t N, const float* W1_packed,
                              const float* b1, int act1, float alpha1, float* save_pre, const float* W2_packed,
                              const float* b2, float* c_out, float* prod_out, float* a_out, float* z2, float* v2,
                              mpStream_t stream);
/* ... and its reverse in one launch: post_bwd as the prologue, pre_bwd as the epilogue of the chain against the transposed
 * kernels (W1T (384,128) = Wa^T, W2T (128,256) = Wd^T as mp_chain_pack_f32 images; grad_pre = the saved c Wd + bd).
 * g_v2 == NULL: no gradient reaches v'' (the last block of an energy model, whose readout sees z only). */
int mp_painn_update_fused_bwd_f32(const float* g_z2, const float* g_v2, const float* uv, const float* prod, const float* a,
                                  const float* c, int64_t N, const float* W1T_packed, int act1, float alpha1,
                                  const float* grad_pre, const float* W2T_packed, float* g_z, float* g_uv,
                                  mpStream_t stream);
int mp_painn_update_post_f32(const float* z, const float* v, const float* uv, const float* prod, const float* a,
                             int64_t N, float* z2, float* v2, mpStream_t stream);
int mp_painn_update_post_bwd_f32(const float* g_z2, const float* g_v2, const float* uv, const float* prod,
                                 const float* a, int64_t N, float* g_a, float* g_prod, mpStream_t stream);
int mp_painn_update_pre_bwd_f32(const float* g_z2, const float* g_v2, const float* uv, const float* c, const float* a,
                                const float* g_prod, const float* g_c, int64_t N, float* g_z, float* g_uv,
                                mpStream_t stream);
/* Reverse of NodePosition -> EdgeDirectionNormalized / NodeDistanceEuclidean (PAiNN.py:116-118; Schnet.py:116-117 with
 * g_rij = 0): g_xyz[n] = scale * (sum_{recv(e)=n} t_e - sum_{send(e)=n} t_e), t_e = g_d r_ij + (g_rij - (g_rij.r_ij) r_ij)/d,
 * g_d (slices,M) and g_rij (slices,M,3) being partial sums that are added first,
 * over the receiver CSR (ptr0/perm0) and the sender CSR (ptr1/perm1); scale = -1 yields the physical force
 * (kgcnn/model/force.py:188-189). */
int mp_edge_geometry_bwd_f32(const float* g_d, const float* g_rij, int slices, const float* rij, const float* dist,
                             const int32_t* ptr0, const int32_t* perm0, const int32_t* ptr1, const int32_t* perm1,
                             int64_t N, int64_t M, float scale, float* g_xyz, mpStream_t stream);

/* Node-side chains of kgcnn/literature/Schnet.py:110-133 / schnet_conv.py:159-165 (F = 128, embedding width 64):
 * node_in:     n = Embedding(Z) W0 + b0 ; x = n Wx
 * node_update: n += ssp(agg W2 + b2) W3 + b3 ; x = n Wx_next ; agg := 0
 * node_last:   n' = n + ssp(agg W2 + b2) W3 + b3 ; h = ssp(ssp(n' Wl0 + bl0) Wl1 + bl1) (N,64) ; agg := 0
 * readout:     out[g] = ssp(sum_{nodes of g} h W_o0 + b_o0) W_o1 + b_o1   (PoolingNodes(sum) + MLP([64,1]));
 *              W_o0 == NULL: out[g] = sum_{nodes of g} (h W_o1 + b_o1)  (last_mlp ending in Dense(1, linear), then
 *              PoolingNodes(sum), use_output_mlp=False: the fork's force_schnet.py configuration)
 * The embedding width is 64 or 128; flags bit 8 (256): `numbers` holds int64 node numbers (the fork's scripts declare
 * int64 inputs) instead of float32.
 * flags bit0: fast softplus as in mp_cfconv_fused_f32; bit1: every weight-matrix pointer is an
 * mp_schnet_node_pack_f32 image of the Keras kernel instead of the kernel itself (biases stay plain): the image stores,
 * per wave and lane, the registers of four consecutive k-steps as one float4, so a workgroup loads its weight slices
 * with 16-B instead of strided 4-B reads. */
int mp_schnet_node_pack_f32(const float* W, int K, int U, float* packed /* K*U floats */, mpStream_t stream);
/* The same slice order with every element as three bf16 pieces (hi + mid + lo = the FP32 value exactly): flags bit 6
 * (value 64, together with bit 1) makes the node kernels of the FORWARD run their GEMMs on the bf16 matrix pipe as an exact
 * FP32 emulation (six products per k block, FP32 accumulate: the error of the FP32 matrix instructions at 2.67x their
 * rate).  K % 32 == 0; the image holds K*U*3/2 floats.
 * flags bit 9 (value 512, bf16-piece build): "several launch sequences share the GPU" - a launch with 256 or more 16-node
 * tiles runs on half the CUs (128 persistent workgroups), leaving the others to the kernels of the other sequences. */
int mp_schnet_node_pack_bf16_f32(const float* W, int K, int U, float* packed /* K*U*3/2 floats */, mpStream_t stream);
/* SchNetInteraction.call's node side alone (schnet_conv.py:162-164), out of place, for the layer API:
 * n_out = n_in + Dense(lin)(Dense(ssp)(agg)); agg is left untouched. */
int mp_schnet_node_residual_f32(const float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                                const float* b3, const float* n_in, float* n_out, int flags, mpStream_t stream);
int mp_schnet_node_in_f32(const float* numbers, int64_t N, const float* emb, int vocab, int emb_dim, const float* W0,
                          const float* b0, const float* Wx, float* n_out, float* x_out, int flags, mpStream_t stream);
/* mp_schnet_node_in_f32 and mp_edge_prepare_i64_f32 in ONE launch (disjoint workgroups): they are independent and at
 * QM9 batch sizes each alone occupies less than half of the chip; falls back to the two launches for large batches. */
int mp_schnet_stage0_f32(const float* numbers, int64_t N, const float* emb, int vocab, int emb_dim, const float* W0,
                         const float* b0, const float* Wx, float* n_out, float* x_out, const int64_t* idx, int64_t M,
                         const int64_t* node_splits, const int64_t* edge_splits, int64_t G, const float* xyz,
                         int32_t* recv, int32_t* send, float* dist, int32_t* flags, int flags_arg, mpStream_t stream);
int mp_schnet_node_update_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                              const float* b3, float* n_inout, const float* Wx_next, float* x_out, int flags,
                              mpStream_t stream);
int mp_schnet_node_last_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3, const float* b3,
                            const float* n_in, const float* Wl0, const float* bl0, const float* Wl1, const float* bl1,
                            float* h_out, int flags, mpStream_t stream);
int mp_schnet_readout_f32(const float* h, const int64_t* node_splits, int64_t G, const float* Wo0, const float* bo0,
                          const float* Wo1, const float* bo1, float* out, mpStream_t stream);

/* ---------------------------------------------------------------- on-GPU SetRange (next: SURVEY 8f.2) ---- */
/* SetRange.call, kgcnn/graph/preprocessor.py:288-314 via define_adjacency_from_distance (kgcnn/graph/adj.py:537-593),
 * exclusive mode, no self loops, for a whole ragged batch of coordinates xyz (N,3): pass 1 counts the edges of every
 * receiving atom and scans them (node_ptr (N+1) int32 = receiver CSR, edge_splits (G+1) int64); the caller reads
 * M = node_ptr[N] to size the outputs; pass 2 writes the (M,2) int64 sample indices in row-major (i,j) order and,
 * optionally, int32 receiver / sender ids and the edge distances (range_attributes).  max_distance < 0 or
 * max_neighbours < 0 disables that criterion. */
int mp_radius_graph_workspace_bytes(int64_t N, size_t* bytes_out_host);
int mp_radius_graph_count_f32(const float* xyz, const int64_t* node_splits, int64_t G, int64_t N, float max_distance,
                              int max_neighbours, int32_t* node_ptr, int64_t* edge_splits, void* ws, size_t ws_bytes,
                              mpStream_t stream);
int mp_radius_graph_fill_f32(const float* xyz, const int64_t* node_splits, int64_t G, int64_t N, float max_distance,
                             int max_neighbours, const int32_t* node_ptr, int64_t M, int64_t* idx_out, int32_t* recv,
                             int32_t* send, float* dist, mpStream_t stream);

/* ---------------------------------------------------------------- backward helpers (forces) -------------- */
/* EnergyForceModel, kgcnn/model/force.py:159-186: F = -dE/dx needs one reverse pass.  Gather-backward is
 * mp_segment_reduce_csr_f32 over the CSR of the gathered column, segment-sum-backward is mp_gather_rows_f32 by the
 * receiver ids, Dense-backward is mp_dense_f32 with the transposed kernel; these are the remaining derivatives. */
int mp_activation_grad_f32(int act, float act_alpha, const float* pre, const float* gy, int64_t n, float* out,
                           mpStream_t stream);                       /* out = gy * act'(pre) */
int mp_sum_axis_f32(const float* x, int64_t R, int64_t D1, int64_t D2, int axis /* 1 or 2 */, float* out,
                    mpStream_t stream);                              /* un-broadcast of mp_binary_f32 operands */
int mp_euclidean_norm_grad_f32(const float* x, const float* gy, int64_t R, int64_t D, int64_t C, int flags, float* gx,
                               mpStream_t stream);                   /* geom.py:181-193 */
int mp_bessel_basis_grad_f32(const float* d, int64_t M, const float* frequencies, int num_radial, float cutoff,
                             int envelope_exponent, const float* gy, float* gd, mpStream_t stream); /* geom.py:772-785 */
int mp_gauss_basis_grad_f32(const float* d, int64_t M, int bins, float distance, float sigma, float offset,
                            const float* gy, float* gd, mpStream_t stream);                         /* geom.py:567-571 */
int mp_cos_cutoff_grad_f32(const float* d, int64_t n, float cutoff, const float* gy, float* gd, mpStream_t stream);

/* The whole fused forward (kgcnn/literature/Schnet.py:104-148 with receiver-sorted edges) as ONE call: stage 0,
 * depth x (cfconv + node update), last node chain, readout, launched in sequence on `stream` from a descriptor of the
 * bound batch slot.  Equivalent to replaying a captured HIP graph of the same eight launches, without the capture:
 * the entry for batches whose shapes change from call to call.  flags: bit0 fast softplus, bit1 node-side
 * weight pointers (W0, Wx, W2, W3, Wl0, Wl1) are mp_schnet_node_pack_f32 images, bits 2-4 as mp_cfconv_fused_f32. */
#define MP_SCHNET_MAX_DEPTH 8
typedef struct mp_schnet_forward_desc {
  int64_t N, M, G;
  int32_t depth, vocab, flags, bins;
  float g_distance, g_sigma, g_offset;
  int32_t emb_dim;                 /* embedding width: 64 (0 = 64) or 128 */
  const float* numbers;            /* (N) node numbers: float32, or int64 with flags bit 8 */
  const float* xyz;                /* (N,3) */
  const int64_t* idx;              /* (M,2) sample indices */
  const int64_t* node_splits;      /* (G+1) */
  const int64_t* edge_splits;      /* (G+1) */
  const float* embedding;          /* (vocab,64) */
  const float* W0;                 /* (64,128) */
  const float* b0;
  const float* Wx[MP_SCHNET_MAX_DEPTH];      /* interaction i: dense1 (128,128), no bias */
  const float* packed[MP_SCHNET_MAX_DEPTH];  /* interaction i: mp_cfconv_pack_f32 image */
  const float* W2[MP_SCHNET_MAX_DEPTH];
  const float* b2[MP_SCHNET_MAX_DEPTH];
  const float* W3[MP_SCHNET_MAX_DEPTH];
  const float* b3[MP_SCHNET_MAX_DEPTH];
  const float* Wl0; const float* bl0; const float* Wl1; const float* bl1;   /* last_mlp */
  const float* Wo0; const float* bo0; const float* Wo1; const float* bo1;   /* output_mlp; Wo0 == NULL: linear head Wo1 */
  int32_t* recv; int32_t* send; float* dist; int32_t* flags_word;           /* (M) work buffers, flag word */
  float* n; float* x; float* agg; float* h; float* out;                     /* (N,128) x3 [agg zeroed], (N,64), (G,1) */
} mp_schnet_forward_desc;
int mp_schnet_forward_launch(const mp_schnet_forward_desc* desc_host, mpStream_t stream);

/* ---------------------------------------------------------------- launch groups ------------------------------ */
/* Disjoint union of k resident ragged batches in one launch (the device-side form of concatenating what two
 * MemoryGraphList.tensor() calls produce, kgcnn/data/base.py:203-239): node numbers (float32, or int64 with z_is_i64),
 * coordinates (N,3), edge sample indices (M,2) int64 - unchanged, kgcnn/layers/base.py:27 - and the two row-split arrays,
 * rebased.  Outputs sized for the sums; node_splits / edge_splits get sum(G) + 1 entries.  Lets one launch sequence of the
 * fused forward serve several independent batches (gcnn_keras_amd/fused.py::SchnetFusedRoute.call_group). */
#define MP_CONCAT_MAX 8
typedef struct mp_batch_src {
  const void* z; const float* xyz; const int64_t* idx; const int64_t* node_splits; const int64_t* edge_splits;
  int64_t N, M, G;
} mp_batch_src;
typedef struct mp_concat_desc {
  int32_t k, z_is_i64;
  mp_batch_src src[MP_CONCAT_MAX];
  void* z; float* xyz; int64_t* idx; int64_t* node_splits; int64_t* edge_splits;
} mp_concat_desc;
int mp_concat_batches(const mp_concat_desc* desc_host, mpStream_t stream);

/* ---------------------------------------------------------------- fused GCN forward -------------------------- */
/* The forward of kgcnn.literature.GCN.make_model (kgcnn/literature/GCN.py:95-109) in 1 + depth launches on 16-node
 * tiles, each launch = one producer of the tile followed by up to three Keras Dense layers on it:
 *   input mode (x != NULL):      t = x (N,K) W_in (K,units_in) + b_in           GCN.py:97  Dense(units, linear)
 *   aggregate mode (x == NULL):  t_i = agg_act( sum_{e: recv(e) = i} weight[e] * h[send[e]] )
 *                                kgcnn/layers/conv/gcn_conv.py:87-90: GatherNodesOutgoing, PoolingWeightedLocalEdges
 *                                (sum, normalize_by_weights=False), Activation; receivers through the CSR `ptr` (N+1)
 *                                over the receiver-sorted edge order, `perm` (nullable) = position -> edge for a list
 *                                that is not sorted (tf.argsort(stable=True), kgcnn/layers/pooling.py:66)
 *   then  t <- act_l(t W_l + b_l)  for l < n_layers (gcn_conv.py:86 lay_dense of the next layer, or GraphMLP,
 *   GCN.py:107), softmax_last: Keras softmax over the last layer's units.  out (N, units of the last layer; units_in
 *   if n_layers == 0).  units_in in {32, 64, 128}, layer widths 1..128.  Sums run in a fixed order: deterministic. */
typedef struct mp_gcn_layer {
  const float* W;   /* (K_l, units) Keras kernel */
  const float* b;   /* (units) or NULL */
  int32_t units, act;
  float alpha;
} mp_gcn_layer;
typedef struct mp_gcn_tile_desc {
  int64_t N;
  const float* x; int64_t K; const float* W_in; const float* b_in;                      /* input mode */
  const float* h; const int32_t* ptr; const int32_t* perm; const int32_t* send;         /* aggregate mode */
  const float* weight; int64_t M; int32_t agg_act; float agg_alpha;
  int32_t units_in, n_layers;
  mp_gcn_layer layer[3];
  int32_t softmax_last;
  float* out;
  const int32_t* tile_start;  /* aggregate mode, nullable: (n_tiles + 1) first node of every tile - strictly increasing,   */
  int64_t n_tiles;            /* tile_start[0] = 0, tile_start[n_tiles] = N, at most 16 nodes per tile (device array; lets */
                              /* the caller cut hub-heavy node ranges into several tiles); NULL: 16 consecutive nodes      */
} mp_gcn_tile_desc;
int mp_gcn_tile_f32(const mp_gcn_tile_desc* desc_host, mpStream_t stream);

/* ---------------------------------------------------------------- Dense chains on 16-row tiles ------------ */
/* One or two Keras Dense layers back to back (kgcnn/layers/modules.py:74-87; PAiNNconv / PAiNNUpdate's
 * Dense(units, act) -> Dense(3 units), kgcnn/layers/conv/painn_conv.py:60-62,187-189, and their reverse forms) in one
 * launch, weights in registers, the 128-wide intermediate in LDS - the latency-bound regime of molecular batches:
 *   m = x W1 + b1 ; [save_pre <- m] ; m = act1(m)        or, with grad_pre:  m = (x W1 + b1) * act1'(grad_pre)
 *   out = m W2 + b2 + addend                              W2_packed null: out = m + addend (U1 columns)
 * K1 in {128,256,384}; two stages: U1 = 128, U2 in {128,256,384}; one stage: U1 in {128,256,384}.  W*_packed are
 * mp_chain_pack_f32 images of the Keras kernels (K, U).  addend may alias out.  mp_chain_supported: 1 if built. */
int mp_chain_supported(int K1, int U1, int U2);
int mp_chain_pack_f32(const float* W, int K, int U, float* packed, mpStream_t stream);
int mp_dense_chain_f32(const float* x, int64_t R, int K1, const float* W1_packed, const float* b1, int U1, int act1,
                       float alpha1, float* save_pre, const float* grad_pre, const float* W2_packed, const float* b2,
                       int U2, const float* addend, float* out, mpStream_t stream);

/* Graph readout in one launch: PoolingNodes(sum) over x (N,K) (kgcnn/layers/pooling.py:215-218) followed by the
 * two-layer output MLP Dense(H, act0) -> Dense(1, linear) (kgcnn/literature/PAiNN.py:146-147): out (G,1).  g_x (nullable,
 * (N,K)): also the reverse of this readout, dE_g/dx_n = W0 (W1 * act0'(pre_g)) written to every node row of graph g
 * (what tape.gradient yields there, kgcnn/model/force.py:159-177).  K, H in {64, 128}. */
int mp_pool_mlp2_f32(const float* x, const int64_t* node_splits, int64_t G, int K, const float* W0, const float* b0, int H,
                     int act0, float alpha0, const float* W1, const float* b1, float* out, float* g_x, mpStream_t stream);

/* ---------------------------------------------------------------- SchNet energy + forces ------------------- */
/* Replaces, for a SchNet energy model, kgcnn/model/force.py:159-201 (GradientTape around the energy model, force =
 * -dE/dx) with a forward that keeps the activation derivatives and a hand-written reverse pass.
 *
 * SAVE variants of the forward node chains: as mp_schnet_node_update_f32 / mp_schnet_node_last_f32 / mp_schnet_readout_f32
 * and additionally store sigmoid(pre-activation) of every shifted softplus on the chain (d2 (N,128), dl0 (N,128),
 * dl1 (N,64)) resp. dE_g/d pooled_g (G,64) of the MLP head.  Null outputs = the plain entry.  Packed images only. */
int mp_schnet_node_update_save_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                                   const float* b3, float* n_inout, const float* Wx_next, float* x_out, float* d2_out,
                                   int flags, mpStream_t stream);
int mp_schnet_node_last_save_f32(float* agg, int64_t N, const float* W2, const float* b2, const float* W3,
                                 const float* b3, const float* n_in, const float* Wl0, const float* bl0,
                                 const float* Wl1, const float* bl1, float* h_out, float* d2_out, float* dl0_out,
                                 float* dl1_out, int flags, mpStream_t stream);
int mp_schnet_readout_grad_f32(const float* h, const int64_t* node_splits, int64_t G, const float* Wo0,
                               const float* bo0, const float* Wo1, const float* bo1, float* out, float* g_pool_out,
                               mpStream_t stream);
/* Reverse node chains on 16-node tiles; every W*T is the mp_schnet_node_pack_f32 image of the TRANSPOSED Keras kernel.
 * head : g_n = ((gh[row] * dl1) Wl1T * dl0) Wl0T ; g_agg = ((g_n W3T) * d2) W2T     gh_row null: gh is one 64-row
 * block: g_n += g_x WxT (g_x rows re-zeroed)     ; g_agg = ((g_n W3T) * d2) W2T */
int mp_schnet_bwd_head_f32(const float* gh, const int32_t* gh_row, const float* dl1, int64_t N, const float* Wl1T,
                           const float* dl0, const float* Wl0T, const float* W3T, const float* d2, const float* W2T,
                           float* g_n, float* g_agg, mpStream_t stream);
int mp_schnet_bwd_block_f32(float* g_x, int64_t N, const float* WxT, float* g_n, const float* W3T, const float* d2,
                            const float* W2T, float* g_agg, mpStream_t stream);
/* out (N,3) = scale * dE/dx from dE/dd (M): sum over the edges touching a node of g_d (x_n - x_other) / d, over the
 * receiver-side and sender-side CSR of the index plan (perm null: that column is sorted).  scale -1: physical force. */
int mp_schnet_force_from_gd_f32(const float* g_d, const float* xyz, const float* dist, const int32_t* recv,
                                const int32_t* send, const int32_t* ptr0, const int32_t* perm0, const int32_t* ptr1,
                                const int32_t* perm1, int64_t N, int64_t M, float scale, float* out, mpStream_t stream);
/* The whole energy + force pass of a bound batch slot in one call (the ~32 launches for depth 6, in sequence on
 * `stream`; capturable into a HIP graph): forward as mp_schnet_forward_launch with the SAVE chains (fwd.x unused: the
 * sender features of every block are kept in xs), then head chain, per block distance gradient + swapped-column cfconv
 * + block chain, then the force kernel.  fwd.flags bit 1 (packed node images) is required. */
typedef struct mp_schnet_force_desc {
  mp_schnet_forward_desc fwd;
  float* xs;                       /* (depth, N, 128) sender features x_i of every block */
  float* d2;                       /* (depth, N, 128) */
  float* dl0; float* dl1;          /* (N,128), (N,64) */
  float* g_pool;                   /* (G,64)  MLP head only */
  const int32_t* node_graph;       /* (N) graph of each node, MLP head only */
  const float* W3T[MP_SCHNET_MAX_DEPTH];
  const float* W2T[MP_SCHNET_MAX_DEPTH];
  const float* WxT[MP_SCHNET_MAX_DEPTH];           /* [0] unused */
  const float* packed_bwd[MP_SCHNET_MAX_DEPTH];    /* mp_cfconv_bwd_pack_f32 images */
  const float* Wl0T; const float* Wl1T;
  const int32_t* seg0; const int32_t* perm0;       /* receiver column sorted + its permutation (null: already sorted) */
  const int32_t* seg1; const int32_t* perm1;       /* sender column sorted + its permutation */
  const int32_t* ptr0; const int32_t* ptr1;        /* (N+1) CSR offsets of both columns */
  float* g_n; float* g_agg; float* g_x; float* g_d;   /* (N,128) x3 [g_x zeroed], (M) */
  float* force;                    /* (N,3) out */
  float force_scale;               /* -1: physical force, +1: dE/dx */
} mp_schnet_force_desc;
int mp_schnet_force_launch(const mp_schnet_force_desc* desc_host, mpStream_t stream);

/* ---------------------------------------------------------------- host batch packer ---------------------- */
/* The data-format side of the path.  Host pointers only; nothing here launches a kernel.
 *
 * mp_pack_rows_host = kgcnn.data.utils.ragged_tensor_from_nested_numpy (kgcnn/data/utils.py:129-157:
 * np.concatenate(list, axis=0, dtype) + row lengths), as called per property by MemoryGraphList.tensor
 * (kgcnn/data/base.py:203-239): rows_host[g] points to counts_host[g] rows of row_elems elements of src_kind; they
 * are concatenated (converted to dst_kind: f64<->f32, i32<->i64, i32/i64->f32) into dst_host, and the int64
 * row_splits (G+1) are written.  Multi-threaded over contiguous graph ranges (threads <= 1: caller's thread). */
int mp_pack_rows_host(const void* const* rows_host, const int64_t* counts_host, int64_t G, int64_t row_elems,
                      int src_kind, int dst_kind, void* dst_host, int64_t* splits_out_host, int threads);

/* Edge indices of a batch: concatenates the per-graph (m_g, K) sample index lists (idx_kind MP_DT_I64 | MP_DT_I32)
 * into the API's int64 (M,K) tensor and, in the same pass, produces what mp_index_prepare_i64 +
 * mp_csr_from_sorted_i32 would compute on the device for this batch: shifted int32 columns cols[k*M + e]
 * (kgcnn/ops/partition.py:140-155), the MP_FLAG_* word, and - if csr_ptr_out_host is given and column 0 is sorted -
 * the CSR offsets ptr[0..N] (zeros otherwise).  Also writes both row_splits (G+1 each). */
int mp_pack_edge_index_host(const void* const* idx_rows_host, int idx_kind, const int64_t* edge_counts_host,
                            const int64_t* node_counts_host, int64_t G, int K, int64_t* idx_out_host,
                            int64_t* edge_splits_out_host, int64_t* node_splits_out_host, int32_t* cols_out_host,
                            int32_t* csr_ptr_out_host, int32_t* flags_out_host, int threads);

/* Staging memory for the packer (pinned = hipHostMalloc: needs a device; 0 = aligned malloc) and the asynchronous
 * host-to-device copy that hands a packed tensor to the engine on the caller's stream. */
int mp_host_alloc(size_t bytes, int pinned, void** out_host);
int mp_host_free(void* p, int pinned);
int mp_memcpy_h2d_async(void* dst_device, const void* src_host, size_t bytes, mpStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPENGINE_H */
