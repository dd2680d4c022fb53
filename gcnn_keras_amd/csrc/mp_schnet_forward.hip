// Whole fused SchNet forward behind ONE C-ABI call: the eight launches of csrc/mp_schnet_node.hip / mp_cfconv.hip in
// sequence on the caller's stream, from a descriptor that holds every pointer and size of a bound batch slot.
//
// Why next to the HIP-graph replay: a captured graph is bound to one batch's buffers and sizes, and capturing costs
// milliseconds - right for a resident batch that is replayed, wrong for a stream of batches of changing shape (a data
// loader feeding new molecules every step).  This entry costs ~22 us of host time per forward, needs no capture, and
// holds no lock, so several batch slots can be fed by a host thread each.  (With four slots in flight both paths run at
// the same 46-47 us per step on an MI355X: there the GPU is the limit, not the submission path.)
#include "mp_common.h"

extern "C" {

int mp_schnet_forward_launch(const mp_schnet_forward_desc* d, mpStream_t stream) {
  MP_REQUIRE(d != nullptr, "mp_schnet_forward_launch: null descriptor");
  MP_REQUIRE(d->depth >= 1 && d->depth <= MP_SCHNET_MAX_DEPTH, "mp_schnet_forward_launch: depth %d not in 1..%d",
             d->depth, MP_SCHNET_MAX_DEPTH);
  int rc = mp_schnet_stage0_f32(d->numbers, d->N, d->embedding, d->vocab, d->emb_dim == 128 ? 128 : 64, d->W0, d->b0,
                                d->Wx[0], d->n, d->x, d->idx, d->M, d->node_splits, d->edge_splits, d->G, d->xyz,
                                d->recv, d->send, d->dist, d->flags_word, d->flags & (3 | 64 | 256 | 512), stream);
  if (rc != MP_OK) return rc;
  for (int i = 0; i < d->depth; ++i) {
    rc = mp_cfconv_gauss_fused_f32(d->x, d->N, d->dist, d->bins, d->g_distance, d->g_sigma, d->g_offset, d->packed[i],
                                   d->recv, d->send, nullptr, d->M, d->flags, d->agg, stream);
    if (rc != MP_OK) return rc;
    if (i + 1 < d->depth) {
      rc = mp_schnet_node_update_f32(d->agg, d->N, d->W2[i], d->b2[i], d->W3[i], d->b3[i], d->n, d->Wx[i + 1], d->x,
                                     d->flags & (3 | 64 | 512), stream);
    } else {
      rc = mp_schnet_node_last_f32(d->agg, d->N, d->W2[i], d->b2[i], d->W3[i], d->b3[i], d->n, d->Wl0, d->bl0, d->Wl1,
                                   d->bl1, d->h, d->flags & (3 | 64 | 512), stream);
    }
    if (rc != MP_OK) return rc;
  }
  return mp_schnet_readout_f32(d->h, d->node_splits, d->G, d->Wo0, d->bo0, d->Wo1, d->bo1, d->out, stream);
}

}  // extern "C"
