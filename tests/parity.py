"""Parity bars shared by the GPU tests.

``rowwise_rel`` is the per-row statement of north_star's "within 1e-5 rel for the float segment-sum": every row (node,
edge or graph) is held to 1e-5 of ITS OWN magnitude, not of the tensor's maximum, so a small row cannot hide behind a
large one.  Rows whose magnitude is below ``floor`` x the tensor scale are measured against that floor (a row that is
zero up to rounding has no meaningful relative error)."""
import numpy as np

RTOL = 1e-5      # BASELINE.json north_star: float segment-sum within 1e-5 relative
FLOOR = 1e-3     # rows smaller than this fraction of the tensor scale are measured against the floor


def _rows(a):
    a = np.asarray(a, dtype=np.float64)
    return a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)


def rowwise_rel(got, ref, floor=FLOOR):
    """max over rows of  |got_row - ref_row|_inf / max(|ref_row|_inf, floor * |ref|_inf)."""
    g, r = _rows(got), _rows(ref)
    assert g.shape == r.shape, (g.shape, r.shape)
    if r.size == 0:
        return 0.0
    scale = float(np.max(np.abs(r)))
    den = np.maximum(np.max(np.abs(r), axis=1), floor * max(scale, 1e-30))
    return float(np.max(np.max(np.abs(g - r), axis=1) / den))


def max_rel(got, ref):
    """Round-1 bar, kept for reference: max |diff| over the tensor / max |ref|."""
    return float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)))) / \
        max(float(np.max(np.abs(ref))), 1e-30)


def assert_rows_close(got, ref32, ref64=None, rtol=RTOL, floor=FLOOR, what=""):
    """Per-row bound against the float32 oracle; with the float64 twin given, additionally no row of the engine may be
    further from float64 truth than 4x the float32 oracle's own worst row (and never needs to beat 2e-6)."""
    err = rowwise_rel(got, ref32, floor)
    bar = rtol
    if ref64 is not None:
        e_engine, e_oracle = rowwise_rel(got, ref64, floor), rowwise_rel(ref32, ref64, floor)
        assert e_engine <= max(4 * e_oracle, 2e-6), "%s: engine %.3g vs oracle %.3g from float64" % (
            what, e_engine, e_oracle)
        # two float32 pipelines cannot agree better than the float32 oracle agrees with float64 truth: where the oracle
        # itself is further than rtol / 2 from its twin (deep recurrent models with O(1) random weights), that distance is
        # the noise floor of the comparison
        bar = max(rtol, 2 * e_oracle)
    assert err <= bar, "%s: worst row off by %.3g relative (bar %.1g)" % (what, err, bar)
    return err
