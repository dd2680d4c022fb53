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


BAR_CAP = 5e-5   # the comparison's noise floor (2x the float32 oracle's own distance from float64) may raise a bar to here, no further


def assert_rows_close(got, ref32, ref64=None, rtol=RTOL, floor=FLOOR, what="", cap=BAR_CAP):
    """Per-row bound against the float32 oracle; with the float64 twin given, additionally no row of the engine may be
    further from float64 truth than 4x the float32 oracle's own worst row (and never needs to beat 2e-6).

    Two float32 pipelines cannot agree better than the float32 oracle agrees with float64 truth: where the oracle itself
    is further than ``rtol / 2`` from its twin (deep recurrent models with O(1) random weights) twice that distance is the
    noise floor of the comparison - but never more than ``cap``: a test that needs a looser bar has to say so in its own
    call (``cap=...``), and the effective bar is printed whenever it is not ``rtol``."""
    err = rowwise_rel(got, ref32, floor)
    bar = rtol
    if ref64 is not None:
        e_engine, e_oracle = rowwise_rel(got, ref64, floor), rowwise_rel(ref32, ref64, floor)
        assert e_engine <= max(min(4 * e_oracle, max(cap, rtol)), 2e-6), \
            "%s: engine %.3g vs oracle %.3g from float64" % (what, e_engine, e_oracle)
        bar = max(rtol, min(2 * e_oracle, cap))
        if bar > rtol:
            print("[parity] %s: effective bar %.2e (float32 oracle is %.2e from float64; engine %.2e from float64, %.2e "
                  "from the float32 oracle)" % (what, bar, e_oracle, e_engine, err))
    assert err <= bar, "%s: worst row off by %.3g relative (bar %.1g)" % (what, err, bar)
    return err


FORCE_RTOL = 2e-5


def assert_forces_close(force, ref32, ref64, node_splits, what="", rtol=FORCE_RTOL, floor=FLOOR, cap=BAR_CAP):
    """Forces ``(N, 3)`` of EVERY molecule against the analytic float64 reference (oracle/torch_force_oracle.py:
    ``ref64``; ``ref32`` = the same autograd restatement run in float32, i.e. what a float32 tape delivers).

    Molecules of a batch do not interact and their force scales differ by orders of magnitude (random coordinates put
    some atom pairs very close), so every molecule is measured against ITS OWN largest force component:

    * max-norm, every molecule: ``max |F - F64| <= max(rtol, 2 x e32) x max |F64|``, ``e32`` = the float32 reference's
      own max-norm error on that molecule, and never more than ``cap``;
    * per atom row, every molecule: worst ``|dF_row|_inf / max(|F64_row|_inf, floor x molecule scale)``.  A force row is a
      sum of cancelling pair terms, so in float32 the small rows of a molecule carry errors of ~1e-7 of the molecule's
      scale whoever computes them (at ``floor`` = 1e-3 the float32 reference itself has rows 1e-4 off on config 3), and
      WHICH molecule draws the bad row is rounding luck.  The statement that holds for a correct float32 pipeline is
      rank-wise: sorted over molecules, the engine's k-th largest row error is within ``max(rtol, 4 x`` the float32
      reference's k-th largest``)`` for every k (``assert_rows_close`` allows the engine 4x the oracle's distance from
      float64 for the worst row only; here the same factor binds at every rank), and the engine's median is within
      ``max(rtol, 2 x`` the reference's median``)``.
    Prints both distributions; returns (worst max-norm error, worst row error)."""
    f, r32, r64 = (np.asarray(a, np.float64).reshape(-1, 3) for a in (force, ref32, ref64))
    assert f.shape == r64.shape == r32.shape, (f.shape, r32.shape, r64.shape)
    ns = np.asarray(node_splits)
    norm_e, row_e, row_32 = [], [], []
    for g in range(len(ns) - 1):
        lo, hi = int(ns[g]), int(ns[g + 1])
        if hi == lo:
            continue
        a, b32, b64 = f[lo:hi], r32[lo:hi], r64[lo:hi]
        scale = float(np.max(np.abs(b64)))
        if scale == 0.0:
            assert float(np.max(np.abs(a))) <= 1e-30, "%s: molecule %d has zero reference forces" % (what, g)
            continue
        e_norm, e32_norm = float(np.max(np.abs(a - b64))) / scale, float(np.max(np.abs(b32 - b64))) / scale
        bar = min(max(rtol, 2 * e32_norm), max(cap, rtol))
        assert e_norm <= bar, "%s: molecule %d forces off by %.3g of its scale %.3g (float32 reference %.3g, bar %.3g)" % (
            what, g, e_norm, scale, e32_norm, bar)
        norm_e.append(e_norm)
        row_e.append(rowwise_rel(a, b64, floor))
        row_32.append(rowwise_rel(b32, b64, floor))
    if not norm_e:
        return 0.0, 0.0
    eng, ref = np.sort(np.array(row_e))[::-1], np.sort(np.array(row_32))[::-1]
    print("[parity] %s: %d molecules; max-norm worst %.2e median %.2e; atom rows (floor %.0e of the molecule): engine "
          "worst %.2e median %.2e, %d above %.0e | float32 reference worst %.2e median %.2e, %d above %.0e" % (
              what, len(norm_e), max(norm_e), float(np.median(norm_e)), floor, eng[0], float(np.median(eng)),
              int(np.sum(eng > rtol)), rtol, ref[0], float(np.median(ref)), int(np.sum(ref > rtol)), rtol))
    bad = np.nonzero(eng > np.maximum(rtol, 4 * ref))[0]
    assert bad.size == 0, "%s: rank %d atom-row error %.3g vs float32 reference %.3g at the same rank" % (
        what, int(bad[0]), eng[bad[0]], ref[bad[0]])
    assert np.median(eng) <= max(rtol, 2 * float(np.median(ref))), "%s: median atom-row error %.3g vs reference %.3g" % (
        what, float(np.median(eng)), float(np.median(ref)))
    return max(norm_e), float(eng[0])
