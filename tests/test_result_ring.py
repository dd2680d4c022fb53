"""Result ring of the fused routes (gcnn_keras_amd/result_ring.py): a result set is handed out only while nobody else
holds one of its tensors or a view of them.  Host logic only - CPU tensors, no graph capture."""
import torch

from gcnn_keras_amd.result_ring import ResultRing


def _ring(size=2):
    made, captured = [], []

    def make():
        made.append(1)
        return torch.zeros(4, 1), torch.zeros(5, 3)

    def capture(bufs):
        captured.append(bufs)
        return object()

    return ResultRing(size=size), make, capture, made, captured


def test_held_results_are_never_handed_out_again():
    ring, make, capture, made, captured = _ring(size=2)
    (e1, f1), g1 = ring.acquire(make, capture)
    (e2, f2), g2 = ring.acquire(make, capture)
    assert e1.data_ptr() != e2.data_ptr() and g1 is not g2 and len(made) == 2 and len(captured) == 2
    assert ring.acquire(make, capture) is None          # both sets held: the caller falls back to a copy
    del e1, f1
    (e3, f3), g3 = ring.acquire(make, capture)          # dropped set comes round again, same graph, nothing new made
    assert g3 is g1 and len(made) == 2 and len(captured) == 2
    del e3
    assert ring.acquire(make, capture) is None          # f3 (the second tensor of the set) is still held
    view = f3[:2]
    del f3
    assert ring.acquire(make, capture) is None          # ... and so is a view of it
    del view
    got = ring.acquire(make, capture)
    assert got is not None and got[1] is g1


def test_loop_that_drops_its_results_cycles_through_the_ring():
    ring, make, capture, made, captured = _ring(size=3)
    seen = set()
    out = None
    for _ in range(10):
        out = ring.acquire(make, capture)[0][0]         # `out` keeps the previous result alive during the call
        seen.add(out.data_ptr())
    assert len(made) == 3 and len(seen) == 3            # the ring fills up first, then its sets are taken round robin


def test_a_caller_holding_only_the_storage_blocks_the_set():
    """``out.untyped_storage()`` kept by the caller (no tensor, no view): torch re-uses the storage's Python wrapper, so use
    count and tensor reference count look idle - the wrapper's own reference count gives the holder away."""
    ring, make, capture, made, captured = _ring(size=1)
    (e1, f1), g1 = ring.acquire(make, capture)
    storage = e1.untyped_storage()
    del e1, f1
    assert ring.acquire(make, capture) is None
    del storage
    got = ring.acquire(make, capture)
    assert got is not None and got[1] is g1
