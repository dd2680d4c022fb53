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


def test_work_arena_size_classes():
    """Capacity classes of the route's work-set arena: an eighth of the size's power of two apart, never finer than the
    unit, never below the request - batches of a dataset (a few per cent apart in N and M) share classes at every scale."""
    from gcnn_keras_amd.fused import WorkArena
    b = WorkArena._bucket
    for x in (1, 17, 511, 512, 513, 2301, 11573, 26190, 130950, 2_000_000):
        for unit in (64, 512, 8192):
            step = max(unit, (1 << (x.bit_length() - 1)) >> 3)
            c = b(x, unit)
            assert c >= x and c % step == 0 and c - x < step
    assert b(11400, 512) == b(11573, 512) == b(11700, 512) == 12288      # launch groups of five config-2 batches
    assert b(128000, 8192) == b(131000, 8192) == 131072
    assert b(0, 512) == 512 and b(2301, 512) == 2560 and b(26190, 8192) == 32768
