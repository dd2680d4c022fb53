"""Reverse node chains of the SchNet energy + force pass (csrc/mp_schnet_bwd.hip) and the SAVE builds of the forward
chains, through the C-ABI, against float64 torch arithmetic of the same formulas on the CPU.  f32 tolerance written per
test (k-ordered fma chains of length <= 128: 2e-5 of the output scale)."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import _ffi

pytestmark = pytest.mark.gpu


def _pack(w):
    """mp_schnet_node_pack_f32 image of a (K, U) matrix."""
    w = w.contiguous()
    out = torch.empty(w.numel(), dtype=torch.float32, device="cuda")
    _ffi.call("mp_schnet_node_pack_f32", _ffi.ptr(w), int(w.shape[0]), int(w.shape[1]), _ffi.ptr(out), _ffi.stream())
    torch.cuda.synchronize()
    return out


def _rand(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).cuda()


def _close(got, want, tol=2e-5):
    want = want.double().cpu()
    err = float((got.double().cpu() - want).abs().max())
    assert err <= tol * max(float(want.abs().max()), 1e-6), (err, float(want.abs().max()))


@pytest.mark.parametrize("n,per_graph", [(1, False), (37, True), (1344, False), (9000, True)])
def test_bwd_head_chain(n, per_graph):
    rng = np.random.default_rng(n)
    g = 7
    wl1, wl0, w3, w2 = (_rand(rng, 128, 64, scale=0.1), _rand(rng, 128, 128, scale=0.1), _rand(rng, 128, 128, scale=0.1),
                        _rand(rng, 128, 128, scale=0.1))
    dl1, dl0, d2 = (torch.sigmoid(_rand(rng, n, 64)), torch.sigmoid(_rand(rng, n, 128)), torch.sigmoid(_rand(rng, n, 128)))
    gh = _rand(rng, g if per_graph else 1, 64)
    row = torch.from_numpy(rng.integers(0, g, size=n).astype(np.int32)).cuda() if per_graph else None
    g_n, g_agg = torch.empty(n, 128, device="cuda"), torch.empty(n, 128, device="cuda")
    images = [_pack(w.t()) for w in (wl1, wl0, w3, w2)]      # kept alive: the kernel reads them after this line
    _ffi.call("mp_schnet_bwd_head_f32", _ffi.ptr(gh), _ffi.ptr(row), _ffi.ptr(dl1), n, _ffi.ptr(images[0]),
              _ffi.ptr(dl0), _ffi.ptr(images[1]), _ffi.ptr(images[2]), _ffi.ptr(d2), _ffi.ptr(images[3]),
              _ffi.ptr(g_n), _ffi.ptr(g_agg), _ffi.stream())
    torch.cuda.synchronize()
    f = lambda t: t.double().cpu()
    g_h = f(gh)[row.cpu().long()] if per_graph else f(gh).expand(n, 64)
    want_n = (((g_h * f(dl1)) @ f(wl1).T) * f(dl0)) @ f(wl0).T
    want_agg = ((want_n @ f(w3).T) * f(d2)) @ f(w2).T
    _close(g_n, want_n)
    _close(g_agg, want_agg)


@pytest.mark.parametrize("n", [1, 16, 1153, 9001])
def test_bwd_block_chain_updates_in_place_and_rezeroes_its_input(n):
    rng = np.random.default_rng(n + 1)
    wx, w3, w2 = _rand(rng, 128, 128, scale=0.1), _rand(rng, 128, 128, scale=0.1), _rand(rng, 128, 128, scale=0.1)
    d2 = torch.sigmoid(_rand(rng, n, 128))
    g_x, g_n = _rand(rng, n, 128), _rand(rng, n, 128)
    g_x0, g_n0 = g_x.clone(), g_n.clone()
    g_agg = torch.empty(n, 128, device="cuda")
    images = [_pack(w.t()) for w in (wx, w3, w2)]
    _ffi.call("mp_schnet_bwd_block_f32", _ffi.ptr(g_x), n, _ffi.ptr(images[0]), _ffi.ptr(g_n), _ffi.ptr(images[1]),
              _ffi.ptr(d2), _ffi.ptr(images[2]), _ffi.ptr(g_agg), _ffi.stream())
    torch.cuda.synchronize()
    f = lambda t: t.double().cpu()
    want_n = f(g_n0) + f(g_x0) @ f(wx).T
    _close(g_n, want_n)
    _close(g_agg, ((want_n @ f(w3).T) * f(d2)) @ f(w2).T)
    assert not bool(g_x.any())          # consumed rows are zero again for the next swapped cfconv


def test_force_from_distance_gradient_over_both_csrs():
    from gcnn_keras_amd import synth
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.qm9_like_batch(num_graphs=9, seed=3)
    rng = np.random.default_rng(0)
    idx = b["edge_indices"].copy()
    es = b["edge_splits"]
    for g in range(len(es) - 1):   # unsorted receivers: both columns need their permutation
        idx[es[g]:es[g + 1]] = idx[es[g]:es[g + 1]][rng.permutation(es[g + 1] - es[g])]
    node = RaggedTensor.from_numpy(b["node_number"], b["node_splits"])
    ridx = RaggedTensor.from_numpy(idx, es)
    plan = ridx.index_plan(node)
    ptr0, perm0, _ = plan.csr(0)
    ptr1, perm1, _ = plan.csr(1)
    assert perm0 is not None and perm1 is not None
    n, m = len(b["node_number"]), len(idx)
    shift = np.repeat(b["node_splits"][:-1], np.diff(es))
    recv, send = (idx[:, 0] + shift).astype(np.int32), (idx[:, 1] + shift).astype(np.int32)
    xyz = b["node_coordinates"].astype(np.float32)
    xyz[send[5]] = xyz[recv[5]]          # one coincident pair: divide_no_nan - no contribution
    dist = np.linalg.norm(xyz[recv] - xyz[send], axis=1).astype(np.float32)
    g_d = rng.standard_normal(m).astype(np.float32)
    out = torch.empty(n, 3, device="cuda")
    dev = lambda a: torch.from_numpy(a).cuda()
    t = [dev(g_d), dev(xyz), dev(dist), dev(recv), dev(send)]
    _ffi.call("mp_schnet_force_from_gd_f32", *[_ffi.ptr(v) for v in t], _ffi.ptr(ptr0), _ffi.ptr(perm0), _ffi.ptr(ptr1),
              _ffi.ptr(perm1), n, m, -1.0, _ffi.ptr(out), _ffi.stream())
    torch.cuda.synchronize()
    want = np.zeros((n, 3))
    for e in range(m):
        if dist[e] == 0.0:
            continue
        t_e = g_d[e] * (xyz[recv[e]].astype(np.float64) - xyz[send[e]]) / dist[e]
        want[recv[e]] += t_e
        want[send[e]] -= t_e
    _close(out, torch.from_numpy(-want), tol=1e-5)


@pytest.mark.parametrize("bins,graphs", [(20, 9), (25, 3), (7, 2)])
def test_cfconv_distance_gradient_against_float64(bins, graphs):
    """mp_cfconv_gauss_dist_grad_f32: g_d[e] = sum_f g_out[recv(e), f] x[send(e), f] dw_f/dd with
    w = ssp(gauss(d) W1 + b1) W2 + b2, against the closed form in float64 (NumPy).  The K = 128 chain of the kernel runs on
    the bf16 matrix pipe as an FP32 emulation: bar 2e-5 of the largest |g_d|, and accumulate mode adds exactly."""
    from gcnn_keras_amd import synth
    b = synth.qm9_like_batch(num_graphs=graphs, seed=11)
    rng = np.random.default_rng(bins)
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    shift = np.repeat(b["node_splits"][:-1], np.diff(b["edge_splits"]))
    recv, send = (b["edge_indices"][:, 0] + shift).astype(np.int32), (b["edge_indices"][:, 1] + shift).astype(np.int32)
    xyz = b["node_coordinates"].astype(np.float64)
    dist = np.linalg.norm(xyz[recv] - xyz[send], axis=1).astype(np.float32)
    w1 = (rng.standard_normal((bins, 128)) * 0.3).astype(np.float32)
    b1 = rng.standard_normal(128).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((128, 128)) * 0.1).astype(np.float32)
    x = rng.standard_normal((n, 128)).astype(np.float32)
    g_out = rng.standard_normal((n, 128)).astype(np.float32)
    distance, sigma, offset = 5.0, 0.4, 0.0
    dev = lambda a: torch.from_numpy(a).cuda()
    t = {k: dev(v) for k, v in dict(w1=w1, b1=b1, w2=w2, x=x, g=g_out, d=dist, recv=recv, send=send).items()}
    image = torch.empty(_ffi.lib().mp_cfconv_bwd_packed_floats(), dtype=torch.float32, device="cuda")
    _ffi.call("mp_cfconv_bwd_pack_f32", _ffi.ptr(t["w1"]), _ffi.ptr(t["b1"]), bins, _ffi.ptr(t["w2"]), _ffi.ptr(image),
              _ffi.stream())
    g_d = torch.full((m,), 3.0, device="cuda")
    for accumulate in (0, 1):
        _ffi.call("mp_cfconv_gauss_dist_grad_f32", _ffi.ptr(t["x"]), _ffi.ptr(t["g"]), n, _ffi.ptr(t["d"]), bins, distance,
                  sigma, offset, _ffi.ptr(image), _ffi.ptr(t["recv"]), _ffi.ptr(t["send"]), m, accumulate, _ffi.ptr(g_d),
                  _ffi.stream())
    torch.cuda.synchronize()
    d64 = dist.astype(np.float64)
    mu = np.arange(bins, dtype=np.float64) / bins * distance
    gamma = 1.0 / sigma / sigma / 2.0
    v = (d64[:, None] - offset) - mu[None, :]
    gauss = np.exp(-gamma * v * v)
    dgauss = gauss * (-2.0 * gamma * v)
    pre = gauss @ w1.astype(np.float64) + b1
    dw = ((dgauss @ w1.astype(np.float64)) / (1.0 + np.exp(-pre))) @ w2.astype(np.float64)       # (M, 128): dw/dd
    want = np.sum(g_out[recv].astype(np.float64) * x[send].astype(np.float64) * dw, axis=1)
    got = g_d.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - 2.0 * want)) <= 2e-5 * np.max(np.abs(2.0 * want))
