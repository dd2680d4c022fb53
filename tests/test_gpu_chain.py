"""mp_dense_chain_f32 (csrc/mp_chain.hip): one or two Dense layers per launch on 16-row tiles, through the C-ABI, against
float64 torch arithmetic of the same formula (tolerance 2e-5 of the output scale: k-ordered f32 fma chains, K <= 384)."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import _ffi

pytestmark = pytest.mark.gpu


def _pack(w):
    w = w.contiguous()
    out = torch.empty(w.numel(), dtype=torch.float32, device="cuda")
    _ffi.call("mp_chain_pack_f32", _ffi.ptr(w), int(w.shape[0]), int(w.shape[1]), _ffi.ptr(out), _ffi.stream())
    torch.cuda.synchronize()
    return out


def _rand(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).cuda()


def _close(got, want, tol=2e-5):
    want = want.double().cpu()
    err = float((got.double().cpu() - want).abs().max())
    assert err <= tol * max(float(want.abs().max()), 1e-6), (err, float(want.abs().max()))


_ACT64 = {"linear": lambda x: x, "swish": lambda x: x * torch.sigmoid(x), "relu": torch.relu,
          "shifted_softplus": lambda x: torch.nn.functional.softplus(x) - np.log(2.0)}
_GRAD64 = {"linear": lambda x: torch.ones_like(x),
           "swish": lambda x: torch.sigmoid(x) + x * torch.sigmoid(x) * (1 - torch.sigmoid(x)),
           "shifted_softplus": torch.sigmoid}


@pytest.mark.parametrize("rows,k1,u2,act", [(1, 128, 384, "swish"), (1344, 128, 384, "swish"), (1344, 256, 384, "swish"),
                                            (37, 128, 128, "shifted_softplus"), (9001, 256, 128, "relu"),
                                            (4099, 384, 256, "linear")])
def test_two_stage_forward_keeps_the_pre_activation(rows, k1, u2, act):
    rng = np.random.default_rng(rows + k1)
    x, w1, b1, w2, b2 = _rand(rng, rows, k1), _rand(rng, k1, 128, scale=0.1), _rand(rng, 128), _rand(rng, 128, u2, scale=0.1), _rand(rng, u2)
    assert _ffi.lib().mp_chain_supported(k1, 128, u2) == 1
    images = [_pack(w1), _pack(w2)]
    pre, out = torch.empty(rows, 128, device="cuda"), torch.empty(rows, u2, device="cuda")
    addend = _rand(rng, rows, u2)
    _ffi.call("mp_dense_chain_f32", _ffi.ptr(x), rows, k1, _ffi.ptr(images[0]), _ffi.ptr(b1), 128, _ffi.activation_code(act),
              0.0, _ffi.ptr(pre), None, _ffi.ptr(images[1]), _ffi.ptr(b2), u2, _ffi.ptr(addend), _ffi.ptr(out), _ffi.stream())
    torch.cuda.synchronize()
    f = lambda t: t.double().cpu()
    want_pre = f(x) @ f(w1) + f(b1)
    _close(pre, want_pre)
    _close(out, _ACT64[act](want_pre) @ f(w2) + f(b2) + f(addend))


@pytest.mark.parametrize("rows,k1,u2,act", [(1344, 384, 256, "swish"), (1344, 384, 128, "swish"), (18, 384, 128, "shifted_softplus")])
def test_two_stage_reverse_multiplies_by_the_activation_derivative(rows, k1, u2, act):
    rng = np.random.default_rng(rows + u2)
    g, w1, w2, saved = _rand(rng, rows, k1), _rand(rng, k1, 128, scale=0.1), _rand(rng, 128, u2, scale=0.1), _rand(rng, rows, 128)
    images = [_pack(w1), _pack(w2)]
    out = _rand(rng, rows, u2)          # in-place accumulation: addend aliases out
    out0 = out.clone()
    _ffi.call("mp_dense_chain_f32", _ffi.ptr(g), rows, k1, _ffi.ptr(images[0]), None, 128, _ffi.activation_code(act), 0.0,
              None, _ffi.ptr(saved), _ffi.ptr(images[1]), None, u2, _ffi.ptr(out), _ffi.ptr(out), _ffi.stream())
    torch.cuda.synchronize()
    f = lambda t: t.double().cpu()
    _close(out, ((f(g) @ f(w1)) * _GRAD64[act](f(saved))) @ f(w2) + f(out0))


@pytest.mark.parametrize("rows,k,u", [(4032, 128, 256), (4032, 256, 128), (5, 128, 384), (1000, 384, 128)])
def test_single_stage(rows, k, u):
    rng = np.random.default_rng(rows + k + u)
    x, w, add = _rand(rng, rows, k), _rand(rng, k, u, scale=0.1), _rand(rng, rows, u)
    image = _pack(w)
    out = torch.empty(rows, u, device="cuda")
    _ffi.call("mp_dense_chain_f32", _ffi.ptr(x), rows, k, _ffi.ptr(image), None, u, 0, 0.0, None, None, None, None, 0,
              _ffi.ptr(add), _ffi.ptr(out), _ffi.stream())
    torch.cuda.synchronize()
    _close(out, x.double().cpu() @ w.double().cpu() + add.double().cpu())


def test_unbuilt_shapes_are_refused():
    lib = _ffi.lib()
    assert lib.mp_chain_supported(384, 128, 384) == 0 and lib.mp_chain_supported(256, 384, 0) == 0
    assert lib.mp_chain_supported(100, 128, 0) == 0 and lib.mp_chain_supported(128, 256, 128) == 0
    x, out = torch.zeros(16, 384, device="cuda"), torch.zeros(16, 384, device="cuda")
    w = torch.zeros(384 * 384, device="cuda")
    with pytest.raises(ValueError):
        _ffi.call("mp_dense_chain_f32", _ffi.ptr(x), 16, 384, _ffi.ptr(w), None, 384, 0, 0.0, None, None, None, None, 0,
                  None, _ffi.ptr(out), _ffi.stream())


@pytest.mark.parametrize("k,h,act", [(128, 128, "swish"), (128, 64, "shifted_softplus"), (64, 64, "relu"), (64, 128, "linear")])
def test_pool_mlp2_readout_and_its_reverse(k, h, act):
    """mp_pool_mlp2_f32: PoolingNodes(sum) + Dense(h, act) + Dense(1) per graph, and dE_g/dx rows, vs float64 torch
    (autograd for the reverse); ragged graphs incl. an empty one in the middle and at the end (written as bias-only rows)."""
    rng = np.random.default_rng(k + h)
    sizes = [5, 0, 21, 1, 9, 33, 0]
    splits = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)).cuda()
    n, g = int(sum(sizes)), len(sizes)
    x, w0, b0, w1, b1 = _rand(rng, n, k), _rand(rng, k, h, scale=0.2), _rand(rng, h), _rand(rng, h, 1), _rand(rng, 1)
    out, gx = torch.empty(g, 1, device="cuda"), torch.full((n, k), 7.0, device="cuda")
    _ffi.call("mp_pool_mlp2_f32", _ffi.ptr(x), _ffi.ptr(splits), g, k, _ffi.ptr(w0), _ffi.ptr(b0), h,
              _ffi.activation_code(act), 0.0, _ffi.ptr(w1), _ffi.ptr(b1), _ffi.ptr(out), _ffi.ptr(gx), _ffi.stream())
    torch.cuda.synchronize()
    f = lambda t: t.double().cpu()
    x64 = f(x).requires_grad_(True)
    seg = torch.repeat_interleave(torch.arange(g), torch.tensor(sizes))
    pooled = torch.zeros(g, k, dtype=torch.float64).index_add(0, seg, x64)
    want = _ACT64[act](pooled @ f(w0) + f(b0)) @ f(w1) + f(b1)
    want.sum().backward()
    _close(out, want.detach())
    _close(gx, x64.grad)
