"""Reverse-mode rules for the engine ops, used only by ``EnergyForceModel`` (kgcnn/model/force.py:159-186).

``torch.autograd`` supplies the tape (plumbing); every forward AND backward computation is an engine kernel:
gather-backward = segment-sum over the CSR of the gathered index column, segment-sum-backward = gather by the receiver
ids, Dense-backward = Dense with the transposed kernel, plus the elementwise derivative kernels of csrc/mp_backward.hip.
Only input gradients are produced (forces need dE/dx, not dE/dW).
"""
import torch

from . import _ffi


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def _rows_elems(t):
    rows = int(t.shape[0])
    elems = 1
    for d in t.shape[1:]:
        elems *= int(d)
    return rows, max(elems, 1)


class GatherRows(torch.autograd.Function):
    """rows of ``values`` at plan columns ``colsel`` -> (M, len(colsel), ...)."""

    @staticmethod
    def forward(ctx, values, plan, colsel):
        from .layers.gather import _gather_rows_raw
        ctx.plan, ctx.colsel, ctx.shape = plan, tuple(colsel), tuple(values.shape)
        return _gather_rows_raw(values, plan, colsel)

    @staticmethod
    def backward(ctx, g):
        from .ops.segment import _segment_reduce_raw
        plan = ctx.plan
        total = None
        for j, col in enumerate(ctx.colsel):
            ptr, perm, _ = plan.csr(col)
            gj = g[:, j].contiguous()
            part = _segment_reduce_raw(_ffi.MP_SUM, gj, ptr, perm, plan.N, None, False)
            if total is None:
                total = part
            else:
                from .layers.modules import _binary_raw
                total = _binary_raw(_ffi.MP_ADD, total, part)
        return total.view(ctx.shape), None, None


class SegmentSum(torch.autograd.Function):
    """CSR segment sum / mean (weights, if any, are constants)."""

    @staticmethod
    def forward(ctx, data, op, ptr, perm, n_out, weight, seg_ids):
        from .ops.segment import _segment_reduce_raw
        if op not in (_ffi.MP_SUM, _ffi.MP_MEAN):
            raise NotImplementedError("gradients are implemented for sum / mean pooling")
        ctx.op, ctx.ptr, ctx.n_out, ctx.weight, ctx.seg_ids = op, ptr, n_out, weight, seg_ids
        ctx.shape = tuple(data.shape)
        return _segment_reduce_raw(op, data, ptr, perm, n_out, weight, False)

    @staticmethod
    def backward(ctx, g):
        gc = g.contiguous()
        rows, elems = _rows_elems(gc)
        m = int(ctx.seg_ids.numel())
        if ctx.op == _ffi.MP_MEAN:
            cnt = (ctx.ptr[1:ctx.n_out + 1] - ctx.ptr[:ctx.n_out]).to(torch.float32).clamp(min=1.0)
            from .layers.modules import _binary_raw
            gc = _binary_raw(_ffi.MP_MUL, gc.view(rows, elems), (1.0 / cnt).view(rows, 1)).view(gc.shape)
        out = torch.empty((m,) + tuple(gc.shape[1:]), dtype=torch.float32, device=gc.device)
        _ffi.call("mp_gather_rows_f32", _ffi.ptr(gc), rows, elems, _ffi.ptr(ctx.seg_ids.contiguous()), m, 1,
                  _ffi.int32_array([0]), _ffi.ptr(out), _ffi.stream())
        if ctx.weight is not None:
            from .layers.modules import _binary_raw
            out = _binary_raw(_ffi.MP_MUL, out.view(m, elems), ctx.weight.contiguous().view(m, 1)).view(out.shape)
        return out, None, None, None, None, None, None


class PoolGraph(torch.autograd.Function):
    """Per-graph sum / mean; backward repeats the graph row over its nodes (GatherState kernel)."""

    @staticmethod
    def forward(ctx, values, op, row_splits, g_rows):
        if op not in (_ffi.MP_SUM, _ffi.MP_MEAN):
            raise NotImplementedError("gradients are implemented for sum / mean pooling")
        ctx.op, ctx.splits, ctx.g, ctx.n = op, row_splits, g_rows, int(values.shape[0])
        rows, elems = _rows_elems(values)
        out = torch.empty((g_rows,) + tuple(values.shape[1:]), dtype=torch.float32, device=values.device)
        _ffi.call("mp_pool_graph_f32", op, _ffi.ptr(values.contiguous()), _ffi.ptr(row_splits), g_rows, elems, None,
                  _ffi.ptr(out), _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        gc = g.contiguous()
        _, elems = _rows_elems(gc)
        if ctx.op == _ffi.MP_MEAN:
            cnt = (ctx.splits[1:] - ctx.splits[:-1]).to(torch.float32).clamp(min=1.0)
            from .layers.modules import _binary_raw
            gc = _binary_raw(_ffi.MP_MUL, gc.view(ctx.g, elems), (1.0 / cnt).view(ctx.g, 1)).view(gc.shape)
        out = torch.empty((ctx.n,) + tuple(gc.shape[1:]), dtype=torch.float32, device=gc.device)
        _ffi.call("mp_repeat_rows_f32", _ffi.ptr(gc), _ffi.ptr(ctx.splits), ctx.g, elems, ctx.n, _ffi.ptr(out),
                  _ffi.stream())
        return out, None, None, None


class Dense(torch.autograd.Function):
    """y = act(x W + b); backward dx = (dy * act'(pre)) W^T."""

    @staticmethod
    def forward(ctx, x, kernel, bias, act_code, alpha):
        from .layers.modules import _dense_raw
        pre = _dense_raw(x, kernel, bias, 0, 0.0)
        ctx.kernel, ctx.act, ctx.alpha = kernel, act_code, alpha
        if act_code == 0:
            ctx.pre = None
            return pre
        ctx.pre = pre
        out = torch.empty_like(pre)
        _ffi.call("mp_activation_f32", act_code, float(alpha), _ffi.ptr(pre), pre.numel(), _ffi.ptr(out), _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        from .layers.modules import _dense_raw
        gc = g.contiguous()
        if ctx.pre is not None:
            gp = torch.empty_like(gc)
            _ffi.call("mp_activation_grad_f32", ctx.act, float(ctx.alpha), _ffi.ptr(ctx.pre), _ffi.ptr(gc), gc.numel(),
                      _ffi.ptr(gp), _ffi.stream())
            gc = gp
        wt = ctx.kernel.t().contiguous()  # (units, in): layout change only
        return _dense_raw(gc, wt, None, 0, 0.0), None, None, None, None


class Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act_code, alpha):
        xc = x.contiguous()
        ctx.x, ctx.act, ctx.alpha = xc, act_code, alpha
        out = torch.empty_like(xc)
        _ffi.call("mp_activation_f32", act_code, float(alpha), _ffi.ptr(xc), xc.numel(), _ffi.ptr(out), _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        gc = g.contiguous()
        out = torch.empty_like(gc)
        _ffi.call("mp_activation_grad_f32", ctx.act, float(ctx.alpha), _ffi.ptr(ctx.x), _ffi.ptr(gc), gc.numel(),
                  _ffi.ptr(out), _ffi.stream())
        return out, None, None


def _unbroadcast(g, shape):
    """Sum ``g`` over the axes that were broadcast from ``shape`` (middle and / or last axis of the 3-D view)."""
    if tuple(g.shape) == tuple(shape):
        return g
    if g.dim() != len(shape) or int(g.shape[0]) != int(shape[0]):
        raise NotImplementedError("gradient of a row-broadcast operand is outside the force path")
    r = int(g.shape[0])
    cur = g.contiguous()
    last_b = int(shape[-1]) == 1 and int(g.shape[-1]) != 1
    mid_g = 1
    for d in g.shape[1:-1]:
        mid_g *= int(d)
    mid_s = 1
    for d in shape[1:-1]:
        mid_s *= int(d)
    mid_b = g.dim() > 2 and mid_s == 1 and mid_g != 1
    d2 = int(cur.shape[-1])
    if last_b:
        out = torch.empty((r, mid_g), dtype=torch.float32, device=g.device)
        _ffi.call("mp_sum_axis_f32", _ffi.ptr(cur), r, mid_g, d2, 2, _ffi.ptr(out), _ffi.stream())
        cur, d2 = out, 1
    if mid_b:
        src = cur.view(r, mid_g, d2)
        out = torch.empty((r, d2), dtype=torch.float32, device=g.device)
        _ffi.call("mp_sum_axis_f32", _ffi.ptr(src), r, mid_g, d2, 1, _ffi.ptr(out), _ffi.stream())
        cur = out
    return cur.reshape(shape)


class Binary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, op):
        from .layers.modules import _binary_raw
        ctx.op, ctx.sa, ctx.sb = op, tuple(a.shape), tuple(b.shape)
        ctx.a = a if (op == _ffi.MP_MUL and b.requires_grad) else None
        ctx.b = b if (op == _ffi.MP_MUL and a.requires_grad) else None
        ctx.ga, ctx.gb = a.requires_grad, b.requires_grad
        return _binary_raw(op, a, b)

    @staticmethod
    def backward(ctx, g):
        from .layers.modules import _binary_raw
        gc = g.contiguous()
        ga = gb = None
        if ctx.ga:
            ga = gc if ctx.op != _ffi.MP_MUL else _binary_raw(_ffi.MP_MUL, gc, ctx.b)
            ga = _unbroadcast(ga, ctx.sa)
        if ctx.gb:
            if ctx.op == _ffi.MP_MUL:
                gb = _binary_raw(_ffi.MP_MUL, gc, ctx.a)
            elif ctx.op == _ffi.MP_SUB:
                gb = torch.zeros_like(gc)
                gb = _binary_raw(_ffi.MP_SUB, gb, gc)
            else:
                gb = gc
            gb = _unbroadcast(gb, ctx.sb)
        return ga, gb, None


class ConcatLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *values):
        from .layers.modules import _concat_last_raw
        ctx.widths = [int(v.shape[-1]) for v in values]
        return _concat_last_raw(list(values))

    @staticmethod
    def backward(ctx, g):
        gc = g.contiguous()
        total = int(gc.shape[-1])
        rows = gc.numel() // max(total, 1)
        outs, off = [], 0
        for w in ctx.widths:
            o = torch.empty(tuple(gc.shape[:-1]) + (w,), dtype=gc.dtype, device=gc.device)
            _ffi.call("mp_copy_cols_f32", _ffi.ptr(gc), total, off, _ffi.ptr(o), w, 0, rows, w, _ffi.stream())
            outs.append(o)
            off += w
        return tuple(outs)


class SplitLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, num):
        from .layers.modules import _split_last_raw
        return tuple(_split_last_raw(value, num))

    @staticmethod
    def backward(ctx, *gs):
        from .layers.modules import _concat_last_raw
        return _concat_last_raw([g.contiguous() for g in gs]), None


class EuclideanNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, d, c, flags, out_shape):
        xc = x.contiguous()
        ctx.x, ctx.rdc, ctx.flags = xc, (r, d, c), flags
        out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        _ffi.call("mp_euclidean_norm_f32", _ffi.ptr(xc), r, d, c, flags, _ffi.ptr(out), _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        r, d, c = ctx.rdc
        gc = g.contiguous()
        gx = torch.empty_like(ctx.x)
        _ffi.call("mp_euclidean_norm_grad_f32", _ffi.ptr(ctx.x), _ffi.ptr(gc), r, d, c, ctx.flags, _ffi.ptr(gx),
                  _ffi.stream())
        return gx, None, None, None, None, None


class ScalarProduct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, axis):
        ac, bc = a.contiguous(), b.contiguous()
        ctx.a, ctx.b, ctx.axis = ac, bc, axis
        shape = list(ac.shape)
        r = 1
        for s in shape[:axis]:
            r *= int(s)
        c = 1
        for s in shape[axis + 1:]:
            c *= int(s)
        out = torch.empty(shape[:axis] + shape[axis + 1:], dtype=torch.float32, device=a.device)
        _ffi.call("mp_scalar_product_f32", _ffi.ptr(ac), _ffi.ptr(bc), r, int(shape[axis]), c, _ffi.ptr(out),
                  _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        from .layers.modules import _binary_raw
        ge = g.contiguous().unsqueeze(ctx.axis)
        return _binary_raw(_ffi.MP_MUL, ge, ctx.b), _binary_raw(_ffi.MP_MUL, ge, ctx.a), None


class BesselBasis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, freq, num_radial, cutoff, exponent):
        dc = d.contiguous()
        ctx.d, ctx.freq, ctx.args = dc, freq, (num_radial, cutoff, exponent)
        out = torch.empty(tuple(dc.shape[:-1]) + (num_radial,), dtype=torch.float32, device=d.device)
        _ffi.call("mp_bessel_basis_f32", _ffi.ptr(dc), dc.numel(), _ffi.ptr(freq), num_radial, float(cutoff),
                  int(exponent), _ffi.ptr(out), _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        num_radial, cutoff, exponent = ctx.args
        gd = torch.empty_like(ctx.d)
        _ffi.call("mp_bessel_basis_grad_f32", _ffi.ptr(ctx.d), ctx.d.numel(), _ffi.ptr(ctx.freq), num_radial,
                  float(cutoff), int(exponent), _ffi.ptr(g.contiguous()), _ffi.ptr(gd), _ffi.stream())
        return gd, None, None, None, None


class GaussBasis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, bins, distance, sigma, offset):
        dc = d.contiguous()
        ctx.d, ctx.args = dc, (bins, distance, sigma, offset)
        out = torch.empty(tuple(dc.shape[:-1]) + (bins,), dtype=torch.float32, device=d.device)
        _ffi.call("mp_gauss_basis_f32", _ffi.ptr(dc), dc.numel(), bins, distance, sigma, offset, _ffi.ptr(out),
                  _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        bins, distance, sigma, offset = ctx.args
        gd = torch.empty_like(ctx.d)
        _ffi.call("mp_gauss_basis_grad_f32", _ffi.ptr(ctx.d), ctx.d.numel(), bins, distance, sigma, offset,
                  _ffi.ptr(g.contiguous()), _ffi.ptr(gd), _ffi.stream())
        return gd, None, None, None, None


class CosCutoff(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, cutoff):
        dc = d.contiguous()
        ctx.d, ctx.cutoff = dc, cutoff
        out = torch.empty_like(dc)
        _ffi.call("mp_cos_cutoff_f32", _ffi.ptr(dc), dc.numel(), float(cutoff), _ffi.ptr(out), _ffi.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        gd = torch.empty_like(ctx.d)
        _ffi.call("mp_cos_cutoff_grad_f32", _ffi.ptr(ctx.d), ctx.d.numel(), float(ctx.cutoff),
                  _ffi.ptr(g.contiguous()), _ffi.ptr(gd), _ffi.stream())
        return gd, None
