"""mp_dense_f32 (LDS-tiled, csrc/mp_dense.hip) vs mp_dense_chain_f32 (16-row tiles, weights in registers, csrc/mp_chain.hip)
per shape: average launch time with HIP events on the launch stream.  Decides the row threshold of layers/modules.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcnn_keras_amd import _ffi
from gcnn_keras_amd.engine import _HipTimer

timer = _HipTimer()
print("rows K U  tiled_us chain_us  tiled_TF chain_TF")
for k, u in ((128, 128), (128, 256), (128, 384), (256, 128), (384, 128)):
    for rows in (1344, 4032, 18432, 84000, 225000, 1000000):
        x = torch.randn(rows, k, device="cuda")
        w = torch.randn(k, u, device="cuda") * 0.1
        b = torch.randn(u, device="cuda")
        out = torch.empty(rows, u, device="cuda")
        img = torch.empty(k * u, device="cuda")
        _ffi.call("mp_chain_pack_f32", _ffi.ptr(w), k, u, _ffi.ptr(img), _ffi.stream())
        torch.cuda.synchronize()
        tiled = lambda: _ffi.call("mp_dense_f32", _ffi.ptr(x), rows, k, _ffi.ptr(w), _ffi.ptr(b), u, 2, 0.0, _ffi.ptr(out),
                                  _ffi.stream())
        chain = lambda: _ffi.call("mp_dense_chain_f32", _ffi.ptr(x), rows, k, _ffi.ptr(img), _ffi.ptr(b), u, 2, 0.0, None,
                                  None, None, None, 0, None, _ffi.ptr(out), _ffi.stream())
        iters = 50 if rows < 100000 else 20
        t1, t2 = timer.time_ms(tiled, iters) * 1e3, timer.time_ms(chain, iters) * 1e3
        fl = 2.0 * rows * k * u
        print("%8d %4d %4d  %9.1f %9.1f  %7.1f %7.1f" % (rows, k, u, t1, t2, fl / t1 / 1e6, fl / t2 / 1e6), flush=True)
