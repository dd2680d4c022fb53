"""Times the PaiNN message kernels (forward / reverse) of config 3 alone, with the real basis size and - as a floor of the
kernel's structure - with a 2-function basis read from the same buffers (the filter work almost gone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import _HipTimer
from gcnn_keras_amd.literature import PAiNN
from gcnn_keras_amd.model.force import EnergyForceModel
from gcnn_keras_amd.ragged import RaggedTensor

b = synth.md17_like_batch(num_graphs=int(sys.argv[1]) if len(sys.argv) > 1 else 64, seed=2345)
ins = [RaggedTensor.from_numpy(b["node_number"], b["node_splits"]), RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
       RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]
n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
force = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True, output_to_tensor=False,
                         output_squeeze_states=True)
force(ins), force(ins)
torch.cuda.synchronize()
slot = energy.fused.slot_of(ins, grad=True)
p, blk = slot.p, slot.blk[1]
timer = _HipTimer()
for B in (slot.B, 2):
    def msg():
        _ffi.call("mp_painn_message_f32", _ffi.ptr(blk["s"]), _ffi.ptr(slot.vs[0]), n, _ffi.ptr(slot.rbf), B, None,
                  _ffi.ptr(slot.rij), _ffi.ptr(p["conv1/w/kernel"]), _ffi.ptr(p["conv1/w/bias"]), _ffi.ptr(slot.ptr0),
                  _ffi.ptr(slot.perm0), _ffi.ptr(slot.send), m, _ffi.ptr(slot.zs[0]), _ffi.ptr(blk["zp"]),
                  _ffi.ptr(blk["vp"]), _ffi.stream())
    def msg_bwd():
        _ffi.call("mp_painn_message_bwd_f32", _ffi.ptr(blk["s"]), _ffi.ptr(slot.vs[0]), n, _ffi.ptr(slot.rbf),
                  _ffi.ptr(slot.rbfd), B, None, None, _ffi.ptr(slot.rij), _ffi.ptr(p["conv1/w/kernel"]),
                  _ffi.ptr(p["conv1/w/bias"]), _ffi.ptr(slot.ptr1), _ffi.ptr(slot.perm1), _ffi.ptr(slot.recv), m,
                  _ffi.ptr(slot.g_zp), _ffi.ptr(slot.g_vp), _ffi.ptr(slot.g_s), _ffi.ptr(slot.gv), _ffi.ptr(slot.g_d),
                  _ffi.ptr(slot.g_rij), 0, _ffi.stream())
    print("B=%d: message %.2f us, message reverse %.2f us (N=%d, M=%d)" % (B, timer.time_ms(msg, 100) * 1e3,
                                                                           timer.time_ms(msg_bwd, 100) * 1e3, n, m))
