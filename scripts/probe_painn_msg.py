"""Times the PaiNN message kernels (forward / reverse) of config 3 alone through the C ABI, HIP events on the launch
stream; MPENGINE_PAINN_VALU=1 selects the VALU builds (A/B)."""
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
B = slot.B
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
def msg_tiles():
    tl = slot.tiles0
    _ffi.call("mp_painn_message_tiles_f32", _ffi.ptr(blk["s"]), _ffi.ptr(slot.vs[0]), n, _ffi.ptr(slot.rbf), B, None,
              _ffi.ptr(slot.rij), _ffi.ptr(slot.w["conv1/w/F"]), _ffi.ptr(slot.ptr0), _ffi.ptr(slot.send), m,
              _ffi.ptr(tl["table"]), tl["count"], tl["max_rows"], tl["max_edges"], _ffi.ptr(slot.zs[0]), _ffi.ptr(blk["zp"]),
              _ffi.ptr(blk["vp"]), _ffi.stream())
if slot.tiles0 is not None:
    msg()
    torch.cuda.synchronize()
    ref_z, ref_v = blk["zp"].clone(), blk["vp"].clone()
    for trial in range(6):
        blk["zp"].fill_(7.0); blk["vp"].fill_(7.0)
        for _ in range(1 if trial < 3 else 50):
            msg_tiles()
        torch.cuda.synchronize()
        bad_v = ((blk["vp"] - ref_v).abs().amax(dim=(1, 2)) > 1e-5).nonzero().flatten().tolist()
        bad_z = ((blk["zp"] - ref_z).abs().amax(dim=1) > 1e-5).nonzero().flatten().tolist()
        if bad_v or bad_z:
            print("trial %d: bad v rows %s bad z rows %s" % (trial, bad_v[:20], bad_z[:20]))
            r = bad_v[0] if bad_v else bad_z[0]
            dd = (blk["vp"][r] - ref_v[r]).abs()
            print("   row %d: v diff per (k): %s, nonzero features %d" % (r, dd.amax(dim=1).tolist(), int((dd > 1e-5).sum())))
            tab = slot.tiles0["table"].cpu().numpy()
            for rr_ in bad_v[:6]:
                trow = [t for t in tab if t[0] <= rr_ < t[1]][0]
                feats = (blk["vp"][rr_] - ref_v[rr_]).abs().amax(dim=0).gt(1e-6).nonzero().flatten().tolist()
                print("   row %d: tile %s, position in tile %d, features %s" % (rr_, trow[:6].tolist(), rr_ - trow[0], feats))
    msg_tiles()
    torch.cuda.synchronize()
    print("tiles: %d tiles, max rows %d, max edges %d; message %.2f us; max |diff| vs gather kernel: z %.2e (scale %.2e) v %.2e (scale %.2e)" % (
        slot.tiles0["count"], slot.tiles0["max_rows"], slot.tiles0["max_edges"], timer.time_ms(msg_tiles, 200) * 1e3,
        float((blk["zp"] - ref_z).abs().max()), float(ref_z.abs().max()), float((blk["vp"] - ref_v).abs().max()),
        float(ref_v.abs().max())))
print("%s: message %.2f us, message reverse %.2f us (N=%d, M=%d, B=%d)" % (
    "MFMA gather" if os.environ.get("MPENGINE_PAINN_MFMA_GATHER") == "1" else "VALU", timer.time_ms(msg, 200) * 1e3,
    timer.time_ms(msg_bwd, 200) * 1e3, n, m, B))

# accuracy of the message step against a float64 evaluation of painn_conv.py:99-113 (torch-CPU), per output row
def f64_reference():
    d = lambda t: t.detach().cpu().double()
    s_, v_, rbf, rij = d(blk["s"]), d(slot.vs[0]), d(slot.rbf), d(slot.rij)
    W, bw = d(p["conv1/w/kernel"]), d(p["conv1/w/bias"])
    recv, send = slot.recv.cpu().long(), slot.send.cpu().long()
    w = rbf @ W + bw
    sw = s_[send] * w
    sw1, sw2, sw3 = sw[:, :128], sw[:, 128:256], sw[:, 256:]
    dz = torch.zeros(n, 128, dtype=torch.float64).index_add_(0, recv, sw1)
    dv = torch.zeros(n, 3, 128, dtype=torch.float64).index_add_(0, recv, sw2[:, None, :] * v_[send] + sw3[:, None, :] * rij[:, :, None])
    return d(slot.zs[0]) + dz, v_ + dv
rz, rv = f64_reference()
def row_err(got, ref):
    g, r = got.detach().cpu().double().reshape(n, -1), ref.reshape(n, -1)
    den = torch.maximum(r.abs().amax(1), 1e-3 * r.abs().max())
    return float(((g - r).abs().amax(1) / den).max()), float(((g - r).abs().amax(1) / den).median())
msg(); torch.cuda.synchronize()
print("accuracy vs float64 (worst row, median row): z' %.2e %.2e  v' %.2e %.2e" % (row_err(blk["zp"], rz) + row_err(blk["vp"], rv)))
