import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.literature import PAiNN
from helpers import mol_inputs, painn_weight_list
from oracle import kgcnn_oracle as ko
b = synth.md17_like_batch(num_graphs=3, seed=77)
rng = np.random.default_rng(3)
idx = b["edge_indices"].copy()
for g in range(3):
    lo, hi = b["edge_splits"][g], b["edge_splits"][g + 1]
    idx[lo:hi] = idx[lo:hi][rng.permutation(hi - lo)]
srt = []
for g in range(3):
    lo, hi = b["edge_splits"][g], b["edge_splits"][g + 1]
    blk = idx[lo:hi]; srt.append(blk[np.lexsort((blk[:, 1], blk[:, 0]))])
for name, ii in (("shuffled", idx), ("sorted", np.concatenate(srt))):
    bb = dict(b, edge_indices=ii)
    for cutoff in (5.0, None):
        for seed in (8, 9):
            p = synth.painn_params(seed=seed, random_bias=True)
            model = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"}, conv_args={"units": 128, "cutoff": cutoff, "conv_pool": "sum"})
            model.set_weights(painn_weight_list(p))
            e64 = ko.painn_forward(ko.to_dtype(p, np.float64), ko.R(bb["node_number"], bb["node_splits"]), ko.R(bb["node_coordinates"].astype(np.float64), bb["node_splits"]), ko.R(bb["edge_indices"], bb["edge_splits"]), depth=3, equiv_method="eps", cutoff=cutoff).ravel()
            e32 = ko.painn_forward(p, ko.R(bb["node_number"], bb["node_splits"]), ko.R(bb["node_coordinates"], bb["node_splits"]), ko.R(bb["edge_indices"], bb["edge_splits"]), depth=3, equiv_method="eps", cutoff=cutoff).ravel()
            got = model(mol_inputs(bb)).cpu().numpy().ravel()
            lay = model(mol_inputs(bb), fused=False).cpu().numpy().ravel()
            print("%-8s cutoff %-4s seed %d: fused %s | layers %s | oracle32 %s" % (name, cutoff, seed, np.abs(got - e64) / np.abs(e64).max(), np.abs(lay - e64) / np.abs(e64).max(), np.abs(e32 - e64) / np.abs(e64).max()))
