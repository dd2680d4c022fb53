"""MEGNet model builder (mirror of kgcnn/literature/Megnet.py:24-193, ``make_model``; the crystal variant is out of scope).

Inputs ``[node_attributes, node_coordinates | edge_distance, edge_indices, graph_attributes]``; three MEGNet blocks with
feed-forward MLPs and skip connections, Set2Set readouts of nodes and edges, output MLP on ``[nodes | edges | state]``.
Weights are created in construction order (embeddings, first feed-forward triple, then per block: [feed-forward triple,]
block; the two Dense + Set2Set readouts; output MLP)."""
import torch

from ..layers.conv.megnet_conv import MEGnetBlock
from ..layers.geom import GaussBasisLayer, NodeDistanceEuclidean, NodePosition
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import Dense, LazyAdd, OptionalInputEmbedding, _binary_raw, _concat_last_raw
from ..layers.pool.set2set import PoolingSet2Set
from ..layers.pooling import PoolingGlobalEdges, PoolingNodes
from ..model.utils import Model, update_model_kwargs
from .. import _ffi

__model_version__ = "2022.11.25"

model_default = {
    "name": "Megnet",
    "inputs": [{"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None, 3), "name": "node_coordinates", "dtype": "float32", "ragged": True},
               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True},
               {"shape": [], "name": "graph_attributes", "dtype": "float32", "ragged": False}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64},
                        "graph": {"input_dim": 100, "output_dim": 64}},
    "make_distance": True, "expand_distance": True,
    "gauss_args": {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4},
    "meg_block_args": {"node_embed": [64, 32, 32], "edge_embed": [64, 32, 32],
                       "env_embed": [64, 32, 32], "activation": "kgcnn>softplus2"},
    "set2set_args": {"channels": 16, "T": 3, "pooling_method": "sum", "init_qstar": "0"},
    "node_ff_args": {"units": [64, 32], "activation": "kgcnn>softplus2"},
    "edge_ff_args": {"units": [64, 32], "activation": "kgcnn>softplus2"},
    "state_ff_args": {"units": [64, 32], "activation": "kgcnn>softplus2"},
    "nblocks": 3, "has_ff": True, "dropout": None, "use_set2set": True,
    "verbose": 10,
    "output_embedding": "graph",
    "output_mlp": {"use_bias": [True, True, True], "units": [32, 16, 1],
                   "activation": ["kgcnn>softplus2", "kgcnn>softplus2", "linear"]}
}


@update_model_kwargs(model_default)
def make_model(inputs: list = None, input_embedding: dict = None, expand_distance: bool = None,
               make_distance: bool = None, gauss_args: dict = None, meg_block_args: dict = None,
               set2set_args: dict = None, node_ff_args: dict = None, edge_ff_args: dict = None,
               state_ff_args: dict = None, use_set2set: bool = None, nblocks: int = None, has_ff: bool = None,
               dropout: float = None, name: str = None, verbose: int = None, output_embedding: str = None,
               output_mlp: dict = None):
    r"""Build MEGNet (kgcnn/literature/Megnet.py:50-193).  Output ``(batch, L)`` graph embedding."""
    if output_embedding != "graph":
        raise ValueError("Unsupported output embedding for mode `Megnet`.")
    if dropout is not None:
        raise NotImplementedError("dropout is a training-time layer; the engine is forward / inference only")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    embed_u = OptionalInputEmbedding(**input_embedding["graph"], use_embedding=len(inputs[3]["shape"]) < 1)
    lay_pos, lay_dist = (NodePosition(), NodeDistanceEuclidean()) if make_distance else (None, None)
    lay_gauss = GaussBasisLayer(**gauss_args) if expand_distance else None
    ff = [(GraphMLP(**node_ff_args), GraphMLP(**edge_ff_args), MLP(**state_ff_args))]
    blocks = []
    for i in range(nblocks):
        if has_ff and i > 0:
            ff.append((GraphMLP(**node_ff_args), GraphMLP(**edge_ff_args), MLP(**state_ff_args)))
        blocks.append(MEGnetBlock(**meg_block_args))
    adds = [(LazyAdd(), LazyAdd()) for _ in range(nblocks)]
    if use_set2set:
        dense_v, dense_e = Dense(set2set_args["channels"], activation="linear"), Dense(set2set_args["channels"],
                                                                                        activation="linear")
        read_v, read_e = PoolingSet2Set(**set2set_args), PoolingSet2Set(**set2set_args)
    else:
        dense_v = dense_e = None
        read_v, read_e = PoolingNodes(), PoolingGlobalEdges()
    out_mlp = MLP(**output_mlp)

    def forward(model_inputs, **kwargs):
        node_input, xyz_input, edi, env_input = model_inputs
        n = embed_n(node_input)
        uenv = embed_u(env_input)
        if make_distance:
            pos1, pos2 = lay_pos([xyz_input, edi])
            ed = lay_dist([pos1, pos2])
        else:
            ed = xyz_input
        if expand_distance:
            ed = lay_gauss(ed)
        vp, ep, up = ff[0][0](n), ff[0][1](ed), ff[0][2](uenv)
        vp2, ep2, up2 = vp, ep, up
        for i in range(nblocks):
            if has_ff and i > 0:
                vp2, ep2, up2 = ff[i][0](vp), ff[i][1](ep), ff[i][2](up)
            vp2, ep2, up2 = blocks[i]([vp2, ep2, edi, up2])
            vp = adds[i][0]([vp2, vp])          # skip connections (Megnet.py:163-165)
            ep = adds[i][1]([ep2, ep])
            up = _binary_raw(_ffi.MP_ADD, up2.contiguous(), up.contiguous())
        if use_set2set:
            vp, ep = read_v(dense_v(vp)), read_e(dense_e(ep))
        else:
            vp, ep = read_v(vp), read_e(ep)
        flat = lambda t: t.reshape(int(t.shape[0]), -1).contiguous()    # ks.layers.Flatten
        final_vec = _concat_last_raw([flat(vp), flat(ep), up.contiguous()])
        return out_mlp(final_vec)

    # weights in construction order, so set_weights() works before a first call
    node_dim = input_embedding["node"]["output_dim"] if len(inputs[0]["shape"]) < 2 else inputs[0]["shape"][-1]
    env_dim = input_embedding["graph"]["output_dim"] if len(inputs[3]["shape"]) < 1 else inputs[3]["shape"][-1]
    edge_dim = gauss_args["bins"] if expand_distance else inputs[1]["shape"][-1]
    embed_n.ensure_built((None, None))
    embed_u.ensure_built((None,))
    last = lambda args: args["units"][-1] if isinstance(args["units"], (list, tuple)) else args["units"]
    dims = (node_dim, edge_dim, env_dim)
    block_out = (meg_block_args["node_embed"][-1], meg_block_args["edge_embed"][-1], meg_block_args["env_embed"][-1])
    layers = [embed_n, embed_u]
    k = 0
    for i in range(nblocks):
        if i == 0 or has_ff:
            trio = ff[k]
            k += 1
            trio[0].ensure_built((None, None, dims[0]))
            trio[1].ensure_built((None, None, dims[1]))
            trio[2].ensure_built((None, dims[2]))
            layers += list(trio)
            cur = (last(node_ff_args), last(edge_ff_args), last(state_ff_args))
        else:
            cur = dims
        blocks[i].ensure_built([(None, None, cur[0]), (None, None, cur[1]), (None, None, 2), (None, cur[2])])
        layers.append(blocks[i])
        dims = block_out
    if use_set2set:
        dense_v.ensure_built((None, None, dims[0]))
        dense_e.ensure_built((None, None, dims[1]))
        read_v.ensure_built((None, None, set2set_args["channels"]))
        read_e.ensure_built((None, None, set2set_args["channels"]))
        layers += [dense_v, dense_e, read_v, read_e]
        final_dim = 4 * set2set_args["channels"] + dims[2]
    else:
        final_dim = dims[0] + dims[1] + dims[2]
    out_mlp.ensure_built((None, final_dim))
    layers.append(out_mlp)
    model = Model(name, forward, layers, config={"nblocks": nblocks, "meg_block_args": meg_block_args,
                                                  "set2set_args": set2set_args, "use_set2set": use_set2set})
    model.__kgcnn_model_version__ = __model_version__
    model.auto_graph = True   # re-bound inputs replay the whole layer sequence from one HIP graph (model/utils.py)
    return model
