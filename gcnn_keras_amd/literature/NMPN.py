"""NMPN model builder (mirror of kgcnn/literature/NMPN.py:24-173, ``make_model``; Gilmer et al. 2017, the message-passing
neural network with edge networks, GRU node updates and a Set2Set readout; the crystal variant is out of scope)."""
from ..layers.casting import ChangeTensorType
from ..layers.conv.mpnn_conv import GRUUpdate, MatMulMessages, TrafoEdgeNetMessages
from ..layers.gather import GatherNodesIngoing, GatherNodesOutgoing
from ..layers.geom import GaussBasisLayer, NodeDistanceEuclidean, NodePosition
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import Dense, LazyConcatenate, OptionalInputEmbedding
from ..layers.pool.set2set import PoolingSet2Set
from ..layers.pooling import PoolingLocalEdges, PoolingNodes
from ..model.utils import Model, update_model_kwargs

__model_version__ = "2022.11.25"

model_default = {
    "name": "NMPN",
    "inputs": [{"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None,), "name": "edge_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64},
                        "edge": {"input_dim": 5, "output_dim": 64}},
    "geometric_edge": False, "make_distance": False, "expand_distance": False,
    "gauss_args": {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4},
    "set2set_args": {"channels": 32, "T": 3, "pooling_method": "sum", "init_qstar": "0"},
    "pooling_args": {"pooling_method": "segment_sum"},
    "edge_mlp": {"use_bias": True, "activation": "swish", "units": [64, 64, 64]},
    "use_set2set": True, "depth": 3, "node_dim": 64,
    "verbose": 10,
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": [True, True, False], "units": [25, 10, 1],
                   "activation": ["selu", "selu", "sigmoid"]},
}


@update_model_kwargs(model_default)
def make_model(inputs: list = None, input_embedding: dict = None, geometric_edge: bool = None,
               make_distance: bool = None, expand_distance: bool = None, gauss_args: dict = None,
               set2set_args: dict = None, pooling_args: dict = None, edge_mlp: dict = None, use_set2set: bool = None,
               node_dim: int = None, depth: int = None, verbose: int = None, name: str = None,
               output_embedding: str = None, output_to_tensor: bool = None, output_mlp: dict = None):
    r"""Build NMPN (kgcnn/literature/NMPN.py:45-173).  Inputs ``[node_attributes, edge_attributes | edge_distance |
    node_coordinates, edge_indices]``."""
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported output embedding for mode `NMPN`")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    embed_e = OptionalInputEmbedding(**input_embedding["edge"], use_embedding=len(inputs[1]["shape"]) < 2) \
        if not geometric_edge else None
    lay_pos, lay_dist = (NodePosition(), NodeDistanceEuclidean()) if make_distance else (None, None)
    lay_gauss = GaussBasisLayer(**gauss_args) if expand_distance else None
    dense_n = Dense(node_dim, activation="linear")
    mlp_in, trafo_in = GraphMLP(**edge_mlp), TrafoEdgeNetMessages(target_shape=(node_dim, node_dim))
    mlp_out, trafo_out = GraphMLP(**edge_mlp), TrafoEdgeNetMessages(target_shape=(node_dim, node_dim))
    gru = GRUUpdate(node_dim)
    gather_out, gather_in = GatherNodesOutgoing(), GatherNodesIngoing()
    matmul_in, matmul_out = MatMulMessages(), MatMulMessages()
    cat_eu, cat_n = LazyConcatenate(axis=-1), LazyConcatenate(axis=-1)
    pool_e = PoolingLocalEdges(**pooling_args)
    if output_embedding == "graph":
        dense_set = Dense(set2set_args["channels"], activation="linear") if use_set2set else None
        readout = PoolingSet2Set(**set2set_args) if use_set2set else PoolingNodes(**pooling_args)
        out_mlp = MLP(**output_mlp)
        cast = None
    else:
        dense_set = readout = None
        out_mlp = GraphMLP(**output_mlp)
        cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") if output_to_tensor else None

    def forward(model_inputs, **kwargs):
        node_input, edge_input, edi = model_inputs
        n0 = embed_n(node_input)
        ed = embed_e(edge_input) if embed_e is not None else edge_input
        if make_distance:
            pos1, pos2 = lay_pos([ed, edi])
            ed = lay_dist([pos1, pos2])
        if expand_distance:
            ed = lay_gauss(ed)
        n = dense_n(n0)
        # the edge networks do not depend on the node state: their (M, F, F) matrices are made once, used `depth` times
        edge_net_in = trafo_in(mlp_in(ed))
        edge_net_out = trafo_out(mlp_out(ed))
        for _ in range(depth):
            n_in = gather_out([n, edi])
            n_out = gather_in([n, edi])
            m_in = matmul_in([edge_net_in, n_in])
            m_out = matmul_out([edge_net_out, n_out])
            eu = pool_e([n, cat_eu([m_in, m_out]), edi])
            n = gru([n, eu])
        n = cat_n([n0, n])
        if output_embedding == "graph":
            if use_set2set:
                out = readout(dense_set(n))
                out = out.reshape(int(out.shape[0]), -1).contiguous()   # ks.layers.Flatten
            else:
                out = readout(n)
            return out_mlp(out)
        out = out_mlp(n)
        return cast(out) if cast is not None else out

    n0_dim = input_embedding["node"]["output_dim"] if len(inputs[0]["shape"]) < 2 else inputs[0]["shape"][-1]
    if expand_distance:
        e_dim = gauss_args["bins"]
    elif make_distance:
        e_dim = 1
    elif not geometric_edge and len(inputs[1]["shape"]) < 2:
        e_dim = input_embedding["edge"]["output_dim"]
    else:
        e_dim = inputs[1]["shape"][-1]
    embed_n.ensure_built((None, None))
    if embed_e is not None:
        embed_e.ensure_built((None, None))
    dense_n.ensure_built((None, None, n0_dim))
    units = edge_mlp["units"]
    last_units = units[-1] if isinstance(units, (list, tuple)) else units
    for mlp_, trafo in ((mlp_in, trafo_in), (mlp_out, trafo_out)):
        mlp_.ensure_built((None, None, e_dim))
        trafo.ensure_built((None, None, last_units))
    gru.ensure_built([(None, None, node_dim), (None, None, 2 * node_dim)])
    layers = [embed_n] + ([embed_e] if embed_e is not None else []) + [dense_n, mlp_in, trafo_in, mlp_out, trafo_out, gru]
    cat_dim = n0_dim + node_dim
    if output_embedding == "graph":
        if use_set2set:
            dense_set.ensure_built((None, None, cat_dim))
            readout.ensure_built((None, None, set2set_args["channels"]))
            layers += [dense_set, readout]
            out_mlp.ensure_built((None, 2 * set2set_args["channels"]))
        else:
            out_mlp.ensure_built((None, cat_dim))
    else:
        out_mlp.ensure_built((None, None, cat_dim))
    layers.append(out_mlp)
    model = Model(name, forward, layers, config={"depth": depth, "node_dim": node_dim, "edge_mlp": edge_mlp,
                                                  "set2set_args": set2set_args, "use_set2set": use_set2set})
    model.__kgcnn_model_version__ = __model_version__
    model.auto_graph = True   # re-bound inputs replay the whole layer sequence from one HIP graph (model/utils.py)
    return model
