"""GraphSAGE model builder (mirror of kgcnn/literature/GraphSAGE.py:20-137, ``make_model``).

Per block: neighbour messages ``MLP_e([n_j (|| e_ij)])`` are pooled at the receiver, concatenated to the node's own
features, passed through ``MLP_n`` and layer-normalised (``GraphLayerNormalization``, ``mp_layer_norm_f32``).  The LSTM
aggregator of the reference wraps a Keras LSTM and stays out of scope.
"""
from ..layers.casting import ChangeTensorType
from ..layers.gather import GatherNodesOutgoing
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import LazyConcatenate, OptionalInputEmbedding
from ..layers.norm import GraphLayerNormalization
from ..layers.pooling import PoolingLocalMessages, PoolingNodes
from ..model.utils import Model, update_model_kwargs

__model_version__ = "2022.11.25"

hyper_model_default = {
    "name": "GraphSAGE",
    "inputs": [{"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None,), "name": "edge_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64},
                        "edge": {"input_dim": 5, "output_dim": 64}},
    "node_mlp_args": {"units": [100, 50], "use_bias": True, "activation": ["relu", "linear"]},
    "edge_mlp_args": {"units": [100, 50], "use_bias": True, "activation": ["relu", "linear"]},
    "pooling_args": {"pooling_method": "segment_mean"}, "gather_args": {},
    "concat_args": {"axis": -1},
    "use_edge_features": True, "pooling_nodes_args": {"pooling_method": "mean"},
    "depth": 3, "verbose": 10,
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": [True, True, False], "units": [25, 10, 1],
                   "activation": ["relu", "relu", "sigmoid"]}
}
model_default = hyper_model_default


def _width(spec, embedding):
    return embedding["output_dim"] if len(spec["shape"]) < 2 else spec["shape"][-1]


@update_model_kwargs(hyper_model_default)
def make_model(inputs: list = None, input_embedding: dict = None, node_mlp_args: dict = None,
               edge_mlp_args: dict = None, pooling_args: dict = None, pooling_nodes_args: dict = None,
               gather_args: dict = None, concat_args: dict = None, use_edge_features: bool = None,
               depth: int = None, name: str = None, verbose: int = None, output_embedding: str = None,
               output_to_tensor: bool = None, output_mlp: dict = None):
    r"""Build GraphSAGE (kgcnn/literature/GraphSAGE.py:95-135).  Model inputs ``[node_attributes, edge_attributes,
    edge_indices]``."""
    if pooling_args["pooling_method"] in ("LSTM", "lstm"):
        raise NotImplementedError("the LSTM aggregator wraps a Keras LSTM and is out of scope on this engine")
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported output embedding for `GraphSAGE`")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    embed_e = OptionalInputEmbedding(**input_embedding["edge"], use_embedding=len(inputs[1]["shape"]) < 2)
    blocks = [{"gather": GatherNodesOutgoing(**gather_args), "cat_e": LazyConcatenate(**concat_args),
               "mlp_e": GraphMLP(**edge_mlp_args), "pool": PoolingLocalMessages(**pooling_args),
               "cat_n": LazyConcatenate(**concat_args), "mlp_n": GraphMLP(**node_mlp_args),
               "norm": GraphLayerNormalization()} for _ in range(depth)]
    pool_nodes = PoolingNodes(**pooling_nodes_args) if output_embedding == "graph" else None
    out_mlp = MLP(**output_mlp) if output_embedding == "graph" else GraphMLP(**output_mlp)
    cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") \
        if (output_embedding == "node" and output_to_tensor) else None

    def forward(model_inputs, **kwargs):
        node_input, edge_input, edi = model_inputs
        n, ed = embed_n(node_input), embed_e(edge_input)
        for blk in blocks:
            eu = blk["gather"]([n, edi])
            if use_edge_features:
                eu = blk["cat_e"]([eu, ed])
            nu = blk["pool"]([n, blk["mlp_e"](eu), edi])
            n = blk["norm"](blk["mlp_n"](blk["cat_n"]([n, nu])))
        if output_embedding == "graph":
            return out_mlp(pool_nodes(n))
        out = out_mlp(n)
        return cast(out) if cast is not None else out

    node_width, edge_width = _width(inputs[0], input_embedding["node"]), _width(inputs[1], input_embedding["edge"])
    embed_n.ensure_built((None, None))
    embed_e.ensure_built((None, None))
    units_e = edge_mlp_args["units"] if isinstance(edge_mlp_args["units"], list) else [edge_mlp_args["units"]]
    units_n = node_mlp_args["units"] if isinstance(node_mlp_args["units"], list) else [node_mlp_args["units"]]
    width = node_width
    layers = [embed_n, embed_e]
    for blk in blocks:
        blk["mlp_e"].ensure_built((None, None, width + (edge_width if use_edge_features else 0)))
        blk["mlp_n"].ensure_built((None, None, width + units_e[-1]))
        blk["norm"].ensure_built((None, None, units_n[-1]))
        width = units_n[-1]
        layers += [blk["mlp_e"], blk["mlp_n"], blk["norm"]]
    out_mlp.ensure_built((None, width) if output_embedding == "graph" else (None, None, width))
    model = Model(name, forward, layers + [out_mlp], config={"depth": depth, "node_mlp_args": node_mlp_args,
                                                             "edge_mlp_args": edge_mlp_args})
    model.__kgcnn_model_version__ = __model_version__
    model.auto_graph = True   # re-bound inputs replay the whole layer sequence from one HIP graph (model/utils.py)
    return model
