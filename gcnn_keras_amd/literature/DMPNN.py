"""DMPNN model builder (mirror of kgcnn/literature/DMPNN.py:23-174, ``make_model``): directed message passing on edges.

Edge states start as ``Dense([n_j || e_ij])``; each of ``depth`` rounds replaces an edge's state by
``act(Dense(sum of the states arriving at its sender, minus its reverse edge) + h0)``
(``DMPNNPPoolingEdgesDirected``); nodes then read ``Dense([sum of incoming states || n])``.
The optional graph-state input of the reference is not wired (no BASELINE config uses it).
"""
from ..layers.casting import ChangeTensorType
from ..layers.conv.dmpnn_conv import DMPNNPPoolingEdgesDirected
from ..layers.gather import GatherNodesOutgoing
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import Activation, Dense, Dropout, LazyAdd, LazyConcatenate, OptionalInputEmbedding
from ..layers.pooling import PoolingLocalEdges, PoolingNodes
from ..model.utils import Model, update_model_kwargs

__model_version__ = "2022.11.25"

model_default = {
    "name": "DMPNN",
    "inputs": [
        {"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
        {"shape": (None,), "name": "edge_attributes", "dtype": "float32", "ragged": True},
        {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True},
        {"shape": (None, 1), "name": "edge_indices_reverse", "dtype": "int64", "ragged": True}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64},
                        "edge": {"input_dim": 5, "output_dim": 64},
                        "graph": {"input_dim": 100, "output_dim": 64}},
    "pooling_args": {"pooling_method": "sum"},
    "use_graph_state": False,
    "edge_initialize": {"units": 128, "use_bias": True, "activation": "relu"},
    "edge_dense": {"units": 128, "use_bias": True, "activation": "linear"},
    "edge_activation": {"activation": "relu"},
    "node_dense": {"units": 128, "use_bias": True, "activation": "relu"},
    "verbose": 10, "depth": 5, "dropout": {"rate": 0.1},
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": [True, True, False], "units": [64, 32, 1],
                   "activation": ["relu", "relu", "linear"]}
}


def _width(spec, embedding):
    return embedding["output_dim"] if len(spec["shape"]) < 2 else spec["shape"][-1]


@update_model_kwargs(model_default)
def make_model(name: str = None, inputs: list = None, input_embedding: dict = None, pooling_args: dict = None,
               edge_initialize: dict = None, edge_dense: dict = None, edge_activation: dict = None,
               node_dense: dict = None, dropout: dict = None, depth: int = None, verbose: int = None,
               use_graph_state: bool = False, output_embedding: str = None, output_to_tensor: bool = None,
               output_mlp: dict = None):
    r"""Build DMPNN (kgcnn/literature/DMPNN.py:119-171).  Model inputs ``[node_attributes, edge_attributes,
    edge_indices, edge_indices_reverse]``; the last holds, per edge, the position of its reverse edge in the graph's
    edge list (``-1`` if there is none)."""
    if use_graph_state:
        raise NotImplementedError("the graph-state input of DMPNN is not wired on this engine")
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported graph embedding for mode `DMPNN`.")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    embed_e = OptionalInputEmbedding(**input_embedding["edge"], use_embedding=len(inputs[1]["shape"]) < 2)
    gather_out = GatherNodesOutgoing()
    concat = LazyConcatenate(axis=-1)
    dense_h0 = Dense(**edge_initialize)
    dense_edge = Dense(**edge_dense)                      # one layer shared by all rounds, like the reference
    pool_directed = [DMPNNPPoolingEdgesDirected() for _ in range(depth)]
    add, act = LazyAdd(), Activation(**edge_activation)
    drop = Dropout(**dropout) if dropout is not None else None
    pool_edges = PoolingLocalEdges(**pooling_args)
    dense_node = Dense(**node_dense)
    pool_nodes = PoolingNodes(**pooling_args) if output_embedding == "graph" else None
    out_mlp = MLP(**output_mlp) if output_embedding == "graph" else GraphMLP(**output_mlp)
    cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") \
        if (output_embedding == "node" and output_to_tensor) else None

    def forward(model_inputs, **kwargs):
        node_input, edge_input, edi, ed_pairs = model_inputs
        n, ed = embed_n(node_input), embed_e(edge_input)
        h0 = dense_h0(concat([gather_out([n, edi]), ed]))
        h = h0
        for lay in pool_directed:
            h = act(add([dense_edge(lay([n, h, edi, ed_pairs])), h0]))
            if drop is not None:
                h = drop(h)
        hv = dense_node(concat([pool_edges([n, h, edi]), n]))
        if output_embedding == "graph":
            return out_mlp(pool_nodes(hv))
        out = out_mlp(hv)
        return cast(out) if cast is not None else out

    node_width, edge_width = _width(inputs[0], input_embedding["node"]), _width(inputs[1], input_embedding["edge"])
    embed_n.ensure_built((None, None))
    embed_e.ensure_built((None, None))
    dense_h0.ensure_built((None, None, node_width + edge_width))
    dense_edge.ensure_built((None, None, edge_initialize["units"]))
    dense_node.ensure_built((None, None, edge_dense["units"] + node_width))
    out_mlp.ensure_built((None, node_dense["units"]) if output_embedding == "graph" else (None, None, node_dense["units"]))
    model = Model(name, forward, [embed_n, embed_e, dense_h0, dense_edge, dense_node, out_mlp],
                  config={"depth": depth, "edge_initialize": edge_initialize, "edge_dense": edge_dense,
                          "node_dense": node_dense})
    model.__kgcnn_model_version__ = __model_version__
    model.auto_graph = True   # re-bound inputs replay the whole layer sequence from one HIP graph (model/utils.py)
    return model
