"""GCN model builder (mirror of kgcnn/literature/GCN.py:21-112, ``make_model``)."""
from ..layers.casting import ChangeTensorType
from ..layers.conv.gcn_conv import GCN
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import Dense, OptionalInputEmbedding
from ..layers.pooling import PoolingNodes
from ..model.utils import Model, update_model_kwargs
from .. import fused_gcn as _fused

__model_version__ = "2022.11.25"

model_default = {
    "name": "GCN",
    "inputs": [{"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64},
                        "edge": {"input_dim": 10, "output_dim": 64}},
    "gcn_args": {"units": 100, "use_bias": True, "activation": "relu", "pooling_method": "sum",
                 "is_sorted": False, "has_unconnected": True},
    "depth": 3, "verbose": 10,
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": [True, True, False], "units": [25, 10, 1],
                   "activation": ["relu", "relu", "sigmoid"]}
}


@update_model_kwargs(model_default)
def make_model(inputs: list = None, input_embedding: dict = None, depth: int = None, gcn_args: dict = None,
               name: str = None, verbose: int = None, output_embedding: str = None, output_to_tensor: bool = None,
               output_mlp: dict = None):
    r"""Build GCN (kgcnn/literature/GCN.py:37-112).  Model inputs ``[node_attributes, edge_weights (.., 1),
    edge_indices]``."""
    if inputs[1]["shape"][-1] != 1:
        raise ValueError("No edge features available for GCN, only edge weights of pre-scaled adjacency matrix, \
                         must be shape (batch, None, 1), but got (without batch-dimension): ", inputs[1]["shape"])
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported output embedding for `GCN`")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    embed_e = OptionalInputEmbedding(**input_embedding["edge"], use_embedding=len(inputs[1]["shape"]) < 2)
    dense0 = Dense(gcn_args["units"], use_bias=True, activation="linear")
    gcns = [GCN(**gcn_args) for _ in range(depth)]
    pool = PoolingNodes() if output_embedding == "graph" else None
    out_mlp = MLP(**output_mlp) if output_embedding == "graph" else GraphMLP(**output_mlp)
    cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") \
        if (output_embedding == "node" and output_to_tensor) else None

    def forward(model_inputs, fused=None, **kwargs):
        # Fused route (csrc/mp_gcn.hip: input GEMM + depth x (aggregate + next Dense / output MLP) = 1 + depth launches,
        # graph-replayed for a re-bound input set) whenever configuration and inputs fit it and no gradient is requested;
        # ``fused=False`` forces the layer sequence below, ``fused=True`` insists on the kernels.
        if route is not None and fused is not False and route.accepts(model_inputs):
            return route(model_inputs)
        if fused is True:
            raise ValueError("this GCN configuration / these inputs do not fit the fused kernels")
        node_input, edge_input, edi = model_inputs
        n = embed_n(node_input)
        ed = embed_e(edge_input)
        n = dense0(n)
        for lay in gcns:
            n = lay([n, ed, edi])
        if output_embedding == "graph":
            return out_mlp(pool(n))
        out = out_mlp(n)
        return cast(out) if cast is not None else out

    units = gcn_args["units"]
    in_dim = input_embedding["node"]["output_dim"] if len(inputs[0]["shape"]) < 2 else inputs[0]["shape"][-1]
    embed_n.ensure_built((None, None))
    embed_e.ensure_built((None, None))
    dense0.ensure_built((None, None, in_dim))
    for lay in gcns:
        lay.ensure_built([(None, None, units), (None, None, 1), (None, None, 2)])
    out_mlp.ensure_built((None, units) if output_embedding == "graph" else (None, None, units))
    route = None
    if _fused.supports({"inputs": inputs, "gcn_args": gcn_args, "depth": depth, "output_embedding": output_embedding,
                        "output_mlp": output_mlp}):
        route = _fused.GcnFusedRoute(dense0, gcns, out_mlp, cast)
    model = Model(name, forward, [embed_n, embed_e, dense0] + gcns + [out_mlp], config={"depth": depth,
                                                                                         "gcn_args": gcn_args})
    model.__kgcnn_model_version__ = __model_version__
    model.fused = route       # None: this configuration always runs the layer sequence
    # the layer sequence (graph output, other widths, gradients) is replayed from one HIP graph for re-bound inputs
    # (model/utils.py); a call the fused route takes manages its own graphs
    model.auto_graph = route is None
    return model
