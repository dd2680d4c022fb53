"""GAT model builder (mirror of kgcnn/literature/GAT.py:20-123, ``make_model``) on the engine's attention heads."""
from ..layers.casting import ChangeTensorType
from ..layers.conv.gat_conv import AttentionHeadGAT
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import Activation, Dense, LazyAverage, LazyConcatenate, OptionalInputEmbedding
from ..layers.pooling import PoolingNodes
from ..model.utils import Model, update_model_kwargs

__model_version__ = "2022.11.25"

model_default = {
    "name": "GAT",
    "inputs": [{"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None,), "name": "edge_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64},
                        "edge": {"input_dim": 5, "output_dim": 64}},
    "attention_args": {"units": 32, "use_final_activation": False, "use_edge_features": True,
                       "has_self_loops": True, "activation": "kgcnn>leaky_relu", "use_bias": True},
    "pooling_nodes_args": {"pooling_method": "mean"},
    "depth": 3, "attention_heads_num": 5,
    "attention_heads_concat": False, "verbose": 10,
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": [True, True, False], "units": [25, 10, 1],
                   "activation": ["relu", "relu", "sigmoid"]}
}


def _feature_width(spec, embedding):
    return embedding["output_dim"] if len(spec["shape"]) < 2 else spec["shape"][-1]


@update_model_kwargs(model_default)
def make_model(inputs: list = None, input_embedding: dict = None, attention_args: dict = None,
               pooling_nodes_args: dict = None, depth: int = None, attention_heads_num: int = None,
               attention_heads_concat: bool = None, name: str = None, verbose: int = None,
               output_embedding: str = None, output_to_tensor: bool = None, output_mlp: dict = None,
               head_class=AttentionHeadGAT):
    r"""Build GAT (kgcnn/literature/GAT.py:89-121).  Model inputs ``[node_attributes, edge_attributes, edge_indices]``;
    per block ``attention_heads_num`` heads are concatenated or averaged (then activated).  ``head_class`` lets the
    GATv2 builder reuse this wiring."""
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported output embedding for `GAT`")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    embed_e = OptionalInputEmbedding(**input_embedding["edge"], use_embedding=len(inputs[1]["shape"]) < 2)
    units = attention_args["units"]
    dense0 = Dense(units=units, activation="linear")
    heads = [[head_class(**attention_args) for _ in range(attention_heads_num)] for _ in range(depth)]
    combine = LazyConcatenate(axis=-1) if attention_heads_concat else LazyAverage()
    head_act = None if attention_heads_concat else Activation(activation=attention_args["activation"])
    pool = PoolingNodes(**pooling_nodes_args) if output_embedding == "graph" else None
    out_mlp = MLP(**output_mlp) if output_embedding == "graph" else GraphMLP(**output_mlp)
    cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") \
        if (output_embedding == "node" and output_to_tensor) else None

    def forward(model_inputs, **kwargs):
        node_input, edge_input, edi = model_inputs
        nk = dense0(embed_n(node_input))
        ed = embed_e(edge_input)
        for block in heads:
            nk = combine([head([nk, ed, edi]) for head in block])
            if head_act is not None:
                nk = head_act(nk)
        if output_embedding == "graph":
            return out_mlp(pool(nk))
        out = out_mlp(nk)
        return cast(out) if cast is not None else out

    node_width = _feature_width(inputs[0], input_embedding["node"])
    edge_width = _feature_width(inputs[1], input_embedding["edge"])
    embed_n.ensure_built((None, None))
    embed_e.ensure_built((None, None))
    dense0.ensure_built((None, None, node_width))
    width = units
    for block in heads:
        for head in block:
            head.ensure_built([(None, None, width), (None, None, edge_width), (None, None, 2)])
            _build_head(head, width, edge_width)
        width = units * attention_heads_num if attention_heads_concat else units
    out_mlp.ensure_built((None, width) if output_embedding == "graph" else (None, None, width))
    flat_heads = [head for block in heads for head in block]
    model = Model(name, forward, [embed_n, embed_e, dense0] + flat_heads + [out_mlp],
                  config={"depth": depth, "attention_args": attention_args, "attention_heads_num": attention_heads_num})
    model.__kgcnn_model_version__ = __model_version__
    model.auto_graph = True   # re-bound inputs replay the whole layer sequence from one HIP graph (model/utils.py)
    return model


def _build_head(head, node_width, edge_width):
    """Create the head's weights for known input widths (the layers build lazily otherwise)."""
    units = head.units
    head.lay_linear_trafo.ensure_built((None, None, node_width))
    edge_part = edge_width if head.use_edge_features else 0
    if hasattr(head, "lay_alpha_activation"):      # GATv2: logits from the raw node features
        head.lay_alpha_activation.ensure_built((None, None, 2 * node_width + edge_part))
        head.lay_alpha.ensure_built((None, None, units))
    else:
        head.lay_alpha.ensure_built((None, None, 2 * units + edge_part))
