"""GIN model builder (mirror of kgcnn/literature/GIN.py:19-118, ``make_model``).

``depth`` blocks of ``GIN`` aggregation (fused gather-reduce kernel) + ``GraphMLP``; for graph outputs every block's
embedding (and the input embedding) is pooled (``PoolingNodes``, default mean), sent through its own ``last_mlp`` and the
results are summed before the output MLP (GIN.py:97-102).  The reference's default ``gin_mlp`` switches on batch
normalisation, a training-time construct the engine does not carry: the default here is the same MLP without it, and
asking for normalisation raises.
"""
from ..layers.casting import ChangeTensorType
from ..layers.conv.gin_conv import GIN
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import Dense, LazyAdd, OptionalInputEmbedding
from ..layers.pooling import PoolingNodes
from ..model.utils import Model, update_model_kwargs

__model_version__ = "2022.11.25"

model_default = {
    "name": "GIN",
    "inputs": [{"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 64}},
    "gin_mlp": {"units": [64, 64], "use_bias": True, "activation": ["relu", "linear"],
                "use_normalization": False, "normalization_technique": "graph_batch"},
    "gin_args": {},
    "depth": 3, "dropout": 0.0, "verbose": 10,
    "last_mlp": {"use_bias": [True, True, True], "units": [64, 64, 64],
                 "activation": ["relu", "relu", "linear"]},
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": True, "units": 1,
                   "activation": "softmax"}
}


@update_model_kwargs(model_default)
def make_model(inputs: list = None, input_embedding: dict = None, depth: int = None, gin_args: dict = None,
               gin_mlp: dict = None, last_mlp: dict = None, dropout: float = None, name: str = None,
               verbose: int = None, output_embedding: str = None, output_to_tensor: bool = None,
               output_mlp: dict = None):
    r"""Build GIN (kgcnn/literature/GIN.py:82-116).  Model inputs ``[node_attributes, edge_indices]``; dropout is the
    identity at inference."""
    assert len(inputs) == 2
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported output embedding for mode `GIN`")
    embed_n = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    n_units = gin_mlp["units"][-1] if isinstance(gin_mlp["units"], list) else int(gin_mlp["units"])
    dense0 = Dense(n_units, use_bias=True, activation="linear")
    gins = [GIN(**gin_args) for _ in range(depth)]
    mlps = [GraphMLP(**gin_mlp) for _ in range(depth)]
    if output_embedding == "graph":
        pools = [PoolingNodes() for _ in range(depth + 1)]
        lasts = [MLP(**last_mlp) for _ in range(depth + 1)]
        out_mlp = MLP(**output_mlp)
        add, cast = LazyAdd(), None
    else:
        pools, lasts = [], [GraphMLP(**last_mlp)]
        out_mlp = GraphMLP(**output_mlp)
        add = None
        cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") if output_to_tensor else None

    def forward(model_inputs, **kwargs):
        node_input, edi = model_inputs
        n = dense0(embed_n(node_input))
        embeddings = [n]
        for gin, mlp in zip(gins, mlps):
            n = mlp(gin([n, edi]))
            embeddings.append(n)
        if output_embedding == "graph":
            parts = [last(pool(x)) for last, pool, x in zip(lasts, pools, embeddings)]
            return out_mlp(add(parts))
        out = out_mlp(lasts[0](n))
        return cast(out) if cast is not None else out

    in_width = input_embedding["node"]["output_dim"] if len(inputs[0]["shape"]) < 2 else inputs[0]["shape"][-1]
    embed_n.ensure_built((None, None))
    dense0.ensure_built((None, None, in_width))
    for gin, mlp in zip(gins, mlps):
        gin.ensure_built([(None, None, n_units), (None, None, 2)])
        mlp.ensure_built((None, None, n_units))
    last_units = last_mlp["units"][-1] if isinstance(last_mlp["units"], list) else int(last_mlp["units"])
    for last in lasts:
        last.ensure_built((None, n_units) if output_embedding == "graph" else (None, None, n_units))
    out_mlp.ensure_built((None, last_units) if output_embedding == "graph" else (None, None, last_units))
    model = Model(name, forward, [embed_n, dense0] + [lay for pair in zip(gins, mlps) for lay in pair] + lasts + [out_mlp],
                  config={"depth": depth, "gin_mlp": gin_mlp, "gin_args": gin_args})
    model.__kgcnn_model_version__ = __model_version__
    model.auto_graph = True   # re-bound inputs replay the whole layer sequence from one HIP graph (model/utils.py)
    return model
