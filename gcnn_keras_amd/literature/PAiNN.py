"""PaiNN model builder (mirror of kgcnn/literature/PAiNN.py:24-155, ``make_model``; crystal variant and the
normalisation branches are out of scope)."""
from ..layers.casting import ChangeTensorType
from ..layers.conv.painn_conv import EquivariantInitialize, PAiNNconv, PAiNNUpdate
from ..layers.geom import (BesselBasisLayer, CosCutOffEnvelope, EdgeDirectionNormalized, NodeDistanceEuclidean,
                           NodePosition)
from ..layers.mlp import MLP, GraphMLP
from ..layers.modules import LazyAdd, OptionalInputEmbedding
from ..layers.pooling import PoolingNodes
from .. import fused_painn as _fused
from ..model.utils import Model, update_model_kwargs

__model_version__ = "2022.11.25"

model_default = {
    "name": "PAiNN",
    "inputs": [
        {"shape": (None,), "name": "node_attributes", "dtype": "float32", "ragged": True},
        {"shape": (None, 3), "name": "node_coordinates", "dtype": "float32", "ragged": True},
        {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}
    ],
    "input_embedding": {"node": {"input_dim": 95, "output_dim": 128}},
    "equiv_initialize_kwargs": {"dim": 3, "method": "zeros"},
    "bessel_basis": {"num_radial": 20, "cutoff": 5.0, "envelope_exponent": 5},
    "pooling_args": {"pooling_method": "sum"},
    "conv_args": {"units": 128, "cutoff": None, "conv_pool": "sum"},
    "update_args": {"units": 128},
    "equiv_normalization": False, "node_normalization": False,
    "depth": 3,
    "verbose": 10,
    "output_embedding": "graph", "output_to_tensor": True,
    "output_mlp": {"use_bias": [True, True], "units": [128, 1], "activation": ["swish", "linear"]}
}


@update_model_kwargs(model_default)
def make_model(inputs: list = None, input_embedding: dict = None, equiv_initialize_kwargs: dict = None,
               bessel_basis: dict = None, depth: int = None, pooling_args: dict = None, conv_args: dict = None,
               update_args: dict = None, equiv_normalization: bool = None, node_normalization: bool = None,
               name: str = None, verbose: int = None, output_embedding: str = None, output_to_tensor: bool = None,
               output_mlp: dict = None):
    r"""Build PaiNN (kgcnn/literature/PAiNN.py:45-155).  Model inputs ``[node_attributes, node_coordinates,
    bond_indices]`` (+ optional ``equiv_initial``); output ``(batch, L)`` for ``output_embedding="graph"``."""
    if equiv_normalization or node_normalization:
        raise NotImplementedError("GraphLayerNormalization / GraphBatchNormalization are outside the hot path")
    if output_embedding not in ("graph", "node"):
        raise ValueError("Unsupported output embedding for mode `PAiNN`")
    embed = OptionalInputEmbedding(**input_embedding["node"], use_embedding=len(inputs[0]["shape"]) < 2)
    equiv_init = EquivariantInitialize(**equiv_initialize_kwargs) if len(inputs) <= 3 else None
    lay_pos, lay_dir, lay_dist = NodePosition(), EdgeDirectionNormalized(), NodeDistanceEuclidean()
    lay_env = CosCutOffEnvelope(conv_args["cutoff"])
    lay_rbf = BesselBasisLayer(**bessel_basis)
    convs = [PAiNNconv(**conv_args) for _ in range(depth)]
    updates = [PAiNNUpdate(**update_args) for _ in range(depth)]
    adds = [[LazyAdd() for _ in range(4)] for _ in range(depth)]
    pool = PoolingNodes(**pooling_args) if output_embedding == "graph" else None
    out_mlp = MLP(**output_mlp) if output_embedding == "graph" else GraphMLP(**output_mlp)
    cast = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor") \
        if (output_embedding == "node" and output_to_tensor) else None

    def forward(model_inputs, fused=None, **kwargs):
        # Fused pipeline (gcnn_keras_amd/fused_painn.py: per block five MFMA GEMMs + three fused kernels, one HIP graph per
        # bound batch) whenever configuration and inputs fit it and no gradient is requested; ``fused=False`` forces the
        # layer sequence below.  Forces take the fused reverse pass through ``EnergyForceModel`` (model/force.py).
        if route is not None and fused is not False and route.accepts(model_inputs):
            return route(model_inputs)
        if fused is True:
            raise ValueError("this PAiNN configuration / these inputs do not fit the fused pipeline")
        node_input, xyz_input, edi = model_inputs[:3]
        z = embed(node_input)
        v = model_inputs[3] if len(model_inputs) > 3 else equiv_init(z)
        pos1, pos2 = lay_pos([xyz_input, edi])
        rij = lay_dir([pos1, pos2])
        d = lay_dist([pos1, pos2])
        env = lay_env(d)
        rbf = lay_rbf(d)
        for i in range(depth):
            ds, dv = convs[i]([z, v, rbf, env, rij, edi])
            z = adds[i][0]([z, ds])
            v = adds[i][1]([v, dv])
            ds, dv = updates[i]([z, v])
            z = adds[i][2]([z, ds])
            v = adds[i][3]([v, dv])
        if output_embedding == "graph":
            return out_mlp(pool(z))
        out = out_mlp(z)
        return cast(out) if cast is not None else out

    f = conv_args["units"]
    emb_dim = input_embedding["node"]["output_dim"] if len(inputs[0]["shape"]) < 2 else inputs[0]["shape"][-1]
    embed.ensure_built((None, None))
    for i in range(depth):
        convs[i].ensure_built([(None, None, emb_dim), (None, None, 3, f), (None, None, bessel_basis["num_radial"]),
                               (None, None, 1), (None, None, 3), (None, None, 2)])
        updates[i].ensure_built([(None, None, f), (None, None, 3, f)])
    out_mlp.ensure_built((None, f) if output_embedding == "graph" else (None, None, f))
    layers = [embed, lay_rbf]
    for i in range(depth):
        layers += [convs[i], updates[i]]
    layers.append(out_mlp)

    def fused_tensors():
        """The model's live weight tensors under the names of ``synth.painn_params``."""
        p = {"embedding": embed.embeddings, "bessel/frequencies": lay_rbf.frequencies}
        for i in range(depth):
            c, u = "conv%d/" % i, "update%d/" % i
            for name, lay in (("dense1", convs[i].lay_dense1), ("phi", convs[i].lay_phi), ("w", convs[i].lay_w)):
                p[c + name + "/kernel"], p[c + name + "/bias"] = lay.kernel, lay.bias
            p[u + "dense1/kernel"], p[u + "dense1/bias"] = updates[i].lay_dense1.kernel, updates[i].lay_dense1.bias
            p[u + "lin_u/kernel"], p[u + "lin_v/kernel"] = updates[i].lay_lin_u.kernel, updates[i].lay_lin_v.kernel
            p[u + "a/kernel"], p[u + "a/bias"] = updates[i].lay_a.kernel, updates[i].lay_a.bias
        for k, d in enumerate(out_mlp.mlp_dense_layer_list):
            p["output_mlp/%d/kernel" % k], p["output_mlp/%d/bias" % k] = d.kernel, d.bias
        return p

    merged = {"inputs": inputs, "input_embedding": input_embedding, "equiv_initialize_kwargs": equiv_initialize_kwargs,
              "bessel_basis": bessel_basis, "depth": depth, "pooling_args": pooling_args, "conv_args": conv_args,
              "update_args": update_args, "equiv_normalization": equiv_normalization,
              "node_normalization": node_normalization, "output_embedding": output_embedding, "output_mlp": output_mlp}
    route = None
    if _fused.supports(merged) and embed.use_embedding:
        route = _fused.PainnFusedRoute(fused_tensors, merged)
    model = Model(name, forward, layers, config={"depth": depth, "conv_args": conv_args, "update_args": update_args})
    model.__kgcnn_model_version__ = __model_version__
    model.fused = route   # None: this configuration always runs the layer path
    return model
