"""MEGNet block (mirror of kgcnn/layers/conv/megnet_conv.py:10-129): edge, node and graph-state updates of one block.

The three updates are three-layer Dense chains on concatenated inputs; the graph state reaches edges and nodes through
``GatherState`` (a per-graph row repeated over the graph's edges / nodes, ``mp_repeat_rows_f32``) and is updated from
the per-graph pools of the new edges and nodes (``mp_pool_graph_f32``).
"""
from ..base import GraphBaseLayer
from ..gather import GatherNodes, GatherState
from ..modules import Dense, LazyConcatenate
from ..pooling import PoolingGlobalEdges, PoolingLocalEdges, PoolingNodes

_KERNEL_KEYS = ("kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint", "bias_constraint",
                "kernel_initializer", "bias_initializer")


class MEGnetBlock(GraphBaseLayer):

    def __init__(self, node_embed=None, edge_embed=None, env_embed=None, pooling_method="mean", use_bias=True,
                 activation="kgcnn>softplus2", kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 kernel_constraint=None, bias_constraint=None, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        scope = locals()
        kernel_args = {key: scope[key] for key in _KERNEL_KEYS}
        kernel_args["use_bias"] = use_bias
        self.pooling_method = pooling_method
        self.node_embed = list(node_embed) if node_embed is not None else [16, 16, 16]
        self.edge_embed = list(edge_embed) if edge_embed is not None else [16, 16, 16]
        self.env_embed = list(env_embed) if env_embed is not None else [16, 16, 16]
        self.use_bias = use_bias

        def chain(widths):
            return [Dense(units=widths[0], activation=activation, **kernel_args),
                    Dense(units=widths[1], activation=activation, **kernel_args),
                    Dense(units=widths[2], activation="linear", **kernel_args)]

        # attribute order = weight order of the reference (node, edge, environment chains)
        self.lay_phi_n, self.lay_phi_n_1, self.lay_phi_n_2 = chain(self.node_embed)
        self.lay_esum = PoolingLocalEdges(pooling_method=pooling_method)
        self.lay_gather_un = GatherState()
        self.lay_conc_nu = LazyConcatenate(axis=-1)
        self.lay_phi_e, self.lay_phi_e_1, self.lay_phi_e_2 = chain(self.edge_embed)
        self.lay_gather_n = GatherNodes()
        self.lay_gather_ue = GatherState()
        self.lay_conc_enu = LazyConcatenate(axis=-1)
        self.lay_usum_e = PoolingGlobalEdges(pooling_method=pooling_method)
        self.lay_usum_n = PoolingNodes(pooling_method=pooling_method)
        self.lay_conc_u = LazyConcatenate(axis=-1)
        self.lay_phi_u, self.lay_phi_u_1, self.lay_phi_u_2 = chain(self.env_embed)

    def build(self, input_shape):
        """Weights of the three Dense chains (so ``set_weights`` works before a first call): the edge chain reads
        [n_i || n_j, e, u], the node chain [pooled e', n, u], the state chain [pooled e', pooled n', u]."""
        super().build(input_shape)
        fn, fe, fu = int(input_shape[0][-1]), int(input_shape[1][-1]), int(input_shape[3][-1])

        def build_chain(layers, widths, in_dim):
            for lay, w in zip(layers, widths):
                lay.ensure_built((None, None, in_dim))
                in_dim = w

        build_chain((self.lay_phi_e, self.lay_phi_e_1, self.lay_phi_e_2), self.edge_embed, 2 * fn + fe + fu)
        build_chain((self.lay_phi_n, self.lay_phi_n_1, self.lay_phi_n_2), self.node_embed, self.edge_embed[-1] + fn + fu)
        build_chain((self.lay_phi_u, self.lay_phi_u_1, self.lay_phi_u_2), self.env_embed,
                    self.edge_embed[-1] + self.node_embed[-1] + fu)

    @staticmethod
    def _run(layers, x, **kwargs):
        for lay in layers:
            x = lay(x, **kwargs)
        return x

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes (batch,[N],F), edges (batch,[M],Fe), edge_index (batch,[M],2), state (batch,Fu)]`` ->
        ``(nodes, edges, state)`` updated."""
        node, edge, edge_index, env = inputs
        # edge update: phi_e([n_i || n_j, e_ij, u])
        e_n = self.lay_gather_n([node, edge_index], **kwargs)
        e_u = self.lay_gather_ue([env, edge], **kwargs)
        ep = self._run((self.lay_phi_e, self.lay_phi_e_1, self.lay_phi_e_2),
                       self.lay_conc_enu([e_n, edge, e_u], **kwargs), **kwargs)
        # node update: phi_n([pool_i e'_ij, n_i, u])
        vb = self.lay_esum([node, ep, edge_index], **kwargs)
        v_u = self.lay_gather_un([env, node], **kwargs)
        vp = self._run((self.lay_phi_n, self.lay_phi_n_1, self.lay_phi_n_2),
                       self.lay_conc_nu([vb, node, v_u], **kwargs), **kwargs)
        # state update: phi_u([pool e', pool n', u])
        es = self.lay_usum_e(ep, **kwargs)
        vs = self.lay_usum_n(vp, **kwargs)
        up = self._run((self.lay_phi_u, self.lay_phi_u_1, self.lay_phi_u_2),
                       self.lay_conc_u([es, vs, env], **kwargs), **kwargs)
        return vp, ep, up

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method, "node_embed": self.node_embed, "use_bias": self.use_bias,
                       "edge_embed": self.edge_embed, "env_embed": self.env_embed})
        dense = self.lay_phi_n.get_config()
        config.update({key: dense[key] for key in _KERNEL_KEYS + ("activation",)})
        return config
