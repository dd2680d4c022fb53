"""``MessagePassingBase`` template (mirror of kgcnn/layers/message.py:9-104): gather -> message -> aggregate -> update.
Subclass it and implement ``message_function`` and ``update_nodes``; aggregation defaults to ``PoolingLocalEdges``."""
from .base import GraphBaseLayer
from .gather import GatherEmbeddingSelection
from .pooling import PoolingLocalEdges


class MessagePassingBase(GraphBaseLayer):

    def __init__(self, pooling_method: str = "sum", **kwargs):
        super().__init__(**kwargs)
        self.pooling_method = pooling_method
        self.lay_gather = GatherEmbeddingSelection([0, 1])
        self.lay_pool_default = PoolingLocalEdges(pooling_method=self.pooling_method)

    def message_function(self, inputs, **kwargs):
        r"""inputs: ``[nodes_in, nodes_out, edges]`` -> messages ``(batch, [M], F)``."""
        raise NotImplementedError(
            "A method to generate messages must be implemented in sub-class of `MessagePassingBase`.")

    def aggregate_message(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, edge_index]`` -> aggregated messages per node."""
        return self.lay_pool_default(inputs, **kwargs)

    def update_nodes(self, inputs, **kwargs):
        r"""inputs: ``[nodes, node_updates]`` -> updated nodes."""
        raise NotImplementedError(
            "A method to update nodes must be implemented in sub-class of `MessagePassingBase`.")

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, edge_index]`` -> updated node embeddings ``(batch, [N], F)``."""
        nodes, edges, edge_index = inputs
        n_in, n_out = self.lay_gather([nodes, edge_index], **kwargs)
        msg = self.message_function([n_in, n_out, edges], **kwargs)
        pool_n = self.aggregate_message([nodes, msg, edge_index], **kwargs)
        n_new = self.update_nodes([nodes, pool_n], **kwargs)
        return n_new

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method})
        return config
