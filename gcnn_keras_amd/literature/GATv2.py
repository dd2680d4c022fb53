"""GATv2 model builder (mirror of kgcnn/literature/GATv2.py): the GAT wiring with ``AttentionHeadGATV2`` heads."""
from ..layers.conv.gat_conv import AttentionHeadGATV2
from ..model.utils import update_model_kwargs
from . import GAT

__model_version__ = "2022.11.25"

model_default = dict(GAT.model_default, name="GATv2")


@update_model_kwargs(model_default)
def make_model(**kwargs):
    r"""Build GATv2: per block ``attention_heads_num`` ``AttentionHeadGATV2`` heads, concatenated or averaged
    (kwargs already merged with ``model_default``; the undecorated GAT builder does the wiring)."""
    return GAT.make_model.__wrapped__(head_class=AttentionHeadGATV2, **kwargs)
