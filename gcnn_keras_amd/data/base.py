"""``MemoryGraphList``: the list-of-graph-dicts container of the reference, reduced to what the hot path's callers use
(kgcnn/data/base.py:28-239): list behaviour, ``obtain_property`` and ``tensor(items)`` - the latter on the native packer."""
import numpy as np

from .packer import BatchPacker


class GraphDict(dict):
    """kgcnn.graph.base.GraphDict is a dict of NumPy arrays with helper methods; the path needs ``get`` / ``set`` /
    ``apply_preprocessor`` only (kgcnn/graph/base.py)."""

    def set(self, key, value):
        self[key] = np.asarray(value)
        return self

    def apply_preprocessor(self, name, **kwargs):
        if not callable(name):
            raise TypeError("preprocessors are callables on this engine (graph -> dict of new properties)")
        out = name(self, **kwargs)
        if isinstance(out, dict):
            self.update(out)
        return self


class MemoryGraphList(list):

    def __init__(self, graphs=()):
        super().__init__(g if isinstance(g, GraphDict) else GraphDict(g) for g in graphs)
        self._packers = {}

    def copy(self):
        return MemoryGraphList(GraphDict({k: np.array(v) for k, v in g.items()}) for g in self)

    def obtain_property(self, key):
        """List of the property over all graphs; ``None`` where a graph lacks it (kgcnn/data/base.py:129-150)."""
        return [g.get(key) for g in self]

    def tensor(self, items, device="cuda"):
        """Packed device tensors for the reference's item descriptors (kgcnn/data/base.py:219-239): a list for a list of
        items, a dict for a dict of items, one tensor for a single item."""
        single = isinstance(items, dict) and "name" in items
        spec = [items] if single else items
        if not isinstance(spec, (list, tuple, dict)):
            raise TypeError("Wrong type, expected e.g. [{'name': 'edge_indices', 'ragged': True}, {...}, ...]")
        key = repr(spec)
        if key not in self._packers:
            names = [it["name"] for it in (spec.values() if isinstance(spec, dict) else spec)]
            index_item = next((n for n in names if n.endswith("_indices")), None)
            self._packers[key] = BatchPacker(spec, index_item=index_item, device=device)
        batch = self._packers[key].pack(self).wait()
        if single:
            return batch[spec[0]["name"]]
        if isinstance(spec, dict):
            return {k: batch[k] for k in spec}
        return [batch[it["name"]] for it in spec]
