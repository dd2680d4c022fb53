"""Model keyword handling with the contract of kgcnn/model/utils.py:69-142 (written independently of it):

* ``update_model_kwargs_logic(defaults, user, update_recursive)`` returns a fresh dictionary = deep copy of the defaults
  overlaid with the user's entries; a top-level user key that the defaults do not know raises ``ValueError``; where both
  sides hold a dictionary the overlay recurses (``update_recursive`` levels deep, below that it replaces wholesale); a
  dictionary replaced by a non-dictionary, and a nested key the defaults lack, are accepted with a warning.
* ``@update_model_kwargs(model_default)`` applies that to the keyword arguments of a ``make_model`` and, like the
  reference, takes the logger level from the merged ``verbose`` entry; positional arguments pass through untouched
  (with an error log, since they cannot be merged).
"""
import copy
import functools
import logging
from math import inf

module_logger = logging.getLogger(__name__)
module_logger.setLevel(logging.WARNING)


def _overlay(base, patch, levels_left):
    """Write ``patch`` over ``base`` in place; ``levels_left`` counts how many more dictionary levels may be merged
    rather than replaced."""
    for name, new_value in patch.items():
        old_value = base.get(name, _overlay)        # the function object doubles as the 'absent' marker
        if old_value is _overlay:
            module_logger.warning("Model kwargs: Unknown key %s with value %s", name, new_value)
            base[name] = new_value
        elif isinstance(old_value, dict) and isinstance(new_value, dict) and levels_left > 0:
            _overlay(old_value, new_value, levels_left - 1)
        else:
            if isinstance(old_value, dict) and not isinstance(new_value, dict):
                module_logger.warning("Model kwargs: Overwriting dictionary of %s with %s", name, new_value)
            base[name] = new_value
    return base


def update_model_kwargs_logic(default_kwargs: dict = None, user_kwargs: dict = None, update_recursive=inf):
    defaults = default_kwargs or {}
    user = user_kwargs or {}
    unknown = [name for name in user if name not in defaults]
    if unknown:
        raise ValueError("Model kwarg %s not in default arguments %s" % (unknown[0], list(defaults)))
    return _overlay(copy.deepcopy(defaults), user, update_recursive)


def update_model_kwargs(model_default, update_recursive=inf):
    def decorate(make_model):
        @functools.wraps(make_model)
        def with_defaults(*args, **kwargs):
            merged = update_model_kwargs_logic(model_default, kwargs, update_recursive)
            if "verbose" in merged:
                module_logger.setLevel(merged["verbose"])
            module_logger.info("Updated model kwargs: %s", merged)
            if args:
                module_logger.error("Can only update kwargs, not %s", args)
            return make_model(*args, **merged)
        return with_defaults
    return decorate


class Model:
    """Minimal stand-in for ``ks.models.Model``: an ordered list of layers plus a forward function.

    ``auto_graph``: a layer-path model issues one engine call per Keras layer (a dozen to a few hundred launches), and at
    QM9 / Cora sizes the host cannot keep the GPU busy that way.  With ``auto_graph=True`` the model remembers, per
    distinct input set (identified by the storage of its tensors), how often it was called: the first call runs eagerly
    (building and caching the index plans on the ragged inputs), the second captures the same call sequence into ONE HIP
    graph (``engine.GraphedModel``), later calls replay it and return a fresh copy of the output.  Calls that ask for
    gradients, pass keyword arguments, or whose capture fails (a layer that needs a host read per call) stay eager.
    """

    def __init__(self, name, forward, layers, config=None, auto_graph=False, max_graphs=4):
        self.name = name
        self._forward = forward
        self.layers = layers
        self.config = config or {}
        self.auto_graph = bool(auto_graph)
        self.max_graphs = int(max_graphs)
        self._graphs = {}      # input identity -> [calls, GraphedModel | None | False]
        self.last_route = None  # "eager" | "graph"

    def __call__(self, inputs, **kwargs):
        if self.auto_graph and not kwargs:
            key = self._graph_key(inputs)
            if key is not None:
                return self._call_graphed(key, inputs)
        self.last_route = "eager"
        return self._forward(inputs, **kwargs)

    # -- graph replay of re-bound inputs ----------------------------------------------------------------------------------
    @staticmethod
    def _graph_key(inputs):
        import torch
        from .. import _ffi
        from ..ragged import RaggedTensor
        if not _ffi.has_gpu() or not isinstance(inputs, (list, tuple)):
            return None
        key = []
        for x in inputs:
            parts = (x.values, x.row_splits) if isinstance(x, RaggedTensor) else (x,)
            for t in parts:
                if not torch.is_tensor(t) or not t.is_cuda or (torch.is_grad_enabled() and t.requires_grad):
                    return None
                key.append((t.data_ptr(), tuple(t.shape), t._version if t.dtype == torch.int64 else 0))
        return tuple(key)

    def _call_graphed(self, key, inputs):
        from ..ragged import RaggedTensor
        entry = self._graphs.get(key)
        if entry is None:
            while len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            entry = self._graphs[key] = [0, None, inputs]   # the entry keeps the inputs alive: addresses stay an identity
        entry[0] += 1
        if entry[0] == 1 or entry[1] is False:
            self.last_route = "eager"
            return self._forward(inputs)
        if entry[1] is None:
            from ..engine import GraphedModel
            try:
                entry[1] = GraphedModel(self._forward, inputs, grad=False)
            except Exception:   # e.g. a layer that reads a size back per call cannot be captured: stay eager
                import torch
                torch.cuda.synchronize()
                entry[1] = False
                self.last_route = "eager"
                return self._forward(inputs)
        out = entry[1]()
        self.last_route = "graph"
        if isinstance(out, RaggedTensor):
            return out.with_values(out.values.clone())
        if isinstance(out, (list, tuple)):
            return type(out)(o.clone() if hasattr(o, "clone") else o for o in out)
        return out.clone()

    def release_graphs(self):
        self._graphs.clear()

    predict = __call__

    @property
    def weights(self):
        out = []
        for lay in self.layers:
            out.extend((lay.name + "/" + n, t) for n, t in lay.weights)
        return out

    def get_weights(self):
        return [t.detach().cpu().numpy() for _, t in self.weights]

    def set_weights(self, arrays):
        ws = self.weights
        if len(ws) != len(arrays):
            raise ValueError("Model %s expects %d weight arrays, got %d" % (self.name, len(ws), len(arrays)))
        import numpy as np
        import torch
        for (n, t), a in zip(ws, arrays):
            a = np.asarray(a, dtype=np.float32)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("Shape mismatch for %s: %s vs %s" % (n, tuple(a.shape), tuple(t.shape)))
            t.copy_(torch.from_numpy(a))
