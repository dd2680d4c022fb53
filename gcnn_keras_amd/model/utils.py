"""Model keyword handling (mirror of kgcnn/model/utils.py:69-142): nested merge of user kwargs into the model
defaults, used as ``@update_model_kwargs(model_default)`` on every ``make_model``."""
import copy
import functools
import logging
from math import inf

module_logger = logging.getLogger(__name__)
module_logger.setLevel(logging.WARNING)


def update_model_kwargs_logic(default_kwargs: dict = None, user_kwargs: dict = None, update_recursive=inf):
    out = {}
    if default_kwargs is None:
        default_kwargs = {}
    if user_kwargs is None:
        user_kwargs = {}
    for iter_key in user_kwargs.keys():
        if iter_key not in default_kwargs:
            raise ValueError("Model kwarg {0} not in default arguments {1}".format(iter_key, default_kwargs.keys()))
    out.update(copy.deepcopy(default_kwargs))

    def _nested_update(dict1, dict2, max_depth=inf, depth=0):
        for key, values in dict2.items():
            if key not in dict1:
                module_logger.warning("Model kwargs: Unknown key {0} with value {1}".format(key, values))
                dict1[key] = values
                continue
            if not isinstance(dict1[key], dict):
                dict1[key] = values
                continue
            if not isinstance(values, dict):
                module_logger.warning("Model kwargs: Overwriting dictionary of {0} with {1}".format(key, values))
                dict1[key] = values
                continue
            if depth < max_depth:
                dict1[key] = _nested_update(dict1[key], values, max_depth=max_depth, depth=depth + 1)
            else:
                dict1[key] = values
        return dict1

    return _nested_update(out, user_kwargs, update_recursive, 0)


def update_model_kwargs(model_default, update_recursive=inf):
    def model_update_decorator(func):
        @functools.wraps(func)
        def update_wrapper(*args, **kwargs):
            updated_kwargs = update_model_kwargs_logic(model_default, kwargs, update_recursive)
            if len(args) > 0:
                module_logger.error("Can only update kwargs, not %s" % args)
            return func(*args, **updated_kwargs)
        return update_wrapper
    return model_update_decorator


class Model:
    """Minimal stand-in for ``ks.models.Model``: an ordered list of layers plus a forward function."""

    def __init__(self, name, forward, layers, config=None):
        self.name = name
        self._forward = forward
        self.layers = layers
        self.config = config or {}

    def __call__(self, inputs, **kwargs):
        return self._forward(inputs, **kwargs)

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
