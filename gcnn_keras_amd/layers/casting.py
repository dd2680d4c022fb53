"""``ChangeTensorType`` ragged <-> padded / mask / partition (mirror of kgcnn/layers/casting.py:13-102)."""
import torch

from .. import _ffi
from ..ops.partition import change_partition_by_name
from .base import GraphBaseLayer


class ChangeTensorType(GraphBaseLayer):

    def __init__(self, input_tensor_type: str = "RaggedTensor", output_tensor_type: str = "RaggedTensor",
                 partition_type: str = "row_length", shape=None, default_value=None, boolean_mask: bool = False,
                 **kwargs):
        super().__init__(**kwargs)
        self.partition_type = partition_type
        self.input_tensor_type = str(input_tensor_type)
        self.output_tensor_type = str(output_tensor_type)
        self.shape = shape
        self.default_value = default_value
        self.boolean_mask = boolean_mask
        self._str_type_ragged = ["ragged", "RaggedTensor"]
        self._str_type_tensor = ["Tensor", "tensor"]
        self._str_type_mask = ["padded", "masked", "mask"]
        self._str_type_partition = ["disjoint", "row_partition", "values_partition", "values"]

    def _to_padded(self, inputs, with_mask):
        if self.shape is not None or self.default_value not in (None, 0, 0.0):
            raise NotImplementedError("explicit shape / default_value are outside the hot path")
        _ffi.require_device(inputs.values, inputs.row_splits)
        splits = inputs.row_splits_host()
        g = inputs.nrows()
        nmax = int((splits[1:] - splits[:-1]).max()) if g > 0 else 0
        vals = inputs.values.contiguous()
        elems = 1
        for d in vals.shape[1:]:
            elems *= int(d)
        padded = torch.empty((g, nmax) + tuple(vals.shape[1:]), dtype=torch.float32, device=vals.device)
        mask = torch.empty_like(padded) if with_mask else None
        _ffi.call("mp_ragged_to_padded_f32", _ffi.ptr(vals), _ffi.ptr(inputs.row_splits), g, nmax, max(elems, 1),
                  _ffi.ptr(padded), _ffi.ptr(mask), _ffi.stream())
        return padded, mask

    def call(self, inputs, **kwargs):
        if self.input_tensor_type in self._str_type_ragged:
            if self.output_tensor_type in self._str_type_ragged:
                return inputs
            if self.output_tensor_type in self._str_type_tensor:
                return self._to_padded(inputs, False)[0]
            if self.output_tensor_type in self._str_type_mask:
                padded, mask = self._to_padded(inputs, True)
                return padded, (mask.to(torch.bool) if self.boolean_mask else mask)
            if self.output_tensor_type in self._str_type_partition:
                inputs = self.assert_ragged_input_rank(inputs, ragged_rank=1)
                part = change_partition_by_name(inputs.row_splits, "row_splits",
                                                self.partition_type if self.partition_type != "row_length"
                                                else "row_lengths")
                return [inputs.values, part]
        raise NotImplementedError(
            "Unsupported conversion from '%s' to '%s'." % (self.input_tensor_type, self.output_tensor_type))

    def get_config(self):
        config = super().get_config()
        config.update({"partition_type": self.partition_type, "input_tensor_type": self.input_tensor_type,
                       "output_tensor_type": self.output_tensor_type, "shape": self.shape,
                       "default_value": self.default_value, "boolean_mask": self.boolean_mask})
        return config
