"""Data-format side of the hot path: host packing of graph lists into the engine's ragged tensors."""
from .utils import ragged_tensor_from_nested_numpy  # noqa: F401
from .packer import BatchPacker, HostBuffer, PackedBatch  # noqa: F401
