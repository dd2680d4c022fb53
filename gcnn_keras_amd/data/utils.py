"""``ragged_tensor_from_nested_numpy`` with the reference's signature (kgcnn/data/utils.py:129-157), on the native packer."""
import numpy as np

from .packer import pack_rows, to_device


def ragged_tensor_from_nested_numpy(numpy_list, dtype=None, row_splits_dtype="int64", device="cuda"):
    r"""List of per-graph arrays (equal trailing shape) -> ``RaggedTensor`` of shape ``(batch, None, ...)``.

    The reference concatenates with NumPy and hands the result to ``tf.RaggedTensor.from_row_lengths``
    (kgcnn/data/utils.py:156-157); here the concatenation (and the optional ``dtype`` conversion) is one pass of
    ``mp_pack_rows_host`` into staging memory followed by one host-to-device copy.  ``row_splits_dtype`` other than
    int64 is rejected: the engine's partitions are int64 like the reference's default."""
    if np.dtype(row_splits_dtype) != np.dtype("int64"):
        raise ValueError("row_splits are int64 on this engine (reference default, kgcnn/data/utils.py:129)")
    values, splits = pack_rows(numpy_list, dtype=dtype)
    return to_device(values, splits, device=device)
