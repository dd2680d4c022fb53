"""Axis normalisation (contract of ``kgcnn.ops.axis.get_positive_axis``, reference kgcnn/ops/axis.py:4-36)."""


def get_positive_axis(axis, ndims, axis_name="axis", ndims_name="ndims"):
    """Return ``axis`` as a non-negative index into a rank-``ndims`` tensor.

    Raises ``TypeError`` for a non-int axis and ``ValueError`` when it is out of ``[-ndims, ndims)`` or when a
    negative axis is given without a known rank."""
    if not isinstance(axis, int):
        raise TypeError("%s must be an int; got %s" % (axis_name, type(axis).__name__))
    if ndims is None:
        if axis < 0:
            raise ValueError("%s may only be negative if %s is statically known." % (axis_name, ndims_name))
        return axis
    if not -ndims <= axis < ndims:
        raise ValueError("%s=%s out of bounds: expected %s<=%s<%s" % (axis_name, axis, -ndims, axis_name, ndims))
    return axis % ndims if ndims else axis
