"""Axis normalisation helper (same contract as kgcnn/ops/axis.py:4-36)."""


def get_positive_axis(axis, ndims, axis_name="axis", ndims_name="ndims"):
    if not isinstance(axis, int):
        raise TypeError("%s must be an int; got %s" % (axis_name, type(axis).__name__))
    if ndims is not None:
        if 0 <= axis < ndims:
            return axis
        elif -ndims <= axis < 0:
            return axis + ndims
        raise ValueError("%s=%s out of bounds: expected %s<=%s<%s" % (axis_name, axis, -ndims, axis_name, ndims))
    elif axis < 0:
        raise ValueError("%s may only be negative if %s is statically known." % (axis_name, ndims_name))
    return axis
