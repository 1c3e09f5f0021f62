"""Tiny NamedTuple stand-in: the reference returns Julia NamedTuples `(; a, b, ...)`;
the host mirror returns `NT(a=..., b=...)` with both attribute and key access."""


class NT(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class Cube:
    """Stand-in for a YAXArray: data plus a `.properties` dict (e.g. {"_FillValue": 1e20}),
    which is all the reference reads from its inputs (src/velocities.jl:120,
    src/gridcellgeometry.jl:270-271)."""

    def __init__(self, data, **properties):
        self.data = data
        self.properties = dict(properties)

    def __array__(self, dtype=None, copy=None):
        import numpy as np

        return np.asarray(self.data, dtype=dtype)


def data_and_props(x):
    import numpy as np

    props = getattr(x, "properties", {}) or {}
    data = x.data if isinstance(x, Cube) else x
    if isinstance(data, np.ma.MaskedArray):
        data = data.astype(np.float64).filled(np.nan)
    return data, props
