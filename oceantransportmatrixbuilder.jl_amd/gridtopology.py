"""Grid topology detection (host side).

Mirrors `getgridtopology` / `isapprox_lon` of the reference
(src/gridtopology.jl:23-53).  The index shifts themselves (i₊₁ … k₋₁,
src/gridtopology.jl:57-68,94) live in the HIP kernels (csrc/otmb_topology.h).
"""
import numpy as np

BIPOLAR = 0
TRIPOLAR = 1
UNKNOWN = 2
NAMES = {BIPOLAR: "BipolarGridTopology", TRIPOLAR: "TripolarGridTopology", UNKNOWN: "UnknownGridTopology"}


def _rot180(a):
    return a[::-1, ::-1]


def isapprox_lon(a, b):
    """gridtopology.jl:23-26: isapprox(mod(a-b+180,360)-180, 0, atol=eps(180.0)) on the array norm."""
    d = np.mod(a - b + 180.0, 360.0) - 180.0
    # isapprox(x, y; atol) for arrays: norm(x-y) <= max(atol, rtol*max(norm(x),norm(y))), rtol=0 when atol>0
    return float(np.linalg.norm(d)) <= np.spacing(180.0)


def _isapprox_arr(a, b):
    """isapprox(A, B) with default rtol = sqrt(eps) on the 2-norm."""
    rtol = np.sqrt(np.finfo(np.float64).eps)
    return float(np.linalg.norm(a - b)) <= rtol * max(float(np.linalg.norm(a)), float(np.linalg.norm(b)))


def getgridtopology(lon_vertices, lat_vertices, lev=None):
    """gridtopology.jl:33-53.  lon/lat_vertices have shape (4, nx, ny), default vertex order."""
    NPlon = lon_vertices[2:4, :, -1]
    NPlat = lat_vertices[2:4, :, -1]
    if np.all(NPlat == 90):
        return BIPOLAR
    if isapprox_lon(NPlon, _rot180(NPlon)) and _isapprox_arr(NPlat, _rot180(NPlat)):
        return TRIPOLAR
    return UNKNOWN
