"""MI355X-native sparse transport-operator assembly (drop-in for the hot path of
TMIP-code/OceanTransportMatrixBuilder.jl v0.8.3).

Directory name: `oceantransportmatrixbuilder.jl_amd/`; import it as `otmb_amd` through the
repo-root shim `otmb_amd.py`.

Public surface (same names and keyword arguments as the reference's exports,
src/OceanTransportMatrixBuilder.jl:31-36):
    makegridmetrics, makeindices, facefluxesfrommasstransport, facefluxes, transportmatrix
The compute path is the HIP library `lib/libotmb_hip.so` (C ABI in include/otmb.h); there is
no CPU fallback -- calls raise if the library is missing.
"""
from ._nt import NT, Cube  # noqa: F401
from .gridmetrics import makegridmetrics  # noqa: F401
from . import gridtopology, synthetic  # noqa: F401


def __getattr__(name):
    # api/capi import the HIP library lazily so that host-only helpers work without it
    if name in ("makeindices", "facefluxesfrommasstransport", "facefluxes", "transportmatrix", "velocity2fluxes",
                "fluxes2velocity", "facefluxesfromvelocities", "interpolateontodefaultCgrid", "lump_and_spray", "as2D", "as3D",
                "spadd", "bolus_GM_velocity",
                "buildTadv", "buildTκH", "buildTκVML", "buildTκVdeep", "buildTkH", "buildTkVML", "buildTkVdeep"):
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
