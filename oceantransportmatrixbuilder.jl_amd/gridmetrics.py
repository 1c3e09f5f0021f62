"""Host-side `makegridmetrics` (numpy).

Mirrors src/gridcellgeometry.jl:265-311 and helpers (:158-178 vertexpermutation,
:182-189 horizontaldistance, :209-222 vertexindices/verticalfacewidth,
:240-255 centroid2edgedistance/midpointonsphere) and Distances.haversine 0.10.
It is O(nx*ny) transcendental work plus O(G) elementwise work and sits on the
boundary of the hot path; the device version is
`otmb_makegridmetrics_dev` once built.

Arrays use Julia's shapes: 3-D (nx,ny,nz), 2-D (nx,ny), vertices (4,nx,ny); they
are returned Fortran-ordered so that their memory is exactly Julia's column-major.
"""
import numpy as np

from . import gridtopology as gt
from ._nt import NT, data_and_props

EARTH_RADIUS = 6371000.0
DIRS = ("south", "east", "north", "west")  # gridcellgeometry.jl:304
_VERTEXINDICES = {"south": (0, 1), "east": (1, 2), "north": (2, 3), "west": (0, 3)}  # :209-215 (0-based)


def haversine(lon1, lat1, lon2, lat2):
    """Distances.Haversine(6371000)((lon1,lat1),(lon2,lat2)), degrees in, metres out."""
    d2r = np.pi / 180.0
    dl = (lon2 - lon1) * d2r  # Δλ = deg2rad(y[1] - x[1])
    p1 = lat1 * d2r           # φ₁ = deg2rad(x[2])
    p2 = lat2 * d2r           # φ₂ = deg2rad(y[2])
    dp = p2 - p1              # Δφ = φ₂ - φ₁: converted first, subtracted after (Distances.jl 0.10, haversine.jl)
    s1 = np.sin(dp / 2)
    s2 = np.sin(dl / 2)
    a = s1 * s1 + np.cos(p1) * np.cos(p2) * (s2 * s2)
    return 2 * (EARTH_RADIUS * np.arcsin(np.minimum(np.sqrt(a), 1.0)))


def midpointonsphere(lonA, latA, lonB, latB):
    """gridcellgeometry.jl:249-255."""
    cross = ~(np.abs(lonA - lonB) < 180)
    lon = (lonA + lonB) / 2 + np.where(cross, 180.0, 0.0)
    lat = (latA + latB) / 2 + 0.0
    return lon, lat


def vertexpermutation(lon_vertices, lat_vertices):
    """gridcellgeometry.jl:158-178: permutation (0-based) that sorts the 4 vertices into
    SW, SE, NE, NW, found from cell (1,1) and its east and north neighbours."""
    assert lon_vertices.shape[0] == 4 and lat_vertices.shape[0] == 4
    pts = [(lon_vertices[v, 0, 0], lat_vertices[v, 0, 0]) for v in range(4)]
    pts_e = {(lon_vertices[v, 1, 0], lat_vertices[v, 1, 0]) for v in range(4)}
    pts_n = {(lon_vertices[v, 0, 1], lat_vertices[v, 0, 1]) for v in range(4)}
    idx_east = [v for v in range(4) if pts[v] in pts_e]
    idx_north = [v for v in range(4) if pts[v] in pts_n]

    def only(xs):
        xs = list(xs)
        if len(xs) != 1:
            raise ValueError("Collection must contain exactly 1 element")
        return xs[0]

    idx3 = only(v for v in idx_east if v in idx_north)
    idx2 = only(v for v in idx_east if v != idx3)
    idx4 = only(v for v in idx_north if v != idx3)
    idx1 = only(v for v in range(4) if v not in (idx2, idx3, idx4))
    return [idx1, idx2, idx3, idx4]


def _neighbour2d(nx, ny, topology, d):
    """(i,j) index arrays (0-based) of the topological neighbour in direction d and a mask of
    `nothing`s.  Pairing south<->j-1, east<->i+1, north<->j+1, west<->i-1 (gridcellgeometry.jl:304-305;
    shifts gridtopology.jl:59-65,95)."""
    if topology == gt.UNKNOWN:
        raise RuntimeError("Unknown grid type")
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    none = np.zeros((nx, ny), dtype=bool)
    if d == "east":
        return np.where(i + 1 < nx, i + 1, 0), j, none
    if d == "west":
        return np.where(i > 0, i - 1, nx - 1), j, none
    if d == "south":
        return i, np.maximum(j - 1, 0), j == 0
    # north
    top = j == ny - 1
    if topology == gt.TRIPOLAR:
        return np.where(top, nx - 1 - i, i), np.where(top, ny - 1, j + 1), none
    return i, np.minimum(j + 1, ny - 1), top


def makegridmetrics(*, areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices):
    """gridcellgeometry.jl:265-311.  Returns NT(area2D, v3D, thkcello, lon_vertices, lat_vertices,
    lon, lat, Z3D, zt, edge_length_2D, distance_to_edge_2D, distance_to_neighbour_2D, gridtopology)."""
    area_raw, area_props = data_and_props(areacello)
    vol_raw, vol_props = data_and_props(volcello)
    fills = [0.0]  # missing/nothing arrive as NaN here and stay NaN
    if "_FillValue" in area_props:
        fills.append(float(area_props["_FillValue"]))
    if "_FillValue" in vol_props:
        fills.append(float(vol_props["_FillValue"]))

    def clean(a):
        a = np.array(a, dtype=np.float64, order="F")
        for f in fills:  # replace() matches with isequal: -0.0 is not isequal to 0 and stays what it is
            a[(a == f) & (np.signbit(a) == np.signbit(f))] = np.nan
        return a

    v3D = clean(vol_raw)  # :275-276
    area2D = clean(area_raw)  # :279-280
    thkcello = np.asfortranarray(v3D / area2D[:, :, None])  # :283
    ZBOT3D = np.cumsum(thkcello, axis=2)  # :284
    Z3D = np.asfortranarray(ZBOT3D - 0.5 * thkcello)  # :285
    zt = np.array(data_and_props(lev)[0], dtype=np.float64)  # :286
    lat = np.array(data_and_props(lat)[0], dtype=np.float64, order="F")
    lon = np.array(data_and_props(lon)[0], dtype=np.float64, order="F")
    lon_vertices = np.array(data_and_props(lon_vertices)[0], dtype=np.float64, order="F")
    lat_vertices = np.array(data_and_props(lat_vertices)[0], dtype=np.float64, order="F")

    perm = vertexpermutation(lon_vertices, lat_vertices)  # :296
    lon_vertices = np.asfortranarray(lon_vertices[perm, :, :])
    lat_vertices = np.asfortranarray(lat_vertices[perm, :, :])

    topology = gt.getgridtopology(lon_vertices, lat_vertices, zt)  # :302
    nx, ny = lon.shape

    edge_length_2D, distance_to_edge_2D, distance_to_neighbour_2D = {}, {}, {}
    for d in DIRS:
        a, b = _VERTEXINDICES[d]
        lonA, latA = lon_vertices[a], lat_vertices[a]
        lonB, latB = lon_vertices[b], lat_vertices[b]
        edge_length_2D[d] = np.asfortranarray(haversine(lonA, latA, lonB, latB))  # :306, :217-222
        mlon, mlat = midpointonsphere(lonA, latA, lonB, latB)
        distance_to_edge_2D[d] = np.asfortranarray(haversine(lon, lat, mlon, mlat))  # :307, :240-247
        if topology == gt.UNKNOWN:
            raise RuntimeError("Unknown grid type")  # :308 calls j₊₁ etc. -> gridtopology.jl:111-116
        ii, jj, none = _neighbour2d(nx, ny, topology, d)
        dist = haversine(lon, lat, lon[ii, jj], lat[ii, jj])  # :308, :182-188
        dist = np.where(none, np.nan, dist)  # :189
        distance_to_neighbour_2D[d] = np.asfortranarray(dist)

    return NT(
        area2D=area2D, v3D=v3D, thkcello=thkcello, lon_vertices=lon_vertices, lat_vertices=lat_vertices,
        lon=lon, lat=lat, Z3D=Z3D, zt=zt, edge_length_2D=edge_length_2D,
        distance_to_edge_2D=distance_to_edge_2D, distance_to_neighbour_2D=distance_to_neighbour_2D,
        gridtopology=NT(kind=topology, name=gt.NAMES[topology], nx=nx, ny=ny, nz=len(zt)),
    )


# ---- Arakawa grid detection (host; decides which device path the velocity fields take) --------------
def getarakawagrid(u_lon, u_lat, v_lon, v_lat, gridmetrics):
    """src/gridcellgeometry.jl:50-95: where do the u and v points of cell (1,1) sit?  Returns ("A"|"B"|"C",
    u_pos, v_pos).  (The reference's relerr > 0.01 branch calls an undefined `warn`; a warning is issued here.)"""
    import warnings

    lon, lat = gridmetrics["lon"], gridmetrics["lat"]
    lv, tv = gridmetrics["lon_vertices"], gridmetrics["lat_vertices"]
    up = (float(np.asarray(u_lon)[0, 0]), float(np.asarray(u_lat)[0, 0]))
    vp = (float(np.asarray(v_lon)[0, 0]), float(np.asarray(v_lat)[0, 0]))
    SW, SE, NE, NW = [(float(lv[q, 0, 0]), float(tv[q, 0, 0])) for q in range(4)]

    def mid(A, B):
        lo, la = midpointonsphere(np.float64(A[0]), np.float64(A[1]), np.float64(B[0]), np.float64(B[1]))
        return (float(lo), float(la))

    cell = dict(C=(float(lon[0, 0]), float(lat[0, 0])), SW=SW, SE=SE, NE=NE, NW=NW, S=mid(SW, SE), N=mid(NE, NW),
                W=mid(SW, NW), E=mid(SE, NE))
    hv = lambda A, B: float(haversine(np.float64(A[0]), np.float64(A[1]), np.float64(B[0]), np.float64(B[1])))
    ud = {k: hv(P, up) for k, P in cell.items()}
    vd = {k: hv(P, vp) for k, P in cell.items()}
    u_pos = min(ud, key=ud.get)  # findmin: first minimum in field order
    v_pos = min(vd, key=vd.get)
    if u_pos == v_pos == "C":
        kind = "A"
    elif u_pos == v_pos and u_pos in ("NE", "NW", "SE", "SW"):
        kind = "B"
    elif u_pos in ("E", "W") and v_pos in ("N", "S"):
        kind = "C"
    else:
        raise RuntimeError("Unknown Arakawa grid type")
    perimeter = hv(SW, SE) + hv(SE, NE) + hv(NE, NW) + hv(NW, SW)
    relerr = (ud[u_pos] + vd[v_pos]) / perimeter
    if relerr > 0.01:
        warnings.warn(f"Relative error in grid positions in {kind}GridCell({u_pos},{v_pos}) is {relerr}")
    return kind, u_pos, v_pos
