"""Second, independent CPU restatement in pure Python.  TEST INFRASTRUCTURE ONLY.

A deliberately literal, slow transliteration of the reference's Julia loops (1-based
indices, `None` for `nothing`/`missing`, explicit push!) used on SMALL grids to pin the C
oracle: both were written separately from the reference text and must agree bit for bit.
Python floats are IEEE binary64 and CPython never contracts a*b+c, so every operation
rounds exactly as Julia's does.

PARITY UNPINNED (no Julia toolchain, no golden vectors in the reference): see
oracle/otmb_oracle.c.
"""
import math

import numpy as np

BIPOLAR, TRIPOLAR, UNKNOWN = 0, 1, 2
NaN = float("nan")


# ---- src/gridtopology.jl ---------------------------------------------------------------------
class Topo:
    def __init__(self, kind, nx, ny, nz):
        self.kind, self.nx, self.ny, self.nz = kind, nx, ny, nz

    def _chk(self):
        if self.kind == UNKNOWN:
            raise RuntimeError("Unknown grid type")  # :111-116

    def ip1(self, C):  # :57
        self._chk()
        i, j, k = C
        return (i + 1, j, k) if i < self.nx else (1, j, k)

    def im1(self, C):  # :58
        self._chk()
        i, j, k = C
        return (i - 1, j, k) if i > 1 else (self.nx, j, k)

    def jp1(self, C):  # :62 and :94
        self._chk()
        i, j, k = C
        if j < self.ny:
            return (i, j + 1, k)
        if self.kind == TRIPOLAR:
            return (self.nx - i + 1, self.ny, k)
        return None

    def jm1(self, C):  # :63
        self._chk()
        i, j, k = C
        return (i, j - 1, k) if j > 1 else None

    def kp1(self, C):  # :67
        self._chk()
        i, j, k = C
        return (i, j, k + 1) if k < self.nz else None

    def km1(self, C):  # :68
        self._chk()
        i, j, k = C
        return (i, j, k - 1) if k > 1 else None


def _at(A, C):
    return float(A[C[0] - 1, C[1] - 1, C[2] - 1])


# ---- makeindices: src/matrixbuilding.jl:10-24 ------------------------------------------------
def makeindices(v3D):
    nx, ny, nz = v3D.shape
    Lwet, Cwet = [], []
    Lwet3D = {}
    L = 0
    for k in range(1, nz + 1):
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                L += 1
                if not math.isnan(v3D[i - 1, j - 1, k - 1]):
                    Lwet.append(L)
                    Cwet.append((i, j, k))
                    Lwet3D[(i, j, k)] = len(Lwet)
    return dict(Lwet=Lwet, Cwet=Cwet, Lwet3D=Lwet3D, N=len(Lwet), shape=(nx, ny, nz))


# ---- nofluxboundaries! / facefluxes: src/velocities.jl:154-255 -------------------------------
def facefluxes(umo, vmo, wet3D, fill, topo):
    nx, ny, nz = umo.shape
    phi_i = np.array(umo, dtype=np.float64)  # Array{Float64} copies, :125-126
    phi_j = np.array(vmo, dtype=np.float64)
    wet = lambda C: bool(wet3D[C[0] - 1, C[1] - 1, C[2] - 1])
    cells = [(i, j, k) for k in range(1, nz + 1) for j in range(1, ny + 1) for i in range(1, nx + 1)]
    for C in cells:  # :161-175
        E = topo.ip1(C)
        N = topo.jp1(C)
        idx = (C[0] - 1, C[1] - 1, C[2] - 1)
        if not wet(C):
            phi_i[idx] = 0.0
            phi_j[idx] = 0.0
        if E is None or not wet(E):
            phi_i[idx] = 0.0
        if N is None or not wet(N):
            phi_j[idx] = 0.0
    allmiss = lambda A: all(math.isnan(x) or x == fill for x in A.ravel())
    assert not allmiss(phi_i) and not allmiss(phi_j)  # :199-200

    def repl(x):  # replace(., NaN => 0.0, FillValue => 0.0) uses isequal
        x = float(x)
        if math.isnan(x):
            return 0.0
        if x == fill and math.copysign(1.0, x) == math.copysign(1.0, fill):
            return 0.0
        return x

    east = np.vectorize(repl)(phi_i).astype(np.float64)
    north = np.vectorize(repl)(phi_j).astype(np.float64)
    west = np.zeros_like(east)
    south = np.zeros_like(north)
    for C in cells:
        idx = (C[0] - 1, C[1] - 1, C[2] - 1)
        W = topo.im1(C)
        if W is not None:
            west[idx] = _at(east, W)  # :206-211
        S = topo.jm1(C)
        if S is not None:
            south[idx] = _at(north, S)  # :219-224
    bottom = np.empty_like(east)
    top = np.empty_like(east)
    for k in range(nz, 0, -1):  # :236-243
        for j in range(ny):
            for i in range(nx):
                b = 0.0 if k == nz else float(top[i, j, k])
                bottom[i, j, k - 1] = b
                top[i, j, k - 1] = (((b + float(west[i, j, k - 1])) + float(south[i, j, k - 1])) - float(east[i, j, k - 1])) - float(north[i, j, k - 1])
    return dict(east=east, west=west, north=north, south=south, top=top, bottom=bottom)


# ---- COO generators: src/matrixbuilding.jl:193-299, 337-479 ---------------------------------
def _jlmax0(x):
    return x if math.isnan(x) else (x if x > 0 else 0.0)


def _jlmin0(x):
    return x if math.isnan(x) else (x if x < 0 else 0.0)


def advection_entries(phi, v3D, rho, idx, topo, upwind=True):
    Is, Js, Vs = [], [], []
    rho_at = (lambda C: float(rho)) if np.ndim(rho) == 0 else (lambda C: _at(rho, C))
    if any(math.isnan(rho_at(C)) for C in idx["Cwet"]):  # :233
        raise RuntimeError("ρ contains NaNs")
    Lwet3D = idx["Lwet3D"]

    def push(wi, Cj, ph, rho_i, v_i):  # pushTadvectionvalues! :193-204
        if Cj is None or Cj not in Lwet3D:
            raise RuntimeError("flux into land or outside the grid")
        wj = Lwet3D[Cj]
        r = (rho_i + rho_at(Cj)) / 2
        m_i = r * v_i
        m_j = r * _at(v3D, Cj)
        Is.append(wi); Js.append(wj); Vs.append(-ph / m_i)
        Is.append(wj); Js.append(wj); Vs.append(ph / m_j)

    for wi, C in enumerate(idx["Cwet"], start=1):  # :237
        v_i = _at(v3D, C)
        rho_i = rho_at(C)
        f = _jlmax0(_at(phi["west"], C)) if upwind else _at(phi["west"], C) / 2
        if f > 0 or f < 0:
            push(wi, topo.im1(C), f, rho_i, v_i)
        f = _jlmin0(_at(phi["east"], C)) if upwind else _at(phi["east"], C) / 2
        if f > 0 or f < 0:
            push(wi, topo.ip1(C), -f, rho_i, v_i)
        f = _jlmax0(_at(phi["south"], C)) if upwind else _at(phi["south"], C) / 2
        if f > 0 or f < 0:
            push(wi, topo.jm1(C), f, rho_i, v_i)
        f = _jlmin0(_at(phi["north"], C)) if upwind else _at(phi["north"], C) / 2
        if f > 0 or f < 0:
            push(wi, topo.jp1(C), -f, rho_i, v_i)
        f = _jlmax0(_at(phi["bottom"], C)) if upwind else _at(phi["bottom"], C) / 2
        if f > 0 or f < 0:
            push(wi, topo.kp1(C), f, rho_i, v_i)
        f = _jlmin0(_at(phi["top"], C)) if upwind else _at(phi["top"], C) / 2
        if C[2] > 1 and (f > 0 or f < 0):  # :290
            push(wi, topo.km1(C), -f, rho_i, v_i)
    return Is, Js, Vs


def _jlmin(a, b):
    if math.isnan(a) or math.isnan(b):
        return NaN
    return a if a < b else b


def _pushmix(Is, Js, Vs, wi, wj, kappa, a, d, V):  # :426-435
    Tval = kappa * a / (d * V)
    Is.append(wi); Js.append(wi); Vs.append(Tval)
    Is.append(wi); Js.append(wj); Vs.append(-Tval)


def hdiff_entries(gm, idx, topo, kappaH, OmegaH=None):
    Is, Js, Vs = [], [], []
    v3D, thk = gm["v3D"], gm["thkcello"]
    edge, dist = gm["edge_length_2D"], gm["distance_to_neighbour_2D"]
    Lwet3D = idx["Lwet3D"]
    ny = topo.ny
    inO = (lambda w: True) if OmegaH is None else (lambda w: bool(OmegaH[w - 1]))
    for wi, C in enumerate(idx["Cwet"], start=1):  # :348
        if not inO(wi):
            continue
        i, j, k = C
        V = _at(v3D, C)
        for d, shift, opp in (("west", topo.im1, "east"), ("east", topo.ip1, "west"),
                              ("south", topo.jm1, "north"), ("north", topo.jp1, "north" if j == ny else "south")):
            Cj = shift(C)
            if Cj is None or Cj not in Lwet3D:
                continue
            wj = Lwet3D[Cj]
            if not inO(wj):
                continue
            aij = _at(thk, C) * float(edge[d][i - 1, j - 1])  # verticalfacearea gridcellgeometry.jl:230-234
            aji = _at(thk, Cj) * float(edge[opp][Cj[0] - 1, Cj[1] - 1])
            a = _jlmin(aij, aji)
            dd = float(dist[d][i - 1, j - 1])
            _pushmix(Is, Js, Vs, wi, wj, kappaH, a, dd, V)
    return Is, Js, Vs


def vdiff_entries(gm, idx, topo, kappaV, Omega=None):
    Is, Js, Vs = [], [], []
    v3D, area2D, zt = gm["v3D"], gm["area2D"], gm["zt"]
    Lwet3D = idx["Lwet3D"]
    inO = (lambda w: True) if Omega is None else (lambda w: bool(Omega[w - 1]))
    for wi, C in enumerate(idx["Cwet"], start=1):  # :450
        if not inO(wi):
            continue
        i, j, k = C
        V = _at(v3D, C)
        a = float(area2D[i - 1, j - 1])
        for shift in (topo.kp1, topo.km1):  # bottom then top
            Cj = shift(C)
            if Cj is None or Cj not in Lwet3D:
                continue
            wj = Lwet3D[Cj]
            if not inO(wj):
                continue
            dd = abs(float(zt[k - 1]) - float(zt[Cj[2] - 1]))
            _pushmix(Is, Js, Vs, wi, wj, kappaV, a, dd, V)
    return Is, Js, Vs


def ml_mask(zt, mlotst, idx):  # :85
    return [bool(float(zt[k - 1]) < float(mlotst[i - 1, j - 1])) for (i, j, k) in idx["Cwet"]]


# ---- SparseArrays semantics, stated from their contract (NOT from sparse!'s loops) ----------
def sparse(Is, Js, Vs, m, n):
    """sparse(I,J,V,m,n): one stored entry per distinct (i,j), value = left fold of + over the
    triplets in input order, zeros kept, rows ascending within a column."""
    acc = {}
    for i, j, v in zip(Is, Js, Vs):
        if (j, i) in acc:
            acc[(j, i)] = acc[(j, i)] + v
        else:
            acc[(j, i)] = v
    keys = sorted(acc)
    counts = [0] * (n + 1)
    for (j, i) in keys:
        counts[j] += 1
    colptr = [1]
    for j in range(1, n + 1):
        colptr.append(colptr[-1] + counts[j])
    return (np.array(colptr, dtype=np.int64), np.array([i for (j, i) in keys], dtype=np.int64),
            np.array([acc[k] for k in keys], dtype=np.float64))


def spadd(A, B, n):
    """A + B = map(+, A, B): union pattern, missing operand is 0.0, exact-zero results dropped."""
    def cols(M):
        p, r, x = M
        return [{int(r[q - 1]): float(x[q - 1]) for q in range(int(p[j]), int(p[j + 1]))} for j in range(n)]
    ca, cb = cols(A), cols(B)
    colptr, rowval, nzval = [1], [], []
    for j in range(n):
        for i in sorted(set(ca[j]) | set(cb[j])):
            s = ca[j].get(i, 0.0) + cb[j].get(i, 0.0)
            if s != 0.0:
                rowval.append(i); nzval.append(s)
        colptr.append(len(rowval) + 1)
    return np.array(colptr, dtype=np.int64), np.array(rowval, dtype=np.int64), np.array(nzval, dtype=np.float64)


def transportmatrix(phi, gm, idx, topo, rho, mlotst, kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5, upwind=True):
    N = idx["N"]

    def chk(Vs, name):
        if any(math.isnan(v) for v in Vs):
            raise RuntimeError(f"{name} contains NaNs.")

    I, J, V = advection_entries(phi, gm["v3D"], rho, idx, topo, upwind); chk(V, "Tadv")
    Tadv = sparse(I, J, V, N, N)
    I, J, V = hdiff_entries(gm, idx, topo, kappaH); chk(V, "TκH")
    TkH = sparse(I, J, V, N, N)
    I, J, V = vdiff_entries(gm, idx, topo, kappaVML, ml_mask(gm["zt"], mlotst, idx)); chk(V, "TκVML")
    TkVML = sparse(I, J, V, N, N)
    I, J, V = vdiff_entries(gm, idx, topo, kappaVdeep, None); chk(V, "TκVdeep")
    TkVdeep = sparse(I, J, V, N, N)
    T = spadd(spadd(spadd(Tadv, TkH, N), TkVML, N), TkVdeep, N)  # :147 left fold
    return {"T": T, "Tadv": Tadv, "TκH": TkH, "TκVML": TkVML, "TκVdeep": TkVdeep}


# ---- Distances.haversine / makegridmetrics pieces: src/gridcellgeometry.jl ------------------
def haversine(A, B, radius=6371000.0):
    d2r = math.pi / 180
    dl = (B[0] - A[0]) * d2r            # Δλ = deg2rad(y[1] - x[1])
    p1, p2 = A[1] * d2r, B[1] * d2r      # φ₁ = deg2rad(x[2]), φ₂ = deg2rad(y[2])
    dp = p2 - p1                         # Δφ = φ₂ - φ₁ (Distances.jl 0.10 haversine.jl: converted first, subtracted after)
    a = math.sin(dp / 2) ** 2 + math.cos(p1) * math.cos(p2) * math.sin(dl / 2) ** 2
    return 2 * (radius * math.asin(min(math.sqrt(a), 1.0)))


def midpointonsphere(A, B):  # :249-255
    if abs(A[0] - B[0]) < 180:
        return ((A[0] + B[0]) / 2, (A[1] + B[1]) / 2)
    return ((A[0] + B[0]) / 2 + 180, (A[1] + B[1]) / 2 + 0)


def gridmetrics_2d(lon, lat, lonv, latv, kind):
    """edge_length_2D, distance_to_edge_2D, distance_to_neighbour_2D (:304-308) by scalar loops."""
    nx, ny = lon.shape
    t = Topo(kind, nx, ny, 1)
    vidx = {"south": (1, 2), "east": (2, 3), "north": (3, 4), "west": (1, 4)}
    shifts = {"south": t.jm1, "east": t.ip1, "north": t.jp1, "west": t.im1}
    el = {d: np.empty((nx, ny)) for d in vidx}
    de = {d: np.empty((nx, ny)) for d in vidx}
    dn = {d: np.empty((nx, ny)) for d in vidx}
    for d, (a, b) in vidx.items():
        for i in range(1, nx + 1):
            for j in range(1, ny + 1):
                A = (float(lonv[a - 1, i - 1, j - 1]), float(latv[a - 1, i - 1, j - 1]))
                B = (float(lonv[b - 1, i - 1, j - 1]), float(latv[b - 1, i - 1, j - 1]))
                Cc = (float(lon[i - 1, j - 1]), float(lat[i - 1, j - 1]))
                el[d][i - 1, j - 1] = haversine(A, B)
                de[d][i - 1, j - 1] = haversine(Cc, midpointonsphere(A, B))
                Jn = shifts[d]((i, j, 1))
                if Jn is None:
                    dn[d][i - 1, j - 1] = NaN
                else:
                    dn[d][i - 1, j - 1] = haversine(Cc, (float(lon[Jn[0] - 1, Jn[1] - 1]), float(lat[Jn[0] - 1, Jn[1] - 1])))
    return el, de, dn


def getarakawagrid(u_lon, u_lat, v_lon, v_lat, lon, lat, lonv, latv):
    """gridcellgeometry.jl:50-95, literally: the position (key of `cell`, in its field order) nearest to the u and v points
    of cell (1,1) -> ("A"|"B"|"C", u_pos, v_pos, relerr)."""
    up = (float(u_lon[0, 0]), float(u_lat[0, 0]))
    vp = (float(v_lon[0, 0]), float(v_lat[0, 0]))
    Cc = (float(lon[0, 0]), float(lat[0, 0]))
    SW, SE, NE, NW = [(float(lonv[q, 0, 0]), float(latv[q, 0, 0])) for q in range(4)]
    cell = dict(C=Cc, SW=SW, SE=SE, NE=NE, NW=NW, S=midpointonsphere(SW, SE), N=midpointonsphere(NE, NW),
                W=midpointonsphere(SW, NW), E=midpointonsphere(SE, NE))
    ud = {k: haversine(P, up) for k, P in cell.items()}
    vd = {k: haversine(P, vp) for k, P in cell.items()}
    u_pos = min(ud, key=ud.get)  # findmin: the first minimum
    v_pos = min(vd, key=vd.get)
    if u_pos == v_pos == "C":
        kind = "A"
    elif u_pos == v_pos and u_pos in ("NE", "NW", "SE", "SW"):
        kind = "B"
    elif u_pos in ("E", "W") and v_pos in ("N", "S"):
        kind = "C"
    else:
        raise RuntimeError("Unknown Arakawa grid type")
    per = haversine(SW, SE) + haversine(SE, NE) + haversine(NE, NW) + haversine(NW, SW)
    return kind, u_pos, v_pos, (ud[u_pos] + vd[v_pos]) / per


def bgrid_to_cgrid(u, v, fill, lonv, latv):
    """interpolateontodefaultCgrid(…, ::BGridCell), gridcellgeometry.jl:106-140, by scalar loops."""
    nx, ny, nz = u.shape
    rep = lambda x: 0.0 if (x == fill or (x != x and fill != fill)) else float(x)  # replace(u, _FillValue => 0.0)
    u2 = np.empty(u.shape, order="F")
    v2 = np.empty(u.shape, order="F")
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                us = 0.0 if j == 0 else rep(u[i, j - 1, k])
                vw = 0.0 if i == 0 else rep(v[i - 1, j, k])
                u2[i, j, k] = 0.5 * (rep(u[i, j, k]) + us)
                v2[i, j, k] = 0.5 * (rep(v[i, j, k]) + vw)
    pts = [np.empty((nx, ny), order="F") for _ in range(4)]
    for j in range(ny):
        for i in range(nx):
            SE = (float(lonv[1, i, j]), float(latv[1, i, j]))
            NE = (float(lonv[2, i, j]), float(latv[2, i, j]))
            NW = (float(lonv[3, i, j]), float(latv[3, i, j]))
            pts[0][i, j], pts[1][i, j] = midpointonsphere(NE, SE)  # zip(NE_points, SE_points), :131
            pts[2][i, j], pts[3][i, j] = midpointonsphere(NW, NE)  # zip(NW_points, NE_points), :132
    return u2, pts[0], pts[1], v2, pts[2], pts[3]


# ---- velocity2fluxes / fluxes2velocity: src/velocities.jl:10-39, :50-74 -----------------------------
def nanmean2(a, b):  # :89-93 (false * NaN == 0.0 in Julia)
    wa, wb = not math.isnan(a), not math.isnan(b)
    num = (a if wa else 0.0) + (b if wb else 0.0)
    den = int(wa) + int(wb)
    return num / den if den else NaN


def nanmin2(a, b):  # :108
    return b if math.isnan(a) else (a if math.isnan(b) else min(a, b))


def velocity_flux(a_i, a_j, rho, gm, topo, to_velocity=False):
    thk, edge = gm["thkcello"], gm["edge_length_2D"]
    nx, ny, nz = thk.shape
    oi, oj = np.zeros((nx, ny, nz)), np.zeros((nx, ny, nz))
    scalar = np.ndim(rho) == 0
    for k in range(1, nz + 1):
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                C = (i, j, k)
                E, N = topo.ip1(C), topo.jp1(C)
                if N is None:
                    raise RuntimeError("flux into land or outside the grid")  # x[nothing]
                mE = float(rho) if scalar else nanmean2(_at(rho, C), _at(rho, E))
                mN = float(rho) if scalar else nanmean2(_at(rho, C), _at(rho, N))
                tE, tN = nanmin2(_at(thk, C), _at(thk, E)), nanmin2(_at(thk, C), _at(thk, N))
                ee, en = float(edge["east"][i - 1, j - 1]), float(edge["north"][i - 1, j - 1])
                if not to_velocity:
                    oi[i - 1, j - 1, k - 1] = _at(a_i, C) * mE * tE * ee
                    oj[i - 1, j - 1, k - 1] = _at(a_j, C) * mN * tN * en
                else:
                    oi[i - 1, j - 1, k - 1] = _at(a_i, C) / (mE * tE * ee)
                    oj[i - 1, j - 1, k - 1] = _at(a_j, C) / (mN * tN * en)
    return oi, oj


# ---- bolus_GM_velocity: src/RediGM.jl:46-79, src/triads.jl:69-146, src/dyads.jl:29-78 (unpinned) -----------
def _gnan(A, I):  # getindexornan, gridtopology.jl:69
    return NaN if I is None else _at(A, I)


def _nanmean(vals):  # sum(w * v for ...) / sum(weights) with Bool weights
    s, n = None, 0
    for x in vals:
        w = not math.isnan(x)
        t = x if w else 0.0
        s = t if s is None else s + t
        n += int(w)
    return s / n if n else NaN


def _div(a, b):  # IEEE division (Python raises on /0)
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0:
            return NaN
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def triad_slope(chi, Z, dist2d, topo, C, shift):  # triads.jl:84-133
    N, S, E = topo.km1(C), topo.kp1(C), shift(C)
    if E is None:
        raise RuntimeError("k₋₁(nothing)")
    NE, SE = topo.km1(E), topo.kp1(E)
    vC, vN, vS, vE, vNE, vSE = _at(chi, C), _gnan(chi, N), _gnan(chi, S), _gnan(chi, E), _gnan(chi, NE), _gnan(chi, SE)
    dCN, dCS = abs(_gnan(Z, N) - _at(Z, C)), abs(_gnan(Z, S) - _at(Z, C))
    dCE = float(dist2d[C[0] - 1, C[1] - 1])
    dENE, dESE = abs(_gnan(Z, NE) - _gnan(Z, E)), abs(_gnan(Z, SE) - _gnan(Z, E))
    CN, CS, CE = _div(vN - vC, dCN), _div(vC - vS, dCS), _div(vE - vC, dCE)
    ENE, ESE = _div(vNE - vE, dENE), _div(vE - vSE, dESE)
    return _nanmean([_div(CE, CN), _div(CE, CS), _div(CE, ENE), _div(CE, ESE)])


def dyad_deriv(chi, Z, topo, C):  # dyads.jl:38-65
    N, S = topo.km1(C), topo.kp1(C)
    dCN, dCS = abs(_gnan(Z, N) - _at(Z, C)), abs(_gnan(Z, S) - _at(Z, C))
    return _nanmean([_div(_gnan(chi, N) - _at(chi, C), dCN), _div(_at(chi, C) - _gnan(chi, S), dCS)])


def bolus_gm_velocity(rho, gm, idx, topo, kappaGM=600.0, maxslope=0.01):
    Z, dn2 = gm["Z3D"], gm["distance_to_neighbour_2D"]
    shape = rho.shape
    Si, Sj = np.full(shape, NaN), np.full(shape, NaN)
    for C in idx["Cwet"]:
        Si[C[0] - 1, C[1] - 1, C[2] - 1] = triad_slope(rho, Z, dn2["east"], topo, C, topo.ip1)
        Sj[C[0] - 1, C[1] - 1, C[2] - 1] = triad_slope(rho, Z, dn2["north"], topo, C, topo.jp1)
    clamp = lambda x: maxslope if x > maxslope else (-maxslope if x < -maxslope else x)
    Ki, Kj = np.full(shape, NaN), np.full(shape, NaN)
    for q in np.ndindex(shape):
        si, sj = clamp(float(Si[q])), clamp(float(Sj[q]))
        r2 = si * si + sj * sj
        taper = 0.5 * (1 + math.tanh((0.004 - math.sqrt(r2)) / 0.001)) if r2 == r2 else NaN
        Ki[q] = kappaGM * (taper * si)
        Kj[q] = kappaGM * (taper * sj)
    u, v = np.full(shape, NaN), np.full(shape, NaN)
    for C in idx["Cwet"]:
        u[C[0] - 1, C[1] - 1, C[2] - 1] = dyad_deriv(Ki, Z, topo, C)
        v[C[0] - 1, C[1] - 1, C[2] - 1] = dyad_deriv(Kj, Z, topo, C)
    return u, v


# ---- lump_and_spray: src/extratools.jl:38-119 (literal transliteration, small grids only) -------------------------
class AsymmetricConnectivity(Exception):
    """Graphs.SimpleGraph(adjmx) throws ArgumentError("Adjacency / distance matrices must be symmetric")."""


def _connected_components(nv, adj):
    """Graphs.connected_components: label = smallest vertex of the component (vertices are visited in ascending
    order and a search labels everything reachable); components are listed in the order their label first appears
    while scanning the vertices, i.e. by ascending smallest vertex, members ascending."""
    label = [0] * (nv + 1)
    for u in range(1, nv + 1):
        if label[u]:
            continue
        label[u] = u
        queue = [u]
        while queue:
            src = queue.pop(0)
            for v in sorted(adj[src]):
                if not label[v]:
                    label[v] = u
                    queue.append(v)
    comps, seen = [], {}
    for v in range(1, nv + 1):
        if label[v] not in seen:
            seen[label[v]] = len(comps)
            comps.append([])
        comps[seen[label[v]]].append(v)
    return comps


def lump_and_spray(wet3D, vol, T, mask=None, di=2, dj=2, dk=1):
    """wet3D (nx,ny,nz) bool; vol (N,) volumes of the wet cells; T = (colptr, rowval, nzval) 1-based N x N;
    mask (nx,ny,nz) bool or None.  Returns LUMP (Nc x N), SPRAY (N x Nc) as (colptr, rowval, nzval) and vol_c."""
    wet3D = np.asarray(wet3D, dtype=bool)
    nx, ny, nz = wet3D.shape
    if mask is None:
        mask = np.ones(wet3D.shape, dtype=bool)  # trues(size(wet3D)), :38
    ex, ey, ez = nx + di - 1, ny + dj - 1, nz + dk - 1  # :41-43
    LUMPidx = np.zeros((ex, ey, ez), dtype=np.int64, order="F")

    def Lext(i, j, k):  # 1-based linear index in the extended grid, :47
        return i + ex * ((j - 1) + ey * (k - 1))

    wet3Dext = np.zeros((ex, ey, ez), dtype=bool)  # :48-49
    wet3Dext[:nx, :ny, :nz] = wet3D
    Lwet = [Lext(i + 1, j + 1, k + 1) for k in range(ez) for j in range(ey) for i in range(ex) if wet3Dext[i, j, k]]  # :50
    colptr, rowval, _ = T
    N = len(colptr) - 1
    conn = set()  # connectivitymatrix = sparse(Lwet[i], Lwet[j], true, ...), :46, :52
    for c in range(N):
        for q in range(int(colptr[c]), int(colptr[c + 1])):
            conn.add((Lwet[int(rowval[q - 1]) - 1], Lwet[c]))
    flat = LUMPidx.reshape(-1, order="F")  # a view: flat[L-1] is LUMPidx at extended linear index L
    wetext = wet3Dext.reshape(-1, order="F")
    c = 2  # :55
    for k in range(1, nz + 1):  # for 𝑖 in eachindex(C), :57: i fastest
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                if LUMPidx[i - 1, j - 1, k - 1] > 0 and mask[i - 1, j - 1, k - 1]:  # :61
                    continue
                if mask[i - 1, j - 1, k - 1]:  # :62
                    Li = [Lext(i + a, j + b, k + d) for d in range(dk) for b in range(dj) for a in range(di)]  # :64
                    for L in Li:  # :66-68
                        if not wetext[L - 1]:
                            flat[L - 1] = 1
                    wetidx = [L for L in Li if wetext[L - 1]]  # :70
                    nv = len(wetidx)
                    adj = {u: set() for u in range(1, nv + 1)}
                    for a in range(nv):  # view(connectivitymatrix, wetidx, wetidx) -> SimpleGraph, :71-72
                        for b in range(nv):
                            ab, ba = (wetidx[a], wetidx[b]) in conn, (wetidx[b], wetidx[a]) in conn
                            if ab != ba:
                                raise AsymmetricConnectivity()
                            if ab and a != b:
                                adj[a + 1].add(b + 1)
                    for comp in _connected_components(nv, adj):  # :74-77
                        for v in comp:
                            flat[wetidx[v - 1] - 1] = c
                        c += 1
                else:  # :78-81
                    LUMPidx[i - 1, j - 1, k - 1] = c
                    c += 1
    # LUMP = sparse(LUMPidx[C][:], 1:length(C), 1), :85
    rows = [int(LUMPidx[i, j, k]) for k in range(nz) for j in range(ny) for i in range(nx)]
    G = nx * ny * nz
    wet = wet3D.reshape(-1, order="F")
    m = max(rows)
    has_wet = [False] * (m + 1)  # wet_c = LUMP * wet .> 0, :88
    for L in range(G):
        if wet[L]:
            has_wet[rows[L]] = True
    newrow = {}
    for r in range(1, m + 1):
        if has_wet[r]:
            newrow[r] = len(newrow) + 1
    Nc = len(newrow)
    lump_rows = [newrow[rows[L]] for L in range(G) if wet[L]]  # LUMP[wet_c, wet], :91
    vol = np.asarray(vol, dtype=np.float64)
    vol_c = np.zeros(Nc)  # vol_c = LUMP * vol, :96: y[I] += 1 * vol[j] for j ascending
    for w, I in enumerate(lump_rows):
        vol_c[I - 1] = vol_c[I - 1] + 1 * vol[w]
    # LUMP = sparse(Diagonal(1 ./ vol_c)) * LUMP * sparse(Diagonal(vol)), :97 -- left to right
    lump_vals = [((1.0 / vol_c[I - 1]) * 1) * vol[w] for w, I in enumerate(lump_rows)]
    LUMP = (np.arange(1, N + 2, dtype=np.int64), np.array(lump_rows, dtype=np.int64), np.array(lump_vals, dtype=np.float64))
    # SPRAY = copy(LUMP'); SPRAY.nzval .= 1, :101-102
    members = [[] for _ in range(Nc)]
    for w, I in enumerate(lump_rows):
        members[I - 1].append(w + 1)
    scp, srv = [1], []
    for I in range(Nc):
        srv.extend(members[I])
        scp.append(len(srv) + 1)
    SPRAY = (np.array(scp, dtype=np.int64), np.array(srv, dtype=np.int64), np.ones(len(srv)))
    return LUMP, SPRAY, vol_c


def as2D(x, wet3D):  # :127-131
    out = np.full(wet3D.shape[:2], np.nan, order="F")
    out.T[np.asarray(wet3D)[:, :, 0].T] = x
    return out


def as3D(x, wet3D):  # :138-142
    out = np.full(wet3D.shape, np.nan, order="F")
    out.reshape(-1, order="F")[np.asarray(wet3D).reshape(-1, order="F")] = x
    return out
