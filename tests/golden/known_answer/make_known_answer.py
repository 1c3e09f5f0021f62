#!/usr/bin/env python3
"""A known-answer fixture small enough to check BY HAND: a 4 x 2 x 2 tripolar grid whose every number is a small dyadic
rational, so that each value of the five matrices is an exact fraction a reader can redo on paper.

This script does not use oracle/ nor the product: it walks the reference's loops in the plainest possible way (one
`emit` per `push!`), records for every triplet WHICH line of src/matrixbuilding.jl pushes it and the arithmetic that
gives its value, and writes
    known_answer.json   inputs, face fluxes, the triplets in emission order, the assembled matrices (machine readable)
    KNOWN_ANSWER.md     the same for a human reader
tests/test_known_answer.py then demands that oracle/otmb_oracle.c, oracle/pyref.py and the HIP library reproduce the
JSON bit for bit.  Re-run after editing:  python tests/golden/known_answer/make_known_answer.py
"""
import json
import os
from fractions import Fraction as F

HERE = os.path.dirname(os.path.abspath(__file__))
nx, ny, nz = 4, 2, 2
FILL = 1.0e20
cells = [(i, j, k) for k in range(1, nz + 1) for j in range(1, ny + 1) for i in range(1, nx + 1)]  # linear (column-major) order

# ---- inputs (1-based (i,j,k) like the reference) -------------------------------------------------------------------
land = {(4, 1, 2), (3, 2, 2)}                      # two land cells in the lower level: v3D = NaN there
v3D = {c: (None if c in land else F(8 if c[2] == 1 else 16)) for c in cells}
thk = {c: (None if c in land else F(2 if c[2] == 1 else 4)) for c in cells}
rho = F(1)                                          # scalar ρ: ρ̄ = (1 + 1)/2 = 1
edge = {d: {(i, j): F(1) for i in range(1, nx + 1) for j in range(1, ny + 1)} for d in ("west", "east", "south", "north")}
edge["east"][(1, 1)] = F(2)                         # makes min(aij, aji) pick the other side once
edge["north"][(2, 2)] = F(2)                        # the fold pairs north edges with north edges (oppdir, :407)
dist = {d: {(i, j): F(2) for i in range(1, nx + 1) for j in range(1, ny + 1)} for d in ("west", "east", "south", "north")}
dist["south"] = {ij: (None if ij[1] == 1 else F(2)) for ij in dist["south"]}   # j₋₁ of the first row is `nothing` -> NaN
dist["west"][(3, 1)] = F(4)
area = {(i, j): F(4) for i in range(1, nx + 1) for j in range(1, ny + 1)}
zt = {1: F(1), 2: F(3)}
mlotst = {(i, j): F(5) for i in range(1, nx + 1) for j in range(1, ny + 1)}  # both levels inside the mixed layer ...
mlotst[(1, 1)] = F(2)                                                         # ... except here (only level 1: no pair in Ω)
mlotst[(2, 2)] = None                                                         # missing -> Ω false
kH, kML, kDeep = F(2), F(4), F(1)
umo = {c: F(0) for c in cells}
vmo = {c: F(0) for c in cells}
umo.update({(1, 1, 1): F(6), (2, 1, 1): F(-2), (3, 1, 1): F(4), (4, 1, 1): F(-8), (1, 2, 1): F(2), (2, 2, 1): F(3), (3, 2, 1): F(-1),
            (4, 2, 1): F(5), (1, 1, 2): F(-4), (2, 1, 2): F(2), (3, 1, 2): F(1), (1, 2, 2): F(3), (2, 2, 2): F(7), (4, 2, 2): F(-6)})
vmo.update({(1, 1, 1): F(1), (2, 1, 1): F(-3), (3, 1, 1): F(2), (4, 1, 1): F(4), (1, 2, 1): F(-2), (2, 2, 1): F(6), (3, 2, 1): F(1),
            (4, 2, 1): F(-5), (1, 1, 2): F(2), (2, 1, 2): F(-1), (3, 1, 2): F(5), (1, 2, 2): F(4), (2, 2, 2): F(-3), (4, 2, 2): F(2)})
for c in land:
    umo[c] = vmo[c] = None                          # _FillValue on land


# ---- topology: gridtopology.jl:57-68, tripolar fold :94 -----------------------------------------------------------------
def ip1(c): i, j, k = c; return ((i + 1 if i < nx else 1), j, k)
def im1(c): i, j, k = c; return ((i - 1 if i > 1 else nx), j, k)
def jp1(c): i, j, k = c; return (i, j + 1, k) if j < ny else (nx - i + 1, j, k)
def jm1(c): i, j, k = c; return (i, j - 1, k) if j > 1 else None
def kp1(c): i, j, k = c; return (i, j, k + 1) if k < nz else None
def km1(c): i, j, k = c; return (i, j, k - 1) if k > 1 else None


wetcells = [c for c in cells if v3D[c] is not None]        # makeindices, matrixbuilding.jl:14-20
W = {c: n + 1 for n, c in enumerate(wetcells)}             # Lwet3D: wet rank, 1-based
N = len(wetcells)
iswet = lambda c: c is not None and c in W

# ---- nofluxboundaries! (velocities.jl:161-175) and facefluxes (:203-243) ----------------------------------------------
east, north = {}, {}
for c in cells:
    u, v = umo[c], vmo[c]
    if not iswet(c):
        u = v = F(0)
    if not iswet(ip1(c)):
        u = F(0)
    if not iswet(jp1(c)):
        v = F(0)
    east[c] = F(0) if u is None else u
    north[c] = F(0) if v is None else v
west = {c: east[im1(c)] for c in cells}
south = {c: (north[jm1(c)] if jm1(c) else F(0)) for c in cells}
top, bottom = {}, {}
for (i, j) in [(i, j) for j in range(1, ny + 1) for i in range(1, nx + 1)]:
    for k in range(nz, 0, -1):
        c = (i, j, k)
        bottom[c] = F(0) if k == nz else top[(i, j, k + 1)]
        top[c] = (((bottom[c] + west[c]) + south[c]) - east[c]) - north[c]   # :242
phi = dict(east=east, west=west, north=north, south=south, top=top, bottom=bottom)

trip = {"Tadv": [], "TκH": [], "TκVML": [], "TκVdeep": []}


def emit(op, row, col, val, line, why):
    trip[op].append(dict(row=row, col=col, val=val, line=line, why=why))


# ---- advection_operator_sparse_entries, upwind (matrixbuilding.jl:237-297) ---------------------------------------------
for c in wetcells:
    i_ = W[c]
    dirs = [("west", max(west[c], 0), im1(c), +1, 250), ("east", min(east[c], 0), ip1(c), -1, 259),
            ("south", max(south[c], 0), jm1(c), +1, 268), ("north", min(north[c], 0), jp1(c), -1, 277),
            ("bottom", max(bottom[c], 0), kp1(c), +1, 286), ("top", min(top[c], 0) if c[2] > 1 else F(0), km1(c), -1, 295)]
    for name, f, nb, sign, line in dirs:
        if f == 0:
            continue
        assert iswet(nb), (c, name)
        j_ = W[nb]
        ph = sign * f                                 # the value handed to pushTadvectionvalues! (:193-204)
        mi, mj = rho * v3D[c], rho * v3D[nb]          # ρ̄ = (ρ𝑖 + ρ𝑗)/2 = 1
        emit("Tadv", i_, j_, -ph / mi, line, f"cell {c} receives through its {name} face: ϕ={ph}; -ϕ/(ρ̄ v𝑖) = -{ph}/{mi}")
        emit("Tadv", j_, j_, ph / mj, line, f"the donor {nb} loses it: ϕ/(ρ̄ v𝑗) = {ph}/{mj}")

# ---- horizontal_diffusion_operator_sparse_entries (:348-415) -------------------------------------------------------------
opp = dict(west="east", east="west", south="north", north="south")
for c in wetcells:
    i, j, k = c
    for name, nb, line in (("west", im1(c), 367), ("east", ip1(c), 381), ("south", jm1(c), 395), ("north", jp1(c), 412)):
        if not iswet(nb):
            continue
        od = "north" if (name == "north" and j == ny) else opp[name]   # :407: through the fold both cells meet on their north edge
        aij = thk[c] * edge[name][(i, j)]
        aji = thk[nb] * edge[od][(nb[0], nb[1])]
        a = min(aij, aji)
        d = dist[name][(i, j)]
        val = (kH * a) / (d * v3D[c])
        why = f"{c}->{name} {nb}: a=min({aij},{aji})={a}, d={d}, κa/(dV)={kH * a}/{d * v3D[c]}"
        emit("TκH", W[c], W[c], val, line, why)
        emit("TκH", W[c], W[nb], -val, line, "same push, off-diagonal")

# ---- vertical_diffusion_operator_sparse_entries (:450-477) with Ω = mixed layer (:85) and Ω = everything (:109) ----------
inML = {c: (mlotst[(c[0], c[1])] is not None and zt[c[2]] < mlotst[(c[0], c[1])]) for c in wetcells}
for op, kappa, Om in (("TκVML", kML, inML), ("TκVdeep", kDeep, {c: True for c in wetcells})):
    for c in wetcells:
        if not Om[c]:
            continue
        for name, nb, line in (("bottom", kp1(c), 464), ("top", km1(c), 474)):
            if not iswet(nb) or not Om[nb]:
                continue
            d = abs(zt[c[2]] - zt[nb[2]])
            val = (kappa * area[(c[0], c[1])]) / (d * v3D[c])
            emit(op, W[c], W[c], val, line, f"{c}->{name} {nb}: κ a/(d V) = {kappa * area[(c[0], c[1])]}/{d * v3D[c]}")
            emit(op, W[c], W[nb], -val, line, "same push, off-diagonal")


# ---- sparse(I, J, V, N, N): duplicates summed in emission order, rows ascending, zeros kept; + drops exact zeros ----------
def sparse(ts):
    cols = {}
    for t in ts:
        cols.setdefault(t["col"], {}).setdefault(t["row"], []).append(t["val"])
    colptr, rowval, nzval = [1], [], []
    for c in range(1, N + 1):
        for r in sorted(cols.get(c, {})):
            rowval.append(r)
            nzval.append(sum(cols[c][r][1:], cols[c][r][0]))
        colptr.append(len(rowval) + 1)
    return colptr, rowval, nzval


def add(A, B):
    colptr, rowval, nzval = [1], [], []
    for c in range(N):
        a = {A[1][q]: A[2][q] for q in range(A[0][c] - 1, A[0][c + 1] - 1)}
        b = {B[1][q]: B[2][q] for q in range(B[0][c] - 1, B[0][c + 1] - 1)}
        for r in sorted(set(a) | set(b)):
            sm = a.get(r, F(0)) + b.get(r, F(0))
            if sm != 0:
                rowval.append(r)
                nzval.append(sm)
        colptr.append(len(rowval) + 1)
    return colptr, rowval, nzval


mats = {op: sparse(ts) for op, ts in trip.items()}
mats["T"] = add(add(add(mats["Tadv"], mats["TκH"]), mats["TκVML"]), mats["TκVdeep"])   # :147


def num(x):
    return None if x is None else float(x)


def arr3(d):  # (nx,ny,nz) nested lists [k][j][i] flattened in column-major order
    return [num(d[c]) for c in cells]


def arr2(d):
    return [num(d[(i, j)]) for j in range(1, ny + 1) for i in range(1, nx + 1)]


out = dict(
    shape=[nx, ny, nz], topology="tripolar", fill=FILL, rho=float(rho), kappa=[float(kH), float(kML), float(kDeep)], upwind=True,
    v3D=arr3(v3D), thkcello=arr3(thk), umo=[FILL if umo[c] is None else float(umo[c]) for c in cells],
    vmo=[FILL if vmo[c] is None else float(vmo[c]) for c in cells], area2D=arr2(area), zt=[float(zt[1]), float(zt[2])],
    mlotst=arr2(mlotst), edge_length_2D={d: arr2(edge[d]) for d in edge}, distance_to_neighbour_2D={d: arr2(dist[d]) for d in dist},
    wet_rank={f"{c[0]},{c[1]},{c[2]}": W[c] for c in wetcells}, N=N,
    phi={k: arr3(v) for k, v in phi.items()},
    triplets={op: [dict(row=t["row"], col=t["col"], val=float(t["val"]), frac=str(t["val"]), line=t["line"]) for t in ts] for op, ts in trip.items()},
    matrices={op: dict(colptr=m[0], rowval=m[1], nzval=[float(x) for x in m[2]], nzval_frac=[str(x) for x in m[2]]) for op, m in mats.items()},
)
json.dump(out, open(os.path.join(HERE, "known_answer.json"), "w"), indent=1, ensure_ascii=False)

with open(os.path.join(HERE, "KNOWN_ANSWER.md"), "w", encoding="utf-8") as f:
    f.write("# Known-answer fixture: 4 x 2 x 2 tripolar grid, every number a small fraction\n\n"
            "Made by `make_known_answer.py` (which uses neither `oracle/` nor the product). Cells are `(i,j,k)`, 1-based;\n"
            "`w` is the wet rank (row/column index of the matrices). `line` is the line of `src/matrixbuilding.jl` whose push emits the triplet.\n\n"
            f"Land: {sorted(land)} (v3D = NaN). ρ = 1 (scalar, so ρ̄ = 1), κH = {kH}, κVML = {kML}, κVdeep = {kDeep}, zt = [1, 3],\n"
            "v3D = 8 (k=1) / 16 (k=2), thkcello = 2 / 4, area2D = 4, edge lengths 1 (east of (1,1) and north of (2,2): 2),\n"
            "distances to neighbours 2 (west of (3,1): 4; south of the first row: NaN), mlotst = 5 ((1,1): 2, (2,2): missing).\n\n"
            "## Wet ranks\n\n| cell | w | cell | w |\n|---|---|---|---|\n")
    for q in range(0, N, 2):
        a, b = wetcells[q], wetcells[q + 1] if q + 1 < N else None
        f.write(f"| {a} | {W[a]} | {b if b else ''} | {W[b] if b else ''} |\n")
    f.write("\n## Face fluxes after nofluxboundaries! and the bottom-up recurrence (velocities.jl:161-175, :203-243)\n\n"
            "| cell | east | west | north | south | top | bottom |\n|---|---|---|---|---|---|---|\n")
    for c in cells:
        f.write(f"| {c} | {east[c]} | {west[c]} | {north[c]} | {south[c]} | {top[c]} | {bottom[c]} |\n")
    for op in ("Tadv", "TκH", "TκVML", "TκVdeep"):
        f.write(f"\n## {op}: triplets in emission order\n\n| # | row | col | value | line | arithmetic |\n|---|---|---|---|---|---|\n")
        for q, t in enumerate(trip[op], 1):
            f.write(f"| {q} | {t['row']} | {t['col']} | {t['val']} | :{t['line']} | {t['why']} |\n")
    for op in ("Tadv", "TκH", "TκVML", "TκVdeep", "T"):
        cp, rv, nzv = mats[op]
        f.write(f"\n## {op} as CSC ({N} x {N}, nnz = {len(rv)})\n\n| col | rows : values |\n|---|---|\n")
        for c in range(N):
            ent = ", ".join(f"{rv[q]}: {nzv[q]}" for q in range(cp[c] - 1, cp[c + 1] - 1))
            f.write(f"| {c + 1} | {ent} |\n")
print("N =", N, {op: len(m[1]) for op, m in mats.items()})
