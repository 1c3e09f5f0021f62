"""CPU checker backend for otmb_amd.dist.SlabRunner (TEST INFRASTRUCTURE): computes a slab's columns
with the oracle on the slab's extended local grid so that the distributed orchestration (partition,
halo exchange, global wet ranks, flux chain, colptr bases, concatenation) can run under gloo without
a GPU.  The product backend is otmb_amd.dist.HipSlabBackend."""
import numpy as np
import torch

from oracle import oracle as orc

MATS = orc.MATS


class OracleSlabBackend:
    device = torch.device("cpu")

    def setup(self, s):
        s = dict(s)
        shape3 = (s["nx"], s["ny"], s["nz"])
        for key in ("v3D", "thkcello", "rho", "lwet3d"):  # SlabRunner.setup hands over flat tensors (Julia order)
            if torch.is_tensor(s[key]):
                s[key] = np.asfortranarray(s[key].cpu().numpy().reshape(shape3, order="F"))
        if torch.is_tensor(s["wet_own"]):
            s["wet_own"] = s["wet_own"].cpu().numpy().reshape((s["nx"], s["ny"], s["k_own1"] - s["k_own0"]), order="F") != 0
        self.s = s
        self.nx, self.ny, self.nz = s["nx"], s["ny"], s["nz"]
        self.P = self.nx * self.ny
        self.k0, self.k1 = s["k_own0"], s["k_own1"]
        thk = np.array(s["thkcello"], order="F")
        if self.k0 > 0:
            thk[:, :, 0] = 1.0  # halo thickness is never used for owned columns; keep the oracle's NaN check quiet
        if self.k1 < self.nz:
            thk[:, :, -1] = 1.0
        self.thk = thk
        self.phi = {k: np.zeros((self.nx, self.ny, self.nz), order="F") for k in orc.PHI_ORDER}

    def facefluxes(self, umo, vmo, fill, top_below):
        self.facefluxes_piece(umo, vmo, fill, top_below, 0, self.ny, True)
        return self.facefluxes_finish(top_below)

    def facefluxes_piece(self, umo, vmo, fill, top_below, j0, j1, first):
        """Rows [j0, j1) only, as the HIP backend's otmb_facefluxes_rows_dev: the oracle runs on the whole slab with whatever the plane
        buffer holds -- rows of the plane that have not arrived yet are garbage and only rows [j0, j1) of the result are kept (a cell's
        fluxes depend on its own column's plane value and on inputs of its neighbours) -- so a piece that is used before it has arrived
        shows up as a wrong matrix."""
        nown = self.k1 - self.k0
        u = umo.numpy().reshape(self.nx, self.ny, nown, order="F")
        v = vmo.numpy().reshape(self.nx, self.ny, nown, order="F")
        tb = None if top_below is None else top_below.numpy().copy()
        phi, uv = orc.facefluxes(u, v, self.s["wet_own"].astype(np.uint8), fill, self.s["topology"], top_below=tb, return_flags=True)
        if first:
            for k in orc.PHI_ORDER:
                self.phi[k][:] = 0.0
            self._top_first = np.full((self.nx, self.ny), np.nan, order="F")
        self.uv = uv  # (the validity flags concern umo / vmo of the whole slab: the same in every piece's call)
        for k in orc.PHI_ORDER:
            self.phi[k][:, j0:j1, self.k0:self.k1] = phi[k][:, j0:j1, :]
        self._top_first[:, j0:j1] = phi["top"][:, j0:j1, 0]
        return torch.from_numpy(np.ascontiguousarray(self._top_first.ravel(order="F")))

    def facefluxes_finish(self, top_below):
        if self.k0 > 0:
            self.phi["bottom"][:, :, 0] = self.phi["top"][:, :, self.k0]
        if top_below is not None:
            self.phi["top"][:, :, -1] = top_below.numpy().reshape(self.nx, self.ny, order="F")
        return torch.from_numpy(np.ascontiguousarray(self.phi["top"][:, :, self.k0].ravel(order="F")))

    def plan(self):
        s = self.s
        idx = orc.makeindices(s["v3D"])
        assert np.array_equal(idx["Lwet3D"] != 0, s["lwet3d"] != 0)
        gm = dict(v3D=s["v3D"], thkcello=self.thk, edge_length_2D=s["edge_length_2D"],
                  distance_to_neighbour_2D=s["distance_to_neighbour_2D"], area2D=s["area2D"], zt=s["zt"],
                  gridtopology=dict(kind=s["topology"]))
        kH, kML, kD = s["kappa"]
        tm = orc.transportmatrix(self.phi, gm, idx, s["rho"], s["mlotst"], kH, kML, kD, s["upwind"])
        glob = s["lwet3d"].ravel(order="F")[idx["Lwet"] - 1]  # local wet rank -> global wet rank
        lev = (idx["Lwet"] - 1) // self.P
        own = np.flatnonzero((lev >= self.k0) & (lev < self.k1))
        self.cols = {}
        self.nnz = []
        for m in MATS:
            cp, rv, nz = tm[m]
            a, b = (cp[own[0]] - 1, cp[own[-1] + 1] - 1) if len(own) else (0, 0)
            self.cols[m] = (cp[own[0]: own[-1] + 2] - cp[own[0]] if len(own) else np.array([0]), glob[rv[a:b] - 1], nz[a:b])
            self.nnz.append(int(b - a))
        assert len(own) == s["n_own"]
        if len(own):
            assert glob[own[0]] == s["wet_base"] + 1
        return self.nnz, self.uv

    def fill(self, nnz_base):
        self.out = {m: (self.cols[m][0] + int(nnz_base[k]) + 1, self.cols[m][1], self.cols[m][2]) for k, m in enumerate(MATS)}
        return self.out

    def assemble_async(self):
        """One asynchronous step: like the product backend, a failure is remembered per step and reported by result()."""
        if not hasattr(self, "_steps"):
            self._steps = []
        try:
            self.plan()
            self.fill(np.zeros(5, dtype=np.int64))
            self._steps.append((0, self.uv))
        except orc.OracleError as e:
            self._steps.append(({-1: 1, -2: 2, -3: 3, -4: 4, -5: 5, -6: 6}.get(e.code, 10), self.uv))

    def result(self):
        steps, self._steps = self._steps, []
        bad = [q for q, (st, _) in enumerate(steps) if st]
        return dict(nnz=[0] * 5 if bad else list(self.nnz), u=[int(uv[0]) for _, uv in steps], v=[int(uv[1]) for _, uv in steps],
                    status=steps[bad[0]][0] if bad else 0, step=bad[0] if bad else -1, message="")

    def shift_colptr(self, bases):
        self.out = {m: (self.out[m][0] + int(bases[k]), self.out[m][1], self.out[m][2]) for k, m in enumerate(MATS)}

    def shift_T_colptr(self, delta):
        cp, rv, nz = self.out[MATS[0]]
        self.out[MATS[0]] = (cp + int(delta), rv, nz)

    def sync(self):
        pass
