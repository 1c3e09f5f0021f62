"""Load a golden fixture back into the structures the oracle / API take."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HD = ("west", "east", "south", "north")
PHI = ("east", "west", "north", "south", "top", "bottom")
GOLDEN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    gm = dict(v3D=np.asfortranarray(z["v3D"]), thkcello=np.asfortranarray(z["thkcello"]), area2D=np.asfortranarray(z["area2D"]),
              zt=z["zt"], gridtopology=dict(kind=int(z["topology"])),
              edge_length_2D={d: np.asfortranarray(z[f"edge_{d}"]) for d in HD},
              distance_to_neighbour_2D={d: np.asfortranarray(z[f"dist_{d}"]) for d in HD})
    rho = z["rho"]
    rho = float(rho) if rho.ndim == 0 else np.asfortranarray(rho)
    phi = {k: np.asfortranarray(z[f"phi_{k}"]) for k in PHI}

    def tm(upwind):
        p = "up" if upwind else "ce"
        return [(z[f"{p}_{q}_colptr"], z[f"{p}_{q}_rowval"], z[f"{p}_{q}_nzval"]) for q in range(5)]

    return dict(z=z, gm=gm, rho=rho, phi=phi, tm=tm, umo=np.asfortranarray(z["umo"]), vmo=np.asfortranarray(z["vmo"]),
                fill=float(z["fill"]), mlotst=np.asfortranarray(z["mlotst"]), kappa=tuple(float(x) for x in z["kappa"]))
