#!/usr/bin/env python3
"""Where does a wave of the fill pass spend its life?  Diagnostic build (-DOTMB_DBG_STAMPS: s_memtime stamps in SGPRs,
written by lane 0 of each wave to a buffer of their own; fences around every stamp forbid overlaps the real kernel has,
so read the SHARES, not the length).   gpurun -- python tools/stamps.py [workload] [extra -D flags]
Stamps: 0 entry | 1 Lwet back | 2 stencil loads back | 3 arithmetic done | 4 tile offsets known | 5 stores issued | 6 stores acked
         7 HW_ID | XCC_ID << 32 | 8, 9 s_memrealtime (100 MHz, one clock for the whole chip) at entry / end of the wave
Round 4: a DISPATCH TIMELINE from words 7-9 -- waves resident per XCD / per CU over the kernel, the share of the kernel spent
ramping up and draining."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
extra = sys.argv[2:]
path = os.path.join(b.LIBDIR, "libotmb_hip_stamps.so")
if not (os.environ.get("OTMB_STAMPS_PREBUILT") == "1" and os.path.exists(path)):  # (prebuilt on the CPU box: saves GPU minutes)
    path = b.build(force=True, extra=["-DOTMB_DBG_STAMPS", *extra], name="stamps")
import torch
import otmb_amd
from otmb_amd import capi, synthetic
from otmb_amd.device import DeviceAssembler

capi.use_library(path, lenient=True)
nx, ny, nz, lf = synthetic.PRESETS[wl]
if wl in ("quarterdeg", "tenthdeg"):  # generated on the device
    from otmb_amd import synthetic_device
    dg = synthetic_device.make_device_grid(wl, torch.device("cuda", 0), seed=20260501, rho="array")
    asm = synthetic_device.assembler_for(dg, 0)
    umo, vmo, fill_ = dg.umo, dg.vmo, dg.fill
else:
    g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    fill_ = 1e20
for _ in range(3):
    asm.step_async(umo, vmo, fill_)
asm.finish()
asm.ctx.timing_enable(True)
for _ in range(3):
    asm.step_async(umo, vmo, fill_)
asm.finish()
print({k: round(v[0] / v[1], 4) for k, v in asm.ctx.timing_collect().items()})
ntiles = (asm.N + 255) // 256
nw = ntiles * 4
NST = 10
buf = np.zeros(nw * NST, dtype=np.uint64)
fn = capi.lib().otmb_debug_stamps
fn.restype = C.c_int32
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = fn(asm.ctx.handle, buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(nw, NST).astype(np.int64)
ok = (t[:, :7] > 0).all(axis=1)
t = t[ok]
print(f"{wl}: {nw} waves, {ok.sum()} with all stamps")
names = ["entry->Lwet back", "->stencil loads back", "->arithmetic done", "->offsets known (scan+barrier)", "->stores issued (5 x stage+store)",
         "->stores acked"]
if "-DOTMB_DBG_STAMPS_ORDER" in extra:  # slot 6 = "tile id known" instead of "stores acked"
    e = t[:, 6] - t[:, 0]
    print(f"  entry->tile id known (kernel arguments, tile order) mean {e.mean():9.0f}  median {np.median(e):9.0f}  p90 {np.percentile(e, 90):9.0f}")
    t[:, 6] = t[:, 5]
d = np.diff(t[:, :7], axis=1)
life = t[:, 6] - t[:, 0]
for q, n in enumerate(names):
    print(f"  {n:40s} mean {d[:, q].mean():9.0f}  median {np.median(d[:, q]):9.0f}  p90 {np.percentile(d[:, q], 90):9.0f}  share {d[:, q].sum() / life.sum():6.1%}")
print(f"  wave life: mean {life.mean():.0f} median {np.median(life):.0f} cycles (s_memtime ticks)")
span = t[:, 6].max() - t[:, 0].min()
print(f"  kernel span {span} ticks; sum of wave lives / span = {life.sum() / span:.1f} waves in flight on average ({life.sum() / span / 256:.2f} per CU)")
# start-time profile: how many dispatch rounds
st = np.sort(t[:, 0] - t[:, 0].min())
print("  wave start times (ticks) deciles:", [int(x) for x in np.percentile(st, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])])
hw = t[:, 7] & 0xffffffff
xcc = (t[:, 7] >> 32) & 0xf
cu_key = ((hw >> 8) & 0xff) | (xcc << 8)  # CU id (4 bits), SH (1), SE (3) inside an XCD, and the XCD
print("  distinct (XCD, SE, SH, CU) values:", len(np.unique(cu_key)), " XCDs seen:", sorted(int(x) for x in np.unique(xcc)))

# ---- heavy tiles (tripolar seam row: generic column builder) against the rest ----
try:
    lwt = asm.lwet[: asm.N]
    P_ = nx * ny
    first = lwt[0::256][:ntiles].cpu().numpy() - 1
    last = lwt[torch.clamp(torch.arange(ntiles, device=lwt.device) * 256 + 255, max=asm.N - 1)].cpu().numpy() - 1
    k0_, j0_ = first // P_, (first % P_) // nx
    k1_, j1_ = last // P_, (last % P_) // nx
    heavy_tile = (j0_ == ny - 1) | (j1_ == ny - 1) | (k1_ > k0_)
    tile_of_wave = (np.arange(nw) // 4)[ok]
    hv = heavy_tile[tile_of_wave]
    lf = (t[:, 9] - t[:, 8]) / 100.0
    print(f"  heavy tiles (cells on the seam row): {int(heavy_tile.sum())} of {ntiles}; wave life heavy {lf[hv].mean():.2f} us (max {lf[hv].max():.2f}), "
          f"regular {lf[~hv].mean():.2f} us -> weight {lf[hv].mean() / lf[~hv].mean():.2f}")
    xh = ((t[:, 7] >> 32) & 0xf)[hv]
    print("  heavy waves per XCD:", {int(x): int((xh == x).sum()) for x in np.unique(xh)})
except Exception as e:  # diagnostics only
    print("  (heavy-tile statistics unavailable:", repr(e), ")")

# ---- what makes a tile expensive?  life of a tile (mean of its waves) against the span of grid cells its 256 wet cells cover ----
try:
    lfw = np.full(nw, np.nan); lfw[ok] = (t[:, 9] - t[:, 8]) / 100.0
    tile_life = np.nanmean(lfw.reshape(ntiles, 4), axis=1)
    span = (last - first + 1) / 256.0
    rows_touched = (k1_ * ny + j1_) - (k0_ * ny + j0_) + 1
    good = np.isfinite(tile_life) & ~heavy_tile
    qs = np.quantile(span[good], [0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0])
    print("  regular tiles: span of grid cells / 256 quantiles (0, 25, 50, 75, 90, 99, 100 %):", [round(float(x), 2) for x in qs])
    for lo, hi in zip(qs[:-1], qs[1:]):
        sel = good & (span >= lo) & (span <= hi)
        if sel.sum():
            print(f"    span {lo:6.2f}-{hi:6.2f}: {int(sel.sum()):7d} tiles, rows touched {rows_touched[sel].mean():5.1f}, tile life {tile_life[sel].mean():6.2f} us")
    A = np.stack([np.ones(good.sum()), span[good], rows_touched[good]], axis=1)
    coef, *_ = np.linalg.lstsq(A, tile_life[good], rcond=None)
    pred = A @ coef
    r2 = 1 - ((tile_life[good] - pred) ** 2).sum() / ((tile_life[good] - tile_life[good].mean()) ** 2).sum()
    print(f"  least squares: life = {coef[0]:.2f} + {coef[1]:.3f} x span/256 + {coef[2]:.3f} x rows  (R^2 = {r2:.3f})")
    # per XCD share under the current order: sum of tile lives
    xw = np.full(nw, -1); xw[ok] = (t[:, 7] >> 32) & 0xf
    xt = xw.reshape(ntiles, 4).max(axis=1)
    print("  sum of tile lives per XCD (us):", {int(x): round(float(np.nansum(tile_life[xt == x])), 0) for x in range(8)})
except Exception as e:
    print("  (tile cost statistics unavailable:", repr(e), ")")

# ---- dispatch timeline (s_memrealtime: 100 MHz, the same counter on every XCD) ----
t0, t1 = t[:, 8], t[:, 9]
k0, k1 = t0.min(), t1.max()
span = k1 - k0
print(f"  timeline: first wave entry -> last wave end = {span} ticks of 10 ns = {span / 100:.1f} us; mean wave life {np.mean(t1 - t0) / 100:.2f} us")
nb = 64
edges = np.linspace(k0, k1, nb + 1)
def resident(sel):
    # average number of waves resident in each time bin: sum of overlaps / bin width
    a, b = t0[sel], t1[sel]
    out = np.zeros(nb)
    for q in range(nb):
        lo, hi = edges[q], edges[q + 1]
        out[q] = np.clip(np.minimum(b, hi) - np.maximum(a, lo), 0, None).sum() / (hi - lo)
    return out
allr = resident(np.ones(len(t0), bool))
steady = np.median(allr[nb // 4: 3 * nb // 4])
print(f"  waves resident on the chip, {nb} bins over the kernel (steady state = median of the middle half = {steady:.0f} = {steady / 256:.2f} per CU):")
print("   ", " ".join(f"{x:.0f}" for x in allr))
ramp = np.argmax(allr >= 0.95 * steady)
tail = nb - 1 - np.argmax(allr[::-1] >= 0.95 * steady)
lost = ((steady - allr[:ramp]).clip(0).sum() + (steady - allr[tail + 1:]).clip(0).sum()) / (steady * nb)
print(f"  ramp: residency reaches 95 % of steady after {ramp}/{nb} bins ({ramp / nb:.1%} of the kernel); drain: falls below 95 % for the last {nb - 1 - tail}/{nb} bins ({(nb - 1 - tail) / nb:.1%})")
print(f"  wave-slots left empty in ramp + drain = {lost:.1%} of (steady residency x kernel span)")
for x in sorted(int(v) for v in np.unique(xcc)):
    sel = xcc == x
    r = resident(sel)
    print(f"  XCD {x}: {sel.sum():6d} waves, first entry +{(t0[sel].min() - k0) / 100:6.1f} us, last end {(k1 - t1[sel].max()) / 100:6.1f} us before the kernel's end, "
          f"steady {np.median(r[nb // 4: 3 * nb // 4]):.0f} waves, mean life {np.mean(t1[sel] - t0[sel]) / 100:.2f} us")
# per-CU idle tail: for each CU the time between its last wave's end and the kernel's end
cus = np.unique(cu_key)
tails = np.array([k1 - t1[cu_key == c].max() for c in cus]) / 100.0
heads = np.array([t0[cu_key == c].min() - k0 for c in cus]) / 100.0
print(f"  per CU ({len(cus)}): idle before its first wave mean {heads.mean():.2f} us (max {heads.max():.2f}); idle after its last wave mean {tails.mean():.2f} us "
      f"(median {np.median(tails):.2f}, max {tails.max():.2f}) = {tails.mean() / (span / 100):.1%} of the kernel")
# life of a wave against its start time (do late waves -- less contention -- live shorter?)
order = np.argsort(t0)
for q in range(8):
    sl = order[q * len(order) // 8:(q + 1) * len(order) // 8]
    print(f"  waves starting in octile {q}: mean life {np.mean(t1[sl] - t0[sl]) / 100:6.2f} us")
