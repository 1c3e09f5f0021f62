#!/usr/bin/env python3
"""What the amdgpu hwmon / sysfs files say WHILE the fill pass runs (the idle readings of bench.py's box_probe do not tell fast boxes from slow
ones): sclk / mclk / power / temperature sampled from a thread during a loop of steps.  Diagnostic only.
    python3 tools/probe_hwmon.py [seconds of load]"""
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def files():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        for h in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input", "power1_cap"):
                p = os.path.join(h, name)
                if os.path.exists(p):
                    out[f"{os.path.basename(os.path.dirname(card))}:{name}"] = p
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out[f"{os.path.basename(os.path.dirname(card))}:{name}"] = p
    return out


def read(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    fs = files()
    # (the box shows the sysfs files of every GPU of its host: the card this process loads is the one whose gpu_busy_percent goes up)
    import torch

    from otmb_amd import synthetic_device  # noqa: F401  (package alias set up by the repo's conftest-free import path)
    dev = torch.device("cuda", 0)
    dg = synthetic_device.make_device_grid("access1deg", dev)
    asm = synthetic_device.assembler_for(dg)
    for _ in range(5):
        asm.step_async(dg.umo, dg.vmo, dg.fill)
    asm.finish()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), {k: read(p) for k, p in fs.items() if "input" in k or "average" in k or "busy" in k}))
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    per = []
    while time.perf_counter() - t0 < secs:
        asm.ctx.timing_enable(True)
        for _ in range(100):
            asm.step_async(dg.umo, dg.vmo, dg.fill)
        asm.finish()
        k = asm.ctx.timing_collect()
        asm.ctx.timing_enable(False)
        per.append((round(time.perf_counter() - t0, 3), round(k["tm_kernel<fill>"][0] / k["tm_kernel<fill>"][1], 5)))
    stop.set()
    th.join()
    # the series in runs: (start s, end s, n, mean ms) of consecutive samples within 2 % of the run's first one
    runs = []
    for t, v in per:
        if runs and abs(v - runs[-1][4]) <= 0.02 * runs[-1][4]:
            r = runs[-1]
            r[1], r[2], r[3] = t, r[2] + 1, r[3] + v
        else:
            runs.append([t, t, 1, v, v])
    print(json.dumps({"fill_ms_runs": [(r[0], r[1], r[2], round(r[3] / r[2], 5)) for r in runs], "samples": len(per),
                      "min": min(v for _, v in per), "max": max(v for _, v in per)}))
    busy = {}
    for _, d in samples:
        for k, v in d.items():
            if k.endswith(":gpu_busy_percent") and v.isdigit():
                busy[k.split(":")[0]] = busy.get(k.split(":")[0], 0) + int(v)
    mine = max(busy, key=busy.get) if busy else ""
    print(json.dumps({"card": mine, "partitions": {n: read(os.path.join("/sys/class/drm", mine, "device", n)) for n in ("current_memory_partition", "current_compute_partition")},
                      "power_cap_uW": read(fs.get(f"{mine}:power1_cap", "/nonexistent"))}))
    keys = sorted(k for k in (samples[0][1] if samples else []) if k.startswith(mine + ":"))
    for k in keys:
        vals = []
        for _, d in samples:
            try:
                vals.append(float(d[k]))
            except ValueError:
                pass
        if vals:
            print(json.dumps({"under_load": k, "n": len(vals), "min": min(vals), "max": max(vals), "first": vals[0], "last": vals[-1],
                              "by_time": [(round(t - t0, 2), d[k]) for t, d in samples[:: max(1, len(samples) // 24)]]}))


if __name__ == "__main__":
    main()
