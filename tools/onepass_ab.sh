#!/bin/bash
# Same box, alternating processes: index arrays over the link as Int32 (OTMB_XFER_NARROW=1, default) or as they are (=0); bench.py's
# end-to-end loop (tools/onepass_loop.py), medians of the steady slices.  Writes gpurun_out/r05/onepass_ab.txt
mkdir -p gpurun_out/r05
out=gpurun_out/r05/onepass_ab.txt
: > $out
for r in 1 2 3; do
  for n in 1 0; do
    echo "narrow=$n round $r" >> $out
    OTMB_XFER_NARROW=$n python tools/onepass_loop.py 2>/dev/null >> $out
  done
done
python - <<'PY'
import re, statistics
cur = None
for line in open("gpurun_out/r05/onepass_ab.txt"):
    if line.startswith("narrow="):
        cur = line.strip(); continue
    name = line.split(" ", 1)[0]
    vals = [float(x) for x in re.findall(r"\(([\d.]+), ([\d.]+), ([\d.]+)\)", line) for x in [x[1]]]
    vals = vals[2:]
    if vals:
        print(cur, name, "median %.2f min %.2f max %.2f" % (statistics.median(vals), min(vals), max(vals)))
PY
