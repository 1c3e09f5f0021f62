"""stdin: bench.py's JSON line; stdout: the same with {"tag", "round"} added and the bulky records dropped (tools/slab_probe.sh)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
keep = {k: d[k] for k in ("value", "ms_per_step", "kernels_ms", "roofline", "n_gpus") if k in d}
keep["tag"], keep["round"] = sys.argv[1], int(sys.argv[2])
print(json.dumps(keep))
