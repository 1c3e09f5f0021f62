#!/bin/bash
# isa_stats.sh <object.o> [kernel-name-regex] -- per-kernel instruction counts, VGPR / SGPR / LDS / scratch of the gfx950 code
# object inside a hipcc -c output (unbundles it first).  Evidence for "the hot kernel's ISA did not change" / instruction diets.
set -e
OBJ=$1; PAT=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ
$LLVM/clang-offload-bundler --type=o --unbundle --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.o
$LLVM/llvm-objdump -d --no-show-raw-insn $TMP/dev.o > $TMP/dis.txt
python3 - $TMP/dis.txt "$PAT" <<'PY'
import re, sys, subprocess
dis, pat = sys.argv[1], re.compile(sys.argv[2])
cur, counts = None, {}
for line in open(dis, errors="replace"):
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        cur = m.group(1); counts.setdefault(cur, {}); continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if cur and m:
        op = m.group(1); c = counts[cur]
        c["total"] = c.get("total", 0) + 1
        cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop", "s_barrier")) else
               "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else "smem" if op.startswith(("s_load", "s_buffer")) else "other")
        c[cls] = c.get(cls, 0) + 1
for k, c in counts.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if pat.search(name) and c.get("total"):
        print(f"{name[:110]:110s} " + " ".join(f"{x}={c.get(x, 0)}" for x in ("total", "valu", "salu", "vmem", "smem", "lds", "other")))
PY
$LLVM/llvm-readelf --notes $TMP/dev.o 2>/dev/null | python3 -c "
import re,sys
t=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\.name:|\Z)', t, re.S):
    blk=m.group(0)
    g=lambda k: (re.search(k+r':\s+(\d+)', blk) or [None,'?'])[1]
    if re.search(sys.argv[1], m.group(1)): print(m.group(1)[:70], 'vgpr', g(r'\.vgpr_count'), 'sgpr', g(r'\.sgpr_count'), 'lds', g(r'\.group_segment_fixed_size'), 'scratch', g(r'\.private_segment_fixed_size'))
" "$PAT" | sort -u
rm -rf $TMP
