#!/bin/bash
# Round-4 GPU call 38: sparse(): colptr over long runs of empty columns filled by workgroups (was one thread): parity, then the supporting-kernel table again.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_38
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_general_path.py tests/test_lump_and_spray.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then echo "STOP tests rc=$rc"; tail -30 $OUT/pytest.log; exit 1; fi
timeout -k 10 400 python3 tools/secondary_time.py access1deg > $OUT/secondary_access1deg.jsonl 2> $OUT/err.log; rc=$?
python3 -c "
import json
for l in open('$OUT/secondary_access1deg.jsonl'):
    d=json.loads(l)
    if 'call' in d: print(d['call'], d['ms'], d['frac_of_8TBps'])"
echo "rc=$rc"
