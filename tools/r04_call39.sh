#!/bin/bash
# Round-4 GPU call 39: sparse add with its columns' entries staged in LDS: the whole GPU suite, then the supporting-kernel table.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_39
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then echo "STOP tests rc=$rc"; tail -40 $OUT/pytest.log; exit 1; fi
timeout -k 10 400 python3 tools/secondary_time.py access1deg > $OUT/secondary_access1deg.jsonl 2> $OUT/err.log; rc=$?
python3 -c "
import json
for l in open('$OUT/secondary_access1deg.jsonl'):
    d=json.loads(l)
    if 'call' in d: print(d['call'], d['ms'], d['algorithmic_MB'], d['frac_of_8TBps'])"
echo "rc=$rc"
