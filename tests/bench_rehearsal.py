#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: bench.py's launcher and multi-rank plumbing on a box without GPUs.  Runs bench.main() with the CPU checker
backend of tests/ injected (gloo, oracle-backed slabs); the line it prints says REHEARSAL.  bench.py itself has no way to select
this backend -- no flag, no environment variable -- and imports nothing from tests/."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bench  # noqa: E402
from slab_checker_backend import OracleSlabBackend  # noqa: E402

if __name__ == "__main__":
    sys.exit(bench.main(slab_backend_factory=lambda local_rank: OracleSlabBackend(), cpu_rehearsal=True))
