#!/bin/bash
# One box: the link's duplex rate (tools/micro/pcie_duplex.py) beside the pipelined / two-phase host-pointer transportmatrix (tools/onepass_loop.py)
mkdir -p gpurun_out/r05
{
python tools/micro/pcie_duplex.py 2>/dev/null | tail -1
python tools/onepass_loop.py 2>/dev/null | python -c "
import sys,re,statistics
for line in sys.stdin:
    m=re.findall(r'\(([\d.]+), ([\d.]+), ([\d.]+)\)', line)[2:]
    print(line.split()[0], 'ff', round(statistics.median([float(x[0]) for x in m]),2), 'tm', round(statistics.median([float(x[1]) for x in m]),2))
"
} | tee -a gpurun_out/r05/duplex_vs_onepass.txt
