#!/bin/bash
# usage: tools/kernel_regs.sh csrc/file.hip [filter]   -- VGPRs / spills / scratch / occupancy per kernel (hipcc -Rpass-analysis)
cd "$(dirname "$0")/../multiplexed-image-annotator_amd"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c "$1" -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import sys, re
cur = {}
def flush():
    if cur: print("%-90s VGPR %3s spill %3s scratch %4s occ %s sgpr %s" % (cur.get("name","?")[:90], cur.get("VGPRs"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("SGPRs")))
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +Function Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m and "Function Name" in line or (m and "Name:" in line and "Function" not in line):
        flush(); cur = {"name": m.group(1)}; continue
    m = re.search(r":\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m: cur[m.group(1).strip()] = m.group(2)
flush()
' | { if [ -n "$2" ]; then grep "$2"; else cat; fi; }
