#!/bin/bash
# Per-kernel register / scratch / occupancy table of the current tree (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/resource_usage.sh [EXTRA="-D..."]
cd "$(dirname "$0")/../multi-gpu-path-tracer_amd/csrc" || exit 1
make -s asm "$@" 2>&1 | python3 -c '
import re, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("pt::dev::", "").replace("(pt::RenderParams)", "")
    if "rocprim" in name or "hipcub" in name: continue
    print("%-58s vgpr %3s sgpr %3s scratch %4s B/lane  vgpr-spill %3s sgpr-spill %3s  waves/SIMD %s" % (
        name[:58], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy")))
'
