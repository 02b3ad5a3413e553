#!/bin/bash
# Registers / scratch / LDS per kernel of one source file:  tools/kernel_resources.sh hdn_amd/csrc/conv3x3.hip [filter]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -c -Iinclude -Ihdn_amd/csrc "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys, subprocess
rows, cur = [], None
for l in sys.stdin:
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", l)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name: continue
    print("%-120s vgpr %3d agpr %3d scratch %4d spill %3d lds %6d" % (name[:120], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("VGPRs Spill", -1), r.get("LDS Size", -1)))
' "$2"
