#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel of libsqmc_gpu, from the compiler's own remarks (no GPU needed).
# usage: tools/kernel_resources.sh [extra hipcc flags] > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Iinclude "$@" \
  -Rpass-analysis=kernel-resource-usage -c sqmc_amd/csrc/sqmc_gpu.hip -o /dev/null 2>&1 | python3 -c '
import sys, re
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
print("%-70s %5s %5s %8s %5s %7s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS", "vspill"))
import subprocess
for r in rows:
    try: name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: name = r["name"]
    print("%-70s %5d %5d %8d %5d %7d %6d" % (name[:70], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1), r.get("VGPRs Spill", -1)))
'
