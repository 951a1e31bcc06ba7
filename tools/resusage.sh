#!/bin/bash
# kernel resource usage (VGPRs, spills, LDS, occupancy) of every kernel of libw2dyn_hip.so as hipcc reports it
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c w2dynamics_amd/csrc/w2dyn_hip.hip -o /tmp/w2dyn_res.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys, re
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("spill", r"VGPR Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
import subprocess
for r in rows:
    name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void w2dyn::", "")
    print("%-42s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %4d occ %d lds %d" % (name[:42], r.get("vgpr", -1), r.get("agpr", 0), r.get("sgpr", -1), r.get("spill", 0), r.get("scratch", 0), r.get("occ", -1), r.get("lds", 0)))
'
