#!/bin/bash
# VGPR / spill report of one translation unit (device code only, ~10 s): tools/vgpr.sh tu_scalar [extra hipcc flags]
cd "$(dirname "$0")/../w2dynamics_amd/csrc" || exit 1
TU=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only $TU.hip -o /tmp/$TU.s -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys, re, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
    if "error" in line: print(line, end="")
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void w2dyn::", "")
    print("%-44s vgpr %3d agpr %3d sgpr %3d vspill %3d sspill %3d scratch %4d occ %d" % (name[:44], r.get("vgpr", -1), r.get("agpr", 0), r.get("sgpr", -1), r.get("vspill", 0), r.get("sspill", 0), r.get("scratch", 0), r.get("occ", -1)))
'
