#!/bin/bash
# Builds librt_amd.so (extra compiler flags as arguments) and prints registers,
# scratch, LDS and occupancy of every kernel.   usage: tools/kres.sh [-DRT_...=..] [-o out.so]
cd "$(dirname "$0")/.."
out=raytracer_amd/librt_amd.so
args=()
while [ $# -gt 0 ]; do
  if [ "$1" = "-o" ]; then out=$2; shift 2; else args+=("$1"); shift; fi
done
hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -shared -Wno-unused-value "${args[@]}" \
  -Rpass-analysis=kernel-resource-usage -o "$out" raytracer_amd/csrc/rt_kernels.hip 2>&1 |
python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    if "error" in line or "warning" in line:
        sys.stderr.write(line)
    m = re.search(r"remark: .*?:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
import subprocess
for r in rows:
    try:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    print("%-34s vgpr %3s agpr %2s sgpr %3s scratch %4s lds %6s occ %s" % (name[:34], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
