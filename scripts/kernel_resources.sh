#!/bin/bash
# Registers, spills and occupancy of every kernel of kernels.hip as the compiler reports them (no GPU needed).
#   scripts/kernel_resources.sh [filter-regex] [extra hipcc flags...]
cd "$(dirname "$0")/../pathtracer_amd/csrc" || exit 1
filter="${1:-.}"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -I../../include "$@" -c -o /tmp/kres.o kernels.hip \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        name = name.replace("pt::(anonymous namespace)::", "").replace("void ", ""); name = re.sub(r"\(.*", "", name)
        cur = {"name": name}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print("%-42s %5s %5s %6s %6s %4s %7s %7s" % ("kernel", "VGPR", "SGPR", "spillV", "spillS", "occ", "LDS", "scratch"))
for r in rows:
    print("%-42s %5s %5s %6s %6s %4s %7s %7s" % (r["name"][:42], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
          r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"), r.get("ScratchSize [bytes/lane]")))
' | grep -E "kernel|$filter"
