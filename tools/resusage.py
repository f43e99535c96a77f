#!/usr/bin/env python3
"""Print per-kernel register/LDS usage of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    if flt in name:
        dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dm = re.sub(r"\(anonymous namespace\)::", "", dm)[:90]
        print(f"{dm:92s} VGPR {d.get('VGPRs'):>4} AGPR {d.get('AGPRs'):>4} occ {d.get('Occupancy [waves/SIMD]'):>2} "
              f"spill {d.get('VGPRs Spill')} scratch {d.get('ScratchSize [bytes/lane]')} LDS {d.get('LDS Size [bytes/block]')}")
