#!/usr/bin/env python
"""Per-kernel resource table (VGPRs, scratch, LDS, occupancy) of one HIP source: python tools/kres.py csrc/dense.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize','?'):>5s} "
          f"lds {r.get('LDS Size','?'):>6s} occ {r.get('Occupancy','?')}")
