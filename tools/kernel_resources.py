#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage (gfx950)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "redtail_amd/csrc/rt_capi.hip"
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build  # noqa: E402  (the product's device flags: what is measured is what ships)
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + build.DEVICE_FLAGS + ["-c", src, "-o",
                      "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void rt::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-58s %5s %5s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "spill", "LDS", "occ"))
for r in rows:
    print("%-58s %5s %5s %6s %7s %4s" % (r["name"][:58], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
                                         r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
