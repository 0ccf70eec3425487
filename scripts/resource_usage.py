#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed):
    python scripts/resource_usage.py [> profiles/rNN_resource_usage.txt]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "cutesv_amd", "csrc"), "resource-usage"] + sys.argv[1:], capture_output=True, text=True)
rows, cur = [], None
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass-analysis", line) or re.search(r"remark: +(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
try:
    import cxxfilt                                           # noqa: F401
    dem = cxxfilt.demangle
except Exception:                                           # noqa: BLE001
    def dem(x):
        try:
            return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", x], capture_output=True, text=True).stdout.strip() or x
        except OSError:
            return x
print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950 (make -C cutesv_amd/csrc resource-usage)")
print("# %-58s %5s %5s %7s %7s %8s %5s %7s" % ("kernel", "VGPR", "SGPR", "spillS", "spillV", "scratchB", "occ", "LDS B"))
for r in rows:
    n = dem(r["name"]).replace("csv::", "").replace("void ", "")
    n = re.sub(r"\(.*\)$", "", n)
    print("%-60s %5s %5s %7s %7s %8s %5s %7s" % (n[:60], r.get("VGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")), r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"),
                                                 r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
