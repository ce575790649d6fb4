#!/usr/bin/env python3
"""Print per-kernel VGPR/SGPR/occupancy/spill/LDS from hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void j2k::", "")}
        rows.append(cur)
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = m.group(1)
for r in rows:
    print("%-60s vgpr=%-4s sgpr=%-4s occ=%-2s spill=%-3s scratch=%-5s lds=%s" % (r["name"][:60], r.get("vgpr"), r.get("sgpr"), r.get("occ"), r.get("spill"), r.get("scratch"), r.get("lds")))
