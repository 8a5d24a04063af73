#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of librtcore's kernels (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "raytracer_amd", "csrc")
extra = sys.argv[1:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(csrc, "rtcore.hip"), "-o", "/tmp/rtcore_res.o",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True, cwd=csrc).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'occ':>4s} {'LDS':>7s} {'scratch':>7s}")
for k, v in rows.items():
    print(f"{k[:58]:58s} {v.get('VGPRs', 0):5d} {v.get('AGPRs', 0):5d} {v.get('VGPRs Spill', 0):6d} {v.get('SGPRs Spill', 0):6d} "
          f"{v.get('Occupancy [waves/SIMD]', 0):4d} {v.get('LDS Size [bytes/block]', 0):7d} {v.get('ScratchSize [bytes/lane]', 0):7d}")
