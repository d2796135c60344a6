#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table from hipcc's resource-usage remarks:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c plume_kernels.hip -o /dev/null 2> remarks.txt
    python profiles/kernel_resources.py remarks.txt"""
import re
import sys

t = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", t)
keys = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("SGPRs", "SGPR"), (r"ScratchSize \[bytes/lane\]", "scratch B/lane"), (r"Occupancy \[waves/SIMD\]", "waves/SIMD"), (r"LDS Size \[bytes/block\]", "LDS B/block")]
print(f"{'kernel':58s} " + " ".join(f"{k[1]:>14s}" for k in keys))
for b in blocks[1:]:
    name = b.split()[0]
    vals = []
    for k, _ in keys:
        m = re.search(k + r": (\S+)", b)
        vals.append(m.group(1) if m else "?")
    print(f"{name[:58]:58s} " + " ".join(f"{v:>14s}" for v in vals))
