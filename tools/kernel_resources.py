#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
    python tools/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt
Scratch is bytes per lane; a kernel that spills shows up here before it shows up in a profile."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "ema_amd", "csrc", "k_*.hip")))
print(f"{'kernel':<58} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>15} {'LDS B':>7} {'waves/SIMD':>11} {'VGPR spills':>12}")
for f in files:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include",
                        f"-I{ROOT}/ema_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", f] + os.environ.get("EMA_RES_FLAGS", "").split(),
                       stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    cur = {}
    for line in p.stderr.split("\n"):
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k.startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
            if "rocprim" in name or "hipcub" in name:          # the library's sort / scan kernels under k_sa.hip: not ours to tune
                continue
            print(f"{os.path.basename(f) + ': ' + name:<58} {cur.get('VGPRs', '?'):>5} {cur.get('AGPRs', '?'):>5} {cur.get('TotalSGPRs', '?'):>5} "
                  f"{cur.get('ScratchSize [bytes/lane]', '?'):>15} {v:>7} {cur.get('Occupancy [waves/SIMD]', '?'):>11} {cur.get('VGPRs Spill', '?'):>12}")
