"""The lean-tier kernels that share the chip in the steady state must fit 128 vector registers, so that a SIMD (512 registers per
lane) holds four wavefronts whichever kernels they belong to (wave slots are what the chip is short of; K2a at
162 registers cost the step 3 %).  Compiles the kernels' sources with hipcc's resource remarks -- no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")

# source file -> the kernels (demangled prefixes) that must stay at or below 128 registers with four waves per SIMD
LEAN = {
    "k_align_lane.hip": ["ema_k_align_simple_t<false>"],
    "k_final.hip": ["ema_k_final_t<0>"],
    "k_seed.hip": ["ema_k_seed_t<false, false>", "ema_k_seed_t<false, true>"],      # (product: without pass 3 [r5] / with it; the diagnostic build <true, true> carries its tick counters in registers)
    "k_seed_p3.hip": ["ema_k_seed_p3"],
}


def resources(src):
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include",
                        f"-I{ROOT}/ema_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull,
                        os.path.join(ROOT, "ema_amd", "csrc", src)], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    out, cur = {}, None
    for line in p.stderr.split("\n"):
        m = re.search(r"remark: +(Function Name|VGPRs|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "")
            out[cur] = {}
        elif cur:
            out[cur][m.group(1)] = int(m.group(2))
    return out


@pytest.mark.skipif(HIPCC is None, reason="hipcc not installed")
@pytest.mark.parametrize("src", sorted(LEAN))
def test_lean_kernels_fit_four_waves_per_simd(src):
    res = resources(src)
    for k in LEAN[src]:
        assert k in res, (k, sorted(res))
        assert res[k]["VGPRs"] <= 128, (k, res[k])
        assert res[k]["Occupancy [waves/SIMD]"] >= 4, (k, res[k])
