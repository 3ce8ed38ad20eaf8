"""sha256 over the device code the PMC passes describe (comments and blank lines left out) (ema_amd/csrc/k_*.hip, dev_*.hpp, dev_types.h, opts.h): stored beside a
committed counter table (profiles/<name>.csv.srchash) so that bench.py can tell when the kernels have changed since
(`traffic_source_stale`).   python tools/kernel_hash.py [> profiles/r04_pmc_grch38scale.csv.srchash]"""
import glob, hashlib, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_hash(root=ROOT):
    c = os.path.join(root, "ema_amd", "csrc")
    files = sorted(glob.glob(os.path.join(c, "k_*.hip")) + glob.glob(os.path.join(c, "dev_*.hpp")) + [os.path.join(c, "dev_types.h"), os.path.join(c, "opts.h")])
    h = hashlib.sha256()
    for f in files:      # the CODE: comments and blank lines do not count (a reworded comment must not mark the counters stale)
        src = open(f, encoding="utf-8", errors="replace").read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        lines = [re.sub(r"//.*$", "", ln).rstrip() for ln in src.split("\n")]
        h.update(os.path.basename(f).encode() + b"\0" + "\n".join(ln for ln in lines if ln.strip()).encode() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_hash())
