"""The oracle's answer in tools/bwa_dump.c's format (test tooling): python tools/oracle_dump.py PREFIX PAIRS.txt [max_occ]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
idx, opt = O.Index(sys.argv[1]), O.default_opt()
if len(sys.argv) > 3:
    opt.max_occ = int(sys.argv[3])
for i, line in enumerate(l for l in open(sys.argv[2]) if len(l.split()) == 2):
    r1, r2 = (x.encode() for x in line.split())
    res = O.align_pair(idx, opt, r1, r2)
    print(f"P {i}")
    for m in range(2):
        for d in res[m]:
            cig = "".join(f"{c >> 4}{'MIDSH'[c & 15]}" for c in d["cigar"])
            print("H", m + 1, d["rb"], d["re"], d["qb"], d["qe"], d["rid"], d["score"], d["truesc"], d["sub"], d["csub"], d["w"], d["seedcov"],
                  d["secondary"], d["seedlen0"], "%.9g" % float(np.float32(d["frac_rep"])), d["pos"], d["is_rev"], d["NM"], cig)
