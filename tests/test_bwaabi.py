"""The libbwa link surface (include/ema_bwaabi.h, libema_bwaabi.so): CPU-side checks -- the nine symbols the
reference links from -lbwa are exported, the struct layouts are the ABI's, bwa's defaults and tables are in place -- and, on
the GPU, the reference's own bridge (src/bwabridge.c:204-311) replayed through those symbols against the oracle."""
import ctypes as C
import subprocess

import numpy as np
import pytest

import bwaabi_lib as B

# `nm -u` of the reference's objects against -lbwa
NINE = ["bwa_idx_load", "bwa_idx_destroy", "mem_opt_init", "mem_align1_core", "mem_chain", "mem_matesw", "mem_reg2aln", "bns_fetch_seq",
        "nst_nt4_table"]


def test_the_nine_symbols_are_exported_and_nothing_else_bwa_shaped_is_missing():
    import os
    so = os.path.join(B.ROOT, "ema_amd", "libema_bwaabi.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], stdout=subprocess.PIPE, text=True, check=True).stdout
    defined = {line.split()[-1] for line in out.splitlines() if line.split()[-2] in "TDB"}
    for s in NINE:
        assert s in defined, s
    # the face is built on the engine's C ABI, not on the oracle
    und = subprocess.run(["nm", "-D", "--undefined-only", so], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "ema_engine_open" in und and "orc_" not in und


def test_struct_layouts_are_the_abi():
    L = B.lib()
    for which, size in B.SIZES.items():
        assert L.ema_bwaabi_sizeof(which) == size, which
    # mem_alnreg_t 88 bytes, mem_aln_t 56 with the bit-field word at offset 16 (src/bwabridge.c:159-168)
    assert C.sizeof(B.AlnReg) == 88 and C.sizeof(B.Aln) == 56 and B.Aln.flag2.offset == 16 and B.Aln.cigar.offset == 24
    assert B.AlnReg.frac_rep.offset == 76 and B.AlnReg.hash.offset == 80
    assert B.MemOpt.max_occ.offset == B.MemOpt.split_width.offset + 4 and B.MemOpt.max_mem_intv.offset == 48


def test_defaults_and_table():
    L = B.lib()
    o = L.mem_opt_init().contents
    assert (o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins, o.w, o.zdrop, o.T) == (1, 4, 6, 1, 6, 1, 100, 100, 30)
    assert (o.min_seed_len, o.split_width, o.max_occ, o.max_mem_intv, o.max_chain_gap) == (19, 10, 500, 20, 10000)
    assert (o.mapQ_coef_len, o.mapQ_coef_fac, o.max_matesw) == (50.0, 3, 50)
    assert list(o.mat) == [1, -4, -4, -4, -1, -4, 1, -4, -4, -1, -4, -4, 1, -4, -1, -4, -4, -4, 1, -1, -1, -1, -1, -1, -1]
    t = (C.c_ubyte * 256).in_dll(L, "nst_nt4_table")
    want = [4] * 256
    for ch, v in zip("ACGT", range(4)):
        want[ord(ch)] = want[ord(ch.lower())] = v
    want[ord("-")] = 5
    assert list(t) == want


@pytest.mark.gpu
def test_reference_bridge_replayed_through_the_nine_symbols():
    import oracle_lib as O
    from common import small_ref
    from ema_amd import synth
    prefix, ctg = small_ref("repeats")
    L = B.lib()
    idx = L.bwa_idx_load(prefix.encode(), 7)
    assert idx
    ix = idx.contents
    assert ix.bns.contents.n_seqs == len(ctg) and ix.bns.contents.l_pac == sum(len(c) for c in ctg)
    opt = L.mem_opt_init()
    opt.contents.max_occ = 3000          # reference src/align.c:185
    oidx, oopt = O.Index(prefix), O.default_opt()
    pairs = synth.make_pairs(ctg, 60, seed=91, sub_rate=0.05, indel_rate=0.003, chimeric=0.1)      # noisy: rescues happen
    n_hits = n_rescued = 0
    for p in range(pairs.n):
        got = B.bridge_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        ref = O.align_pair(oidx, oopt, pairs.read(2 * p), pairs.read(2 * p + 1))
        for m in range(2):
            g = [{k: v for k, v in d.items() if k != "mapq"} for d in got[m]]
            r = []
            for d in ref[m]:
                d = {k: d[k] for k in g[0]} if g else dict(d)
                d["frac_rep"] = float(np.float32(d["frac_rep"]))
                r.append(d)
            assert g == r, (p, m)
            n_hits += len(g)
            n_rescued += sum(1 for d in g if d["seedlen0"] == 0)
    assert n_hits > 2 * pairs.n * 0.8 and n_rescued > 0
    # bns_fetch_seq against the genome
    beg, end, rid = C.c_int64(1000), C.c_int64(1100), C.c_int(-1)
    s = L.bns_fetch_seq(ix.bns, ix.pac, C.byref(beg), 1050, C.byref(end), C.byref(rid))
    assert rid.value == 0 and bytes(s[i] for i in range(100)) == ctg[0][1000:1100].tobytes()
    L.bwa_idx_destroy(idx)


@pytest.mark.gpu
def test_options_travel_with_every_call():
    """bwa's default max_occ = 500 vs the reference's 3000 on a repeat-rich read: the face follows the mem_opt_t it is given."""
    import oracle_lib as O
    from common import small_ref
    from ema_amd import synth
    prefix, ctg = small_ref("repeats")
    L = B.lib()
    idx = L.bwa_idx_load(prefix.encode(), 7)
    ix = idx.contents
    oidx = O.Index(prefix)
    pairs = synth.make_pairs(ctg, 40, seed=92)
    for max_occ in (3000, 20, 3000):
        opt = L.mem_opt_init()
        opt.contents.max_occ = max_occ
        oopt = O.default_opt()
        oopt.max_occ = max_occ
        for r in range(0, 2 * pairs.n, 3):
            read = pairs.read(r)
            s = C.create_string_buffer(B.nt4(read), len(read))
            v = L.mem_align1_core(opt, ix.bwt, ix.bns, ix.pac, len(read), s, None)
            ref = O.align1(oidx, oopt, read)
            assert [(v.a[i].rb, v.a[i].re, v.a[i].score) for i in range(v.n)] == [(d["rb"], d["re"], d["score"]) for d in ref]
            B._libc.free(v.a)
    L.bwa_idx_destroy(idx)


@pytest.mark.gpu
def test_c_driver_against_the_face_equals_the_oracle_dump(tmp_path):
    """tools/bwa_dump.c (the reference's bridge in C over the nine symbols, the program tools/diff_vs_bwa.sh also builds against
    a real bwa) compiled against libema_bwaabi.so: its text equals tools/oracle_dump.py's on the same pairs."""
    import os
    import sys
    from common import small_ref
    from ema_amd import synth
    prefix, ctg = small_ref("two_contigs")
    pairs = synth.make_pairs(ctg, 50, seed=93, sub_rate=0.04, indel_rate=0.002)
    txt = tmp_path / "pairs.txt"
    txt.write_text("".join(f"{pairs.read(2 * p).decode()} {pairs.read(2 * p + 1).decode()}\n" for p in range(pairs.n)))
    exe = str(tmp_path / "bwa_dump")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(B.ROOT, "include"), "-o", exe, os.path.join(B.ROOT, "tools", "bwa_dump.c"),
                    "-L" + os.path.join(B.ROOT, "ema_amd"), "-lema_bwaabi", "-Wl,-rpath," + os.path.join(B.ROOT, "ema_amd")], check=True)
    got = subprocess.run([exe, prefix, str(txt)], stdout=subprocess.PIPE, text=True, check=True).stdout
    want = subprocess.run([sys.executable, os.path.join(B.ROOT, "tools", "oracle_dump.py"), prefix, str(txt)], stdout=subprocess.PIPE, text=True,
                          check=True).stdout
    assert got.count("\nH ") > 80 and got == want


def test_diff_tool_without_a_bwa_checkout_says_so():
    import os
    env = {k: v for k, v in os.environ.items() if k != "BWADIR"}
    p = subprocess.run(["bash", os.path.join(B.ROOT, "tools", "diff_vs_bwa.sh")], stdout=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 2 and "BWADIR not set" in p.stdout
