"""The B2 drop-in made real: every UNMODIFIED reference source compiles against the headers this repository ships
under include/bwa_compat/ (with the reference's own flags, Makefile:5: -std=gnu99 -fopenmp, asserts live) and the objects link
against libema_bwaabi.so -- the nine libbwa symbols on the GPU engine -- into an `ema` that starts.  Build container only: the
test is skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "align.c")), reason="the reference tree is not present")


def test_unmodified_reference_compiles_and_links_against_the_b2_face(tmp_path):
    lib = os.path.join(ROOT, "ema_amd", "libema_bwaabi.so")
    assert os.path.exists(lib), "run make first"
    objs = []
    for f in sorted(os.listdir(os.path.join(REF, "src"))):
        if not f.endswith(".c"):
            continue
        o = str(tmp_path / (f[:-2] + ".o"))
        p = subprocess.run(["gcc", "-std=gnu99", "-march=x86-64", "-O1", "-fopenmp", "-fstrict-aliasing", "-Wall", "-Wextra", f"-I{REF}", f"-I{REF}/include",
                            f"-I{ROOT}/include/bwa_compat", "-c", os.path.join(REF, "src", f), "-o", o], stderr=subprocess.PIPE)
        assert p.returncode == 0, f"{f}: {p.stderr.decode()[-3000:]}"
        assert b"conflicting types" not in p.stderr and b"implicit declaration" not in p.stderr, p.stderr.decode()[-3000:]
        objs.append(o)
    assert len(objs) == 9
    for f in ("main", "count", "correct", "format"):
        o = str(tmp_path / f"cpp_{f}.o")
        p = subprocess.run(["g++", "-c", "-std=c++11", "-O1", "-march=x86-64", "-pthread", "-w", f"-I{REF}/cpp", os.path.join(REF, "cpp", f + ".cc"), "-o", o],
                           stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        objs.append(o)
    # what the objects still need from -lbwa is exactly the nine symbols, and the face exports exactly those
    need = set()
    for o in objs[:9]:
        for line in subprocess.run(["nm", "-u", o], stdout=subprocess.PIPE).stdout.decode().split("\n"):
            t = line.split()
            if t and t[-1] in ("bwa_idx_load", "bwa_idx_destroy", "mem_opt_init", "mem_align1_core", "mem_chain", "mem_matesw", "mem_reg2aln",
                               "bns_fetch_seq", "nst_nt4_table"):
                need.add(t[-1])
    assert len(need) == 9
    exe = str(tmp_path / "ema")
    p = subprocess.run(["g++", "-pthread", "-fopenmp", "-o", exe] + objs + [f"-L{ROOT}/ema_amd", "-lema_bwaabi", "-lema_engine",
                        f"-Wl,-rpath,{ROOT}/ema_amd", "-lm", "-lpthread"], stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert b"EMA version 0.6.2" in p.stdout + p.stderr
    p = subprocess.run([exe, "help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"align: choose best alignments" in p.stdout


def test_the_shipped_header_still_gives_its_own_users_the_chain_types(tmp_path):
    """tools/bwa_dump.c and bwaabi.cpp include ema_bwaabi.h directly (no EMA_BWAABI_REFERENCE_BUILD): mem_chain_v must be there."""
    src = tmp_path / "t.c"
    src.write_text('#include "ema_bwaabi.h"\nint main(void) { mem_chain_v v = {0, 0, 0}; mem_seed_t s; (void)s; return (int)v.n + (int)sizeof(mem_chain_t) - 40; }\n')
    p = subprocess.run(["gcc", "-std=gnu99", "-Wall", f"-I{ROOT}/include", "-c", str(src), "-o", str(tmp_path / "t.o")], stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
