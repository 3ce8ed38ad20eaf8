#!/usr/bin/env python3
"""Writes tests/golden/sam/: golden SAM text produced by the reference's OWN host code.

    python tests/golden/make_sam_vectors.py        (build container only: needs /root/reference)

`$TMPDIR/ema_ref/ema_refhost` (oracle/Makefile, target `refhost`) is every unmodified reference source -- src/align.c, bwabridge.c,
samdict.c, samrecord.c, split.c, techs.c, util.c, main.c, cpp/*.cc -- compiled where it lies against the B2 headers under
include/bwa_compat/ and linked to the nine libbwa symbols over the CPU oracle (oracle/bwaface.c).  This script runs
`ema align -s <bucket> -r <ref> -t 1` (and `-x`, `-p haplotag`, `-R`, `-i`) on a few tiny inputs and commits, per case,
the inputs (bucket files), the command line, and the whole output (header + body).  Everything after the nine engine calls in
that output -- barcode grouping, append_alignments' filters and scores, record order, clouds, the dictionary's collision rule,
EM, XA, duplicate marking, MAPQ, every SAM field -- is the reference's code, not a restatement of it.

What the vectors pin: the product's host stages (host_ingest / host_append / host_clouds / host_sam .cpp) and the oracle's
restatements (ingest.c, clouds.c, sam.c), both compared with these files byte for byte (tests/test_golden_sam.py on the CPU;
tests/test_gpu_golden_sam.py through ema_sam_header + ema_stream_sam on the GPU).  What they do NOT pin: the engine's arithmetic
(lh3/bwa is absent; the oracle restates it on both sides).

The references are committed as gzipped FASTA (the index is rebuilt by ema_index_build wherever the tests run).
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from ema_amd import synth, build_index      # noqa: E402

OUT = os.path.join(HERE, "sam")
REFHOST = os.path.join(os.environ.get("EMA_REF_OUT") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "ema_ref"), "ema_refhost")


def ref_plain():
    return synth.make_genome([110000, 50000], seed=41, short_rep=0.08, long_rep=0.04, segdup=0.02)


def ref_dups():
    """Exact 3 kb copies 15 kb apart (two candidates of one read in one cloud: bad clouds, XF:i:1) and 90 kb apart (candidates of
    equal likelihood in different clouds: XA entries)."""
    ctg = synth.make_genome([160000, 40000], seed=43, short_rep=0.0, long_rep=0.0, segdup=0.0)
    g = ctg[0]
    for k in range(6):
        src = 4000 + 24000 * k
        dst = (src + (15000 if k % 2 == 0 else 90000)) % (len(g) - 4000)
        g[dst:dst + 3000] = g[src:src + 3000]
    return ctg


def write_bucket(path, ctg, n_pairs, seed, per_bc, haplotag=False, dup_frac=0.1, junk_frac=0.03, len1=127, len2=150, **kw):
    pairs = synth.make_pairs(ctg, n_pairs, seed=seed, pairs_per_barcode=per_bc, len1=len1, len2=len2, **kw)
    rng = random.Random(seed)
    if haplotag:
        nrng = np.random.default_rng(seed)
        codes = {}
        for i in range(pairs.n):
            key = pairs.barcodes[i].tobytes()
            if key not in codes:
                a, c, b, d = (int(x) for x in nrng.integers(1, 97, 4))
                codes[key] = ("A%02dC%02dB%02dD%02d" % (a, c, b, d)).encode()
        bcs = [codes[pairs.barcodes[i].tobytes()] for i in range(pairs.n)]
    else:
        bcs = [pairs.barcodes[i].tobytes() for i in range(pairs.n)]
    lines = []
    for i in range(pairs.n):
        r1, r2 = pairs.read(2 * i), pairs.read(2 * i + 1)
        q1 = bytes(rng.choice(b"#,5:AFF") for _ in r1)
        q2 = bytes(rng.choice(b"#,5:AFF") for _ in r2)
        if rng.random() < junk_frac:      # a mate that aligns nowhere
            r2 = bytes(rng.choice(b"ACGT") for _ in range(len(r2)))
        lines.append(b" ".join([bcs[i], b"@s%d" % i, r1, q1, r2, q2]) + b"\n")
        if rng.random() < dup_frac:      # a PCR duplicate under another name
            lines.append(b" ".join([bcs[i], b"@dup%d" % i, r1, q1, r2, q2]) + b"\n")
    rng.shuffle(lines)
    with open(path, "wb") as f:
        f.write(b"".join(lines))
    return len(lines)


def write_fastq(d, ctg, n_pairs, seed, per_bc, style, haplotag=False, interleaved=False, **kw):
    """Barcode-sorted FASTQ as `ema align -1 [-2]` takes it: the barcode after the last ':' of the read name."""
    pairs = synth.make_pairs(ctg, n_pairs, seed=seed, pairs_per_barcode=per_bc, **kw)
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    codes = {}
    recs = []
    for i in range(pairs.n):
        key = pairs.barcodes[i].tobytes()
        if haplotag:
            if key not in codes:
                a, c, b, dd = (int(x) for x in nrng.integers(1, 97, 4))
                codes[key] = ("A%02dC%02dB%02dD%02d" % (a, c, b, dd)).encode()
            bc = codes[key]
        else:
            bc = key
        r1, r2 = pairs.read(2 * i), pairs.read(2 * i + 1)
        q1 = bytes(rng.choice(b"#,5:AFF") for _ in r1)
        q2 = bytes(rng.choice(b"#,5:AFF") for _ in r2)
        if style in ("tru", "cpt"):      # integer barcodes: extract_bc_truseq reads atoi() of the name, extract_bc_cptseq the digits after ":BC"
            if key not in codes:
                codes[key] = len(codes) + 1
            bc = b"%06d" % codes[key]
            names = (b"@%d_s%d" % (codes[key], i),) * 2 if style == "tru" else (b"@s%d:BC%d" % (i, codes[key]),) * 2
        else:
            names = (b"@s%d 1:N:0:" % i + bc, b"@s%d 2:N:0:" % i + bc) if style == "longranger" else (b"@s%d:" % i + bc, b"@s%d:" % i + bc)
        recs.append((bc, names, r1, q1, r2, q2))
    recs.sort(key=lambda r: r[0])      # equal barcodes side by side, as the reference expects
    m1 = b"".join(n[0] + b"\n" + r1 + b"\n+\n" + q1 + b"\n" for _bc, n, r1, q1, _r2, _q2 in recs)
    m2 = b"".join(n[1] + b"\n" + r2 + b"\n+\n" + q2 + b"\n" for _bc, n, _r1, _q1, r2, q2 in recs)
    if interleaved:
        both = b"".join(n[0] + b"\n" + r1 + b"\n+\n" + q1 + b"\n" + n[1] + b"\n" + r2 + b"\n+\n" + q2 + b"\n" for _bc, n, r1, q1, r2, q2 in recs)
        open(os.path.join(d, "reads.fq"), "wb").write(both)
        return ["reads.fq"]
    open(os.path.join(d, "r1.fq"), "wb").write(m1)
    open(os.path.join(d, "r2.fq"), "wb").write(m2)
    return ["r1.fq", "r2.fq"]


CASES = [
    # name, reference, [bucket specs], extra argv
    ("10x_full_em", "plain", [dict(n_pairs=150, seed=501, per_bc=50, sub_rate=0.015, indel_rate=0.002, chimeric=0.05)], []),
    ("10x_small_barcodes_rg", "plain", [dict(n_pairs=90, seed=502, per_bc=8, sub_rate=0.02, chimeric=0.08, junk_frac=0.15, dup_frac=0.25)],
     ["-R", "@RG\\tID:rgA\\tSM:sample1", "-i", "7"]),
    ("exact_dups_bad_clouds_xa", "dups", [dict(n_pairs=260, seed=503, per_bc=65, sub_rate=0.003, dup_frac=0.0, junk_frac=0.0)], []),
    ("haplotag", "plain", [dict(n_pairs=120, seed=504, per_bc=40, haplotag=True, sub_rate=0.01)], ["-p", "haplotag"]),
    ("x_two_buckets", "plain", [dict(n_pairs=80, seed=505, per_bc=40), dict(n_pairs=70, seed=506, per_bc=35, chimeric=0.1)], []),
    ("10x_150bp_mates", "dups", [dict(n_pairs=100, seed=507, per_bc=34, len1=150, len2=150, sub_rate=0.01, indel_rate=0.004)], []),
    # -d: the density optimiser on bad clouds; the reference seeds rand() from time(), which ema_refhost sees as 1500000000 (oracle/bwaface.c)
    ("density_opt_exact_dups", "dups", [dict(n_pairs=320, seed=508, per_bc=80, sub_rate=0.003, dup_frac=0.0, junk_frac=0.0)], ["-d"]),
    # -1 / -2: barcode-sorted FASTQ, the barcode in the read name (src/align.c:637-744, src/techs.c:5-69)
    ("fastq_two_files", "plain", [dict(fastq=True, n_pairs=110, seed=511, per_bc=37, style="plain", sub_rate=0.01, chimeric=0.05)], []),
    ("fastq_interleaved_longranger_names", "plain", [dict(fastq=True, n_pairs=90, seed=512, per_bc=45, style="longranger", interleaved=True)], []),
    ("fastq_interleaved_haplotag", "plain", [dict(fastq=True, n_pairs=80, seed=513, per_bc=40, style="plain", haplotag=True, interleaved=True)], ["-p", "haplotag"]),
    # -p tru / cpt: integer barcodes in the names, the many-clouds EM, other distance thresholds and error rates (src/techs.c:56-107)
    ("tru_fastq_many_clouds", "dups", [dict(fastq=True, n_pairs=240, seed=514, per_bc=60, style="tru", sub_rate=0.004)], ["-p", "tru"]),
    ("cpt_fastq_density_opt", "dups", [dict(fastq=True, n_pairs=260, seed=515, per_bc=65, style="cpt", sub_rate=0.004, interleaved=True)], ["-p", "cpt", "-d"]),
    ("density_opt_x_two_buckets", "dups", [dict(n_pairs=200, seed=509, per_bc=100, sub_rate=0.004, dup_frac=0.05), dict(n_pairs=180, seed=510, per_bc=60, sub_rate=0.004)], ["-d"]),
]


def main():
    if not os.path.exists(REFHOST):
        sys.exit(f"{REFHOST} is missing: run `make -C oracle refhost` (needs /root/reference)")
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    refs = {"plain": ref_plain(), "dups": ref_dups()}
    work = tempfile.mkdtemp(prefix="ema_samvec_")
    for name, ctg in refs.items():
        fa = os.path.join(work, f"{name}.fa")
        synth.write_fasta(fa, ctg)
        build_index(fa)
        with open(fa, "rb") as f, gzip.GzipFile(os.path.join(OUT, f"ref_{name}.fa.gz"), "wb", mtime=0) as z:
            z.write(f.read())
    manifest = []
    for name, ref, specs, extra in CASES:
        d = os.path.join(OUT, name)
        os.makedirs(d)
        buckets = []
        fastq = bool(specs[0].get("fastq"))
        for k, spec in enumerate(specs):
            if fastq:
                spec = {x: y for x, y in spec.items() if x != "fastq"}
                buckets = write_fastq(d, refs[ref], **spec)
                break
            b = f"bucket{k}"
            write_bucket(os.path.join(d, b), refs[ref], **spec)
            buckets.append(b)
        # run in a scratch directory that holds the reference under the neutral name `ref.fa` so that the @PG line is stable
        run = tempfile.mkdtemp(prefix="ema_samrun_")
        for ext in ("", ".fai", ".bwt", ".sa", ".pac", ".ann", ".amb"):
            os.symlink(os.path.join(work, f"{ref}.fa{ext}"), os.path.join(run, f"ref.fa{ext}"))
        for b in buckets:
            shutil.copy(os.path.join(d, b), os.path.join(run, b))
        if fastq:
            argv = ["ema", "align", "-1", buckets[0]] + (["-2", buckets[1]] if len(buckets) == 2 else []) + ["-r", "ref.fa", "-t", "1", "-o", "out.sam"] + extra
        elif len(buckets) == 1:
            argv = ["ema", "align", "-s", buckets[0], "-r", "ref.fa", "-t", "1", "-o", "out.sam"] + extra
        else:
            argv = ["ema", "align", "-r", "ref.fa", "-t", "1", "-o", "out.sam"] + extra + ["-x"] + buckets
        p = subprocess.run(argv, executable=REFHOST, cwd=run, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode != 0:
            sys.exit(f"{name}: ema_refhost failed ({p.returncode}): {p.stderr.decode()[-2000:]}")
        text = open(os.path.join(run, "out.sam"), "rb").read()
        with open(os.path.join(d, "expected.sam"), "wb") as f:
            f.write(text)
        body = [l for l in text.split(b"\n") if l and not l.startswith(b"@")]
        info = {"name": name, "ref": ref, "buckets": buckets, "argv": argv, "lines": len(body),
                "with_xa": sum(b"\tXA:Z:" in l for l in body), "bad_cloud": sum(b"\tXF:i:1" in l for l in body),
                "duplicates": sum(int(l.split(b"\t")[1]) & 1024 != 0 for l in body),
                "unmapped": sum(int(l.split(b"\t")[1]) & 4 != 0 for l in body)}
        manifest.append(info)
        print(info)
        shutil.rmtree(run)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_sam_vectors.py", "binary": "ema_refhost, built outside the repository by oracle/Makefile (reference src/*.c + cpp/*.cc, unmodified, over oracle/bwaface.c)",
                   "cases": manifest}, f, indent=1)
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
