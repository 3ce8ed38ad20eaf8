"""`ema preproc` behind include/ema_preproc.h against the REFERENCE's own implementation (cpp/correct.cc:271-633): the committed
golden vectors (tests/golden/preproc_vectors.json, written by the reference compiled into $TMPDIR/ema_ref/ref_preproc) everywhere, and
fresh random inputs through that binary where it exists (the build container).  Every bucket file byte for byte."""
import base64
import hashlib
import json
import os
import random
import subprocess

import pytest

import count_cases as K
from ema_amd import count as ema_count, preproc as ema_preproc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as _O
REF = os.path.join(_O.REF_OUT, "ref_preproc")


def well_formed(fq: str) -> str:
    """preproc's reference has undefined output for a quality line that is not as long as its read: the cases keep them equal."""
    out, lines = [], fq.split("\n")
    for i in range(0, len(lines) - 7, 8):
        rec = lines[i:i + 8]
        if len(rec[1]) != len(rec[3]):
            rec[3] = (rec[3] + "F" * len(rec[1]))[:len(rec[1])]
        out.extend(rec)
    return "\n".join(out) + ("\n" if out else "")


def run_product(tmp, wl_text, fq_text, haplotag, **kw):
    tmp.mkdir(exist_ok=True)
    wl = tmp / "wl.txt"; wl.write_text(wl_text)
    fq = tmp / "in.fastq"; fq.write_bytes(fq_text.encode("latin-1"))
    prefix = str(tmp / "cnt")
    ema_count.count_fastq(None if haplotag else str(wl), str(fq), prefix, kw.pop("max_map", 1 << 30), haplotag)
    out = tmp / "buckets"
    st = ema_preproc.preproc_fastq(None if haplotag else str(wl), [prefix + ".ema-ncnt"], str(out), str(fq), is_haplotag=haplotag, **kw)
    return {f: open(out / f, "rb").read() for f in sorted(os.listdir(out))}, st


def run_reference(tmp, wl_text, fq_text, haplotag, do_h2=False, buffer_size=10 << 20, do_bx_format=False, n_threads=1, n_buckets=500, max_map=1 << 30):
    tmp.mkdir(exist_ok=True)
    wl = tmp / "wl.txt"; wl.write_text(wl_text)
    prefix = str(tmp / "cnt")
    data = fq_text.encode("latin-1")
    subprocess.run([os.path.join(os.path.dirname(REF), "ref_count"), str(wl), prefix, str(max_map), str(int(haplotag))], input=data, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = tmp / "buckets"
    subprocess.run([REF, str(wl), str(out), str(int(do_h2)), str(buffer_size), str(int(do_bx_format)), str(n_threads), str(n_buckets),
                    str(int(haplotag)), prefix + ".ema-ncnt"], input=data, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return {f: open(out / f, "rb").read() for f in sorted(os.listdir(out))}


def digest(files):
    return {f: [len(b), hashlib.sha256(b).hexdigest()] for f, b in files.items()}


def test_golden_vectors_of_the_reference(tmp_path):
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "preproc_vectors.json")))
    assert len(doc["cases"]) >= 4
    for i, c in enumerate(doc["cases"]):
        got, st = run_product(tmp_path / f"c{i}", c["whitelist"], c["fastq"], bool(c["haplotag"]), **c["args"])
        assert digest(got) == c["expect"], f"{c['name']}: bucket files differ from the reference's"
        assert st["pairs_written"] + st["pairs_nobc"] + st["pairs_skipped"] > 0 or c["fastq"] == ""


def test_statistics_threads_and_errors(tmp_path):
    rng = random.Random(9)
    wl = K.whitelist(rng, 40)
    wl_text = "\n".join(wl) + "\n"
    fq = well_formed(K.tenx_fastq(12, wl, 600))
    a, st1 = run_product(tmp_path / "t1", wl_text, fq, False, n_threads=1, n_buckets=7, do_h2=True)
    b, st7 = run_product(tmp_path / "t7", wl_text, fq, False, n_threads=7, n_buckets=7, do_h2=True)
    assert a == b and st1 == st7      # the correction step's result does not depend on its threads
    assert st1["h1_corrected"] > 0 and st1["pairs_written"] > 0 and st1["pairs_nobc"] > 0 and st1["pairs_skipped"] > 0
    assert sorted(a) == ["ema-bin-%03d" % i for i in range(7)] + ["ema-nobc"]
    # a bucket line: BC NAME R1 Q1 R2 Q2 with mate 1 trimmed by 16 + 7 bases
    line = next(l for f in sorted(a) if f != "ema-nobc" for l in a[f].decode().split("\n") if l)
    bc, name, r1, q1, r2, q2 = line.split(" ")
    assert len(bc) == 16 and len(r1) == len(q1) and len(r2) == len(q2) and name.startswith("@r")
    bad = K.record("x 1", wl[0] + "ACGT" * 8, "F" * 40) + fq      # a quality line SHORTER than its read: the reference writes whatever its buffer held, an error here
    with pytest.raises(RuntimeError, match="quality line"):
        run_product(tmp_path / "bad", wl_text, bad, False, n_buckets=3)
    longer = K.record("x 1", wl[0] + "ACGT" * 8, "F" * 47 + "#,") + fq      # a LONGER one is cut to the read's length by the reference's next write: reproduced
    got, _ = run_product(tmp_path / "longer", wl_text, longer, False, n_buckets=3)
    assert any(b" @x " in v and b"#," not in v.split(b" @x ")[1].split(b"\n")[0] for v in got.values())
    if os.path.exists(REF):
        assert got == run_reference(tmp_path / "longer_ref", wl_text, longer, False, n_buckets=3)
    with pytest.raises(RuntimeError, match="not an ema-ncnt"):
        ema_preproc.preproc_fastq(str(tmp_path / "t1" / "wl.txt"), [str(tmp_path / "t1" / "wl.txt")], str(tmp_path / "o"), str(tmp_path / "t1" / "in.fastq"))


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference's preproc ($TMPDIR/ema_ref/ref_preproc) is built where /root/reference exists")
@pytest.mark.parametrize("seed,kw", [(41, dict(n_buckets=5)), (42, dict(n_buckets=11, do_h2=True, n_threads=3)), (43, dict(n_buckets=3, do_bx_format=True)),
                                     (44, dict(n_buckets=4, buffer_size=2000, max_map=72 * 30)), (45, dict(n_buckets=2, do_h2=True, do_bx_format=True))])
def test_random_inputs_against_the_reference_binary(tmp_path, seed, kw):
    rng = random.Random(seed)
    wl = K.whitelist(rng, rng.choice([6, 80, 400]))
    wl_text = "\n".join(wl) + "\n"
    fq = well_formed(K.tenx_fastq(seed, wl, rng.choice([3, 300, 1200])))
    got, _ = run_product(tmp_path / "p", wl_text, fq, False, **dict(kw))
    want = run_reference(tmp_path / "r", wl_text, fq, False, **dict(kw))
    assert sorted(got) == sorted(want)
    for f in want:
        assert got[f] == want[f], f"{f} differs"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference's preproc ($TMPDIR/ema_ref/ref_preproc) is built where /root/reference exists")
def test_haplotag_against_the_reference_binary(tmp_path):
    """Haplotag mode (96^4 whitelisted codes: half a minute per program run), including the reference's test of the BX tag against the
    previous pair's last line: the first pair of the stream is dropped."""
    fq = K.haplotag_fastq(51, 200)
    got, st = run_product(tmp_path / "p", "", fq, True, n_buckets=4)
    want = run_reference(tmp_path / "r", "", fq, True, n_buckets=4)
    assert got == want and st["pairs_skipped"] >= 1
    # `ema count` in this mode as well (tests/test_count.py leaves it to this test: the whitelist is built four times here already)
    assert open(tmp_path / "p" / "cnt.ema-ncnt", "rb").read() == open(tmp_path / "r" / "cnt.ema-ncnt", "rb").read()
    assert not os.path.exists(tmp_path / "p" / "cnt.ema-fcnt") and not os.path.exists(tmp_path / "r" / "cnt.ema-fcnt")


def test_buckets_feed_the_bucket_reader(tmp_path):
    """Raw interleaved FASTQ (mate 1 = 16 bp barcode + 7 bp + read) -> `ema count` -> `ema preproc` -> the bucket reader of
    include/ema_ingest.h: every pair comes back, under its barcode, with mate 1 trimmed -- the two ends of the workflow fit."""
    import numpy as np
    from ema_amd import ingest, synth
    ctg = synth.make_genome([150000], seed=3)
    pairs = synth.make_pairs(ctg, 900, seed=8)
    wl = sorted({pairs.barcodes[i].tobytes().decode() for i in range(pairs.n)})
    lines = []
    for i in range(pairs.n):
        r1, r2 = pairs.read(2 * i).decode(), pairs.read(2 * i + 1).decode()
        m1 = pairs.barcodes[i].tobytes().decode() + "ACGTACG" + r1
        lines += [f"@s{i} 1:N:0", m1, "+", "F" * len(m1), f"@s{i} 2:N:0", r2, "+", "F" * len(r2)]
    fq = tmp_path / "raw.fastq"; fq.write_text("\n".join(lines) + "\n")
    wlp = tmp_path / "wl.txt"; wlp.write_text("\n".join(wl) + "\n")
    ema_count.count_fastq(str(wlp), str(fq), str(tmp_path / "c"))
    st = ema_preproc.preproc_fastq(str(wlp), [str(tmp_path / "c.ema-ncnt")], str(tmp_path / "b"), str(fq), n_threads=3, n_buckets=4)
    assert st["pairs_written"] == pairs.n and st["pairs_nobc"] == 0 and st["no_change"] == pairs.n
    seen = {}
    for f in sorted(os.listdir(tmp_path / "b")):
        if f == "ema-nobc":
            assert os.path.getsize(tmp_path / "b" / f) == 0
            continue
        bk = ingest.read_bucket(str(tmp_path / "b" / f))
        for p in range(bk.n_pairs):
            seen[bk.ident(p).decode()] = (ingest.decode_barcode(int(bk.bc[p])), bk.read(2 * p), bk.read(2 * p + 1))
    assert len(seen) == pairs.n
    for i in range(pairs.n):
        bc, r1, r2 = seen[f"@s{i}"]
        assert bc == pairs.barcodes[i].tobytes() and r1 == pairs.read(2 * i) and r2 == pairs.read(2 * i + 1)
