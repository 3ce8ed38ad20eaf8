"""Bucket files to SAM text on the GPU, end to end (`ema align -s` / `-x`'s body, reference src/main.c:380-406,
src/align.c:213-628): ONE C-ABI call, ema_stream_sam (include/ema_stream.h) -- reader, engine, append stage, clouds / EM /
duplicates, formatter -- against the same chain made of oracle parts only (oracle reader, aligner, append stage, cloud stage,
formatter): byte-identical SAM bodies, 10x and haplotag, cloud numbers per bucket and running on (-x)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from common import small_ref
from ema_amd import engine as E
from ema_amd import ingest, sam, stream, synth
from test_clouds import make_bucket, oracle_selection
from test_sam_format import oracle_text

pytestmark = pytest.mark.gpu


def oracle_sam(prefix, path, names, so, haplotag=False):
    """The oracle's SAM body for one bucket file, and the cloud counter after it."""
    bucket = ingest.read_bucket(path, bc_len=12 if haplotag else 16, is_haplotag=haplotag)      # checked against the oracle's reader below
    want, _groups = O.read_special_fastq(path, 12 if haplotag else 16, haplotag)
    assert [w[0] for w in want] == bucket.bc.tolist() and all(bucket.read(2 * i) == w[2] for i, w in enumerate(want))
    idx, opt = O.Index(prefix), O.default_opt()
    cand_off, cands, cigar, recs, pair_off = [0], [], [], [], [0]
    for p in range(bucket.n_pairs):
        r1, r2 = bucket.read(2 * p), bucket.read(2 * p + 1)
        res = O.align_pair(idx, opt, r1, r2)
        base = [len(cands)]
        for m in range(2):
            for d in res[m]:
                c = np.zeros((), dtype=E.CAND_DTYPE)
                for f in O.REG_FIELDS:
                    c[f] = d[f]
                c["pos"], c["is_rev"], c["NM"], c["n_cigar"], c["cigar_off"] = d["pos"], d["is_rev"], d["NM"], len(d["cigar"]), len(cigar)
                cigar.extend(d["cigar"])
                cands.append(c)
            cand_off.append(len(cands))
            base.append(len(cands))
        for e in O.append_alignments(idx, opt, r1, r2):
            a = np.zeros((), dtype=E.ALN_REC_DTYPE)
            a["pair"], a["mate"], a["unique"], a["cand"] = p, e["mate"], e["unique"], base[e["mate"]] + e["cand"]
            a["clip"], a["clip_edit_dist"], a["mapq"], a["score_mapq"], a["score"] = e["clip"], e["clip_edit_dist"], e["mapq"], e["score_mapq"], e["score"]
            recs.append(a)
        pair_off.append(len(recs))
    batch = E.Batch(np.array(cand_off, np.uint64), np.array(cands, dtype=E.CAND_DTYPE) if cands else np.zeros(0, E.CAND_DTYPE),
                    np.array(cigar, np.uint32), np.zeros(2 * bucket.n_pairs, np.int32))
    rec = np.array(recs, dtype=E.ALN_REC_DTYPE) if recs else np.zeros(0, E.ALN_REC_DTYPE)
    arr, n, keep, rows, next_id = oracle_selection(bucket, batch, rec, np.array(pair_off, np.uint64), names)
    return oracle_text(arr, n, so), next_id, keep


def _run(tmp_path, kind, sizes, haplotag, continue_ids, **kw):
    prefix, ctg = small_ref(kind)
    paths = []
    for k, n in enumerate(sizes):
        d = tmp_path / f"b{k}"
        d.mkdir()
        _p, _c, _bucket = make_bucket(d, kind, n, 900 + k, 40, haplotag, sub_rate=0.015, chimeric=0.04, **kw)
        paths.append(str(d / "bucket"))
    _check(tmp_path, prefix, ctg, paths, sum(sizes), haplotag, continue_ids)


def _check(tmp_path, prefix, ctg, paths, n_pairs, haplotag, continue_ids):
    names = [f"chr{i + 1}".encode() for i in range(len(ctg))]
    eng = E.Engine(prefix)
    assert [c[0].encode() for c in eng.contigs()] == names
    out = str(tmp_path / "out.sam")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    bst, sst = stream.stream_sam(eng, paths, fd, rg_id=b"rg1", is_haplotag=haplotag, bc_len=12 if haplotag else 16, continue_cloud_ids=continue_ids)
    os.close(fd)
    eng.close()
    got = open(out, "rb").read()
    so = sam.default_opts()
    so.rg_id = b"rg1"
    if haplotag:
        so.is_haplotag, so.bc_len = 1, 12
    want, shift = b"", 0
    for k, path in enumerate(paths):
        text, n_clouds, _keep = oracle_sam(prefix, path, names, so, haplotag)
        if continue_ids and shift:      # -x: the cloud counter runs on; renumber the oracle's per-bucket MI values
            lines = []
            for line in text.split(b"\n"):
                at = line.find(b"\tMI:i:")
                if at >= 0:
                    end = line.find(b"\t", at + 1)
                    line = line[:at] + b"\tMI:i:%d" % (int(line[at + 6:end]) + shift) + line[end:]
                lines.append(line)
            text = b"\n".join(lines)
        want += text
        assert sst[k]["lines"] == text.count(b"\n") and sst[k]["clouds"] == n_clouds
        if continue_ids:
            shift += n_clouds
    assert got == want and got.count(b"\n") > 1.6 * n_pairs
    assert all(s["rc"] == 0 and s["capacity_flags"] == 0 for s in bst)


def test_three_10x_buckets_to_sam(tmp_path):
    _run(tmp_path, "repeats", [300, 260, 340], False, False)


def test_host_formatter_route(tmp_path, tuning):
    """ema_stream_sam's text comes from the device's formatter (csrc/k_sam.hip) by default -- every other test of this file; tuned off
    ("sam_device_format=0") the cloud stage emits its per-record structs and ema_sam_write formats them on the host's threads."""
    tuning(sam_device_format=0)
    _run(tmp_path, "repeats", [300, 260], False, False)


def test_a_bucket_the_device_reader_declines_shares_a_pass_with_ones_it_took(tmp_path, tuning):
    """Three small buckets make one pass.  The middle one has a NUL byte in an ignored seventh field: the device's reader hands such a file
    to the host reader (same bucket), so the pass mixes device-resident buckets with a host-resident one -- the stager fetches the
    others' reads back and takes the host path.  The SAM text must be what the host reader + host formatter write for the UNCHANGED
    files (a seventh field is never read; that route is the one the other tests of this file compare with the oracle)."""
    prefix, ctg = small_ref("repeats")
    paths = []
    for k, n in enumerate((220, 180, 200)):
        d = tmp_path / f"b{k}"
        d.mkdir()
        make_bucket(d, "repeats", n, 700 + k, 40, False, sub_rate=0.015, chimeric=0.04)
        paths.append(str(d / "bucket"))

    def run(ps, out):
        eng = E.Engine(prefix)
        fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        bst, _sst = stream.stream_sam(eng, ps, fd, rg_id=b"rg1")
        os.close(fd)
        eng.close()
        assert all(s["rc"] == 0 and s["capacity_flags"] == 0 for s in bst)
        return open(out, "rb").read()
    tuning(sam_device_format=0)
    want = run(paths, str(tmp_path / "host.sam"))
    tuning()
    assert run(paths, str(tmp_path / "dev.sam")) == want      # reader and formatter on the device
    lines = open(paths[1], "rb").read().split(b"\n")
    lines[3] += b" seventh\0field"
    odd = str(tmp_path / "odd_bucket")
    open(odd, "wb").write(b"\n".join(lines))
    _b, on_device = ingest.read_bucket_device(odd)
    assert not on_device
    assert run([paths[0], odd, paths[2]], str(tmp_path / "mixed.sam")) == want
    tuning(sam_device_reader=0)      # host reader, device formatter: the bucket's arrays go up with the records
    assert run(paths, str(tmp_path / "up.sam")) == want


def test_x_mode_cloud_numbers_run_on(tmp_path):
    _run(tmp_path, "two_contigs", [200, 240], False, True)


def test_haplotag_bucket_to_sam(tmp_path):
    _run(tmp_path, "two_contigs", [320], True, False)


def test_haplotag_250bp_bucket_to_sam(tmp_path):
    """BASELINE configs[4] as written: 2 x 250 bp haplotag reads (the documented MAX_READ_LEN deviation, SURVEY 0.4: the reference
    stops at 200 bases, the engine and the oracle take 255), bucket file -> SAM text."""
    _run(tmp_path, "two_contigs", [260], True, False, len1=250, len2=250, indel_rate=0.002)


def test_raw_fastq_through_count_and_preproc_to_sam(tmp_path):
    """The whole workflow through C-ABI calls only: raw interleaved FASTQ (mate 1 = barcode + 7 bases + read; one barcode in twelve
    with a wrong base) -> ema_count_fastq -> ema_preproc_fastq (three buckets) -> ema_stream_sam, against the oracle chain on the
    bucket files preproc wrote (preproc and count themselves are pinned to the reference's own code in tests/test_preproc.py)."""
    from ema_amd import count as ema_count, preproc as ema_preproc
    prefix, ctg = small_ref("two_contigs")
    pairs = synth.make_pairs(ctg, 700, seed=321, pairs_per_barcode=40, sub_rate=0.012, chimeric=0.03)
    rng = np.random.default_rng(4)
    wl = sorted({pairs.barcodes[i].tobytes().decode() for i in range(pairs.n)})
    lines = []
    for i in range(pairs.n):
        bc = pairs.barcodes[i].tobytes().decode()
        if i % 12 == 5:
            p = int(rng.integers(0, 16)); bc = bc[:p] + "ACGT"[("ACGT".index(bc[p]) + 1) % 4] + bc[p + 1:]
        m1 = bc + "ACGTACG" + pairs.read(2 * i).decode()
        r2 = pairs.read(2 * i + 1).decode()
        lines += [f"@s{i} 1:N:0", m1, "+", "F" * len(m1), f"@s{i} 2:N:0", r2, "+", "F" * len(r2)]
    fq = tmp_path / "raw.fastq"; fq.write_text("\n".join(lines) + "\n")
    wlp = tmp_path / "wl.txt"; wlp.write_text("\n".join(wl) + "\n")
    ema_count.count_fastq(str(wlp), str(fq), str(tmp_path / "c"))
    st = ema_preproc.preproc_fastq(str(wlp), [str(tmp_path / "c.ema-ncnt")], str(tmp_path / "b"), str(fq), n_threads=2, n_buckets=3)
    assert st["h1_corrected"] > 20 and st["pairs_written"] + st["pairs_nobc"] == pairs.n
    paths = [str(tmp_path / "b" / f"ema-bin-{k:03d}") for k in range(3)]
    _check(tmp_path, prefix, ctg, paths, st["pairs_written"], False, True)


@pytest.mark.parametrize("draws", ["one_stream_for_the_run", "a_stream_per_bucket"])
def test_hundred_buckets_streamed_with_the_density_optimiser(tmp_path, draws):
    """[r6] draws: the process's one rand() stream through all buckets in order (one cloud thread), or every bucket its own stream seeded
    seed + k (ema_cloud_opts.seed_private: each bucket an `ema align -s` process of its own; three cloud threads) -- the oracle chain
    reseeds per bucket then.
    BASELINE configs[2]'s shape: a hundred-odd barcode buckets through ONE ema_stream_sam call with `-d` on
    (ema_cloud_opts.density_opt; rand() seeded once, as the reference's first bad cloud does), the SAM text against the all-oracle
    chain -- oracle/ingest.c, the oracle's candidates and append stage, oracle/clouds.c WITH its restatement of src/split.c (pinned
    to the reference's own -d output by tests/test_golden_sam.py), oracle/sam.c -- byte for byte, bucket after bucket.  A reference
    with exact 3 kb copies 15 kb apart and 80 pairs per barcode, so that clouds holding a read in two places exist and the optimiser
    has moves to make (every bucket's text differs from its text without -d on this workload); the test insists that -d changed the output."""
    from ema_amd import clouds
    prefix, ctg = small_ref("exact_dups")
    sizes = [160 + 16 * (k % 5) for k in range(104)]
    paths = []
    for k, n in enumerate(sizes):
        d = tmp_path / f"b{k}"
        d.mkdir()
        make_bucket(d, "exact_dups", n, 2000 + k, 80, False, sub_rate=0.003)
        paths.append(str(d / "bucket"))
    names = [f"chr{i + 1}".encode() for i in range(len(ctg))]
    seed = 1500000000
    eng = E.Engine(prefix)
    out = str(tmp_path / "out.sam")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    per_bucket = draws == "a_stream_per_bucket"
    clouds.reseed(seed if not per_bucket else 99)
    bst, sst = stream.stream_sam(eng, paths, fd, rg_id=b"rg1", density_opt=True, density_seed=seed if per_bucket else None)
    os.close(fd)
    eng.close()
    got = open(out, "rb").read()
    so = sam.default_opts()
    so.rg_id = b"rg1"
    texts = {}
    for on in (True, False):
        O.clouds_density(on, seed=seed if on else None)
        try:
            texts[on] = []
            for k, path in enumerate(paths):
                if on and per_bucket:
                    O.clouds_density(True, seed=seed + k)
                texts[on].append(oracle_sam(prefix, path, names, so)[0])
        finally:
            O.clouds_density(False)
    want = b"".join(texts[True])
    at = 0
    for k, t in enumerate(texts[True]):      # bucket by bucket, so that a difference names its bucket
        assert got[at:at + len(t)] == t, f"bucket {k} differs"
        at += len(t)
    assert got == want and got.count(b"\n") > 1.6 * sum(sizes)
    assert want != b"".join(texts[False]), "-d changed nothing: the test has no teeth"
    assert all(s["rc"] == 0 and s["capacity_flags"] == 0 for s in bst) and len(bst) == 104
