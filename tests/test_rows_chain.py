"""The three C-ABI headers fit together: a bucket file goes through the reader (ema_ingest.h), the candidates of its
pairs -- the oracle's here, standing in for the engine's, which the GPU suite shows to be identical -- through the
append stage (ema_engine.h), and the best surviving record of every mate through the SAM formatter (ema_sam.h).  The
cloud/EM stage between the last two is the reference's and is not modelled: gamma = 1, one cloud per barcode group.
Checked here: the plumbing (offsets, orders, ownership) and the SAM text against the data it was made from."""
import ctypes as C
import random

import numpy as np

import oracle_lib as O
from common import small_ref
from ema_amd import engine as E
from ema_amd import ingest, sam, synth
from test_append_alignments import batch_from_oracle


def test_bucket_to_sam_lines(tmp_path):
    prefix, ctg = small_ref("two_contigs")
    pairs = synth.make_pairs(ctg, 60, seed=71, sub_rate=0.01, pairs_per_barcode=5)
    path = str(tmp_path / "bucket.fq")
    synth.write_special_fastq(path, pairs)
    lines = open(path, "rb").read().splitlines(keepends=True)
    random.Random(1).shuffle(lines)
    open(path, "wb").write(b"".join(lines))
    bucket = ingest.read_bucket(path)
    ordered = synth.Pairs(bucket.bases, bucket.off)
    batch = batch_from_oracle(prefix, ordered)
    rec, pair_off = E.append_alignments(batch, bucket.off)
    names = [b"chr1", b"chr2"]
    ctg_off = np.concatenate([[0], np.cumsum([len(c) for c in ctg])])
    keep, recs = [], {}
    for p in range(bucket.n_pairs):
        group = int(np.searchsorted(bucket.group_off, p, side="right")) - 1
        for m in range(2):
            mine = [r for r in rec[pair_off[p]:pair_off[p + 1]] if int(r["mate"]) == m]
            if not mine:
                continue
            best = max(mine, key=lambda r: float(r["score"]))
            c = batch.cand[int(best["cand"])]
            cig = np.ascontiguousarray(batch.cigar[int(c["cigar_off"]):int(c["cigar_off"]) + int(c["n_cigar"])], dtype=np.uint32)
            chrom = int(np.searchsorted(ctg_off, int(c["pos"]), side="right")) - 1
            s = sam.SamRec()
            s.ident, s.chrom, s.chrom_id = bucket.ident(p), names[chrom], chrom
            s.pos = int(c["pos"]) - int(ctg_off[chrom]) + 1
            s.mapq, s.score_mapq, s.gamma = int(best["mapq"]), int(best["score_mapq"]), 1.0
            s.mate, s.rev, s.duplicate, s.cloud_id, s.cloud_bad, s.bc = m, int(c["is_rev"]), 0, group, 0, int(bucket.bc[p])
            s.read, s.qual, s.read_len = bucket.read(2 * p + m), bucket.qual(2 * p + m), len(bucket.read(2 * p + m))
            s.mate_read, s.mate_qual, s.mate_read_len = bucket.read(2 * p + 1 - m), bucket.qual(2 * p + 1 - m), len(bucket.read(2 * p + 1 - m))
            s.aln_pos, s.aln_rev, s.edit_dist, s.n_cigar = s.pos - 1, int(c["is_rev"]), int(c["NM"]), len(cig)
            s.cigar = cig.ctypes.data_as(C.POINTER(C.c_uint32))
            keep += [cig, s]
            recs[(p, m)] = s
    n_lines = 2 * bucket.n_pairs
    arr = (sam.SamLine * n_lines)()
    for p in range(bucket.n_pairs):      # the reference prints (best, best_mate) then (best_mate, best)
        a, b = recs.get((p, 0)), recs.get((p, 1))
        assert a is not None or b is not None
        if a is not None:
            arr[2 * p].rec = C.pointer(a)
            arr[2 * p + 1].mate = C.pointer(a)
        if b is not None:
            arr[2 * p].mate = C.pointer(b)
            arr[2 * p + 1].rec = C.pointer(b)
    text = sam.format_lines(arr, n_lines, sam.default_opts())
    out = text.split(b"\n")
    assert out[-1] == b"" and len(out) == n_lines + 1
    n_mapped = 0
    for k, line in enumerate(out[:-1]):
        f = line.split(b"\t")
        p, m = k // 2, k % 2
        flag = int(f[1])
        assert f[0] == bucket.ident(p) and bool(flag & 64) == (m == 0) and bool(flag & 128) == (m == 1)
        bx = [t for t in f[11:] if t.startswith(b"BX:Z:")][0][5:]
        assert bx == ingest.decode_barcode(int(bucket.bc[p])) + b"-1"
        if flag & 4:
            assert f[5] == b"*" and f[9] == bucket.read(2 * p + m)
            continue
        n_mapped += 1
        s = recs[(p, m)]
        want_seq = bucket.read(2 * p + m)
        if flag & 16:
            want_seq = want_seq[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        assert f[9] == want_seq and f[2] == s.chrom and int(f[3]) == s.pos
        # the CIGAR's query length is the read's length
        num, qlen = b"", 0
        for ch in f[5]:
            if chr(ch).isdigit():
                num += bytes([ch])
            else:
                if chr(ch) in "MIS":
                    qlen += int(num)
                num = b""
        assert qlen == len(want_seq)
    assert n_mapped > 1.8 * bucket.n_pairs * 0.9
