"""Random buckets and selected records for the SAM formatter on the device (csrc/k_sam.hip), in both forms: the compact records the
kernels take (ema_sam_desc / ema_sam_xa / sel_at over an ema_bucket and a CIGAR array) and the same lines as ema_sam_rec / ema_sam_line
for the host formatter, which tests/test_sam_format.py pins against the oracle's restatement of print_sam_record.  Shared by the
interpreter test (CPU) and the GPU test."""
import ctypes as C
import math
import random

import numpy as np

from ema_amd import ingest, sam

CHROMS = [b"chr1", b"chr2", b"chrX", b"chrUn_KI270742v1", b"c"]


def rand_cigar(rng, read_len):
    ops, left = [], read_len
    if rng.random() < 0.3:
        k = rng.randrange(1, 20); ops.append((k, rng.choice((3, 4)))); left -= k
    tail = None
    if rng.random() < 0.3:
        k = rng.randrange(1, 20); tail = (k, rng.choice((3, 4))); left -= k
    while left > 0:
        k = rng.randrange(1, left + 1)
        ops.append((k, 0)); left -= k
        if left > 0 and rng.random() < 0.5:
            if rng.random() < 0.5:
                ops.append((rng.choice((1, 5, 9, 10, 99, 100, 12345)), 2))
            else:
                j = rng.randrange(1, min(5, left) + 1); ops.append((j, 1)); left -= j
    if tail:
        ops.append(tail)
    return ops


class Case:
    """n_pairs pairs, every one selected (with or without a mate record); keeps every buffer alive."""

    def __init__(self, seed, n_pairs, haplotag, bases=b"ACGTN"):
        rng = random.Random(seed)
        self.haplotag = haplotag
        names, reads, quals = [], [], []
        for p in range(n_pairs):
            names.append(b"@" + bytes(rng.choice(b"abcXYZ0123456789:_/") for _ in range(rng.choice((0, 1, 3, 4, 5, 17, 40, 149)))))
            for _ in range(2):
                n = rng.choice((1, 2, 3, 4, 5, 7, 8, 30, 100, 150, 151, 250, 255))
                reads.append(bytes(rng.choice(bases) for _ in range(n)))
                quals.append(bytes(rng.choice(b"#,:FGH!~") for _ in range(n)))
        self.ids = np.frombuffer(b"".join(names) + b"\0" * 8, np.uint8).copy()
        self.id_off = np.cumsum([0] + [len(n) for n in names]).astype(np.uint32)
        self.bases = np.frombuffer(b"".join(reads) + b"\0" * 8, np.uint8).copy()
        self.quals = np.frombuffer(b"".join(quals) + b"\0" * 8, np.uint8).copy()
        self.off = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint32)
        if haplotag:
            self.bc = np.array([rng.choice((0, 127, 9, 10, 99, 100)) << 24 | rng.randrange(128) << 16 | rng.randrange(128) << 8 | rng.randrange(128) for _ in range(n_pairs)], np.uint64)
        else:
            self.bc = np.array([rng.getrandbits(32) for _ in range(n_pairs)], np.uint64)
        self.group_off = np.array([0, n_pairs], np.uint64)
        bk = ingest._Bucket()
        bk.n_pairs, bk.n_groups = n_pairs, 1
        bk.group_off = self.group_off.ctypes.data_as(C.POINTER(C.c_uint64))
        bk.bc = self.bc.ctypes.data_as(C.POINTER(C.c_uint64))
        bk.off = self.off.ctypes.data_as(C.POINTER(C.c_uint32))
        bk.bases = C.cast(self.bases.ctypes.data, C.POINTER(C.c_char))
        bk.quals = C.cast(self.quals.ctypes.data, C.POINTER(C.c_char))
        bk.id_off = self.id_off.ctypes.data_as(C.POINTER(C.c_uint32))
        bk.ids = C.cast(self.ids.ctypes.data, C.POINTER(C.c_char))
        self.bk = bk
        # the batch's CIGAR array: the records name a stretch [cigar_lo, cigar_hi) in its middle
        self.cigar_lo = rng.choice((0, 1, 7, 1000))
        cig = [0xdead] * self.cigar_lo
        descs, xas, sel_at = [], [], []
        self.keep, recs = [], []

        def one(p, mate_no, has_mate):
            d = np.zeros((), sam.DESC_DTYPE)
            rlen = len(reads[2 * p + mate_no])
            ops = rand_cigar(rng, rlen) if rng.random() < 0.95 else []
            gamma = rng.choice((0.0, 1.0, 0.999999, 0.9999991, 0.5, 0.9, 1e-7, 1.2345678e-5, rng.random(), 1 - 10 ** -rng.uniform(0, 7)))
            bwa_mapq = rng.randrange(0, 255)
            gm = int(-10 * math.log10(1 - gamma)) if gamma <= 0.999999 else 60
            d["pair"], d["rid"], d["pos"] = p, rng.randrange(len(CHROMS)), rng.choice((1, rng.randrange(1, 5000), rng.randrange(1, 2_000_000_000), 4_000_000_000))
            d["cigar_off"], d["n_cigar"], d["edit_dist"] = len(cig), len(ops), rng.choice((0, 1, 9, 10, 123))
            cig.extend(n << 4 | t for n, t in ops)
            d["cloud_id"], d["xa"] = rng.choice((0, 7, 99999, 2_000_000_000)), -1
            d["mate"], d["rev"], d["duplicate"], d["cloud_bad"] = mate_no, rng.randrange(2), int(rng.random() < 0.2), rng.randrange(2)
            d["mapq"], d["has_mate"] = max(0, min(60, gm, bwa_mapq)), has_mate
            g = b"%.5g" % gamma
            d["gamma"], d["gamma_len"] = g, len(g)
            r = sam.SamRec()
            r.ident, r.chrom, r.chrom_id, r.pos = bytes(names[p][1:]), CHROMS[int(d["rid"])], int(d["rid"]), int(d["pos"])
            r.mapq, r.score_mapq, r.gamma = bwa_mapq, 60, gamma
            r.mate, r.rev, r.duplicate = mate_no, int(d["rev"]), int(d["duplicate"])
            r.cloud_id, r.cloud_bad, r.bc = int(d["cloud_id"]), int(d["cloud_bad"]), int(self.bc[p])
            r.read, r.qual, r.read_len = reads[2 * p + mate_no], quals[2 * p + mate_no], rlen
            r.mate_read, r.mate_qual, r.mate_read_len = reads[2 * p + 1 - mate_no], quals[2 * p + 1 - mate_no], len(reads[2 * p + 1 - mate_no])
            r.aln_pos, r.aln_rev, r.edit_dist, r.n_cigar = int(d["pos"]) - 1, int(d["rev"]), int(d["edit_dist"]), len(ops)
            ca = (C.c_uint32 * max(1, len(ops)))(*[n << 4 | t for n, t in ops])
            self.keep.append(ca)
            r.cigar = ca
            if rng.random() < 0.3:
                x = np.zeros((), sam.XA_DTYPE)
                xops = rand_cigar(rng, rlen)
                x["rid"], x["pos"], x["cigar_off"], x["n_cigar"] = rng.randrange(len(CHROMS)), rng.choice((5, 4_000_000_000, rng.randrange(1, 10**9))), len(cig), len(xops)
                x["edit_dist"], x["rev"] = rng.randrange(0, 30), rng.randrange(2)
                cig.extend(n << 4 | t for n, t in xops)
                d["xa"] = len(xas)
                xas.append(x)
                al = sam.SamAlt()
                xa_c = (C.c_uint32 * len(xops))(*[n << 4 | t for n, t in xops])
                al.chrom, al.pos, al.edit_dist, al.rev, al.n_cigar, al.cigar = CHROMS[int(x["rid"])], int(x["pos"]), int(x["edit_dist"]), int(x["rev"]), len(xops), xa_c
                self.keep += [al, xa_c]
                r.alts, r.n_alts = C.pointer(al), 1
            descs.append(d)
            recs.append(r)
            return r

        lines = []
        for p in range(n_pairs):
            first_mate = rng.randrange(2)
            has_mate = rng.random() < 0.7
            sel_at.append(len(descs))
            r1 = one(p, first_mate, int(has_mate))
            r2 = one(p, 1 - first_mate, 0) if has_mate else None
            if r2 is not None and rng.random() < 0.5:      # a proper-looking pair: same contig, opposite strands, near each other
                descs[-1]["rid"] = descs[-2]["rid"]; r2.chrom, r2.chrom_id = r1.chrom, r1.chrom_id
                descs[-1]["rev"] = 1 - descs[-2]["rev"]; r2.rev = r2.aln_rev = int(descs[-1]["rev"])
                near = (int(descs[-2]["pos"]) + rng.choice((-800, -750, -36, -35, 0, 35, 36, 300, 750, 751))) % 2**32 or 1
                descs[-1]["pos"] = near; r2.pos, r2.aln_pos = near, near - 1
            lines.append((r1, r2))
        self.cigar_hi = len(cig)
        cig += [0xbeef] * 3
        self.cigar = np.array(cig, np.uint32)
        self.descs = np.array(descs, sam.DESC_DTYPE)
        self.xas = np.array(xas, sam.XA_DTYPE) if xas else np.zeros(1, sam.XA_DTYPE)
        self.n_xas = len(xas)
        self.sel_at = np.array(sel_at, np.uint32)
        self.n_sel = n_pairs
        arr = (sam.SamLine * (2 * n_pairs))()
        for i, (r1, r2) in enumerate(lines):
            arr[2 * i].rec = C.pointer(r1)
            arr[2 * i].mate = C.pointer(r2) if r2 is not None else None
            arr[2 * i + 1].rec = C.pointer(r2) if r2 is not None else None
            arr[2 * i + 1].mate = C.pointer(r1)
        self.keep.append(recs)
        self.lines = arr

    def opts(self, rg=b"rg1\tSM:x", bx=b"1"):
        so = sam.default_opts()
        so.rg_id, so.bx_index = rg, bx
        so.is_haplotag, so.bc_len = int(self.haplotag), 12 if self.haplotag else 16
        return so

    def host_text(self, so):
        return sam.format_lines(self.lines, len(self.lines), so)

    def oracle_text(self, so):
        """The same lines through the oracle's stdio restatement of print_sam_record (oracle/sam.c)."""
        from test_sam_format import oracle_text
        return oracle_text(self.lines, len(self.lines), so)

    def cigar_ptr(self):
        return self.cigar.ctypes.data + 4 * self.cigar_lo
