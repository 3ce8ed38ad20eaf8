"""tests/golden/sam/ (written by tests/golden/make_sam_vectors.py): SAM text produced by the reference's own host code
(every unmodified src/*.c over the nine libbwa symbols of oracle/bwaface.c).  Helpers shared by the CPU and the GPU test."""
from __future__ import annotations

import gzip
import json
import os
import tempfile

import numpy as np

import oracle_lib as O
from ema_amd import build_index
from ema_amd import engine as E
from ema_amd import clouds, ingest, sam, stream

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "sam")
_REFS = {}


def cases():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)["cases"]


def reference(name):
    """The committed FASTA unpacked as <tmp>/ref.fa with its index rebuilt (ema_index_build); returns (prefix, [(name, len)])."""
    if name not in _REFS:
        d = tempfile.mkdtemp(prefix=f"ema_gold_{name}_")
        prefix = os.path.join(d, "ref.fa")
        with gzip.open(os.path.join(GOLD, f"ref_{name}.fa.gz"), "rb") as z, open(prefix, "wb") as f:
            f.write(z.read())
        build_index(prefix)
        contigs = []
        with open(prefix + ".fai") as f:
            for line in f:
                t = line.split("\t")
                contigs.append((t[0].encode(), int(t[1])))
        _REFS[name] = (prefix, contigs)
    return _REFS[name]


class Run:
    """What the reference's command line of one case means for the C ABI's options."""

    def __init__(self, case):
        argv = case["argv"]
        self.argv = [a.encode() for a in argv]
        self.platform = argv[argv.index("-p") + 1] if "-p" in argv else "10x"
        self.po = stream.platform_opts(self.platform)      # barcode length, thresholds, error rate, name style (src/techs.c:70-130)
        self.haplotag = self.po["is_haplotag"]
        self.bc_len = self.po["bc_len"]
        self.x_mode = "-x" in argv
        self.density_opt = "-d" in argv      # the reference seeds rand() from time(): 1500000000 as ema_refhost saw it (oracle/bwaface.c)
        self.density_seed = 1500000000
        self.rg_line = b"@RG\tID:rg1\tSM:sample1"      # the reference's default, src/main.c:25: never NULL
        if "-R" in argv:      # main.c:282 escape(): backslash-t etc. become the characters
            s = argv[argv.index("-R") + 1]
            self.rg_line = s.replace("\\t", "\t").replace("\\n", "\n").replace("\\r", "\r").replace("\\\\", "\\").encode()
        self.rg_id = None if self.rg_line is None else self.rg_line[self.rg_line.index(b"ID:") + 3:]      # src/align.c:255 -> samrecord.c:260-264
        self.bx_index = argv[argv.index("-i") + 1].encode() if "-i" in argv else b"1"
        self.paths = [os.path.join(GOLD, case["name"], b) for b in case["buckets"]]
        self.fastq = "-1" in argv      # `-1 a.fq [-2 b.fq]`: ONE input, of one or two files
        if self.fastq:
            self.fastq_mate = self.paths[1] if "-2" in argv else None
            self.paths = self.paths[:1]
        self.expected = open(os.path.join(GOLD, case["name"], "expected.sam"), "rb").read()

    def read(self, path):
        """The input as the product's reader lays it out (bucket file, or FASTQ as -1 / -2 take it)."""
        if self.fastq:
            return ingest.read_fastq(path, self.fastq_mate, bc_len=self.bc_len, is_haplotag=self.haplotag, name_style=self.po["fastq_name_style"])
        return ingest.read_bucket(path, bc_len=self.bc_len, is_haplotag=self.haplotag)

    def sam_opts(self):
        so = sam.default_opts()
        so.rg_id, so.bx_index = self.rg_id, self.bx_index
        so.is_haplotag, so.bc_len = int(self.haplotag), self.bc_len
        return so

    def cloud_opts(self):
        co = clouds.default_opts()
        co.dist_thresh, co.many_clouds, co.density_opt = self.po["dist_thresh"], int(self.po["many_clouds"]), int(self.density_opt)
        co.n_density_probs = len(self.po["density_probs"])
        for i, p in enumerate(self.po["density_probs"]):
            co.density_probs[i] = p
        return co

    def header(self, contigs):
        return sam.header(contigs, self.rg_line, b"0.6.2", self.argv)


def oracle_batch(prefix, bucket, error_rate=0.001):
    """Candidates and append_alignments records of every pair of a bucket from the CPU oracle, in the engine's layout."""
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
                c["aln_score"], c["aln_sub"] = d["score"], max(d["sub"], d["csub"])
                cigar.extend(d["cigar"])
                cands.append(c)
            cand_off.append(len(cands))
            base.append(len(cands))
        for e in O.append_alignments(idx, opt, r1, r2, error_rate=error_rate):
            a = np.zeros((), dtype=E.ALN_REC_DTYPE)
            a["pair"], a["mate"], a["unique"], a["cand"] = p, e["mate"], e["unique"], base[e["mate"]] + e["cand"]
            a["clip"], a["clip_edit_dist"], a["mapq"], a["score_mapq"], a["score"] = e["clip"], e["clip_edit_dist"], e["mapq"], e["score_mapq"], e["score"]
            recs.append(a)
        pair_off.append(len(recs))
    batch = E.Batch(np.array(cand_off, np.uint64), np.array(cands, dtype=E.CAND_DTYPE) if cands else np.zeros(0, E.CAND_DTYPE),
                    np.array(cigar, np.uint32), np.zeros(2 * bucket.n_pairs, np.int32))
    rec = np.array(recs, dtype=E.ALN_REC_DTYPE) if recs else np.zeros(0, E.ALN_REC_DTYPE)
    return batch, rec, np.array(pair_off, np.uint64)
