"""Host stages against SAM text written by the REFERENCE'S OWN host code (tests/golden/sam/, see make_sam_vectors.py there:
unmodified src/align.c, bwabridge.c, samdict.c, samrecord.c, util.c, techs.c, main.c linked over the CPU oracle's nine
libbwa symbols; `ema align -s/-x ... -t 1`).  Two chains, both byte for byte, header included:

  * the ORACLE's restatements -- oracle/ingest.c (read_special_fastq, src/align.c:759-806), the append stage
    (src/align.c:986-1061), oracle/clouds.c (src/align.c:347-608, src/samdict.c), oracle/sam.c (src/samrecord.c:104-284,
    src/align.c:193-212) -- which every GPU parity test uses as its checker;
  * the PRODUCT's host stages -- ema_ingest_read_bucket, ema_batch_append_alignments, ema_clouds_select, ema_sam_format,
    ema_sam_header -- fed with the oracle's candidates in place of the engine's (no GPU here; tests/test_gpu_golden_sam.py
    runs ema_stream_sam on the engine against the same files).

The engine's arithmetic is the oracle's on both sides of these comparisons; what they pin is the reading of the reference's
host code."""
import ctypes as C

import pytest

import emu_lib

import oracle_lib as O
from golden_sam_lib import Run, cases, oracle_batch, reference
from ema_amd import clouds, ingest, sam
from ema_amd import engine as E
from test_clouds import oracle_selection
from test_sam_format import oracle_text

CASES = cases()


def split(text):
    lines = text.split(b"\n")
    n_head = sum(1 for l in lines if l.startswith(b"@"))
    return b"\n".join(lines[:n_head]) + b"\n", b"\n".join(lines[n_head:])


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])      # ([r5] the -d cases too: oracle/clouds.c restates src/split.c)
def test_oracle_chain_equals_the_reference_host_code(case):
    run = Run(case)
    # -d: the platform's density model and ONE srand() per run with the value the reference's time() gave (tests/golden/make_sam_vectors.py)
    O.clouds_density(run.density_opt, probs=list(run.po["density_probs"]), seed=run.density_seed if run.density_opt else None)
    try:
        _oracle_chain(case, run)
    finally:
        O.clouds_density(False)


def _oracle_chain(case, run):
    prefix, contigs = reference(case["ref"])
    names = [n for n, _ in contigs]
    want_head, want_body = split(run.expected)
    head = O.sam_header(contigs, run.rg_line, b"0.6.2", run.argv)
    assert head == want_head
    so = run.sam_opts()
    body, first = b"", 0
    for path in run.paths:
        bucket = run.read(path)
        if not run.fastq:      # (the oracle restates the bucket reader only; for FASTQ input the product's reader feeds its chain)
            want, _groups = O.read_special_fastq(path, run.bc_len, run.haplotag)
            assert [w[0] for w in want] == bucket.bc.tolist() and all(bucket.read(2 * i) == w[2] for i, w in enumerate(want))
        batch, rec, pair_off = oracle_batch(prefix, bucket, error_rate=run.po["error_rate"])
        arr, n, _keep, _rows, next_id = oracle_selection(bucket, batch, rec, pair_off, names, dist_thresh=run.po["dist_thresh"],
                                                         many_clouds=run.po["many_clouds"], first_cloud_id=first)
        body += oracle_text(arr, n, so)
        first = next_id      # src/align.c:19: the cloud counter is static, so it runs on across the files of an -x run
    assert body == want_body
    assert body.count(b"\n") == case["lines"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_product_host_stages_equal_the_reference_host_code(case):
    run = Run(case)
    prefix, contigs = reference(case["ref"])
    names = [n for n, _ in contigs]
    want_head, want_body = split(run.expected)
    assert run.header(contigs) == want_head
    so = run.sam_opts()
    body, dev_body, first = b"", b"", 0
    if run.density_opt:
        clouds.reseed(run.density_seed)      # once per run, as the reference's first bad cloud does
    for path in run.paths:
        bucket = run.read(path)
        batch, _orec, _opair_off = oracle_batch(prefix, bucket)
        rec, pair_off = E.append_alignments(batch, bucket.off, error_rate=run.po["error_rate"])      # the product's append stage on the oracle's candidates
        co = run.cloud_opts()
        co.first_cloud_id, co.n_threads, co.emit = first, 3, 2      # lines for the host formatter AND the compact records for the device's
        sel = clouds.select(bucket, batch, rec, pair_off, names, co)
        body += sam.format_lines(sel.lines, sel.n_lines, so)
        dev_body += emu_lib.sam_format_selection(sel, names, so)      # k_sam.hip under the host interpreter (the GPU run: test_gpu_golden_sam.py)
        first = sel.next_cloud_id
    assert body == want_body
    assert dev_body == want_body


def test_the_vectors_cover_what_they_claim():
    by = {c["name"]: c for c in CASES}
    assert by["exact_dups_bad_clouds_xa"]["with_xa"] > 20 and by["exact_dups_bad_clouds_xa"]["bad_cloud"] > 50
    assert by["10x_small_barcodes_rg"]["unmapped"] > 5 and by["10x_small_barcodes_rg"]["duplicates"] > 10
    assert len(by["x_two_buckets"]["buckets"]) == 2 and "-x" in by["x_two_buckets"]["argv"]
    assert "haplotag" in by["haplotag"]["argv"]
    # -d changes what is printed: the same bucket without it selects other alignments
    run = Run(by["density_opt_exact_dups"])
    prefix, contigs = reference("dups")
    bucket = ingest.read_bucket(run.paths[0], bc_len=16, is_haplotag=False)
    batch, _r, _p = oracle_batch(prefix, bucket)
    rec, pair_off = E.append_alignments(batch, bucket.off)
    texts = []
    for d in (0, 1):
        co = clouds.default_opts()
        co.density_opt = d
        clouds.reseed(run.density_seed)
        sel = clouds.select(bucket, batch, rec, pair_off, [n for n, _ in contigs], co)
        texts.append(sam.format_lines(sel.lines, sel.n_lines, run.sam_opts()))
    assert texts[0] != texts[1] and texts[1] == split(run.expected)[1]
