"""The GPU path against SAM text written by the REFERENCE'S OWN host code (tests/golden/sam/, make_sam_vectors.py there).

1. ema_sam_header + ema_stream_sam (bucket files -> SAM text through C-ABI calls only: reader, engine on the GPU, append stage,
   clouds / EM / duplicates, formatter) == expected.sam, header included, byte for byte, for every committed case
   (`ema align -s`, `-x`, `-p haplotag`, `-R`, `-i`).
2. Opt-in only (EMA_RUN_REF_BINARY=1 on a machine that has both a GPU and the binary `make -C oracle ref` builds OUTSIDE the
   repository; never on the pool's GPU boxes, where no reference-built file may travel: SURVEY 8c): ema_ref_gpu (the reference's unmodified objects -- its own bwabridge.c, align.c,
   samdict.c, samrecord.c -- linked against libema_bwaabi.so, the nine-symbol face on this engine, instead of -lbwa): the
   reference binary itself, running on the GPU one call at a time, writes the same file.  That is rows B1 / B2 of SURVEY 8b as
   the reference's maintainers would use them."""
import os
import subprocess

import pytest

from golden_sam_lib import Run, cases, reference
from ema_amd import engine as E
from ema_amd import stream

pytestmark = pytest.mark.gpu
CASES = cases()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GPU = os.path.join(os.environ.get("EMA_REF_OUT") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "ema_ref"), "ema_ref_gpu")
RUN_REF_BINARY = os.environ.get("EMA_RUN_REF_BINARY") == "1" and os.path.exists(REF_GPU)


@pytest.mark.parametrize("case", [c for c in CASES if Run(c).density_opt and not Run(c).x_mode and len(Run(c).paths) == 1], ids=lambda c: c["name"])
def test_a_bucket_with_its_own_stream_of_draws_equals_the_reference_process(case, tmp_path):
    """`-d` with ema_cloud_opts.seed_private: the bucket draws from a glibc random_r stream of its own, seeded with the
    value the reference process's time() gave -- the same SAM text as the reference's srand() / rand(), byte for byte, without
    touching the process's rand() (which is seeded with something else here to prove it)."""
    from ema_amd import clouds
    run = Run(case)
    prefix, contigs = reference(case["ref"])
    eng = E.Engine(prefix)
    out = str(tmp_path / "out.sam")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    head = run.header(contigs)
    assert os.write(fd, head) == len(head)
    clouds.reseed(12345)
    stream.stream_sam(eng, run.paths, fd, rg_id=run.rg_id, platform=run.platform, bx_index=run.bx_index, density_opt=True, density_seed=run.density_seed,
                      fastq_mates=[run.fastq_mate] if run.fastq else None)
    os.close(fd)
    eng.close()
    assert open(out, "rb").read() == run.expected


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_stream_sam_equals_the_reference_host_code(case, tmp_path):
    run = Run(case)
    prefix, contigs = reference(case["ref"])
    eng = E.Engine(prefix)
    assert [(c[0].encode(), c[1]) for c in eng.contigs()] == contigs
    out = str(tmp_path / "out.sam")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    head = run.header(contigs)
    assert os.write(fd, head) == len(head)
    if run.density_opt:
        from ema_amd import clouds
        clouds.reseed(run.density_seed)
    bst, sst = stream.stream_sam(eng, run.paths, fd, rg_id=run.rg_id, platform=run.platform,
                                 continue_cloud_ids=run.x_mode, bx_index=run.bx_index, density_opt=run.density_opt,
                                 fastq_mates=[run.fastq_mate] if run.fastq else None)
    os.close(fd)
    eng.close()
    got = open(out, "rb").read()
    assert got == run.expected
    assert sum(s["lines"] for s in sst) == case["lines"] and sum(s["with_xa"] for s in sst) == case["with_xa"]
    assert sum(s["duplicates"] for s in sst) == case["duplicates"] and sum(s["unmapped_mates"] for s in sst) == case["unmapped"]
    assert all(s["rc"] == 0 and s["capacity_flags"] == 0 for s in bst)


@pytest.mark.skipif(not RUN_REF_BINARY, reason="opt-in: EMA_RUN_REF_BINARY=1 and a reference-built ema_ref_gpu outside the repository (no reference binary travels to the GPU box)")
@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("10x_small_barcodes_rg", "x_two_buckets", "haplotag", "fastq_two_files")],
                         ids=lambda c: c["name"])
def test_the_reference_binary_on_the_gpu_face_writes_the_same_sam(case, tmp_path):
    run = Run(case)
    prefix, _contigs = reference(case["ref"])
    for ext in ("", ".fai", ".bwt", ".fsa", ".sa", ".pac", ".ann", ".amb"):
        if os.path.exists(prefix + ext):
            os.symlink(prefix + ext, str(tmp_path / ("ref.fa" + ext)))
    for b in case["buckets"]:
        os.symlink(os.path.join(os.path.dirname(run.paths[0]), b), str(tmp_path / b))
    p = subprocess.run(case["argv"], executable=REF_GPU, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert open(str(tmp_path / "out.sam"), "rb").read() == run.expected
