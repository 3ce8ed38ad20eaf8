"""Writes tests/golden/preproc_vectors.json: inputs (tests/count_cases.py) and, per bucket file, length + SHA-256 of what the
REFERENCE's `ema count` + `ema preproc` produced for them -- $TMPDIR/ema_ref/ref_count and ref_preproc, i.e.
/root/reference/cpp/count.cc and cpp/correct.cc compiled where they lie (oracle/Makefile, target ref).  Run in the build container.
  python tests/golden/make_preproc_vectors.py"""
import json, os, random, sys, tempfile, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import count_cases as K
import test_preproc as T


def main():
    cases = []
    rng = random.Random(77)
    for seed, n_wl, n, args in ((1, 60, 500, dict(n_buckets=6)), (2, 25, 400, dict(n_buckets=9, do_h2=True, n_threads=2)),
                                (3, 300, 250, dict(n_buckets=3, do_bx_format=True)), (4, 10, 0, dict(n_buckets=2)),
                                (5, 40, 350, dict(n_buckets=4, buffer_size=1500, max_map=72 * 20, do_h2=True))):
        wl = K.whitelist(rng, n_wl)
        wl_text = "\n".join(wl) + "\n"
        fq = T.well_formed(K.tenx_fastq(100 + seed, wl, n))
        with tempfile.TemporaryDirectory() as d:
            want = T.run_reference(pathlib.Path(d) / "r", wl_text, fq, False, **dict(args))
        cases.append({"name": f"10x_{seed}", "whitelist": wl_text, "fastq": fq, "haplotag": 0, "args": args, "expect": T.digest(want)})
    # a stream cut short inside its last pair without a final line end: the missing lines read as what the reference's strings still
    # hold (cpp/correct.cc:427-430,573,596,607-608); 4..7 lines present: the mate is written from stale strings
    wl = K.whitelist(rng, 20)
    wl_text = "\n".join(wl) + "\n"
    lines = T.well_formed(K.tenx_fastq(11, wl, 40)).split("\n")
    for k in (1, 4, 5, 6, 7):
        fq = "\n".join(lines[:8 * 39 + k])
        with tempfile.TemporaryDirectory() as d:
            want = T.run_reference(pathlib.Path(d) / "r", wl_text, fq, False, n_buckets=3)
        cases.append({"name": f"cut_short_after_{k}_lines_no_final_newline", "whitelist": wl_text, "fastq": fq, "haplotag": 0, "args": dict(n_buckets=3), "expect": T.digest(want)})
    json.dump({"made_by": "tests/golden/make_preproc_vectors.py with ref_count + ref_preproc built outside the repository by oracle/Makefile (reference cpp/count.cc, cpp/correct.cc)",
               "cases": cases}, open(os.path.join(R, "tests", "golden", "preproc_vectors.json"), "w"), indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
