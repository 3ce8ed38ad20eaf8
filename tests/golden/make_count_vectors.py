"""Writes tests/golden/count_vectors.json: inputs (tests/count_cases.py) and the bytes the REFERENCE's `ema count` produced for them
-- $TMPDIR/ema_ref/ref_count, i.e. /root/reference/cpp/count.cc compiled where it lies (oracle/Makefile, target ref).  Run in the
build container (the reference tree is needed); the vectors are what pins the product on a machine without it.
  python tests/golden/make_count_vectors.py"""
import base64, json, os, random, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
import count_cases as K

REF = os.path.join(os.environ.get("EMA_REF_OUT") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "ema_ref"), "ref_count")


def run_ref(wl_text, fastq_text, max_map, haplotag):
    with tempfile.TemporaryDirectory() as d:
        wl = os.path.join(d, "wl.txt")
        open(wl, "w").write(wl_text)
        subprocess.run([REF, wl, os.path.join(d, "o"), str(max_map), str(int(haplotag))], input=fastq_text.encode("latin-1"), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for ext in ("ema-fcnt", "ema-ncnt"):
            p = os.path.join(d, "o." + ext)
            out[ext] = base64.b64encode(open(p, "rb").read()).decode() if os.path.exists(p) else None
        return out


def main():
    cases = []
    rng = random.Random(2024)
    for seed, n_wl, n, max_map, nl in ((1, 50, 400, 1 << 30, True), (2, 8, 300, 72 * 40, True), (3, 200, 0, 1 << 30, True), (4, 30, 120, 1 << 30, False)):
        wl = K.whitelist(rng, n_wl)
        wl_text = "\n".join(wl) + "\n"
        if seed == 2:
            wl_text += wl[0] + "\n" + wl[1][:16] + "TTTT\n"      # a duplicate line and a line longer than a barcode
        fq = K.tenx_fastq(seed, wl, n, nl)
        cases.append({"name": f"10x_{seed}", "whitelist": wl_text, "fastq": fq, "max_map_size": max_map, "haplotag": 0, "expect": run_ref(wl_text, fq, max_map, False)})
    # a stream cut short inside its last pair WITHOUT a final line end: the reference's later getlines leave their strings as they were
    # (the read takes the previous pair's qualities, or the name line for its bases: cpp/count.cc:86-108)
    wl = K.whitelist(rng, 20)
    wl_text = "\n".join(wl) + "\n"
    lines = K.tenx_fastq(11, wl, 40).split("\n")
    for k in (1, 2, 3, 5):
        fq = "\n".join(lines[:8 * 39 + k])
        cases.append({"name": f"cut_short_after_{k}_lines_no_final_newline", "whitelist": wl_text, "fastq": fq, "max_map_size": 1 << 20, "haplotag": 0,
                      "expect": run_ref(wl_text, fq, 1 << 20, False)})
    json.dump({"made_by": "tests/golden/make_count_vectors.py with $TMPDIR/ema_ref/ref_count (reference cpp/count.cc)", "cases": cases},
              open(os.path.join(R, "tests", "golden", "count_vectors.json"), "w"), indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
