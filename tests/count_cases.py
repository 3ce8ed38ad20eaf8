"""Inputs of the `ema count` parity tests (tests/test_count.py, tests/golden/make_count_vectors.py): whitelists and
interleaved FASTQ text that reach every branch of reference cpp/count.cc:80-147."""
import random

ACGT = "ACGT"


def whitelist(rng, n):
    out = []
    while len(out) < n:
        bc = "".join(rng.choice(ACGT) for _ in range(16))
        if bc != "A" * 16:
            out.append(bc)
    return out


def record(name, seq, qual, mate_len=40):
    return f"@{name}\n{seq}\n+\n{qual}\n@{name}\n{'C' * mate_len}\n+\n{'F' * mate_len}\n"


def tenx_fastq(seed, wl, n, last_newline=True):
    """10x pairs: whitelisted, one-off and unlisted barcodes, N in the barcode, lower case, qualities below '!' and above the cap,
    mates shorter than 32 bases, a quality line shorter than the barcode, a truncated last record."""
    rng = random.Random(seed)
    txt = []
    for i in range(n):
        kind = rng.random()
        bc = rng.choice(wl) if kind < 0.55 else "".join(rng.choice(ACGT) for _ in range(16))
        if 0.55 <= kind < 0.7:      # one base off a whitelisted barcode
            b = list(rng.choice(wl)); p = rng.randrange(16); b[p] = rng.choice([c for c in ACGT if c != b[p]]); bc = "".join(b)
        if rng.random() < 0.08:
            p = rng.randrange(16); bc = bc[:p] + "N" + bc[p + 1:]
        if rng.random() < 0.05:
            bc = bc.lower()
        L = rng.choice([20, 31, 32, 60, 151])
        seq = (bc + "".join(rng.choice(ACGT) for _ in range(max(0, L - 16))))[:L]
        qual = "".join(rng.choice("#,5:AFIJ~") for _ in range(L))
        r = rng.random()
        if r < 0.04:
            p = rng.randrange(16); qual = qual[:p] + rng.choice(" \x1f") + qual[p + 1:]      # below '!': the pair is dropped
        elif r < 0.07:
            qual = qual[:rng.randrange(4, 14)]      # a quality line shorter than the barcode
        txt.append(record(f"r{i} 1:N:0", seq, qual))
    s = "".join(txt)
    if n and not last_newline:
        s = s[:-1]
    if n and rng.random() < 0.5:
        s += "@tail\nACGT"      # a record cut short: counted as ignored
    return s


def haplotag_fastq(seed, n):
    rng = random.Random(seed)
    txt = []
    for i in range(n):
        a, b, c, d = (rng.randrange(1, 97) for _ in range(4))
        tag = f"BX:Z:A{a:02d}C{c:02d}B{b:02d}D{d:02d}"
        r = rng.random()
        if r < 0.1:
            name = f"h{i}"      # no tag at all
        elif r < 0.2:
            name = f"h{i} {tag[:-1]}"      # tag cut short at the end of the line: not accepted
        elif r < 0.3:
            name = f"h{i}\tRX:Z:x {tag}-1"
        else:
            name = f"h{i} {tag} extra"
        L = rng.choice([31, 40, 100])
        txt.append(record(name, "".join(rng.choice(ACGT) for _ in range(L)), "F" * L))
    return "".join(txt)
