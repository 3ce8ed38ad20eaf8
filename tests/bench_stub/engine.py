import ctypes as C
import os

import numpy as np

import oracle_lib as O
from ema_amd import engine as E


def default_opts():
    return E.default_opts()


gather_pairs = E.gather_pairs


class Engine:
    n_streams = 3

    def __init__(self, prefix, device=0, opts=None, _peer_of=None):
        self.prefix = prefix
        self.idx, self.opt = (O.Index(prefix), O.default_opt()) if _peer_of is None else (_peer_of.idx, _peer_of.opt)
        self.slots = {}
        self._peer = None

    def peer(self):
        if os.environ.get("BENCH_STUB_NO_PEER"):
            return None
        if self._peer is None:
            self._peer = Engine(self.prefix, _peer_of=self)
        return self._peer

    def stage_slot(self, slot, bases, off):
        self.slots[slot] = (np.asarray(bases).copy(), np.asarray(off).copy())

    def run_slot(self, slot):
        assert slot in self.slots

    def run(self, serial=False):
        pass

    def sync(self):
        pass

    def timing(self):
        return {"seed_ms": 1.0, "extend_ms": 1.0, "rescue_ms": 1.0, "final_ms": 1.0, "full_tier_ms": 1.0, "full_ms": [1.0] * 4}

    def index_info(self):
        return {"n_super": 1, "super_shift": 31, "sa_width": 4, "kmer_k": 0}

    def seed_launches_per_series(self):
        return 3

    def close(self):
        pass

    def align(self, bases, off):
        """(Batch, records, pair_off) of one batch through the oracle + the library's host append stage."""
        n = (len(off) - 1) // 2
        cand_off, cands, cigar = [0], [], []
        for p in range(n):
            r1, r2 = bases[off[2 * p]:off[2 * p + 1]].tobytes(), bases[off[2 * p + 1]:off[2 * p + 2]].tobytes()
            res = O.align_pair(self.idx, self.opt, r1, r2)
            for m in range(2):
                for d in res[m]:
                    c = np.zeros((), dtype=E.CAND_DTYPE)
                    for f in O.REG_FIELDS:
                        c[f] = d[f]
                    c["pos"], c["is_rev"], c["NM"], c["n_cigar"], c["cigar_off"] = d["pos"], d["is_rev"], d["NM"], len(d["cigar"]), len(cigar)
                    cigar.extend(d["cigar"])
                    cands.append(c)
                cand_off.append(len(cands))
        cand = np.array(cands, dtype=E.CAND_DTYPE) if cands else np.zeros(0, E.CAND_DTYPE)
        if os.environ.get("BENCH_STUB_CORRUPT") and len(cand):
            cand["score"] += 1      # what a wrong kernel would look like to the spot check
        batch = E.Batch(np.array(cand_off, np.uint64), cand, np.array(cigar, np.uint32), np.zeros(2 * n, np.int32))
        rec, pair_off = E.append_alignments(batch, off)
        return batch, rec, pair_off
