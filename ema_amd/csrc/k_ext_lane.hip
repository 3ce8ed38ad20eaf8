// ema_amd/csrc/k_ext_lane.hip -- K2x: the banded extension of a seed (ksw_extend2 to the left, then to the right, as mem_chain2aln
// runs them for ONE seed), ONE LANE PER SEED.
//
// Replaces, for a batch (un-vendored bwa, reached from reference src/bwabridge.c:236-237 -> mem_align1_core -> mem_chain2aln):
// the two ksw_extend2 calls of the first seed mem_chain2aln extends in a chain -- which is 99.95 % of all extension calls on the
// benchmark mix (the oracle's count: a chain's longest seed is extended, the region it yields covers the chain's other seeds).
//
// Why a lane per seed.  The wave-per-read kernels (k_align.hip) run an extension as a row-parallel DP over 64 lanes: ~130-250
// issue slots per row whatever the number of live cells, and the usual extension -- a re-seeding hit in a diverged repeat copy
// that dies after thirty rows of fifteen cells, or the rest of a read beyond a seed -- has 10-60 live cells per row.  Here a lane
// runs ksw_extend2 as the CPU does, cell by cell over the adaptive band [beg, end), with the H/E row in LDS (one word per
// column, lane-interleaved: no bank conflicts): ~30 instructions per CELL-step for 64 extensions at once.  What a wavefront
// loses is divergence -- a row-step lasts as long as the widest band among its 64 tasks, and a lane whose task has ended waits --
// so finished lanes take new tasks from the queue as soon as a quarter of the wave is idle (persistent lanes), and the row loop,
// the band, the z-drop, the exact early exit (dev_dp.hpp) and the "which column attains the row maximum" rule are per-lane state.
//
// The chain of calls of a seed stays in its lane: left extension (h0 = seed length x a), then right (h0 = the score the left one
// reached), then the region's ends and scores exactly as mem_chain2aln derives them.  The known-outcome shortcuts of dev_dp.hpp
// (no or one mismatch on the diagonal) are taken first.  A task whose band would have to be doubled (max_off >= 3/4 w:
// MAX_BAND_TRY's second round) or whose window does not fit is left "not computed": the consumer (k_align.hip, the replay of
// mem_chain2aln in modes 2 and 3) then runs the wave DP itself, as it does for every seed no task was planned for.
//
// Target bases come from the packed reference (16 words of it per 256 rows, staged in LDS at the task's start), query bases from
// the batch's nt4 bytes.  Bound: instruction issue (integer DP); HBM traffic is the read (<= 255 B) and <= 128 B of reference per task.
#include <hip/hip_runtime.h>
#include "dev_ref.hpp"

#define EMA_XL_TW 32      // words of packed reference per task side (16 bases each: 496 rows whatever the alignment)

namespace {

// signed 6-bit field of `tab` at bit `at`
__device__ __forceinline__ int xl_sbfe6(uint32_t tab, unsigned at)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_sbfe((int)tab, at, 6u);
#else
	return (int)((int32_t)(tab << (26u - at)) >> 26);
#endif
}

// H/E row word of a column: h in bits 0..12, e in 13..25, 6 x the query's nt4 code (0, 6, .. 24) in 26..30
#define XL_HMASK 0x1fffu
#define XL_QMASK 0x7c000000u

struct XlSide {          // one ksw_extend2 call in flight (per lane)
	int qlen, tlen, i, beg, end, w, h0, end_bonus;
	int mx, max_i, max_j, max_ie, gscore, max_off;
	int64_t f0;          // forward-strand position of row 0's base
	int fstep, comp;     // per row: +1 / -1 along the forward strand; 3 if the bases are complemented
	int wbase;           // (f >> 4) of tw[0]
};

}  // namespace

// tasks[0..*n_tasks): seeds to extend; res[task.res]: the region's ends and scores; valid[task.res] = 1 when computed.
// bases: the batch's reads as nt4 bytes.  counter: the launch's claim counter (zero on entry).
template <int QMAX>
__global__ void __launch_bounds__(64)
ema_k_ext_lane(DevIndex ix, DevOpts opt, const uint8_t *__restrict__ bases, const ExtTask *__restrict__ tasks, const int *__restrict__ n_tasks, int tasks_cap,
               ExtRes *__restrict__ res, uint8_t *__restrict__ valid, int *__restrict__ counter, int min_q, unsigned long long *__restrict__ prof)
{
	__shared__ uint32_t lds_all[(QMAX + 1 + EMA_XL_TW) * 64];
	const int lane = (int)ema_lane();
	EMA_LDS uint32_t *const L = (EMA_LDS uint32_t *)(&lds_all[0]) + lane;      // column j at L[j << 6]
	EMA_LDS uint32_t *const TW = L + ((QMAX + 1) << 6);                         // reference word k at TW[k << 6]
	const int total = *n_tasks < tasks_cap ? *n_tasks : tasks_cap;
	const int oe_del = opt.o_del + opt.e_del, oe_ins = opt.o_ins + opt.e_ins, e_del = opt.e_del, e_ins = opt.e_ins;
	const int mxsc = opt.a > 0 ? opt.a : 0;
	const int64_t l_pac = ix.l_pac;
	const unsigned long long t_start = prof ? __builtin_amdgcn_s_memtime() : 0;
	unsigned long long n_rowsteps = 0, n_lanerows = 0, n_cells = 0, n_done = 0, n_dp = 0;

	// per-lane task state
	bool active = false;       // a DP side is running
	int pending = 0;           // 0: none; 1: start the left side; 2: start the right side
	ExtTask tk;
	tk.rbeg = tk.rmax0 = tk.rmax1 = 0; tk.q_off = 0; tk.l_query = tk.qbeg = tk.len = tk.res = 0; tk.pad = 0;
	int side = 0;              // 0 left, 1 right
	int a_score = 0, a_truesc = 0, a_qb = 0, a_qe = 0;
	int64_t a_rb = 0, a_re = 0;
	bool ok = true;
	XlSide S;
	S.qlen = S.tlen = S.i = S.beg = S.end = S.w = S.h0 = S.end_bonus = 0; S.mx = 0; S.max_i = S.max_j = S.max_ie = S.gscore = -1; S.max_off = 0;
	S.f0 = 0; S.fstep = 1; S.comp = 0; S.wbase = 0;
	bool drained = false;

	for (;;) {
		// ---- 1. idle lanes take new tasks (all of them when nothing runs; otherwise once a quarter of the wave is idle)
		{
			const unsigned long long idle = __ballot(!active && pending == 0);
			const int n_idle = __popcll(idle);
			if (!drained && (n_idle == 64 || n_idle >= 16)) {
				int base = 0;
				if (lane == 0) base = atomicAdd(counter, n_idle);
				base = __builtin_amdgcn_readlane(base, 0);
				if (base >= total) drained = true;
				else if ((idle >> lane) & 1) {
					const int t = base + __popcll(idle & ((1ULL << lane) - 1));
					if (t < total) {
						tk = tasks[t];
						ok = true;
						const int ql = tk.qbeg, qr = tk.l_query - tk.qbeg - tk.len;
						const int qm = ql > qr ? ql : qr;
						if (qm > QMAX || qm < min_q) ok = false;      // another launch's class (or nobody's: stays "not computed")
						if (!ok) pending = 0;
						else if (tk.qbeg > 0) pending = 1;
						else {
							a_score = a_truesc = tk.len * opt.a; a_qb = 0; a_rb = tk.rbeg;
							if (qr > 0) pending = 2;
							else { a_qe = tk.l_query; a_re = tk.rbeg + tk.len; pending = 3; }
						}
					}
				}
			}
		}
		// ---- 2. sides that start: the call's parameters, the reference words, the H/E row and the query bases, the shortcut
		while (__ballot(pending == 1 || pending == 2 || pending == 4)) {
			if (pending == 1 || pending == 2) {
				side = pending - 1;
				pending = 0;
				int64_t t0;      // forward-reverse coordinate of the target's row 0; rows step by tstep
				int tstep, q0, qstep;
				if (side == 0) {
					S.qlen = tk.qbeg; S.tlen = (int)(tk.rbeg - tk.rmax0); t0 = tk.rbeg - 1; tstep = -1; q0 = tk.qbeg - 1; qstep = -1;
					S.h0 = tk.len * opt.a; S.end_bonus = opt.pen_clip5;
				} else {
					const int qe = tk.qbeg + tk.len;
					S.qlen = tk.l_query - qe; S.tlen = (int)(tk.rmax1 - (tk.rbeg + tk.len)); t0 = tk.rbeg + tk.len; tstep = 1; q0 = qe; qstep = 1;
					S.h0 = a_score; S.end_bonus = opt.pen_clip3;
				}
				const bool rev = t0 >= l_pac;
				S.comp = rev ? 3 : 0;
				S.f0 = rev ? (l_pac << 1) - 1 - t0 : t0;
				S.fstep = rev ? -tstep : tstep;
				bool dp = S.tlen > 0;      // (ksw_extend2 with no target rows: the loop does not run)
				if (S.tlen > 0) {
					const int64_t f_last = S.f0 + (int64_t)(S.tlen - 1) * S.fstep;
					const int64_t f_lo = S.f0 < f_last ? S.f0 : f_last, f_hi = S.f0 < f_last ? f_last : S.f0;
					S.wbase = (int)(f_lo >> 4);
					const int n_w = (int)(f_hi >> 4) - S.wbase + 1;
					if (n_w > EMA_XL_TW) { ok = false; dp = false; }
					else {
						const uint32_t *pw = reinterpret_cast<const uint32_t *>(ix.pac) + S.wbase;
						for (int k = 0; k < n_w; k += 8) {
							uint32_t v[8];
#pragma unroll
							for (int u = 0; u < 8; ++u) v[u] = k + u < n_w ? pw[k + u] : 0u;
#pragma unroll
							for (int u = 0; u < 8; ++u) if (k + u < n_w) TW[(k + u) << 6] = v[u];
						}
					}
				}
				if (ok) {
					// row -1 of the H/E array (eh[j].h = H(-1, j-1): h0, then an insertion run), the query's bases, and the diagonal's
					// mismatches against the first qlen target bases for the shortcut
					const int h0 = S.h0, qlen = S.qlen;
					const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
					const uint8_t *qsrc = bases + tk.q_off;
					int n_mm = 0, p_mm = -1, n_bad = 0;
					const bool diag = S.tlen >= qlen && h0 > 0;
					for (int j0 = 0; j0 <= qlen; j0 += 8) {
						uint8_t qc[8];
#pragma unroll
						for (int u = 0; u < 8; ++u) { const int j = j0 + u; qc[u] = j < qlen ? qsrc[q0 + j * qstep] : (uint8_t)4; }
#pragma unroll
						for (int u = 0; u < 8; ++u) {
							const int j = j0 + u;
							if (j <= qlen) {
								int hv;
								if (j == 0) hv = h0;
								else if (j == 1) hv = h1v;
								else { const int pred = h1v - (j - 2) * e_ins; hv = pred > e_ins ? h1v - (j - 1) * e_ins : 0; }      // (closed form of the run: dev_dp.hpp)
								const unsigned qv = qc[u] > 4 ? 4u : (unsigned)qc[u];
								L[j << 6] = (uint32_t)hv | (qv * 6u) << 26;
								if (diag && j < qlen) {
									const int64_t f = S.f0 + (int64_t)j * S.fstep;
									const uint32_t wd = TW[((int)(f >> 4) - S.wbase) << 6];
									const int tb = (int)((wd >> ((((unsigned)f >> 2) & 3u) << 3) >> ((~(unsigned)f & 3u) << 1)) & 3u) ^ S.comp;
									if (qv > 3) ++n_bad;
									if ((int)qv != tb) { ++n_mm; p_mm = j; }
								}
							}
						}
					}
					S.i = 0; S.beg = 0; S.end = qlen;
					S.mx = h0; S.max_i = S.max_j = -1; S.max_ie = -1; S.gscore = -1; S.max_off = 0;
					{
						int max_ins = (int)((double)(qlen * mxsc + S.end_bonus - opt.o_ins) / e_ins + 1.);
						max_ins = max_ins > 1 ? max_ins : 1;
						int w = opt.w < max_ins ? opt.w : max_ins;
						int max_del = (int)((double)(qlen * mxsc + S.end_bonus - opt.o_del) / e_del + 1.);
						max_del = max_del > 1 ? max_del : 1;
						S.w = w < max_del ? w : max_del;
					}
					if (diag && n_bad == 0) {      // the known-outcome extensions (dev_dp.hpp, ema_wave_extend_nc: proof there)
						if (n_mm == 0) { S.mx = S.gscore = h0 + qlen * opt.a; S.max_j = S.max_i = S.max_ie = qlen - 1; S.max_off = 0; dp = false; }
						else {
							const int gap_min = oe_del < oe_ins + opt.a ? oe_del : oe_ins + opt.a;
							if (n_mm == 1 && opt.a > 0 && gap_min > opt.a + opt.b && (opt.zdrop <= 0 || opt.a + opt.b <= opt.zdrop) && h0 + p_mm * opt.a - opt.b > 0) {
								S.gscore = h0 + (qlen - 1) * opt.a - opt.b; S.max_ie = qlen - 1; S.max_off = 0;
								if (qlen - 1 >= p_mm + opt.b / opt.a + 1) { S.mx = S.gscore; S.max_i = S.max_j = qlen - 1; }
								else { S.mx = h0 + p_mm * opt.a; S.max_i = S.max_j = p_mm - 1; }
								dp = false;
							}
						}
					}
					if (S.h0 <= 0) { ok = false; dp = false; }      // (ksw_extend2 asserts h0 > 0: never on this path; left to the consumer)
				}
				if (dp && ok) { active = true; if (prof) ++n_dp; }
				else pending = 4;      // the side's outcome is known (or the task is given up)
			}
			// ---- a side has ended: mem_chain2aln's use of the six results
			if (pending == 4) {
				pending = 0;
				if (!ok) pending = 3;
				else {
					const int score = S.mx, qle = S.max_j + 1, tle = S.max_i + 1, gtle = S.max_ie + 1, gscore = S.gscore;
					// MAX_BAND_TRY: the band is doubled and the call repeated unless the score did not move or the maximum stayed within 3/4 of it
					const int prev = side == 0 ? -1 : a_score;
					if (!(score == prev || S.max_off < (opt.w >> 1) + (opt.w >> 2))) { ok = false; pending = 3; }
					else if (side == 0) {
						a_score = score;
						if (gscore <= 0 || gscore <= a_score - opt.pen_clip5) { a_qb = tk.qbeg - qle; a_rb = tk.rbeg - tle; a_truesc = a_score; }
						else { a_qb = 0; a_rb = tk.rbeg - gtle; a_truesc = gscore; }
						if (tk.qbeg + tk.len != tk.l_query) pending = 2;
						else { a_qe = tk.l_query; a_re = tk.rbeg + tk.len; pending = 3; }
					} else {
						const int sc0 = a_score, qe = tk.qbeg + tk.len;
						const int64_t re = tk.rbeg + tk.len;
						a_score = score;
						if (gscore <= 0 || gscore <= a_score - opt.pen_clip3) { a_qe = qe + qle; a_re = re + tle; a_truesc += a_score - sc0; }
						else { a_qe = tk.l_query; a_re = re + gtle; a_truesc += gscore - sc0; }
						pending = 3;
					}
				}
			}
			if (pending == 3) {      // the task's result
				pending = 0;
				if (ok) {
					ExtRes r; r.rb = a_rb; r.re = a_re; r.qb = a_qb; r.qe = a_qe; r.score = a_score; r.truesc = a_truesc;
					res[tk.res] = r;
					valid[tk.res] = 1;
				}
				if (prof) ++n_done;
			}
		}
		// (tasks that ended without a DP side on their first look -- no left flank and no right one)
		if (__ballot(pending == 3)) {
			if (pending == 3) {
				pending = 0;
				if (ok) {
					ExtRes r; r.rb = a_rb; r.re = a_re; r.qb = a_qb; r.qe = a_qe; r.score = a_score; r.truesc = a_truesc;
					res[tk.res] = r;
					valid[tk.res] = 1;
				}
				if (prof) ++n_done;
			}
		}
		const unsigned long long act = __ballot(active);
		if (!act) { if (drained) break; continue; }
		// ---- 3. one row of every running side (ksw_extend2's loop body, literally; eh[] in LDS)
		if (prof) { ++n_rowsteps; n_lanerows += (unsigned long long)__popcll(act); }
		if (active) {
			const int i = S.i, qlen = S.qlen, w = S.w, h0 = S.h0;
			int beg = S.beg, end = S.end;
			const int64_t f = S.f0 + (int64_t)i * S.fstep;
			const uint32_t wd = TW[((int)(f >> 4) - S.wbase) << 6];
			const int tb = (int)((wd >> ((((unsigned)f >> 2) & 3u) << 3) >> ((~(unsigned)f & 3u) << 1)) & 3u) ^ S.comp;
			// the row's scores by query code: 6 bits each (match a, mismatch -b, against an ambiguous base -1)
			uint32_t sc_tab = 0;
#pragma unroll
			for (int q = 0; q < 5; ++q) sc_tab |= ((uint32_t)(q > 3 ? -1 : q == tb ? opt.a : -opt.b) & 63u) << (6 * q);
			if (beg < i - w) beg = i - w;
			if (end > i + w + 1) end = i + w + 1;
			if (end > qlen) end = qlen;
			int h1 = 0;
			if (beg == 0) { h1 = h0 - (opt.o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
			const int h1_init = h1;
			int fgap = 0, mkey = -1;
			int ub = (beg == 0 && h1_init > 0) ? h1_init + qlen * mxsc : -1;
			int rem = (qlen - 1 - beg) * mxsc;
			uint32_t wnext = L[beg << 6];
			int ncell = 0;
			for (int j = beg; j < end; ++j) {
				const uint32_t wv = wnext;
				wnext = L[(j + 1) << 6];
				int M = (int)(wv & XL_HMASK);
				const int e = (int)((wv >> 13) & XL_HMASK);
				const int s = xl_sbfe6(sc_tab, wv >> 26);
				M = M ? M + s : 0;
				int h = M > e ? M : e;
				h = h > fgap ? h : fgap;
				int t = M - oe_del; t = t > 0 ? t : 0;
				int e2 = e - e_del; e2 = e2 > t ? e2 : t;
				L[j << 6] = (uint32_t)h1 | (uint32_t)e2 << 13 | (wv & XL_QMASK);
				h1 = h;
				const int key = h << 9 | j;
				mkey = mkey > key ? mkey : key;
				t = M - oe_ins; t = t > 0 ? t : 0;
				fgap -= e_ins; fgap = fgap > t ? fgap : t;
				const int u = h > 0 ? h + rem : -1;
				ub = ub > u ? ub : u;
				rem -= mxsc;
				++ncell;
			}
			if (prof) n_cells += (unsigned long long)ncell;
			{
				const uint32_t we = L[end << 6];
				L[end << 6] = (uint32_t)h1 | (we & XL_QMASK);      // eh[end].h = h1; eh[end].e = 0
			}
			const int jfin = end > beg ? end : beg;
			if (jfin == qlen) {
				S.max_ie = S.gscore > h1 ? S.max_ie : i;
				S.gscore = S.gscore > h1 ? S.gscore : h1;
			}
			const int m = mkey < 0 ? 0 : mkey >> 9, mj = mkey < 0 ? -1 : mkey & 511;
			bool stop = m == 0;
			if (!stop) {
				if (m > S.mx) {
					S.mx = m; S.max_i = i; S.max_j = mj;
					const int off = mj - i < 0 ? i - mj : mj - i;
					S.max_off = S.max_off > off ? S.max_off : off;
				} else if (opt.zdrop > 0) {
					if (i - S.max_i > mj - S.max_j) { if (S.mx - m - ((i - S.max_i) - (mj - S.max_j)) * e_del > opt.zdrop) stop = true; }
					else { if (S.mx - m - ((mj - S.max_j) - (i - S.max_i)) * e_ins > opt.zdrop) stop = true; }
				}
			}
			// the exact early exit once the query's end has been reached (dev_dp.hpp: no later cell can exceed ub)
			if (!stop && i >= qlen - 1 && S.gscore > 0 && S.gscore > ub) stop = true;
			if (!stop) {
				int j;
				for (j = beg; j < end && (L[j << 6] & ~XL_QMASK) == 0; ++j) {}
				beg = j;
				for (j = end; j >= beg && (L[j << 6] & ~XL_QMASK) == 0; --j) {}
				end = j + 2 < qlen ? j + 2 : qlen;
				S.beg = beg; S.end = end; S.i = i + 1;
				if (S.i >= S.tlen) stop = true;
			}
			if (stop) { active = false; pending = 4; }
		}
	}
	if (prof && lane == 0) {
		atomicAdd(prof + 0, __builtin_amdgcn_s_memtime() - t_start); atomicAdd(prof + 1, 1ULL); atomicAdd(prof + 2, n_rowsteps); atomicAdd(prof + 3, n_lanerows);
	}
	if (prof) {
		unsigned long long c = n_cells, d = n_done, p = n_dp;
		for (int mm = 1; mm < 64; mm <<= 1) { c += __shfl_xor(c, mm); d += __shfl_xor(d, mm); p += __shfl_xor(p, mm); }
		if (lane == 0) { atomicAdd(prof + 4, c); atomicAdd(prof + 5, d); atomicAdd(prof + 6, p); }
	}
}

// ---------------------------------------------------------------------------------------------------------------------------
// Planning for K2a's hand-overs (dev_types.h, HandHdr): one lane per record.  For every chain mem_chain2aln will still visit
// (filtered order, from chain_from on, kept != 0): the seed it extends FIRST -- the last one of ks_introsort's order on
// (length << 32 | index in the chain), i.e. the longest, the later one on ties -- and the chain's window as mode 3 of k_align.hip
// plans it.  One task per such chain, its result slot = record * EMA_HAND_SEEDS + chain (filtered order).
__global__ void __launch_bounds__(256)
ema_k_ext_plan_hand(DevIndex ix, DevOpts opt, const uint8_t *__restrict__ hand, const int *__restrict__ n_hand, ExtTask *__restrict__ tasks, int *__restrict__ n_tasks,
                    int tasks_cap, uint8_t *__restrict__ valid)
{
	const int rec = (int)(blockIdx.x * 256 + threadIdx.x);
	const int lane = (int)ema_lane();
	const int total = *n_hand;
	const bool live = rec < total;
	const uint8_t *h = hand + (size_t)(live ? rec : 0) * EMA_HAND_BYTES;
	HandHdr hd;
	hd.read = hd.n_chn = hd.n_seed = hd.l_query = hd.chain_from = hd.n_av = hd.pad = 0; hd.base_off = 0;
	if (live) hd = *reinterpret_cast<const HandHdr *>(h);
	const uint64_t *hk = reinterpret_cast<const uint64_t *>(h + sizeof(HandHdr));
	const ChainRec *hc = reinterpret_cast<const ChainRec *>(h + sizeof(HandHdr) + EMA_HAND_SEEDS * 8);
	const SeedRec *hs = reinterpret_cast<const SeedRec *>(h + sizeof(HandHdr) + EMA_HAND_SEEDS * (8 + sizeof(ChainRec)));
	const int64_t l_pac = ix.l_pac;
	if (live) {      // nothing computed yet for this record
		uint64_t *v = reinterpret_cast<uint64_t *>(valid + (size_t)rec * EMA_HAND_SEEDS);
#pragma unroll
		for (int k = 0; k < EMA_HAND_SEEDS / 8; ++k) v[k] = 0;
	}
	int cmax = live ? hd.n_chn : 0;
	for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(cmax, m); cmax = cmax > o ? cmax : o; }
	for (int cs = 0; cs < cmax; ++cs) {
		bool emit = false;
		ExtTask t;
		t.rbeg = t.rmax0 = t.rmax1 = 0; t.q_off = 0; t.l_query = t.qbeg = t.len = t.res = 0; t.pad = 0;
		if (live && cs >= hd.chain_from && cs < hd.n_chn) {
			const ChainRec c = hc[(int)(uint32_t)hk[cs]];
			if (c.kept != 0) {
				int64_t rmax0 = l_pac << 1, rmax1 = 0;
				int k = c.first_seed, best_len = -1;
				SeedRec best; best.rbeg = 0; best.qbeg = best.len = best.next = best.pad = 0;
				for (int i = 0; i < c.n; ++i) {
					const SeedRec s = hs[k];
					const int64_t b = s.rbeg - (s.qbeg + ema_cal_max_gap(opt, s.qbeg));
					const int tail = hd.l_query - s.qbeg - s.len;
					const int64_t e = s.rbeg + s.len + (tail + ema_cal_max_gap(opt, tail));
					rmax0 = rmax0 < b ? rmax0 : b;
					rmax1 = rmax1 > e ? rmax1 : e;
					if (s.len >= best_len) { best_len = s.len; best = s; }      // (length << 32 | i): the later seed wins a tie
					k = s.next;
				}
				rmax0 = rmax0 > 0 ? rmax0 : 0;
				rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
				if (rmax0 < l_pac && l_pac < rmax1) {
					if (c.f_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
				}
				ema_clamp_window_rid(ix, rmax0, c.rid, c.f_rbeg >= l_pac, rmax1);
				if (rmax1 - rmax0 <= EMA_RSEQ_CAP && best_len > 0) {
					t.rbeg = best.rbeg; t.rmax0 = rmax0; t.rmax1 = rmax1; t.q_off = hd.base_off; t.l_query = hd.l_query; t.qbeg = best.qbeg; t.len = best.len;
					t.res = rec * EMA_HAND_SEEDS + cs;
					emit = true;
				}
			}
		}
		const unsigned long long em = __ballot(emit);
		if (em) {
			int base = 0;
			if (lane == (__ffsll((long long)em) - 1)) base = atomicAdd(n_tasks, __popcll(em));
			base = __shfl(base, __ffsll((long long)em) - 1);
			const int at = base + __popcll(em & ((1ULL << lane) - 1));
			if (emit && at < tasks_cap) tasks[at] = t;
		}
	}
}

extern "C" void ema_launch_ext_plan_hand(const DevIndex *ix, const DevOpts *opt, const uint8_t *hand, const int *n_hand, int max_records, ExtTask *tasks, int *n_tasks,
                                         int tasks_cap, uint8_t *valid, hipStream_t stream)
{
	if (max_records <= 0) return;
	hipLaunchKernelGGL(ema_k_ext_plan_hand, dim3((unsigned)((max_records + 255) / 256)), dim3(256), 0, stream, *ix, *opt, hand, n_hand, tasks, n_tasks, tasks_cap, valid);
}

// The tasks by the longer of their two queries: up to 63 bases (24 KB of LDS per wavefront), up to 127 (40 KB), up to 255 (72 KB);
// each launch walks the whole list and takes its class.  counters: one claim counter per class (zero on entry).
extern "C" void ema_launch_ext_lane(const DevIndex *ix, const DevOpts *opt, const uint8_t *bases, const ExtTask *tasks, const int *n_tasks, int tasks_cap,
                                    ExtRes *res, uint8_t *valid, int *counters, int n_cu, hipStream_t stream, unsigned long long *prof)
{
	hipLaunchKernelGGL((ema_k_ext_lane<63>), dim3((unsigned)(n_cu * 6)), dim3(64), 0, stream, *ix, *opt, bases, tasks, n_tasks, tasks_cap, res, valid, counters + 0, 0, prof);
	hipLaunchKernelGGL((ema_k_ext_lane<127>), dim3((unsigned)(n_cu * 4)), dim3(64), 0, stream, *ix, *opt, bases, tasks, n_tasks, tasks_cap, res, valid, counters + 1, 64, prof ? prof + 8 : nullptr);
	hipLaunchKernelGGL((ema_k_ext_lane<255>), dim3((unsigned)(n_cu * 2)), dim3(64), 0, stream, *ix, *opt, bases, tasks, n_tasks, tasks_cap, res, valid, counters + 2, 128, prof ? prof + 16 : nullptr);
}

// the lane route needs scores that fit its packing: 13 bits of H / E per column, 6-bit signed substitution scores
extern "C" int ema_ext_lane_supported(const DevOpts *opt)
{
	return opt->a >= 0 && opt->a <= 31 && opt->b >= 0 && opt->b <= 32 && (long)opt->a * (2 * EMA_MAX_READ + 4) < 8000 && opt->e_ins > 0 && opt->e_del > 0;
}
