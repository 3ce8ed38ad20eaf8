// ema_amd/csrc/k_pair.hip -- K3: mate rescue, one wavefront per read pair.
//
// Replaces the two rescue loops of the reference's bridge (reference src/bwabridge.c:242-283) and the
// mem_matesw / ksw_align2 they call in the un-vendored bwa: for up to 50 hits of mate 2 within 25 of its best
// score, look for mate 1 in the window an FR pair with insert in [-35, 500] allows (local Smith-Waterman of
// the reverse-complemented mate), insert what is found into mate 1's region list and re-run the dedup; then
// the same for mate 2 around the hits of the updated mate-1 list, with the threshold taken from mate 1's
// best score BEFORE rescue.
//
// Each mem_matesw call first looks for a region of the other mate that already forms a consistent pair and
// returns at once if there is one -- and every call may change that list -- so the calls of a pair are
// sequential; the wave runs them in order and spreads the consistency test (over regions), the window
// fetch and the DP rows over its lanes.  Region lists are edited in a per-wave slab of HBM scratch and
// written back in place.
#include <hip/hip_runtime.h>
#include "dev_regions.hpp"
#include "dev_prof.hpp"
#ifdef EMA_K34_PROF
__device__ unsigned long long ema_k3_lp[3][12];
extern "C" void ema_k3_prof_read(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ema_k3_lp), sizeof(ema_k3_lp)); static unsigned long long z[36]; (void)hipMemcpyToSymbol(HIP_SYMBOL(ema_k3_lp), z, sizeof(z)); }
#endif
#include <cstring>

#define EMA_PAIR_SLAB_BYTES ((size_t)EMA_AV_CAP * (3 * sizeof(DevReg) + 8) + 2048 * 8 + 1024)

namespace {

struct PairCtx {
	EmaLp lp;               // (make prof-lib: phase clocks, dev_prof.hpp) 0 claim, 1 the pair in, 2 anchors and "already there?", 3 window, 4 local DP forward, 5 backward, 6 insertion + dedup, 7 results out
	const DevIndex *ix;
	const DevOpts *opt;
	uint8_t *rc, *rseq;     // LDS: reverse complement of the mate being rescued; reference window
	uint64_t *bsc;          // scratch for the local DP's b[] list
	EmaRegWork wk;
	int pes_low, pes_high;
	int status;
	int n_sw = 0;           // alignments run so far (mem_matesw's return value)
};

__device__ __forceinline__ int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t &dist)
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// mem_matesw with the reference's insert model (only FR allowed, pes[1] = {low, high}), in three steps so that the alignment --
// which depends on the anchor and the mate's bases only -- can also run ahead of the in-order decisions (K3t / K3r below):
//   matesw_found  an FR-consistent region already in ma = wk.a[0..n_ma)?  (then mem_matesw returns at once)
//   matesw_sw     the rescue window and the local alignment: SW_NONE (window unusable: mem_matesw returns before aligning),
//                 SW_RAN (aligned, nothing to insert) or SW_REGION (b is the region found); status bits in st
//   matesw_apply  b inserted in score order, then mem_sort_dedup_patch without patching (runs whenever the alignment ran)
enum { SW_NONE = 0, SW_RAN = 1, SW_REGION = 2 };

__device__ inline bool matesw_found(const PairCtx &cx, const DevReg &a, int n_ma)
{
	const int lane = (int)ema_lane();
	const int64_t l_pac = cx.ix->l_pac;
	bool found = false;
	for (int base = 0; base < n_ma && !found; base += EMA_WAVE) {
		bool hit = false;
		const int i = base + lane;
		if (i < n_ma) {
			int64_t dist;
			const int r = infer_dir(l_pac, a.rb, cx.wk.a[i].rb, dist);
			hit = r == 1 && dist >= cx.pes_low && dist <= cx.pes_high;
		}
		if (__ballot(hit)) found = true;
	}
	return found;
}

__device__ inline int matesw_sw(PairCtx &cx, const DevReg &a, int l_ms, const uint8_t *ms, DevReg &b, int &st)
{
	const DevIndex &ix = *cx.ix;
	const DevOpts &o = *cx.opt;
	const int lane = (int)ema_lane();
	const int64_t l_pac = ix.l_pac;
	// r = 1: the mate is reverse-complemented and lies at larger coordinates
	for (int i = lane; i < l_ms; i += EMA_WAVE) { const int c = ms[i]; cx.rc[l_ms - 1 - i] = (uint8_t)(c < 4 ? 3 - c : 4); }
	ema_wave_sync();
	int64_t rb = a.rb + cx.pes_low - l_ms, re = a.rb + cx.pes_high;
	if (rb < 0) rb = 0;
	if (re > l_pac << 1) re = l_pac << 1;
	int rid = -1;
	if (rb < re) rid = ema_clamp_window(ix, rb, (rb + re) >> 1, re);
	if (!(a.rid == rid && re - rb >= o.min_seed_len)) return SW_NONE;
	if (re - rb > EMA_RSEQ_CAP) { st |= EMA_ST_RSEQ_OVERFLOW; return SW_NONE; }
	const int tlen = (int)(re - rb);
	EMA_LP_UPTO(cx.lp, 2);
	ema_wave_fetch(ix, rb, re, cx.rseq);
	EMA_LP_UPTO(cx.lp, 3);
	// ksw_align2 with KSW_XSUBO | KSW_XSTART | (l_ms * a < 250 ? KSW_XBYTE : 0) | min_seed_len * a
	const int p = l_ms * o.a < 250 ? 16 : 8;
	const int minsc = o.min_seed_len * o.a;
	const EmaLocalRes r1 = ema_wave_local(o, l_ms, p, EmaSeq{cx.rc, 1}, tlen, EmaSeq{cx.rseq, 1}, minsc, 0x10000, cx.bsc);
	EMA_LP_UPTO(cx.lp, 4);
	int qb = -1, tb = -1;
	if (r1.score >= minsc) {      // start coordinates: the same DP on the reversed prefixes, stopping at the score
		EmaSeq tq{cx.rc + r1.qe, -1};
		EmaSeq tt{cx.rseq, 1, r1.te};
		const EmaLocalRes r2 = ema_wave_local(o, r1.qe + 1, p, tq, tlen, tt, 0x10000, r1.score, cx.bsc);
		if (r1.score == r2.score) { tb = r1.te - r2.te; qb = r1.qe - r2.qe; }
		EMA_LP_UPTO(cx.lp, 5);
	}
	if (!(r1.score >= o.min_seed_len && qb >= 0)) return SW_RAN;
	b.rid = a.rid; b.is_alt = a.is_alt;
	b.qb = l_ms - (r1.qe + 1); b.qe = l_ms - qb;
	b.rb = (l_pac << 1) - (rb + r1.te + 1); b.re = (l_pac << 1) - (rb + tb);
	b.score = r1.score; b.csub = r1.score2; b.secondary = -1;
	b.truesc = 0; b.sub = 0; b.w = 0; b.seedlen0 = 0; b.n_comp = 0; b.frac_rep = 0.f;
	b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
	return SW_REGION;
}

__device__ inline int matesw_apply(PairCtx &cx, int what, const DevReg &b, int n_ma)
{
	const int lane = (int)ema_lane();
	EMA_LP_UPTO(cx.lp, 2);
	++cx.n_sw;
	int n = n_ma;
	if (what == SW_REGION) {
		if (n_ma >= EMA_AV_CAP) { cx.status |= EMA_ST_REG_OVERFLOW; }
		else {
			ema_wave_sync();
			if (lane == 0) {      // keep the list ordered: before the first element with a smaller score
				int at = 0;
				while (at < n_ma && !(cx.wk.a[at].score < b.score)) ++at;
				for (int i = n_ma; i > at; --i) cx.wk.a[i] = cx.wk.a[i - 1];
				cx.wk.a[at] = b;
			}
			ema_wave_sync();
			n = n_ma + 1;
		}
	}
	const int n_out_ = ema_sort_dedup_patch(*cx.ix, *cx.opt, nullptr, n, cx.wk, cx.status);     // runs whenever the SW ran
	EMA_LP_UPTO(cx.lp, 6);
	return n_out_;
}

__device__ inline int matesw(PairCtx &cx, const DevReg &a, int l_ms, const uint8_t *ms, int n_ma)
{
	if (matesw_found(cx, a, n_ma)) return n_ma;
	DevReg b;
	const int what = matesw_sw(cx, a, l_ms, ms, b, cx.status);
	if (what == SW_NONE) return n_ma;
	return matesw_apply(cx, what, b, n_ma);
}

}  // namespace

// K3a: which pairs need any rescue alignment at all?  ONE LANE PER PAIR.  mem_matesw first looks for a region of the mate
// that already forms a consistent pair with the anchor and, failing that, checks that the rescue window is usable;
// only then does it align.  For a properly paired read pair -- most of a bucket -- every anchor passes the first test
// and the pair's region lists stay exactly as K2 left them.  This pass evaluates those tests (a few dozen scalar
// operations per pair; K3b would issue them 64 wide) and lists the pairs for which an alignment would run.
__global__ void __launch_bounds__(256)
ema_k_pair_simple(DevIndex ix, DevOpts opt, int score_delta, int max_rescue, int pes_low, int pes_high,
                  const uint32_t *__restrict__ off, int n_pairs, const int *__restrict__ n_pairs_dev, const int *__restrict__ map,
                  const DevReg *__restrict__ regs, const int *__restrict__ n_regs, const int *__restrict__ status,
                  int *__restrict__ todo, int *__restrict__ n_todo)
{
	const int pair = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (pair >= ema_work_count(n_pairs, n_pairs_dev, 1)) return;
	if (status[2 * pair] | status[2 * pair + 1]) return;      // redone by the full-capacity tier
	const int64_t l_pac = ix.l_pac;
	int n[2], best[2] = {0, 0}, len[2];
	for (int m = 0; m < 2; ++m) {
		const int r = 2 * pair + m, in_r = ema_in_read(map, r);
		len[m] = (int)(off[in_r + 1] - off[in_r]);
		n[m] = n_regs[r];
		const DevReg *src = regs + (size_t)r * opt.reg_cap;
		for (int i = 0; i < n[m]; ++i) { const int sc = src[i].score; best[m] = best[m] > sc ? best[m] : sc; }
	}
	bool need = false;
	for (int dirn = 0; dirn < 2 && !need; ++dirn) {      // reference src/bwabridge.c:263-269, :277-283
		const int anchor = dirn == 0 ? 1 : 0, target = 1 - anchor;
		const DevReg *an = regs + (size_t)(2 * pair + anchor) * opt.reg_cap, *tg = regs + (size_t)(2 * pair + target) * opt.reg_cap;
		int num = 0;
		for (int k = 0; k < n[anchor] && num < max_rescue && !need; ++k) {
			const DevReg a = an[k];
			if (a.score < best[anchor] - score_delta) continue;
			++num;
			bool found = false;
			for (int i = 0; i < n[target] && !found; ++i) {
				int64_t dist;
				const int r = infer_dir(l_pac, a.rb, tg[i].rb, dist);
				found = r == 1 && dist >= pes_low && dist <= pes_high;
			}
			if (found) continue;
			int64_t rb = a.rb + pes_low - len[target], re = a.rb + pes_high;
			if (rb < 0) rb = 0;
			if (re > l_pac << 1) re = l_pac << 1;
			int rid = -1;
			if (rb < re) rid = ema_clamp_window(ix, rb, (rb + re) >> 1, re);
			if (a.rid == rid && re - rb >= opt.min_seed_len) need = true;      // mem_matesw would align here
		}
	}
	if (need) todo[atomicAdd(n_todo, 1)] = pair;
}

// A pair with many rescue attempts (a pair inside a repeat family in the full-capacity tier: up to 50 + 50 local alignments of
// ~700 reference bases each) is a long serial job for the wavefront that owns it: 20 of the full tier's 105 ms (r02).  What an attempt
// ALIGNS depends on its anchor and the mate's bases only; what is sequential is whether the attempt runs at all (a consistent region
// may be there already, possibly put there by an earlier attempt) and the list edit afterwards.  K3b therefore sets such a pair aside:
//   K3t (ema_k_pair_t<1>): one wavefront per attempt runs matesw_sw and leaves its outcome in the pair's record;
//   K3r (ema_k_pair_t<2>, direction 0 then 1): one wavefront per pair replays the attempts of one direction in order --
//        matesw_found on the list as it stands, then matesw_apply with the recorded outcome: no alignment runs here.  After direction 0
//        it lists the attempts of direction 1, whose anchors are the updated mate-1 list; after direction 1 it writes the pair back.
// Lists and arena are K2's (HeavyCtl: idle while K3 runs); a pair that finds no room there is done in place.
#define EMA_PAIR_MAX_RESCUE 64      // slots of recorded outcomes per direction (the reference tries at most 50, src/bwabridge.c:264,278)
struct PairHeavy {
	uint8_t *arena;                    // null: nothing is set aside
	unsigned long long arena_bytes;
	unsigned long long *arena_used;    // bump allocator (bytes), zero on entry
	unsigned long long *pairs;         // record offsets of the pairs set aside
	unsigned long long *tasks;         // [dirn * tasks_cap + i]: record offset / 64 << 32 | anchor index << 8 | slot
	int *n_pairs, *n_tasks;            // n_tasks[2]: per direction
	int pairs_cap, tasks_cap;
	int min_attempts;                  // a pair with at least this many candidate anchors (both directions) is set aside
};
struct PairRes { DevReg b; int32_t what, st; };
struct PairHdr {                       // head of a record; arrays at the offsets given (bytes from the record's start)
	int32_t pair, n[2], len[2], best[2], cap[2], status, done1, pad_;
	uint64_t off_av[2], off_res, bytes;
};

#ifndef EMA_K3_MIN_BLOCKS
#define EMA_K3_MIN_BLOCKS 1      // (4 costs K3b 40 spilled registers and changed nothing, r03at)
#endif
// K3b: the pairs K3a listed.  MODE 0: that; 1: K3t; 2: K3r (above), dirn_arg = the direction it handles.
// regs/n_regs: K2's output, updated in place.  One wave per pair, pairs taken from a shared counter.
template <int MODE>
__global__ void __launch_bounds__(256, EMA_K3_MIN_BLOCKS)
ema_k_pair_t(DevIndex ix, DevOpts opt, int score_delta, int max_rescue, int pes_low, int pes_high,
           const uint8_t *__restrict__ bases, const uint32_t *__restrict__ off, int n_pairs,
           const int *__restrict__ n_pairs_dev, const int *__restrict__ map, DevReg *__restrict__ regs,
           int *__restrict__ n_regs, int *__restrict__ status, const int *__restrict__ todo, const int *__restrict__ n_todo,
           uint8_t *__restrict__ slabs, int *__restrict__ counter, int *dbg, PairHeavy ph, int dirn_arg)
{
#define EMA_DBG(stage, val) do { if (dbg && lane == 0) { __hip_atomic_store(dbg + slot * 4 + 1, (stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(dbg + slot * 4 + 2, (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } } while (0)
	__shared__ uint8_t lds_q[4][2][256];
	__shared__ uint8_t lds_rc[4][256];
	__shared__ uint8_t lds_r[4][EMA_RSEQ_CAP];
	__shared__ int lds_stack[4][3 * 70];
	const int lane = (int)ema_lane();
	const int wib = ema_uni((int)(threadIdx.x >> 6));      // scalar: slab and LDS pointers derived from it stay in SGPRs
	const int slot = (int)(blockIdx.x * (blockDim.x >> 6)) + wib;
	uint8_t *slab = slabs + (size_t)slot * EMA_PAIR_SLAB_BYTES;
	DevReg *av[2] = {(DevReg *)slab, (DevReg *)slab + EMA_AV_CAP};
	PairCtx cx;
	cx.ix = &ix; cx.opt = &opt;
	cx.rc = lds_rc[wib]; cx.rseq = lds_r[wib];
	cx.wk.tmp = (DevReg *)slab + 2 * EMA_AV_CAP;
	cx.wk.keys = (uint64_t *)((DevReg *)slab + 3 * EMA_AV_CAP);
	cx.bsc = cx.wk.keys + EMA_AV_CAP;
	cx.wk.stack = lds_stack[wib];
	cx.wk.rseq = lds_r[wib];
	cx.pes_low = pes_low; cx.pes_high = pes_high;
	if (max_rescue > EMA_PAIR_MAX_RESCUE) ph.arena = nullptr;      // (more attempts than a record has slots for: everything in place)

	EmaClaim claim;      // work items four at a time, with their list entries (dev_common.hpp)
#ifdef EMA_K34_PROF
	cx.lp.start();
#endif
	for (;;) {
		EMA_LP_UPTO(cx.lp, 7);
		int pair = 0;
		unsigned long long list_entry = 0;
		if (MODE == 1) { const int n = ph.n_tasks[dirn_arg]; pair = ema_claim_next(claim, counter, n < ph.tasks_cap ? n : ph.tasks_cap, ph.tasks + (size_t)dirn_arg * ph.tasks_cap, list_entry); }
		else if (MODE == 2) { const int n = *ph.n_pairs; pair = ema_claim_next(claim, counter, n < ph.pairs_cap ? n : ph.pairs_cap, ph.pairs, list_entry); }
		else {
			int t = 0;
			pair = ema_claim_next(claim, counter, todo ? *n_todo : ema_work_count(n_pairs, n_pairs_dev, 1), todo, t);
			if (todo) list_entry = (unsigned long long)(unsigned)t;
		}
		if (pair < 0) break;
		EMA_LP_UPTO(cx.lp, 0); EMA_LP_ITEM(cx.lp);
		if (MODE == 1) {      // K3t: one attempt
			const unsigned long long t = list_entry;
			if (t == ~0ULL) continue;
			uint8_t *rec = ph.arena + (size_t)(t >> 32) * 64;
			const PairHdr *h = reinterpret_cast<const PairHdr *>(rec);
			const int k = (int)((uint32_t)t >> 8), rslot = (int)(t & 0xff);
			const int anchor = dirn_arg == 0 ? 1 : 0, target = 1 - anchor;
			const int in_r = ema_uni(ema_in_read(map, 2 * ema_uni(h->pair) + target));
			const int l_ms = ema_uni(h->len[target]);
			for (int i = lane; i < l_ms; i += EMA_WAVE) lds_q[wib][0][i] = bases[off[in_r] + i];
			ema_wave_sync();
			const DevReg a = ema_uni(reinterpret_cast<const DevReg *>(rec + ema_uni(h->off_av[anchor]))[k]);
			EMA_LP_UPTO(cx.lp, 1);
			DevReg b;
			memset(&b, 0, sizeof(b));
			int st = 0;
			const int what = matesw_sw(cx, a, l_ms, lds_q[wib][0], b, st);
			if (lane == 0) {
				PairRes r; r.b = b; r.what = what; r.st = st;
				reinterpret_cast<PairRes *>(rec + h->off_res)[dirn_arg * EMA_PAIR_MAX_RESCUE + rslot] = r;
			}
			ema_wave_sync();
			continue;
		}
		if (MODE == 2) {      // K3r: the attempts of direction dirn_arg replayed in order
			const unsigned long long t = list_entry;
			if (t == ~0ULL) continue;
			uint8_t *rec = ph.arena + t;
			PairHdr *h = reinterpret_cast<PairHdr *>(rec);
			const int dirn = dirn_arg;
			const int anchor = dirn == 0 ? 1 : 0, target = 1 - anchor;
			pair = ema_uni(h->pair);
			const int n_anchor = ema_uni(h->n[anchor]), best_a = ema_uni(h->best[anchor]);
			int n_t = ema_uni(h->n[target]);
			DevReg *r_anchor = reinterpret_cast<DevReg *>(rec + ema_uni(h->off_av[anchor]));
			DevReg *r_target = reinterpret_cast<DevReg *>(rec + ema_uni(h->off_av[target]));
			const PairRes *res = reinterpret_cast<const PairRes *>(rec + ema_uni(h->off_res)) + dirn * EMA_PAIR_MAX_RESCUE;
			cx.status = ema_uni(h->status);
			cx.wk.a = av[target];
			if (!(dirn == 1 && ema_uni(h->done1))) {
				for (int i = lane; i < n_t; i += EMA_WAVE) av[target][i] = r_target[i];
				ema_wave_sync();
				EMA_LP_UPTO(cx.lp, 1);
				int num = 0;
				for (int k = 0; k < n_anchor && num < max_rescue; ++k) {
					const DevReg a = ema_uni(r_anchor[k]);
					if (a.score < best_a - score_delta) continue;
					const int rslot = num++;
					if (matesw_found(cx, a, n_t)) continue;
					const PairRes r = res[rslot];
					const int what = ema_uni(r.what);
					cx.status |= ema_uni(r.st);
					if (what != SW_NONE) n_t = matesw_apply(cx, what, ema_uni(r.b), n_t);
				}
				ema_wave_sync();
				for (int i = lane; i < n_t; i += EMA_WAVE) r_target[i] = av[target][i];
				if (lane == 0) { h->n[target] = n_t; h->status = cx.status; }
				ema_wave_sync();
			}
			if (dirn == 0) {      // the attempts of direction 1: anchors = the updated list of mate 1, threshold from its best BEFORE rescue
				const int best0 = ema_uni(h->best[0]);
				int cnt = 0;
				for (int base = 0; base < n_t && cnt < max_rescue; base += EMA_WAVE) {
					const int i = base + lane;
					cnt += __popcll(__ballot(i < n_t && av[0][i].score >= best0 - score_delta));
				}
				cnt = cnt < max_rescue ? cnt : max_rescue;
				long long tb = -1;
				if (lane == 0 && cnt > 0) {
					tb = atomicAdd(ph.n_tasks + 1, cnt);
					if (tb + cnt > ph.tasks_cap) {
						for (long long j = tb; j < ph.tasks_cap && j < tb + cnt; ++j) ph.tasks[(size_t)ph.tasks_cap + j] = ~0ULL;
						tb = -1;
					}
				}
				tb = (long long)ema_uni((int64_t)__shfl(tb, 0));
				if (cnt > 0 && tb < 0) {      // no room on the list: direction 1 in place, here
					const int in_r = ema_uni(ema_in_read(map, 2 * pair + 1));
					const int l_ms = ema_uni(h->len[1]);
					for (int i = lane; i < l_ms; i += EMA_WAVE) lds_q[wib][1][i] = bases[off[in_r] + i];
					int n1 = ema_uni(h->n[1]);
					DevReg *r1 = reinterpret_cast<DevReg *>(rec + ema_uni(h->off_av[1]));
					for (int i = lane; i < n1; i += EMA_WAVE) av[1][i] = r1[i];
					ema_wave_sync();
					cx.wk.a = av[1];
					int num = 0;
					for (int k = 0; k < n_t && num < max_rescue; ++k) {
						const DevReg a = ema_uni(av[0][k]);
						if (a.score >= best0 - score_delta) { ++num; n1 = matesw(cx, a, l_ms, lds_q[wib][1], n1); }
					}
					ema_wave_sync();
					for (int i = lane; i < n1; i += EMA_WAVE) r1[i] = av[1][i];
					if (lane == 0) { h->n[1] = n1; h->status = cx.status; h->done1 = 1; }
					ema_wave_sync();
				} else if (cnt > 0) {
					int num = 0;
					for (int k = 0; k < n_t && num < max_rescue; ++k) {      // (lane 0 walks the list: a few dozen entries)
						const int sc = ema_uni(av[0][k].score);
						if (sc < best0 - score_delta) continue;
						if (lane == 0) ph.tasks[(size_t)ph.tasks_cap + tb + num] = (unsigned long long)((size_t)t >> 6) << 32 | (unsigned long long)(uint32_t)k << 8 | (uint32_t)num;
						++num;
					}
				}
				continue;
			}
			// direction 1 done: the pair's lists go back where K2 left them
			for (int m = 0; m < 2; ++m) {
				const int r = 2 * pair + m;
				int cnt = ema_uni(h->n[m]);
				if (m == target) cnt = n_t;
				const DevReg *src = reinterpret_cast<const DevReg *>(rec + ema_uni(h->off_av[m]));
				if (cnt > opt.reg_cap) { cx.status |= EMA_ST_REG_OVERFLOW; cnt = opt.reg_cap; }
				DevReg *dst = regs + (size_t)r * opt.reg_cap;
				for (int i = lane; i < cnt; i += EMA_WAVE) dst[i] = src[i];
				if (lane == 0) { n_regs[r] = cnt; if (cx.status) atomicOr(status + r, cx.status); }
			}
			ema_wave_sync();
			continue;
		}
		if (todo) pair = (int)(unsigned)list_entry;
		if (dbg && lane == 0) __hip_atomic_store(dbg + slot * 4, pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		EMA_DBG(1, 0);
		if (ema_uni(status[2 * pair] | status[2 * pair + 1])) { EMA_DBG(9, 0); continue; }      // redone by the full-capacity tier
		int len[2], n[2], best[2] = {0, 0};
		for (int m = 0; m < 2; ++m) {
			const int r = 2 * pair + m;
			const int in_r = ema_uni(ema_in_read(map, r));
			len[m] = ema_uni((int)(off[in_r + 1] - off[in_r]));
			n[m] = ema_uni(n_regs[r]);
			for (int i = lane; i < len[m]; i += EMA_WAVE) lds_q[wib][m][i] = bases[off[in_r] + i];
			const DevReg *src = regs + (size_t)r * opt.reg_cap;
			int b = 0;
			for (int i = lane; i < n[m]; i += EMA_WAVE) { const DevReg x = src[i]; av[m][i] = x; b = b > x.score ? b : x.score; }
			best[m] = ema_wave_max(b);
		}
		ema_wave_sync();
		EMA_LP_UPTO(cx.lp, 1);
		cx.status = 0;
		if (ph.arena) {      // many attempts ahead?  then the pair is set aside (see above) -- if the lists and the arena have room
			int cand[2];
			for (int m = 0; m < 2; ++m) {
				int c = 0;
				for (int base = 0; base < n[m]; base += EMA_WAVE) { const int i = base + lane; c += __popcll(__ballot(i < n[m] && av[m][i].score >= best[m] - score_delta)); }
				cand[m] = c < max_rescue ? c : max_rescue;
			}
			if (cand[0] + cand[1] >= ph.min_attempts && cand[1] > 0) {
				auto up64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
				PairHdr h;
				memset(&h, 0, sizeof(h));
				h.pair = pair;
				for (int m = 0; m < 2; ++m) { h.n[m] = n[m]; h.len[m] = len[m]; h.best[m] = best[m]; h.cap[m] = n[m] + EMA_PAIR_MAX_RESCUE; }
				h.off_res = up64(sizeof(PairHdr));
				h.off_av[0] = h.off_res + up64((size_t)2 * EMA_PAIR_MAX_RESCUE * sizeof(PairRes));
				h.off_av[1] = h.off_av[0] + up64((size_t)h.cap[0] * sizeof(DevReg));
				h.bytes = h.off_av[1] + up64((size_t)h.cap[1] * sizeof(DevReg));
				long long pi = -1, tb = -1, at = -1;
				if (lane == 0) {
					pi = atomicAdd(ph.n_pairs, 1);
					if (pi >= ph.pairs_cap) pi = -1;
					if (pi >= 0) {
						tb = atomicAdd(ph.n_tasks, cand[1]);
						if (tb + cand[1] > ph.tasks_cap) {
							for (long long j = tb; j < ph.tasks_cap && j < tb + cand[1]; ++j) ph.tasks[j] = ~0ULL;
							tb = -1;
						}
					}
					if (tb >= 0) {
						at = (long long)atomicAdd(ph.arena_used, (unsigned long long)h.bytes);
						if ((unsigned long long)at + h.bytes > ph.arena_bytes) {
							for (long long j = tb; j < tb + cand[1]; ++j) ph.tasks[j] = ~0ULL;
							at = -1;
						}
					}
					if (pi >= 0) ph.pairs[pi] = at >= 0 ? (unsigned long long)at : ~0ULL;
				}
				at = (long long)ema_uni((int64_t)__shfl(at, 0)); tb = (long long)ema_uni((int64_t)__shfl(tb, 0));
				if (at >= 0) {
					uint8_t *rec = ph.arena + at;
					if (lane == 0) *reinterpret_cast<PairHdr *>(rec) = h;
					for (int m = 0; m < 2; ++m) {
						DevReg *d = reinterpret_cast<DevReg *>(rec + h.off_av[m]);
						for (int i = lane; i < n[m]; i += EMA_WAVE) d[i] = av[m][i];
					}
					int num = 0;
					for (int k = 0; k < n[1] && num < max_rescue; ++k) {      // direction 0: the anchors are mate 2's hits
						const int sc = ema_uni(av[1][k].score);
						if (sc < best[1] - score_delta) continue;
						if (lane == 0) ph.tasks[tb + num] = (unsigned long long)((size_t)at >> 6) << 32 | (unsigned long long)(uint32_t)k << 8 | (uint32_t)num;
						++num;
					}
					ema_wave_sync();
					EMA_DBG(9, -num);
					continue;
				}
			}
		}
		// reference src/bwabridge.c:263-269: rescue mate 1 (index 0) from the hits of mate 2, then :277-283 the other way
		for (int dirn = 0; dirn < 2; ++dirn) {
			const int anchor = dirn == 0 ? 1 : 0, target = 1 - anchor;
			const int n_anchor = n[anchor];
			int num = 0;
			cx.wk.a = av[target];
			for (int k = 0; k < n_anchor && num < max_rescue; ++k) {
				EMA_DBG(2 + dirn, k);
				const DevReg a = ema_uni(av[anchor][k]);
				if (a.score >= best[anchor] - score_delta) {
					++num;
					n[target] = matesw(cx, a, len[target], lds_q[wib][target], n[target]);
				}
			}
		}
		EMA_DBG(5, 0);
		for (int m = 0; m < 2; ++m) {
			const int r = 2 * pair + m;
			int cnt = n[m];
			if (cnt > opt.reg_cap) { cx.status |= EMA_ST_REG_OVERFLOW; cnt = opt.reg_cap; }
			DevReg *dst = regs + (size_t)r * opt.reg_cap;
			for (int i = lane; i < cnt; i += EMA_WAVE) dst[i] = av[m][i];
			if (lane == 0) { n_regs[r] = cnt; if (cx.status) atomicOr(status + r, cx.status); }
		}
		ema_wave_sync();
		EMA_DBG(9, 0);
	}
#ifdef EMA_K34_PROF
	EMA_LP_FLUSH(cx.lp, &ema_k3_lp[MODE][0]);
#endif
#undef EMA_DBG
}

// One mem_matesw call in isolation (the libbwa-shaped face, include/ema_bwaabi.h, and its parity test): anchor region `a`,
// the mate's nt4 bases ms[0..l_ms), the mate's region list ma[0..*n_ma) (room for cap), FR window [pes_low, pes_high].
// One wavefront.  slab: EMA_PAIR_SLAB_BYTES of scratch.
__global__ void __launch_bounds__(64)
ema_k_test_matesw(DevIndex ix, DevOpts opt, int pes_low, int pes_high, DevReg a, const uint8_t *__restrict__ ms, int l_ms,
                  DevReg *__restrict__ ma, int *__restrict__ n_ma, int cap, uint8_t *__restrict__ slab, int *__restrict__ status /* [0] status bits, [1] alignments run */)
{
	__shared__ uint8_t lds_q[256];
	__shared__ uint8_t lds_rc[256];
	__shared__ uint8_t lds_r[EMA_RSEQ_CAP];
	__shared__ int lds_stack[3 * 70];
	const int lane = (int)ema_lane();
	PairCtx cx;
	cx.ix = &ix; cx.opt = &opt;
	cx.rc = lds_rc; cx.rseq = lds_r;
	cx.wk.a = (DevReg *)slab;
	cx.wk.tmp = (DevReg *)slab + 2 * EMA_AV_CAP;
	cx.wk.keys = (uint64_t *)((DevReg *)slab + 3 * EMA_AV_CAP);
	cx.bsc = cx.wk.keys + EMA_AV_CAP;
	cx.wk.stack = lds_stack;
	cx.wk.rseq = lds_r;
	cx.wk.mark = nullptr;
	cx.pes_low = pes_low; cx.pes_high = pes_high;
	cx.status = 0;
	int n = ema_uni(*n_ma);
	for (int i = lane; i < l_ms; i += EMA_WAVE) lds_q[i] = ms[i];
	for (int i = lane; i < n; i += EMA_WAVE) cx.wk.a[i] = ma[i];
	ema_wave_sync();
	n = matesw(cx, a, l_ms, lds_q, n);
	ema_wave_sync();
	if (n > cap) { cx.status |= EMA_ST_REG_OVERFLOW; n = cap; }
	for (int i = lane; i < n; i += EMA_WAVE) ma[i] = cx.wk.a[i];
	if (lane == 0) { *n_ma = n; status[0] = cx.status; status[1] = cx.n_sw; }
}

extern "C" void ema_launch_test_matesw(const DevIndex *ix, const DevOpts *opt, int pes_low, int pes_high, const DevReg *a, const uint8_t *ms,
                                       int l_ms, DevReg *ma, int *n_ma, int cap, uint8_t *slab, int *status, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_matesw, dim3(1), dim3(64), 0, s, *ix, *opt, pes_low, pes_high, *a, ms, l_ms, ma, n_ma, cap, slab, status);
}

extern "C" size_t ema_pair_slab_bytes() { return EMA_PAIR_SLAB_BYTES; }

// K3 = K3a (which pairs need a rescue alignment: one lane per pair) then K3b (those pairs, one wavefront each) and, for the pairs K3b
// sets aside, K3t / K3r per direction.  todo: n_pairs ints; n_todo: one int, zero on entry; todo == null runs K3b over every pair.
// heavy (may be null: nothing is set aside): K2's lists and arena, idle while K3 runs; heavy_counters: seven ints, zero on entry
// {pairs set aside, attempts of direction 0, of direction 1, work queues of K3t 0, K3r 0, K3t 1, K3r 1}; arena_used: u64, zero on entry
extern "C" void ema_launch_pair(const DevIndex *ix, const DevOpts *opt, int score_delta, int max_rescue, int pes_low,
                                int pes_high, const uint8_t *bases, const uint32_t *off, int n_pairs, const int *n_pairs_dev, const int *map,
                                DevReg *regs, int *n_regs, int *status, int *todo, int *n_todo, uint8_t *slabs, int *counter, int n_blocks,
                                hipStream_t stream, int *dbg, const HeavyCtl *heavy, int *heavy_counters, unsigned long long *arena_used, int min_attempts)
{
	if (n_pairs <= 0) return;
	PairHeavy ph;
	memset(&ph, 0, sizeof(ph));
	ph.min_attempts = 1 << 30;
	if (heavy && heavy->arena && min_attempts > 0) {
		ph.arena = heavy->arena; ph.arena_bytes = heavy->arena_bytes; ph.arena_used = arena_used;
		ph.pairs = heavy->reads; ph.pairs_cap = heavy->reads_cap; ph.tasks = heavy->tasks; ph.tasks_cap = heavy->tasks_cap / 2;
		ph.n_pairs = heavy_counters; ph.n_tasks = heavy_counters + 1;
		ph.min_attempts = min_attempts;
	}
#define EMA_PAIR_LAUNCH(MODE, ctr, dirn) hipLaunchKernelGGL(ema_k_pair_t<MODE>, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, score_delta, max_rescue, pes_low, pes_high, \
	                   bases, off, n_pairs, n_pairs_dev, map, regs, n_regs, status, todo, n_todo, slabs, (ctr), dbg, ph, (dirn))
	if (todo)
		hipLaunchKernelGGL(ema_k_pair_simple, dim3((n_pairs + 255) / 256), dim3(256), 0, stream, *ix, *opt, score_delta, max_rescue, pes_low,
		                   pes_high, off, n_pairs, n_pairs_dev, map, regs, n_regs, status, todo, n_todo);
	EMA_PAIR_LAUNCH(0, counter, 0);
	if (ph.arena) {
		EMA_PAIR_LAUNCH(1, heavy_counters + 3, 0);
		EMA_PAIR_LAUNCH(2, heavy_counters + 4, 0);
		EMA_PAIR_LAUNCH(1, heavy_counters + 5, 1);
		EMA_PAIR_LAUNCH(2, heavy_counters + 6, 1);
	}
#undef EMA_PAIR_LAUNCH
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_pair_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_pair_t<0>, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
