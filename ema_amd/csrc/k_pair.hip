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

#define EMA_PAIR_SLAB_BYTES ((size_t)EMA_AV_CAP * (3 * sizeof(DevReg) + 8) + 2048 * 8 + 1024)

namespace {

struct PairCtx {
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

// mem_matesw with the reference's insert model (only FR allowed, pes[1] = {low, high}); ma = wk.a[0..n_ma)
__device__ inline int matesw(PairCtx &cx, const DevReg &a, int l_ms, const uint8_t *ms, int n_ma)
{
	const DevIndex &ix = *cx.ix;
	const DevOpts &o = *cx.opt;
	const int lane = (int)ema_lane();
	const int64_t l_pac = ix.l_pac;
	// an FR-consistent region already present?
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
	if (found) return n_ma;
	// r = 1: the mate is reverse-complemented and lies at larger coordinates
	for (int i = lane; i < l_ms; i += EMA_WAVE) { const int b = ms[i]; cx.rc[l_ms - 1 - i] = (uint8_t)(b < 4 ? 3 - b : 4); }
	ema_wave_sync();
	int64_t rb = a.rb + cx.pes_low - l_ms, re = a.rb + cx.pes_high;
	if (rb < 0) rb = 0;
	if (re > l_pac << 1) re = l_pac << 1;
	int rid = -1;
	if (rb < re) rid = ema_clamp_window(ix, rb, (rb + re) >> 1, re);
	if (!(a.rid == rid && re - rb >= o.min_seed_len)) return n_ma;
	if (re - rb > EMA_RSEQ_CAP) { cx.status |= EMA_ST_RSEQ_OVERFLOW; return n_ma; }
	const int tlen = (int)(re - rb);
	ema_wave_fetch(ix, rb, re, cx.rseq);
	++cx.n_sw;
	// ksw_align2 with KSW_XSUBO | KSW_XSTART | (l_ms * a < 250 ? KSW_XBYTE : 0) | min_seed_len * a
	const int p = l_ms * o.a < 250 ? 16 : 8;
	const int minsc = o.min_seed_len * o.a;
	const EmaLocalRes r1 = ema_wave_local(o, l_ms, p, EmaSeq{cx.rc, 1}, tlen, EmaSeq{cx.rseq, 1}, minsc, 0x10000, cx.bsc);
	int qb = -1, tb = -1;
	if (r1.score >= minsc) {      // start coordinates: the same DP on the reversed prefixes, stopping at the score
		EmaSeq tq{cx.rc + r1.qe, -1};
		EmaSeq tt{cx.rseq, 1, r1.te};
		const EmaLocalRes r2 = ema_wave_local(o, r1.qe + 1, p, tq, tlen, tt, 0x10000, r1.score, cx.bsc);
		if (r1.score == r2.score) { tb = r1.te - r2.te; qb = r1.qe - r2.qe; }
	}
	int n = n_ma;
	if (r1.score >= o.min_seed_len && qb >= 0) {
		DevReg b;
		b.rid = a.rid; b.is_alt = a.is_alt;
		b.qb = l_ms - (r1.qe + 1); b.qe = l_ms - qb;
		b.rb = (l_pac << 1) - (rb + r1.te + 1); b.re = (l_pac << 1) - (rb + tb);
		b.score = r1.score; b.csub = r1.score2; b.secondary = -1;
		b.truesc = 0; b.sub = 0; b.w = 0; b.seedlen0 = 0; b.n_comp = 0; b.frac_rep = 0.f;
		b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
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
	return ema_sort_dedup_patch(ix, o, nullptr, n, cx.wk, cx.status);     // runs whenever the SW ran
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

// K3b: the pairs K3a listed.
// regs/n_regs: K2's output, updated in place.  One wave per pair, pairs taken from a shared counter.
__global__ void __launch_bounds__(256)
ema_k_pair(DevIndex ix, DevOpts opt, int score_delta, int max_rescue, int pes_low, int pes_high,
           const uint8_t *__restrict__ bases, const uint32_t *__restrict__ off, int n_pairs,
           const int *__restrict__ n_pairs_dev, const int *__restrict__ map, DevReg *__restrict__ regs,
           int *__restrict__ n_regs, int *__restrict__ status, const int *__restrict__ todo, const int *__restrict__ n_todo,
           uint8_t *__restrict__ slabs, int *__restrict__ counter, int *dbg)
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

	for (;;) {
		int pair = 0;
		if (lane == 0) pair = atomicAdd(counter, 1);
		pair = ema_uni(__shfl(pair, 0));
		if (pair >= (todo ? *n_todo : ema_work_count(n_pairs, n_pairs_dev, 1))) break;
		if (todo) pair = ema_uni(todo[pair]);
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
		cx.status = 0;
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

// K3 = K3a (which pairs need a rescue alignment: one lane per pair) then K3b (those pairs, one wavefront each).
// todo: n_pairs ints; n_todo: one int, zero on entry; todo == null runs K3b over every pair.
extern "C" void ema_launch_pair(const DevIndex *ix, const DevOpts *opt, int score_delta, int max_rescue, int pes_low,
                                int pes_high, const uint8_t *bases, const uint32_t *off, int n_pairs, const int *n_pairs_dev, const int *map,
                                DevReg *regs, int *n_regs, int *status, int *todo, int *n_todo, uint8_t *slabs, int *counter, int n_blocks,
                                hipStream_t stream, int *dbg)
{
	if (n_pairs <= 0) return;
	if (todo)
		hipLaunchKernelGGL(ema_k_pair_simple, dim3((n_pairs + 255) / 256), dim3(256), 0, stream, *ix, *opt, score_delta, max_rescue, pes_low,
		                   pes_high, off, n_pairs, n_pairs_dev, map, regs, n_regs, status, todo, n_todo);
	hipLaunchKernelGGL(ema_k_pair, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, score_delta, max_rescue, pes_low, pes_high,
	                   bases, off, n_pairs, n_pairs_dev, map, regs, n_regs, status, todo, n_todo, slabs, counter, dbg);
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_pair_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_pair, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
