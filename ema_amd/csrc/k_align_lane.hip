// ema_amd/csrc/k_align_lane.hip -- K2a: seeds -> chains -> regions for the reads that need no extension DP, ONE LANE PER READ.
//
// Same stage as k_align.hip (bwa's mem_align1_core after seeding, reached from reference src/bwabridge.c:236-237:
// bwt_sa, mem_chain, mem_chain_flt, mem_chain2aln, mem_sort_dedup_patch).  Nearly half of the reads of a typical
// bucket match the reference exactly (their longest seed spans the whole read, mem_chain2aln extends nothing) and another
// third differ from it in one base (the extension across that base has a known outcome: lane_extend_diag); what
// is left per read is a few hundred scalar operations (chaining a handful of seed occurrences through a small sorted
// table, the chain filter with klib's introsort, the "already covered" tests, de-duplication).  A whole wavefront per
// read (K2b) issues every one of them 64 wide; here every lane runs the sequential algorithm for its own read, on
// lane-interleaved scratch arrays in HBM (element e of a lane's array lives at [e * 64 + lane]).  A read that turns out
// to need the banded extension DP, a region merge test, or more room than the small tables have is left untouched and put on
// the todo list of K2b.
// (Measured dead end, kept out: running the extension DP per lane as well.  With the H/E row in HBM every cell is a
// dependent memory round trip; with the row in LDS it is 39 KB per wave, three waves per CU, ~1000 clocks per cell
// step, and -- lanes of a wave running the longest band and the most rows among their 64 reads -- barely fewer
// instructions per read than the row-parallel wave DP.)
#include <hip/hip_runtime.h>
#include "dev_regions.hpp"

#define EMA_LANE_INTV 24        // seed intervals of a small read
#define EMA_LANE_SEEDS 32       // seed occurrences (and therefore chains)
#define EMA_LANE_REGS EMA_HAND_REGS        // regions before de-duplication (12)

namespace {

template <typename T> struct LaneArr {      // element e of this lane's array at p[e * 64]
	T *p;
	__device__ __forceinline__ T &operator[](int e) const { return p[(size_t)e << 6]; }
	__device__ __forceinline__ LaneArr operator+(int k) const { return LaneArr{p + ((size_t)k << 6)}; }
};

struct LaneScratch {
	LaneArr<SeedRec> seeds;
	LaneArr<ChainRec> chains;
	LaneArr<int64_t> cpos;
	LaneArr<uint64_t> skey, srt, rkeys;
	LaneArr<int32_t> cord, kept, ord, csi;
	LaneArr<DevReg> av, av_tmp;
};

#define EMA_LANE_WAVE_BYTES                                                                                                  \
	((size_t)64 * (EMA_LANE_SEEDS * (sizeof(SeedRec) + sizeof(ChainRec) + 8 + 8 + 8 + 4 + 4 + 4) + EMA_LANE_INTV * 4 +       \
	               EMA_LANE_REGS * (2 * sizeof(DevReg) + 8)) + 1024)

__device__ __forceinline__ LaneScratch lane_carve(uint8_t *wave_base, int lane)
{
	LaneScratch s;
	size_t o = 0;
	auto take = [&](size_t elem, int n) { uint8_t *p = wave_base + o + (size_t)lane * elem; o += ((size_t)64 * elem * n + 63) & ~(size_t)63; return p; };
	s.chains.p = (ChainRec *)take(sizeof(ChainRec), EMA_LANE_SEEDS);
	s.seeds.p = (SeedRec *)take(sizeof(SeedRec), EMA_LANE_SEEDS);
	s.av.p = (DevReg *)take(sizeof(DevReg), EMA_LANE_REGS);
	s.av_tmp.p = (DevReg *)take(sizeof(DevReg), EMA_LANE_REGS);
	s.cpos.p = (int64_t *)take(8, EMA_LANE_SEEDS);
	s.skey.p = (uint64_t *)take(8, EMA_LANE_SEEDS);
	s.srt.p = (uint64_t *)take(8, EMA_LANE_SEEDS);
	s.rkeys.p = (uint64_t *)take(8, EMA_LANE_REGS);
	s.cord.p = (int32_t *)take(4, EMA_LANE_SEEDS);
	s.kept.p = (int32_t *)take(4, EMA_LANE_SEEDS);
	s.csi.p = (int32_t *)take(4, EMA_LANE_SEEDS);
	s.ord.p = (int32_t *)take(4, EMA_LANE_INTV);
	return s;
}

__device__ __forceinline__ int lane_max_gap(const DevOpts &o, int qlen)
{
	const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
	const int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < o.w << 1 ? l : o.w << 1;
}

// ksw_extend2 for the extensions whose outcome is known without the dynamic program (dev_dp.hpp, ema_wave_extend_nc: the
// first qlen target bases differ from the query in at most one position, nothing ambiguous): the lane compares the
// packed read with the packed reference along the diagonal.  Query base j is read base q0 + j * qstep, target base j
// the reference base at forward-reverse coordinate t0 + j * tstep.  Returns false when the DP is needed.
__device__ inline bool lane_extend_diag(const DevIndex &ix, const DevOpts &o, const uint32_t *qp, int qlen, int q0, int qstep, int tlen,
                                        int64_t t0, int tstep, int zdrop, int h0, EmaExtRes &r)
{
	if (!(tlen >= qlen && h0 > 0)) return false;
	int n_mm = 0, p_mm = -1;
	uint32_t qw = 0, nw = 0, pw = 0;
	int qw_at = -1, nw_at = -1;
	int64_t pw_at = -1;
	for (int j = 0; j < qlen; ++j) {
		const int i = q0 + j * qstep;
		if ((i >> 4) != qw_at) { qw_at = i >> 4; qw = qp[qw_at]; }
		if ((i >> 5) != nw_at) { nw_at = i >> 5; nw = qp[16 + nw_at]; }
		if ((nw >> (i & 31)) & 1) return false;      // ambiguous read base
		const uint32_t code = (qw >> ((i & 15) << 1)) & 3;
		const int64_t p = t0 + (int64_t)j * tstep;
		const bool rs = p >= ix.l_pac;
		const int64_t f = rs ? (ix.l_pac << 1) - 1 - p : p;
		if ((f >> 4) != pw_at) { pw_at = f >> 4; pw = *reinterpret_cast<const uint32_t *>(ix.pac + (pw_at << 2)); }
		const uint32_t b = (pw >> ((((uint32_t)f >> 2) & 3) << 3) >> ((~(uint32_t)f & 3) << 1)) & 3;
		if (code != (rs ? 3 - b : b)) { if (++n_mm > 1) return false; p_mm = j; }
	}
	const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
	r.max_off = 0;
	if (n_mm == 0) { r.score = r.gscore = h0 + qlen * o.a; r.qle = r.tle = r.gtle = qlen; return true; }
	const int gap_min = oe_del < oe_ins + o.a ? oe_del : oe_ins + o.a;
	if (!(o.a > 0 && gap_min > o.a + o.b && (zdrop <= 0 || o.a + o.b <= zdrop) && h0 + p_mm * o.a - o.b > 0)) return false;
	r.gscore = h0 + (qlen - 1) * o.a - o.b; r.gtle = qlen;
	if (qlen - 1 >= p_mm + o.b / o.a + 1) { r.score = r.gscore; r.qle = r.tle = qlen; }
	else { r.score = h0 + p_mm * o.a; r.qle = r.tle = p_mm; }
	return true;
}

// mem_chain's loop body for one seed (test_and_merge, or a new chain right after the element the lookup returned)
__device__ inline void lane_chain_insert(const DevOpts &o, int64_t l_pac, const LaneScratch &s, int &n_chain, int &n_seed, int64_t rbeg,
                                         int qbeg, int len, int rid)
{
	int at = 0, lower = -1;
	if (n_chain) {
		int lo = 0, hi = n_chain;
		while (lo < hi) { const int mid = (lo + hi) >> 1; if (s.cpos[mid] < rbeg) lo = mid + 1; else hi = mid; }
		if (lo < n_chain && s.cpos[lo] == rbeg) { lower = s.cord[lo]; at = lo + 1; }
		else if (lo > 0) { lower = s.cord[lo - 1]; at = lo; }
	}
	if (lower >= 0) {
		ChainRec c = s.chains[lower];
		const int64_t qend = c.l_qbeg + c.l_len, rend = c.l_rbeg + c.l_len;
		if (rid != c.rid) {}
		else if (qbeg >= c.f_qbeg && qbeg + len <= qend && rbeg >= c.f_rbeg && rbeg + len <= rend) return;      // contained: absorbed
		else if ((c.l_rbeg < l_pac || c.f_rbeg < l_pac) && rbeg >= l_pac) {}                                  // other strand
		else {
			const int64_t x = qbeg - c.l_qbeg, y = rbeg - c.l_rbeg;
			if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.l_len < o.max_chain_gap && y - c.l_len < o.max_chain_gap) {
				const int id = n_seed++;
				SeedRec sd; sd.rbeg = rbeg; sd.qbeg = qbeg; sd.len = len; sd.next = -1; sd.pad = 0;
				s.seeds[id] = sd;
				s.seeds[c.last_seed].next = id;
				c.last_seed = id; c.l_rbeg = rbeg; c.l_qbeg = qbeg; c.l_len = len; ++c.n;
				s.chains[lower] = c;
				return;
			}
		}
	}
	for (int idx = n_chain - 1; idx >= at; --idx) { s.cpos[idx + 1] = s.cpos[idx]; s.cord[idx + 1] = s.cord[idx]; }
	const int sid = n_seed++, cid = n_chain++;
	SeedRec sd; sd.rbeg = rbeg; sd.qbeg = qbeg; sd.len = len; sd.next = -1; sd.pad = 0;
	s.seeds[sid] = sd;
	ChainRec c;
	c.pos = rbeg; c.f_rbeg = c.l_rbeg = rbeg; c.f_qbeg = c.l_qbeg = qbeg; c.l_len = len;
	c.rid = rid; c.n = 1; c.first_seed = c.last_seed = sid; c.w = 0; c.kept = 0; c.first = -1;
	s.chains[cid] = c;
	s.cpos[at] = rbeg; s.cord[at] = cid;
}

__device__ inline int lane_chain_weight(const LaneArr<SeedRec> &seeds, int first)
{
	int64_t end = 0;
	int w = 0;
	for (int k = first; k >= 0;) {
		const SeedRec sd = seeds[k];
		if (sd.qbeg >= end) w += sd.len;
		else if (sd.qbeg + sd.len > end) w += (int)(sd.qbeg + sd.len - end);
		end = end > sd.qbeg + sd.len ? end : sd.qbeg + sd.len;
		k = sd.next;
	}
	const int tmp = w;
	w = 0; end = 0;
	for (int k = first; k >= 0;) {
		const SeedRec sd = seeds[k];
		if (sd.rbeg >= end) w += sd.len;
		else if (sd.rbeg + sd.len > end) w += (int)(sd.rbeg + sd.len - end);
		end = end > sd.rbeg + sd.len ? end : sd.rbeg + sd.len;
		k = sd.next;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

// mem_patch_reg up to the point where it would run the global alignment: 0 = the regions stay apart, 1 = needs the DP
__device__ inline int lane_patch_needs_dp(const DevIndex &ix, const DevOpts &o, const DevReg &a, const DevReg &b)
{
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb));
	w = w > 0 ? w : -w;
	double r = (double)(a.re - b.rb) / (double)(b.re - a.rb) - (double)(a.qe - b.qb) / (double)(b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) {
		if (w > o.w << 1 || r >= (double)EMA_PATCH_MAX_R_BW) return 0;
	} else if (w > o.w << 2 || r >= (double)(EMA_PATCH_MAX_R_BW * 2)) return 0;
	return 1;
}

// mem_sort_dedup_patch for a short list.  Returns the number of regions kept, or -1 when a pair of regions would have to
// be test-merged by global alignment (the read then goes to K2b).
__device__ inline int lane_sort_dedup(const DevIndex &ix, const DevOpts &o, int n, const LaneScratch &s, int *stack)
{
	if (n <= 1) return n;
	const LaneArr<DevReg> &a = s.av;
	for (int i = 0; i < n; ++i) s.rkeys[i] = (uint64_t)a[i].re << 11 | (uint64_t)i;
	ema_introsort(s.rkeys, n, [](uint64_t x, uint64_t y) { return (x >> 11) < (y >> 11); }, stack);
	for (int i = 0; i < n; ++i) s.av_tmp[i] = a[(int)(s.rkeys[i] & 0x7ff)];
	for (int i = 0; i < n; ++i) { DevReg r = s.av_tmp[i]; r.n_comp = 1; a[i] = r; }
	for (int i = 1; i < n; ++i) {
		DevReg p = a[i];
		{
			const DevReg pr = a[i - 1];
			if (p.rid != pr.rid || p.rb >= pr.re + o.max_chain_gap) continue;
		}
		bool p_dirty = false;
		for (int j = i - 1; j >= 0; --j) {
			const DevReg q = a[j];
			if (!(p.rid == q.rid && p.rb < q.re + o.max_chain_gap)) break;
			if (q.qe == q.qb) continue;
			const int64_t or_ = q.re - p.rb;
			const int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
			const int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
			const int64_t mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
			if ((float)or_ > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq) {
				if (p.score < q.score) { p.qe = p.qb; p_dirty = true; break; }
				a[j].qe = q.qb;
			} else if (q.rb < p.rb && lane_patch_needs_dp(ix, o, q, p)) return -1;
		}
		if (p_dirty) a[i] = p;
	}
	int m = 0;
	for (int i = 0; i < n; ++i) {
		const DevReg r = a[i];
		if (r.qe > r.qb) { if (m != i) a[m] = r; ++m; }
	}
	n = m;
	for (int i = 0; i < n; ++i) {
		const DevReg r = a[i];
		s.rkeys[i] = (uint64_t)(1023 - (r.score < 0 ? 0 : r.score > 1023 ? 1023 : r.score)) << 54 | (uint64_t)(r.rb & 0x7ffffffffLL) << 19 |
		             (uint64_t)(r.qb & 0xff) << 11 | (uint64_t)i;
	}
	ema_introsort(s.rkeys, n, [](uint64_t x, uint64_t y) { return (x >> 11) < (y >> 11); }, stack);
	for (int i = 0; i < n; ++i) s.av_tmp[i] = a[(int)(s.rkeys[i] & 0x7ff)];
	for (int i = 0; i < n; ++i) a[i] = s.av_tmp[i];
	for (int i = 1; i < n; ++i) {
		const DevReg u = a[i], v = a[i - 1];
		if (u.score == v.score && u.rb == v.rb && u.qb == v.qb) a[i].qe = u.qb;
	}
	m = n > 0 ? 1 : 0;
	for (int i = 1; i < n; ++i) {
		const DevReg r = a[i];
		if (r.qe > r.qb) { if (m != i) a[m] = r; ++m; }
	}
	return m;
}

}  // namespace

// One lane = one read; a wave takes 64 consecutive reads at a time from the shared counter.
// todo / n_todo: reads left for K2b's full path (too many seed occurrences for a lane); hand / n_hand: the reads given up at the
// extension, with their chains (both counters zero on entry).  scratch: EMA_LANE_WAVE_BYTES per resident wave.
#ifndef EMA_K2A_MIN_BLOCKS
#define EMA_K2A_MIN_BLOCKS 4      // resident 256-thread blocks per CU the register allocation must allow: 126 registers and 12 spilled instead of 162 -- beside K1 and K2b (128 each) a SIMD then holds four waves of any mix, not three (r03at: the steady state 178 -> 173 ms per step)
#endif
template <bool PROF>      // PROF: the diagnostic build (phase clocks); the product build carries none of its registers
__global__ void __launch_bounds__(256, EMA_K2A_MIN_BLOCKS)
ema_k_align_simple_t(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
                   const int *__restrict__ n_pairs_dev, const int *__restrict__ map, const Intv *__restrict__ intv,
                   const int *__restrict__ n_intv, DevReg *__restrict__ regs, int *__restrict__ n_regs, int *__restrict__ status,
                   uint8_t *__restrict__ scratch, int *__restrict__ counter, int *__restrict__ todo, int *__restrict__ n_todo,
                   uint8_t *__restrict__ hand, int *__restrict__ n_hand, unsigned long long *prof_arg)
{
	unsigned long long *const prof = PROF ? prof_arg : nullptr;
	// diagnostic phase timing (prof != null): shader clocks per phase of this wave (all lanes move together)
	unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
	int phase = 0;
#define EMA_PHASE(idx) do { if (prof) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); acc[phase] += t_now - t_prev; t_prev = t_now; phase = (idx); } } while (0)
	const int lane = (int)(threadIdx.x & 63);
	const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)ema_uni((int)(threadIdx.x >> 6));
	const LaneScratch s = lane_carve(scratch + wave * EMA_LANE_WAVE_BYTES, lane);
	const int n_total = ema_work_count(n_reads, n_pairs_dev, 2);
	const int64_t l_pac = ix.l_pac;
	int stack[3 * 12];      // introsort frames: ranges above 16 elements only, so a 32-element sort needs a couple

	for (;;) {
		int base = 0;
		if (lane == 0) base = atomicAdd(counter, 64);
		base = __shfl(base, 0);
		if (base >= n_total) break;
		EMA_PHASE(1);
		const int read = base + lane;
		if (read >= n_total) continue;
		if (status[read]) { n_regs[read] = 0; continue; }      // over a capacity in K1: the pair is redone by the full-capacity tier
		const int in_read = ema_in_read(map, read);
		const int l_query = (int)(off[in_read + 1] - off[in_read]);
		const uint32_t *qp = qpack + (size_t)in_read * 24;
		const int n_iv = n_intv[read];
		const Intv *raw = intv + (size_t)read * opt.intv_cap;
		bool small = n_iv <= EMA_LANE_INTV;
		if (small) {
			int64_t tot = 0;
			for (int i = 0; i < n_iv; ++i) { const uint64_t x2 = raw[i].x2; tot += x2 > 64 ? 64 : (int64_t)x2; }
			small = tot <= EMA_LANE_SEEDS;      // which also means: no interval above max_occ, frac_rep = 0
		}
		if (!small) {
			todo[atomicAdd(n_todo, 1)] = read;      // K2b's: too many seed occurrences for a lane
			continue;
		}
		// intervals in mem_collect_intv's final order: by (start, end); equal keys are identical entries
		for (int i = 0; i < n_iv; ++i) {
			const uint64_t mine = raw[i].info;
			int rank = 0;
			for (int k = 0; k < n_iv; ++k) { const uint64_t other = raw[k].info; rank += other < mine || (other == mine && k < i); }
			s.ord[rank] = i;
		}
		// ---------------- mem_chain ----------------
		EMA_PHASE(2);
		int n_chain = 0, n_seed = 0;
		for (int i = 0; i < n_iv; ++i) {
			const Intv p = raw[s.ord[i]];
			const int qbeg = (int)(p.info >> 32), slen = (int)((uint32_t)p.info - (uint32_t)(p.info >> 32));
			for (int64_t k = 0; k < (int64_t)p.x2; ++k) {
				const int64_t rbeg = p.x1 == EMA_INTV_BYPOS ? (int64_t)p.x0 : (int64_t)ema_sa(ix, p.x0 + (uint64_t)k);      // (by position: k_seed.hip, "anchors")
				const int rid = ema_intv2rid(ix, rbeg, rbeg + slen);
				if (rid < 0) continue;
				lane_chain_insert(opt, l_pac, s, n_chain, n_seed, rbeg, qbeg, slen, rid);
			}
		}
		// ---------------- mem_chain_flt ----------------
		EMA_PHASE(3);
		const int n_chn = n_chain;
		if (n_chn > 0) {
			for (int i = 0; i < n_chn; ++i) {
				const int id = s.cord[i];
				const int w = lane_chain_weight(s.seeds, s.chains[id].first_seed);
				s.chains[id].w = w;
				s.skey[i] = (uint64_t)(uint32_t)w << 32 | (uint32_t)id;
			}
			ema_introsort(s.skey, n_chn, [](uint64_t x, uint64_t y) { return (x >> 32) > (y >> 32); }, stack);
			int n_kept = 1;
			s.chains[(int)(uint32_t)s.skey[0]].kept = 3; s.kept[0] = 0;
			for (int i = 1; i < n_chn; ++i) {
				const int ci = (int)(uint32_t)s.skey[i];
				const ChainRec a_i = s.chains[ci];
				const int beg_i = a_i.f_qbeg, end_i = a_i.l_qbeg + a_i.l_len;
				bool large_ovlp = false;
				int k;
				for (k = 0; k < n_kept; ++k) {
					const int cj = (int)(uint32_t)s.skey[s.kept[k]];
					const ChainRec a_j = s.chains[cj];
					const int beg_j = a_j.f_qbeg, end_j = a_j.l_qbeg + a_j.l_len;
					const int b_max = beg_j > beg_i ? beg_j : beg_i;
					const int e_min = end_j < end_i ? end_j : end_i;
					if (e_min > b_max && (!ema_ctg_alt(ix, a_j.rid) || ema_ctg_alt(ix, a_i.rid))) {     // a kept ALT chain does not shadow a primary one
						const int li = end_i - beg_i, lj = end_j - beg_j;
						const int min_l = li < lj ? li : lj;
						if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
							large_ovlp = true;
							if (a_j.first < 0) s.chains[cj].first = i;
							if ((float)a_i.w < (float)a_j.w * opt.drop_ratio && a_j.w - a_i.w >= opt.min_seed_len << 1) break;
						}
					}
				}
				if (k == n_kept) { s.kept[n_kept++] = i; s.chains[ci].kept = large_ovlp ? 2 : 3; }
			}
			for (int k = 0; k < n_kept; ++k) {
				const int f = s.chains[(int)(uint32_t)s.skey[s.kept[k]]].first;
				if (f >= 0) s.chains[(int)(uint32_t)s.skey[f]].kept = 1;
			}
		}
		// ---------------- mem_chain2aln for every surviving chain, in filtered order ----------------
		EMA_PHASE(4);
		int n_av = 0, st = 0;
		int chain_from = 0, n_av_from = 0;      // where K2b resumes if this lane gives the read up: the chain being extended, the list at its start
		bool bail = false;
		for (int cs_ = 0; cs_ < n_chn && !bail; ++cs_) {
			const ChainRec c = s.chains[(int)(uint32_t)s.skey[cs_]];
			chain_from = cs_; n_av_from = n_av;
			if (c.kept == 0) continue;
			const int cn = c.n;
			int64_t rmax0 = l_pac << 1, rmax1 = 0;
			{
				int k = c.first_seed;
				for (int t = 0; t < cn; ++t) {
					const SeedRec sd = s.seeds[k];
					s.csi[t] = k;
					const int64_t b = sd.rbeg - (sd.qbeg + lane_max_gap(opt, sd.qbeg));
					const int tail = l_query - sd.qbeg - sd.len;
					const int64_t e = sd.rbeg + sd.len + (tail + lane_max_gap(opt, tail));
					rmax0 = rmax0 < b ? rmax0 : b;
					rmax1 = rmax1 > e ? rmax1 : e;
					s.srt[t] = (uint64_t)(uint32_t)sd.len << 32 | (uint32_t)t;      // score == len
					k = sd.next;
				}
			}
			rmax0 = rmax0 > 0 ? rmax0 : 0;
			rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
			if (rmax0 < l_pac && l_pac < rmax1) {
				if (c.f_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
			}
			ema_clamp_window(ix, rmax0, c.f_rbeg, rmax1);
			if (rmax1 - rmax0 > EMA_RSEQ_CAP) { bail = true; break; }      // K2b reports it
			if (cn > 1) ema_introsort(s.srt, cn, [](uint64_t x, uint64_t y) { return x < y; }, stack);
			for (int k = cn - 1; k >= 0; --k) {
				const SeedRec sd = s.seeds[s.csi[(int)(uint32_t)s.srt[k]]];
				bool covered = false;
				for (int i = 0; i < n_av && !covered; ++i) {      // already covered by an earlier extension of this read?
					const DevReg p = s.av[i];
					if (sd.rbeg < p.rb || sd.rbeg + sd.len > p.re || sd.qbeg < p.qb || sd.qbeg + sd.len > p.qe) continue;
					if ((double)(sd.len - p.seedlen0) > .1 * (double)l_query) continue;
					int qd = sd.qbeg - p.qb; int64_t rd = sd.rbeg - p.rb;
					int max_gap = lane_max_gap(opt, qd < rd ? qd : (int)rd);
					int w = max_gap < p.w ? max_gap : p.w;
					if (qd - rd < w && rd - qd < w) { covered = true; break; }
					qd = p.qe - (sd.qbeg + sd.len); rd = p.re - (sd.rbeg + sd.len);
					max_gap = lane_max_gap(opt, qd < rd ? qd : (int)rd);
					w = max_gap < p.w ? max_gap : p.w;
					if (qd - rd < w && rd - qd < w) covered = true;
				}
				if (covered) {   // ... unless an already-extended, overlapping seed of the chain lies on another diagonal
					bool other = false;
					for (int i = k + 1; i < cn && !other; ++i) {
						const uint64_t key = s.srt[i];
						if (key == 0) continue;
						const SeedRec t = s.seeds[s.csi[(int)(uint32_t)key]];
						if ((double)t.len < (double)sd.len * .95) continue;
						if (sd.qbeg <= t.qbeg && sd.qbeg + sd.len - t.qbeg >= sd.len >> 2 && t.qbeg - sd.qbeg != t.rbeg - sd.rbeg) other = true;
						if (t.qbeg <= sd.qbeg && t.qbeg + t.len - sd.qbeg >= sd.len >> 2 && sd.qbeg - t.qbeg != sd.rbeg - t.rbeg) other = true;
					}
					if (!other) { s.srt[k] = 0; continue; }
				}
				if (n_av >= EMA_LANE_REGS) { bail = true; break; }
				DevReg a;
				a.sub = a.csub = a.secondary = a.n_comp = a.is_alt = 0; a.seedcov = 0;
				const int aw0 = opt.w, aw1 = opt.w;
				a.score = a.truesc = -1;
				a.rid = c.rid;
				// extensions: only those whose outcome the diagonal determines (no or one mismatch); otherwise K2b.  max_off is
				// 0 for them, so mem_chain2aln's band doubling stops after the first try.
				if (sd.qbeg) {     // left extension, both sequences reversed
					EmaExtRes r;
					if (!lane_extend_diag(ix, opt, qp, sd.qbeg, sd.qbeg - 1, -1, (int)(sd.rbeg - rmax0), sd.rbeg - 1, -1, opt.zdrop, sd.len * opt.a, r)) { bail = true; break; }
					a.score = r.score;
					if (r.gscore <= 0 || r.gscore <= a.score - opt.pen_clip5) {
						a.qb = sd.qbeg - r.qle; a.rb = sd.rbeg - r.tle; a.truesc = a.score;
					} else {
						a.qb = 0; a.rb = sd.rbeg - r.gtle; a.truesc = r.gscore;
					}
				} else { a.score = a.truesc = sd.len * opt.a; a.qb = 0; a.rb = sd.rbeg; }
				if (sd.qbeg + sd.len != l_query) {     // right extension
					const int sc0 = a.score, qe = sd.qbeg + sd.len;
					const int64_t re = sd.rbeg + sd.len;
					EmaExtRes r;
					if (!lane_extend_diag(ix, opt, qp, l_query - qe, qe, 1, (int)(rmax1 - re), re, 1, opt.zdrop, sc0, r)) { bail = true; break; }
					a.score = r.score;
					if (r.gscore <= 0 || r.gscore <= a.score - opt.pen_clip3) {
						a.qe = qe + r.qle; a.re = re + r.tle; a.truesc += a.score - sc0;
					} else {
						a.qe = l_query; a.re = re + r.gtle; a.truesc += r.gscore - sc0;
					}
				} else { a.qe = l_query; a.re = sd.rbeg + sd.len; }
				EMA_PHASE(4);
				{   // seedcov: seeds of the chain fully inside the region
					int cov = 0;
					for (int t = 0; t < cn; ++t) {
						const SeedRec u = s.seeds[s.csi[t]];
						if (u.qbeg >= a.qb && u.qbeg + u.len <= a.qe && u.rbeg >= a.rb && u.rbeg + u.len <= a.re) cov += u.len;
					}
					a.seedcov = cov;
				}
				a.w = aw0 > aw1 ? aw0 : aw1;
				a.seedlen0 = sd.len;
				a.frac_rep = 0.f;
				s.av[n_av++] = a;
			}
		}
		EMA_PHASE(6);
		if (!bail) { chain_from = 0; n_av_from = 0; }      // (given up in the dedup below, which edits the list in place: K2b starts over)
		int n_out = bail ? -1 : lane_sort_dedup(ix, opt, n_av, s, stack);
		if (n_out < 0) {
			// K2b redoes the read from mem_chain2aln on: hand it the chains, the filter's order and the seed pool (dev_types.h, HandHdr)
			uint8_t *h = hand + (size_t)atomicAdd(n_hand, 1) * EMA_HAND_BYTES;
			HandHdr hd;
			hd.read = read; hd.n_chn = n_chn; hd.n_seed = n_seed; hd.l_query = l_query; hd.base_off = off[in_read];
			hd.chain_from = chain_from; hd.n_av = n_av_from; hd.pad = 0;
			*reinterpret_cast<HandHdr *>(h) = hd;
			uint64_t *hk = reinterpret_cast<uint64_t *>(h + sizeof(HandHdr));
			ChainRec *hc = reinterpret_cast<ChainRec *>(h + sizeof(HandHdr) + EMA_HAND_SEEDS * 8);
			SeedRec *hs = reinterpret_cast<SeedRec *>(h + sizeof(HandHdr) + EMA_HAND_SEEDS * (8 + sizeof(ChainRec)));
			for (int i = 0; i < n_chn; ++i) { hk[i] = s.skey[i]; hc[i] = s.chains[i]; }
			for (int i = 0; i < n_seed; ++i) hs[i] = s.seeds[i];
			DevReg *hr = reinterpret_cast<DevReg *>(h + sizeof(HandHdr) + EMA_HAND_SEEDS * (8 + sizeof(ChainRec) + sizeof(SeedRec)));
			for (int i = 0; i < n_av_from; ++i) hr[i] = s.av[i];
			continue;
		}
		if (n_out > opt.reg_cap) { st |= EMA_ST_REG_OVERFLOW; n_out = opt.reg_cap; }
		DevReg *dst = regs + (size_t)read * opt.reg_cap;
		for (int i = 0; i < n_out; ++i) { DevReg r = s.av[i]; r.is_alt = ema_ctg_alt(ix, r.rid); dst[i] = r; }      // mem_align1_core's last loop
		n_regs[read] = n_out;
		if (st) atomicOr(status + read, st);
		EMA_PHASE(0);
	}
	if (prof && lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(prof + 16 + i, acc[i]);
#undef EMA_PHASE
}

extern "C" size_t ema_align_lane_wave_bytes() { return EMA_LANE_WAVE_BYTES; }

extern "C" int ema_align_simple_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_align_simple_t<false>, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}

extern "C" void ema_launch_align_simple(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off, int n_reads,
                                        const int *n_pairs_dev, const int *map, const Intv *intv, const int *n_intv, DevReg *regs,
                                        int *n_regs, int *status, uint8_t *scratch, int *counter, int *todo, int *n_todo,
                                        uint8_t *hand, int *n_hand, int n_blocks, hipStream_t stream, unsigned long long *prof)
{
	if (prof)
		hipLaunchKernelGGL(ema_k_align_simple_t<true>, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, qpack, off, n_reads, n_pairs_dev, map, intv,
		                   n_intv, regs, n_regs, status, scratch, counter, todo, n_todo, hand, n_hand, prof);
	else
		hipLaunchKernelGGL(ema_k_align_simple_t<false>, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, qpack, off, n_reads, n_pairs_dev, map, intv,
		                   n_intv, regs, n_regs, status, scratch, counter, todo, n_todo, hand, n_hand, prof);
}
