// ema_amd/csrc/dev_regions.hpp -- region post-processing shared by the extension and the rescue kernels:
// bwa's mem_sort_dedup_patch / mem_patch_reg and the score-only form of bwa_gen_cigar2, one wavefront
// per read.  Control flow is wave-uniform; the global alignment inside mem_patch_reg is the wave DP.
#ifndef EMA_DEV_REGIONS_HPP
#define EMA_DEV_REGIONS_HPP

#include "dev_dp.hpp"
#include "dev_ref.hpp"
#include "dev_sort.hpp"

struct EmaRegWork {
	DevReg *a;        // regions, in place
	DevReg *tmp;      // same capacity as a
	uint64_t *keys;   // same capacity
	int *stack;       // introsort frames (>= 3 * 66 ints)
	uint8_t *rseq;    // EMA_RSEQ_CAP bytes for a reference window
	int *mark = nullptr;   // development aid: this wave's progress words (host-visible) or null
};
#define EMA_MARK(wk, stage, val) do { if ((wk).mark && ema_lane() == 0) { __hip_atomic_store((wk).mark + 1, (stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store((wk).mark + 2, (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } } while (0)

// band of bwa_gen_cigar2
__device__ __forceinline__ int ema_cigar_band(const DevOpts &o, int l_query, int rlen, int w_)
{
	int max_ins = (int)((double)(((l_query + 1) >> 1) * o.a - o.o_ins) / o.e_ins + 1.);
	int max_del = (int)((double)(((l_query + 1) >> 1) * o.a - o.o_del) / o.e_del + 1.);
	int max_gap = max_ins > max_del ? max_ins : max_del;
	max_gap = max_gap > 1 ? max_gap : 1;
	const int diff = rlen > l_query ? rlen - l_query : l_query - rlen;
	int w = (max_gap + diff + 1) >> 1;
	w = w < w_ ? w : w_;
	const int min_w = diff + 3;
	return w > min_w ? w : min_w;
}

// bwa_gen_cigar2(..., &score, no cigar): global score of query[0..l_query) against reference [rb, re).
// Returns false (score untouched) when upstream rejects the request or the window does not fit.
__device__ inline bool ema_global_score(const DevIndex &ix, const DevOpts &o, int w_, int l_query, const uint8_t *query,
                                        int64_t rb, int64_t re, uint8_t *rseq, int &score, int &status)
{
	if (l_query <= 0 || rb >= re || (rb < ix.l_pac && re > ix.l_pac)) return false;
	if (re > ix.l_pac << 1) re = ix.l_pac << 1;
	if (rb < 0) rb = 0;
	const int64_t rlen64 = re - rb;
	if (rlen64 > EMA_RSEQ_CAP) { status |= EMA_ST_RSEQ_OVERFLOW; return false; }
	const int rlen = (int)rlen64;
	ema_wave_fetch(ix, rb, re, rseq);
	const bool rev = rb >= ix.l_pac;    // reversed so that indels end up left-aligned on the forward strand
	if (l_query == rlen && w_ == 0) {
		int part = 0;
		for (int i = (int)ema_lane(); i < l_query; i += EMA_WAVE) part += ema_score(o, rseq[i], query[i]);
		score = ema_wave_sum(part);
		return true;
	}
	const int w = ema_cigar_band(o, l_query, rlen, w_);
	EmaSeq q{rev ? query + l_query - 1 : query, rev ? -1 : 1};
	EmaSeq t{rev ? rseq + rlen - 1 : rseq, rev ? -1 : 1};
	score = ema_wave_global(o, l_query, q, rlen, t, w, nullptr);
	return true;
}

#define EMA_PATCH_MAX_R_BW 0.05f
#define EMA_PATCH_MIN_SC_RATIO 0.90f

// mem_patch_reg: score of merging a (left) and b (right) into one region, 0 if they should stay apart
__device__ inline int ema_patch_reg(const DevIndex &ix, const DevOpts &o, const uint8_t *query, const DevReg &a,
                                    const DevReg &b, int &w_out, uint8_t *rseq, int &status)
{
	if (query == nullptr) return 0;
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb));
	w = w > 0 ? w : -w;
	double r = (double)(a.re - b.rb) / (double)(b.re - a.rb) - (double)(a.qe - b.qb) / (double)(b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) {
		if (w > o.w << 1 || r >= (double)EMA_PATCH_MAX_R_BW) return 0;
	} else if (w > o.w << 2 || r >= (double)(EMA_PATCH_MAX_R_BW * 2)) return 0;
	w += a.w + b.w;
	w = w < o.w << 2 ? w : o.w << 2;
	int score = 0;
	ema_global_score(ix, o, w, b.qe - a.qb, query + a.qb, a.rb, b.re, rseq, score, status);
	const int q_s = (int)((double)(b.qe - a.qb) / (double)((b.qe - b.qb) + (a.qe - a.qb)) * (double)(b.score + a.score) + .499);
	const int r_s = (int)((double)(b.re - a.rb) / (double)((b.re - b.rb) + (a.re - a.rb)) * (double)(b.score + a.score) + .499);
	if ((double)score / (double)(q_s > r_s ? q_s : r_s) < (double)EMA_PATCH_MIN_SC_RATIO) return 0;
	w_out = w;
	return score;
}

// gathers a[] into the order given by the low 11 bits of keys[]
__device__ __forceinline__ void ema_reg_permute(EmaRegWork &wk, int n)
{
	for (int i = (int)ema_lane(); i < n; i += EMA_WAVE) wk.tmp[i] = wk.a[(int)(wk.keys[i] & 0x7ff)];
	ema_wave_sync();
	for (int i = (int)ema_lane(); i < n; i += EMA_WAVE) wk.a[i] = wk.tmp[i];
	ema_wave_sync();
}

// mem_sort_dedup_patch.  query == nullptr disables patching (as in the call from mem_matesw).
// Sequential list edits are made by lane 0 only, separated from the reads of the old values by ema_wave_sync().
__device__ inline int ema_sort_dedup_patch(const DevIndex &ix, const DevOpts &o, const uint8_t *query, int n,
                                           EmaRegWork &wk, int &status)
{
	if (n <= 1) return n;
	const bool leader = ema_lane() == 0;
	DevReg *a = wk.a;
	EMA_MARK(wk, 20, n);
	// sort by END position (ks_introsort(mem_ars2)): keys = re << 11 | index (EMA_AV_CAP = 2048), compared on re only
	for (int i = (int)ema_lane(); i < n; i += EMA_WAVE) wk.keys[i] = (uint64_t)a[i].re << 11 | (uint64_t)i;
	ema_wave_sync();
	if (leader) ema_introsort(wk.keys, n, [](uint64_t x, uint64_t y) { return (x >> 11) < (y >> 11); }, wk.stack);
	ema_wave_sync();
	EMA_MARK(wk, 21, n);
	ema_reg_permute(wk, n);
	EMA_MARK(wk, 22, n);
	for (int i = (int)ema_lane(); i < n; i += EMA_WAVE) a[i].n_comp = 1;
	ema_wave_sync();
	for (int i = 1; i < n; ++i) {
		EMA_MARK(wk, 23, i);
		DevReg p = ema_uni(a[i]);
		const DevReg pr = ema_uni(a[i - 1]);
		if (p.rid != pr.rid || p.rb >= pr.re + o.max_chain_gap) continue;
		bool p_dirty = false;
		for (int j = i - 1; j >= 0; --j) {
			EMA_MARK(wk, 24, i * 1000 + j);
			const DevReg q = ema_uni(a[j]);
			if (!(p.rid == q.rid && p.rb < q.re + o.max_chain_gap)) break;
			if (q.qe == q.qb) continue;
			const int64_t or_ = q.re - p.rb;
			const int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
			const int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
			const int64_t mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
			int score, w;
			if ((float)or_ > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq) {
				if (p.score < q.score) {
					p.qe = p.qb; p_dirty = true;
					break;
				}
				ema_wave_sync();
				if (leader) a[j].qe = q.qb;
			} else if (q.rb < p.rb && (score = ema_patch_reg(ix, o, query, q, p, w, wk.rseq, status)) > 0) {
				p.n_comp += q.n_comp + 1;
				p.seedcov = p.seedcov > q.seedcov ? p.seedcov : q.seedcov;
				p.sub = p.sub > q.sub ? p.sub : q.sub;
				p.csub = p.csub > q.csub ? p.csub : q.csub;
				p.qb = q.qb; p.rb = q.rb;
				p.truesc = p.score = score;
				p.w = w;
				p_dirty = true;
				ema_wave_sync();
				if (leader) a[j].qb = q.qe;
			}
		}
		ema_wave_sync();
		if (p_dirty && leader) a[i] = p;
	}
	ema_wave_sync();
	EMA_MARK(wk, 25, n);
	int m = 0;
	if (leader)
		for (int i = 0; i < n; ++i) {      // drop excluded regions
			const DevReg r = a[i];
			if (r.qe > r.qb) { if (m != i) a[m] = r; ++m; }
		}
	n = ema_uni(__shfl(m, 0));
	EMA_MARK(wk, 26, n);
	// sort by (score desc, rb asc, qb asc) (ks_introsort(mem_ars)).  The three fields are packed into one word
	// above the region's index (10 + 35 + 8 bits: scores < 1024, forward-reverse coordinates < 2^35, reads <= 255),
	// so the comparison on (key >> 11) is mem_ars's comparison and the sort needs no look-ups.
	for (int i = (int)ema_lane(); i < n; i += EMA_WAVE) {
		const DevReg r = a[i];
		wk.keys[i] = (uint64_t)(1023 - (r.score < 0 ? 0 : r.score > 1023 ? 1023 : r.score)) << 54 | (uint64_t)(r.rb & 0x7ffffffffLL) << 19 |
		             (uint64_t)(r.qb & 0xff) << 11 | (uint64_t)i;
	}
	ema_wave_sync();
	if (leader) ema_introsort(wk.keys, n, [](uint64_t x, uint64_t y) { return (x >> 11) < (y >> 11); }, wk.stack);
	ema_wave_sync();
	EMA_MARK(wk, 27, n);
	ema_reg_permute(wk, n);
	EMA_MARK(wk, 28, n);
	m = n > 0 ? 1 : 0;
	if (leader) {
		for (int i = 1; i < n; ++i) {      // identical hits
			const DevReg u = a[i], v = a[i - 1];
			if (u.score == v.score && u.rb == v.rb && u.qb == v.qb) a[i].qe = u.qb;
		}
		for (int i = 1; i < n; ++i) {
			const DevReg r = a[i];
			if (r.qe > r.qb) { if (m != i) a[m] = r; ++m; }
		}
	}
	m = ema_uni(__shfl(m, 0));
	return m;
}

#endif
