// ema_amd/csrc/dev_dp.hpp -- the three dynamic programs of the path as one-wavefront-per-task
// device functions (gfx950, wave64).
//
// All three walk the target row by row and compute one whole DP row per step across the 64 lanes:
// lane t owns NC adjacent query columns (NC = 1..4: the narrowest layout that holds the query; EMA_MAX_READ <= 255
// columns + the boundary column), and
// keeps their H and E values in registers from row to row.  Within a row, M and E depend only on the
// previous row; the horizontal gap state F obeys F(j+1) = max(F(j) - e, T(j)), a max-plus prefix
// scan, which the wave resolves with a 6-step shuffle scan.  This form is exact (integer max/plus),
// so the adaptive band of the extension (per-row [beg,end) shrink, m == 0 break, z-drop, argmax with
// "last column wins") is reproduced row for row -- an anti-diagonal sweep could not, because the band
// of row i+1 is only known after row i.
//
// Replaces (un-vendored bwa, reached from reference src/bwabridge.c:236-237, 267, 281, 304):
//   ema_wave_extend  : ksw_extend2   (seed extension, via mem_chain2aln)
//   ema_wave_global  : ksw_global2   (final alignment + traceback via mem_reg2aln; score only for mem_patch_reg)
//   ema_wave_local   : ksw_u8/ksw_i16 (mate rescue via mem_matesw -> ksw_align2)
#ifndef EMA_DEV_DP_HPP
#define EMA_DEV_DP_HPP

#include "dev_common.hpp"

// EMA_DP_CALL: how the kernels reach the extension / global DPs.  As real calls (noinline) the row loops are register-allocated
// on their own and the callers keep their state across the call instead of through the loops.
#ifndef EMA_DP_CALL
#define EMA_DP_CALL inline
#endif
#define EMA_DP_MINUS_INF (-0x40000000)
#define EMA_NEG_BIG (-0x7f000000)      // "no element" in max scans; never reached by real values

__device__ __forceinline__ int ema_wave_max(int v)
{
	return __builtin_amdgcn_readlane(ema_wave_incl_scan_max(v, EMA_NEG_BIG), 63);
}
// exclusive prefix max over lanes (lane 0 gets EMA_NEG_BIG)
__device__ __forceinline__ int ema_wave_exscan_max(int v)
{
	return ema_wave_shr1(ema_wave_incl_scan_max(v, EMA_NEG_BIG), EMA_NEG_BIG);
}
// value of column j's owner (j wave-uniform); vals[c] = this lane's column NC*lane+c
template <int NC>
__device__ __forceinline__ int ema_col_get_nc(const int vals[NC], int j)
{
	const int c = j % NC;
	int mine = vals[0];
#pragma unroll
	for (int k = 1; k < NC; ++k) mine = c == k ? vals[k] : mine;
	return __builtin_amdgcn_readlane(mine, j / NC);
}

// bwa_fill_scmat(a, b): +a on a match, -b on a mismatch, -1 against an ambiguous base -- computed, not looked up:
// a per-lane table index would turn every DP cell into a memory access.
__device__ __forceinline__ int ema_score(const DevOpts &o, int t, int q)
{
	return (t > 3 || q > 3) ? -1 : (t == q ? o.a : -o.b);
}

struct EmaSeq {          // a byte sequence read forwards (step 1) or backwards (step -1); with pivot >= 0 (step 1)
	const uint8_t *p;    // the prefix [0, pivot] is read reversed and the rest in place (ksw_align2's revseq on the target)
	int step;
	int pivot = -1;
	__device__ __forceinline__ int at(int i) const
	{
		if (pivot >= 0) return i <= pivot ? p[pivot - i] : p[i];
		return p[(long)i * step];
	}
};

// The row loops need one wave-uniform target base per row.  Fetching it from memory row by row puts a load and a
// wait on the critical path of every row, so the wave keeps 256 rows' worth in ONE register per lane (lane t holds
// rows 4t..4t+3 of the current 256-row chunk, loaded together) and extracts row i with a readlane + shift.
struct EmaRowBases {
	int pack;
	__device__ __forceinline__ void load(const EmaSeq &t, int chunk_base, int tlen)
	{
		int p = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int i = chunk_base + (int)ema_lane() * 4 + k;
			if (i < tlen) p |= t.at(i) << (8 * k);
		}
		pack = p;
	}
	__device__ __forceinline__ int get(int i) const      // i wave-uniform, inside the loaded chunk
	{
		return (__builtin_amdgcn_readlane(pack, (i & 255) >> 2) >> ((i & 3) << 3)) & 0xff;
	}
};

struct EmaExtRes { int score, qle, tle, gtle, gscore, max_off; };

// ksw_extend2: banded extension from a seed with score h0, NC query columns per lane (qlen + 1 <= 64 NC).
template <int NC>
__device__ inline EmaExtRes ema_wave_extend_nc(const DevOpts &o, int qlen, EmaSeq query, int tlen, EmaSeq target, int w,
                                               int end_bonus, int zdrop, int h0)
{
	const int lane = (int)ema_lane();
	const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins, e_del = o.e_del, e_ins = o.e_ins;
	int hh[NC], ee[NC], qb[NC];
	// row -1 of the H/E arrays (index j holds H(-1, j-1); index 0 is the boundary column)
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int j = lane * NC + c;
		int v = 0;
		if (j == 0) v = h0;
		else if (j <= qlen) {
			// H[1] = max(h0 - oe_ins, 0); H[j] = H[j-1] - e_ins while H[j-1] > e_ins
			const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
			if (j == 1) v = h1v;
			else {
				// closed form of the run: H[j] = h1v - (j-1)*e_ins as long as every predecessor exceeded e_ins
				const int cand = h1v - (j - 1) * e_ins;
				const int pred = h1v - (j - 2) * e_ins;        // H[j-1] if the run reached it
				v = (pred > e_ins) ? cand : 0;
			}
		}
		hh[c] = v; ee[c] = 0;
		qb[c] = j < qlen ? query.at(j) : 4;
	}
	if (tlen >= qlen && h0 > 0) {
		// Exact shortcuts: the DP's outcome is known when the first qlen target bases differ from the query in at most one
		// position and no base is ambiguous.
		//  * No difference: every row's maximum is its diagonal cell h0 + (i+1)a, strictly above every gapped cell of the
		//    row, so max = gscore = h0 + qlen*a at (qlen-1, qlen-1), max_off 0, whatever the band.
		//  * One mismatch at position p (the usual reason a seed ended): a gapped cell (i, j) scores at most
		//    h0 + min(i,j)+1 matches - the cheapest gap, the diagonal cell at least h0 + (i+1)a - (a+b); with
		//    min(o_del+e_del, o_ins+e_ins+a) > a+b the diagonal stays the strict maximum of every row i < qlen, so the row
		//    maxima are h0+(i+1)a up to p and h0+(i+1)a-(a+b) from p on.  The running maximum (updated on strict increase
		//    only) therefore ends on the last row if the diagonal recovers above the pre-mismatch prefix before the query
		//    ends (qlen-1 >= p + b/a + 1), else it stays at (p-1, p-1); max_off stays 0; the end-to-end score is the last
		//    diagonal cell; z-drop (a+b <= zdrop) does not fire on the diagonal.
		int n_bad = 0;
		unsigned long long mm_mask[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			bool mm = false;
			if (j < qlen) {
				const int t = target.at(j);
				if (qb[c] > 3 || t > 3) ++n_bad;
				mm = qb[c] != t;
			}
			mm_mask[c] = __ballot(mm);
		}
		if (!__ballot(n_bad != 0)) {
			int n_mm = 0, p_mm = -1;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				n_mm += __popcll(mm_mask[c]);
				if (mm_mask[c]) p_mm = (__ffsll((long long)mm_mask[c]) - 1) * NC + c;
			}
			if (n_mm == 0) {
				EmaExtRes r;
				r.score = r.gscore = h0 + qlen * o.a; r.qle = r.tle = r.gtle = qlen; r.max_off = 0;
				return r;
			}
			const int gap_min = oe_del < oe_ins + o.a ? oe_del : oe_ins + o.a;
			if (n_mm == 1 && o.a > 0 && gap_min > o.a + o.b && (zdrop <= 0 || o.a + o.b <= zdrop) && h0 + p_mm * o.a - o.b > 0) {
				EmaExtRes r;
				r.gscore = h0 + (qlen - 1) * o.a - o.b; r.gtle = qlen; r.max_off = 0;
				if (qlen - 1 >= p_mm + o.b / o.a + 1) { r.score = r.gscore; r.qle = r.tle = qlen; }
				else { r.score = h0 + p_mm * o.a; r.qle = r.tle = p_mm; }
				return r;
			}
		}
	}
	int max_ins, max_del;
	{
		const int mx = o.a > 0 ? o.a : 0;      // largest entry of the scoring matrix
		max_ins = (int)((double)(qlen * mx + end_bonus - o.o_ins) / e_ins + 1.);
		max_ins = max_ins > 1 ? max_ins : 1;
		w = w < max_ins ? w : max_ins;
		max_del = (int)((double)(qlen * mx + end_bonus - o.o_del) / e_del + 1.);
		max_del = max_del > 1 ? max_del : 1;
		w = w < max_del ? w : max_del;
	}
	int mx_sc = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	EmaRowBases rows;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 255) == 0) rows.load(target, i, tlen);
		const int tb = rows.get(i);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1_init = 0;
		if (beg == 0) { h1_init = h0 - (o.o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		// per cell: M, E from the previous row; T = what F(j+1) may open with
		int M[NC], g[NC];
		int run = EMA_NEG_BIG;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j >= beg && j < end;
			const int s = ema_score(o, tb, qb[c]);
			int m = hh[c] ? hh[c] + s : 0;
			M[c] = m;
			int t = m - oe_ins; t = t > 0 ? t : 0;
			g[c] = in ? t + j * e_ins : EMA_NEG_BIG;
			run = max(run, g[c]);
		}
		int pre = ema_wave_exscan_max(run);     // max g over all columns of lower lanes
		int h[NC];
		int m_key = -1;                         // (h << 9 | j), max over in-range cells
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j >= beg && j < end;
			int f = pre == EMA_NEG_BIG ? 0 : pre - (j - 1) * e_ins;   // F(i,j) = max(0, max_{k<j} g_k - (j-1) e)
			f = f > 0 ? f : 0;
			int hv = M[c] > ee[c] ? M[c] : ee[c];
			hv = hv > f ? hv : f;
			h[c] = in ? hv : 0;
			if (in) m_key = max(m_key, (hv << 9) | j);
			pre = max(pre, g[c]);
		}
		// new E for in-range cells; new H array: index j takes H(i, j-1)
		const int up = ema_wave_shr1(h[NC - 1], 0);   // H(i, 4*lane - 1)
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j >= beg && j < end;
			if (in) {
				int t = M[c] - oe_del; t = t > 0 ? t : 0;
				int e = ee[c] - e_del;
				ee[c] = e > t ? e : t;
			}
			const int left = c == 0 ? up : h[c - 1];      // H(i, j-1)
			if (in) hh[c] = j == beg ? h1_init : left;
			else if (j == end) { hh[c] = end > beg ? left : h1_init; ee[c] = 0; }
		}
		const int h1_fin = end > beg ? ema_col_get_nc<NC>(h, end - 1) : h1_init;
		const int jfin = end > beg ? end : beg;
		if (jfin == qlen) {
			max_ie = gscore > h1_fin ? max_ie : i;
			gscore = gscore > h1_fin ? gscore : h1_fin;
		}
		m_key = ema_wave_max(m_key);
		const int m = m_key < 0 ? 0 : m_key >> 9, mj = m_key < 0 ? -1 : m_key & 511;
		if (m == 0) break;
		if (m > mx_sc) {
			mx_sc = m; max_i = i; max_j = mj;
			const int off = mj - i < 0 ? i - mj : mj - i;
			max_off = max_off > off ? max_off : off;
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) {
				if (mx_sc - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
			} else {
				if (mx_sc - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
			}
		}
		// Exact early exit once the query's end has been reached (gscore > 0).  No later cell can exceed
		//   ub = max over this row's live cells of H(i,j) + (qlen-1-j) * a      (and the boundary column's h1 + qlen * a):
		// a cell of row i+1 takes H(i,j-1) + s <= H(i,j-1) + a, or E <= H(i,j), or F <= a cell to its left in its own row, so
		// max_j H(i',j) + (qlen-1-j) a does not grow from row to row.  With gscore > ub (and mx_sc >= gscore) the rows that remain
		// cannot raise the maximum (strict >) nor touch gscore / max_ie (>=), and max_off only moves with the maximum: the six
		// results are final.  This is the usual end of an extension: the diagonal has run off the query and what is left in the
		// band are gap states decaying by e_del a row, for up to cal_max_gap more rows.
		if (i >= qlen - 1 && gscore > 0) {
			const int mx = o.a > 0 ? o.a : 0;
			int ub = (beg == 0 && h1_init > 0) ? h1_init + qlen * mx : -1;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				const int j = lane * NC + c;
				if (j >= beg && j < end && h[c] > 0) ub = max(ub, h[c] + (qlen - 1 - j) * mx);
			}
			if (gscore > ema_wave_max(ub)) break;
		}
		// next row's range.  Sequentially: beg = first j in [beg,end) with H or E non-zero (else end);
		// then j = last non-zero index in [beg,end] (else beg-1); end = min(j+2, qlen).
		int first = 1 << 20, last = -1;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool nz = (j >= beg && j <= end) && (hh[c] != 0 || ee[c] != 0);
			const unsigned long long b = __ballot(nz);
			if (b) {
				const int f_ = (__ffsll((long long)b) - 1) * NC + c;
				const int l_ = (63 - __clzll((long long)b)) * NC + c;
				first = first < f_ ? first : f_;
				last = last > l_ ? last : l_;
			}
		}
		beg = first < end ? first : end;
		{
			const int jl = last >= beg ? last : beg - 1;
			end = jl + 2 < qlen ? jl + 2 : qlen;
		}
	}
	EmaExtRes r;
	r.score = mx_sc; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

// ksw_extend2, qlen <= 255.  Extensions are mostly the rest of a read beyond its seed: the narrowest column layout
// that holds the query keeps the per-row work proportional to it.
__device__ EMA_DP_CALL EmaExtRes ema_wave_extend(const DevOpts &o, int qlen, EmaSeq query, int tlen, EmaSeq target, int w,
                                            int end_bonus, int zdrop, int h0)
{
	if (qlen < 64) return ema_wave_extend_nc<1>(o, qlen, query, tlen, target, w, end_bonus, zdrop, h0);
	if (qlen < 128) return ema_wave_extend_nc<2>(o, qlen, query, tlen, target, w, end_bonus, zdrop, h0);
	if (qlen < 192) return ema_wave_extend_nc<3>(o, qlen, query, tlen, target, w, end_bonus, zdrop, h0);
	return ema_wave_extend_nc<4>(o, qlen, query, tlen, target, w, end_bonus, zdrop, h0);
}

// ksw_global2: banded global alignment of query (columns) against target (rows).
// z != nullptr: the direction matrix (n_col x tlen bytes, n_col = min(qlen, 2w+1)) is written for traceback.
template <int NC>
__device__ inline int ema_wave_global_nc(const DevOpts &o, int qlen, EmaSeq query, int tlen, EmaSeq target, int w, uint8_t *z)
{
	const int lane = (int)ema_lane();
	const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins, e_del = o.e_del, e_ins = o.e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	int hh[NC], ee[NC], qb[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int j = lane * NC + c;
		int v = EMA_DP_MINUS_INF;
		if (j == 0) v = 0;
		else if (j <= qlen && j <= w) v = -(o.o_ins + e_ins * j);
		hh[c] = v; ee[c] = EMA_DP_MINUS_INF;
		qb[c] = j < qlen ? query.at(j) : 4;
	}
	EmaRowBases rows;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 255) == 0) rows.load(target, i, tlen);
		const int tb = rows.get(i);
		const int beg = i > w ? i - w : 0;
		const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int h1_init = beg == 0 ? -(o.o_del + e_del * (i + 1)) : EMA_DP_MINUS_INF;
		int M[NC], g[NC];
		int run = EMA_NEG_BIG;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j >= beg && j < end;
			const int m = hh[c] + ema_score(o, tb, qb[c]);
			M[c] = m;
			g[c] = in ? (m - oe_ins) + j * e_ins : EMA_NEG_BIG;
			run = max(run, g[c]);
		}
		int pre = ema_wave_exscan_max(run);
		int h[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j >= beg && j < end;
			// F(i,j) = max(MINUS_INF - (j-beg) e, max_{beg<=k<j} g_k - (j-1) e)
			int f = EMA_DP_MINUS_INF - (j - beg) * e_ins;
			if (pre != EMA_NEG_BIG) f = max(f, pre - (j - 1) * e_ins);
			const int m = M[c], e = ee[c];
			int d = m >= e ? 0 : 1;
			int hv = m >= e ? m : e;
			d = hv >= f ? d : 2;
			hv = hv >= f ? hv : f;
			h[c] = hv;
			if (in) {
				int t = m - oe_del, e2 = e - e_del;
				d |= e2 > t ? 1 << 2 : 0;
				ee[c] = e2 > t ? e2 : t;
				t = m - oe_ins;
				d |= (f - e_ins) > t ? 2 << 4 : 0;
				if (z) z[(size_t)i * n_col + (j - beg)] = (uint8_t)d;
			}
			pre = max(pre, g[c]);
		}
		const int up = ema_wave_shr1(h[NC - 1], 0);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j >= beg && j < end;
			const int left = c == 0 ? up : h[c - 1];
			if (in) hh[c] = j == beg ? h1_init : left;
			else if (j == end) { hh[c] = end > beg ? left : h1_init; ee[c] = EMA_DP_MINUS_INF; }
		}
	}
	return ema_col_get_nc<NC>(hh, qlen);
}

// ksw_global2 when the band fits the wavefront (2w + 1 <= 64) and reaches the last column in the last row (|tlen - qlen| <= w: always,
// for the bands bwa_gen_cigar2 chooses): the lanes lie ACROSS THE BAND, lane b on the diagonal j - i = b - w, i.e. on column
// j = i - w + b in row i.  The bands of the final alignments are a few cells wide (inferred from the score: a mismatch or two, one
// short gap), so the column layout above computes two or three cells per lane for 64 lanes to fill a dozen; here a row is one cell per
// lane whatever the read's length.  The cell's diagonal predecessor is the lane's own cell of the row before (no shift), E comes
// from the lane above (one DPP shift per row), F from the lanes below (the same max-plus scan), the query slides down the lanes by
// one base per row.  Same recurrences, initial values, tie rules and direction bytes as ema_wave_global_nc, cell for cell.
__device__ inline int ema_wave_global_band(const DevOpts &o, int qlen, EmaSeq query, int tlen, EmaSeq target, int w, uint8_t *z)
{
	const int lane = (int)ema_lane();
	const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins, e_del = o.e_del, e_ins = o.e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	// row 0: this lane's column is lane - w.  hd = H(i-1, j-1) of the cell the lane computes in row i; ein = E(i, j) as it enters row i.
	int hd, ein = EMA_DP_MINUS_INF, qb;
	{
		const int j = lane - w;
		hd = j == 0 ? 0 : (j >= 1 && j <= qlen && j <= w) ? -(o.o_ins + e_ins * j) : EMA_DP_MINUS_INF;
		qb = (j >= 0 && j < qlen) ? query.at(j) : 4;
	}
	EmaRowBases rows, qrows;      // the query base that enters at lane 63 in row i is q[i + 64 - w]: taken from a pack like the target's
	int q_chunk = -1;
	int h_last = EMA_DP_MINUS_INF;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 255) == 0) rows.load(target, i, tlen);
		const int tb = rows.get(i);
		const int j = i - w + lane;
		const int beg = i > w ? i - w : 0;
		const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const bool in = j >= beg && j < end;
		if (j == 0) hd = i == 0 ? 0 : -(o.o_del + e_del * i);      // H(i-1, -1): the boundary column (in the band while i <= w)
		const int m = hd + ema_score(o, tb, qb);
		const int g = in ? (m - oe_ins) + j * e_ins : EMA_NEG_BIG;
		const int pre = ema_wave_exscan_max(g);
		int f = EMA_DP_MINUS_INF - (j - beg) * e_ins;
		if (pre != EMA_NEG_BIG) f = max(f, pre - (j - 1) * e_ins);
		const int e = ein;
		int d = m >= e ? 0 : 1;
		int hv = m >= e ? m : e;
		d = hv >= f ? d : 2;
		hv = hv >= f ? hv : f;
		int e_next = EMA_DP_MINUS_INF;
		if (in) {
			int t = m - oe_del;
			const int e2 = e - e_del;
			d |= e2 > t ? 1 << 2 : 0;
			e_next = e2 > t ? e2 : t;
			t = m - oe_ins;
			d |= (f - e_ins) > t ? 2 << 4 : 0;
			if (z) z[(size_t)i * n_col + (j - beg)] = (uint8_t)d;
			hd = hv;      // the diagonal predecessor of this lane's next cell, (i + 1, j + 1)
		}
		if (i == tlen - 1) h_last = hv;
		ein = ema_wave_shl1(e_next, EMA_DP_MINUS_INF);      // cell (i + 1, j + 1) has (i, j + 1) above it: the lane above's
		// the query slides down by one lane; lane 63 takes q[i + 1 - w + 63]
		{
			const int jn = i + 1 - w + 63;
			int nb = 4;
			if (jn >= 0 && jn < qlen) {
				if ((jn >> 8) != q_chunk) { q_chunk = jn >> 8; qrows.load(query, q_chunk << 8, qlen); }
				nb = qrows.get(jn);
			}
			const int down = ema_wave_shl1(qb, 4);
			qb = lane == 63 ? nb : down;
		}
	}
	// eh[qlen].h after the last row = H(tlen - 1, qlen - 1): the lane on diagonal (qlen - 1) - (tlen - 1)
	return __builtin_amdgcn_readlane(h_last, qlen - tlen + w);
}

// ksw_global2 with the narrowest column layout that holds the query (qlen + 1 <= 64 NC)
__device__ EMA_DP_CALL int ema_wave_global(const DevOpts &o, int qlen, EmaSeq query, int tlen, EmaSeq target, int w, uint8_t *z)
{
	if (2 * w + 1 <= 64 && tlen >= 1 && qlen >= 1 && tlen - qlen <= w && qlen - tlen <= w) return ema_wave_global_band(o, qlen, query, tlen, target, w, z);
	if (qlen < 64) return ema_wave_global_nc<1>(o, qlen, query, tlen, target, w, z);
	if (qlen < 128) return ema_wave_global_nc<2>(o, qlen, query, tlen, target, w, z);
	if (qlen < 192) return ema_wave_global_nc<3>(o, qlen, query, tlen, target, w, z);
	return ema_wave_global_nc<4>(o, qlen, query, tlen, target, w, z);
}

// Traceback of ema_wave_global's direction matrix (ksw_global2's backtrack): wave-uniform, sequential.
// Ops are produced from the alignment's end towards its start and written backwards into cig[0..cap),
// so cig[first..cap) is the CIGAR in forward order (BAM packing len<<4|op, M=0 I=1 D=2).  Returns `first`,
// or -1 if cap is too small.
// ZP: const uint8_t * (the matrix where the DP wrote it) or const EMA_LDS uint8_t * (staged: k_final.hip).
// Run by the WHOLE wavefront with wave-uniform scalars (lane 0 alone stores the operations): as a loop of one lane under an exec
// mask its ~30 dependent vector instructions per step cost more than the read they wait for.  Runs of matches are taken 64 cells at a
// time (below); what stays one read per step are the cells in and around gaps.
template <typename ZP>
__device__ inline int ema_traceback(ZP z, int qlen, int tlen, int w, uint32_t *cig, int cap)
{
	const bool leader = ema_lane() == 0;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0, pos = cap, last_op = -1;
	uint32_t cur = 0;
	const int lane = (int)ema_lane();
	while (i >= 0 && k >= 0) {
		if (which == 0) {
			// In the match state the walk goes down the diagonal for as long as the cells say "from the diagonal": lane l looks at the
			// cell l steps ahead (same offset inside its row's band), and the whole run -- usually everything up to the next gap or the
			// alignment's start -- becomes one step instead of one dependent read per base.
			const int ii = i - lane, kk = k - lane;
			bool m = false;
			if (ii >= 0 && kk >= 0) m = (z[(size_t)ii * n_col + (kk - (ii > w ? ii - w : 0))] & 3) == 0;
			const unsigned long long run_mask = __ballot(m);
			const int r = run_mask == ~0ULL ? 64 : __ffsll((long long)~run_mask) - 1;
			if (r > 0) {
				if (last_op == 0) cur += (uint32_t)r << 4;
				else {
					if (last_op >= 0) { if (pos == 0) return -1; --pos; if (leader) cig[pos] = cur; }
					cur = (uint32_t)r << 4; last_op = 0;
				}
				i -= r; k -= r;
				continue;
			}
		}
		which = ema_uni((int)z[(size_t)i * n_col + (k - (i > w ? i - w : 0))]) >> (which << 1) & 3;
		const int op = which == 0 ? 0 : which == 1 ? 2 : 1;
		if (op == last_op) cur += 1u << 4;
		else {
			if (last_op >= 0) { if (pos == 0) return -1; --pos; if (leader) cig[pos] = cur; }
			cur = 1u << 4 | (uint32_t)op; last_op = op;
		}
		if (which == 0) { --i; --k; } else if (which == 1) --i; else --k;
	}
	if (i >= 0) {
		if (last_op == 2) cur += (uint32_t)(i + 1) << 4;
		else { if (last_op >= 0) { if (pos == 0) return -1; --pos; if (leader) cig[pos] = cur; } cur = (uint32_t)(i + 1) << 4 | 2u; last_op = 2; }
	}
	if (k >= 0) {
		if (last_op == 1) cur += (uint32_t)(k + 1) << 4;
		else { if (last_op >= 0) { if (pos == 0) return -1; --pos; if (leader) cig[pos] = cur; } cur = (uint32_t)(k + 1) << 4 | 1u; last_op = 1; }
	}
	if (last_op >= 0) { if (pos == 0) return -1; --pos; if (leader) cig[pos] = cur; }
	return pos;
}

struct EmaLocalRes { int score, te, qe, score2, te2; };

// One pass of ksw_u8 / ksw_i16 (see oracle/dp.c for why the striped SSE2 kernel equals this):
// Gotoh local alignment over qpad = ceil(qlen/p)*p columns (p = 16 for the 8-bit kernel, 8 for the 16-bit one),
// padded columns scoring 0.  minsc / endsc as in ksw (0x10000 = off).  qpad <= 256.
// bsc: scratch for ksw's b[] list (one u64 per target row at most).
template <int NC>
__device__ inline EmaLocalRes ema_wave_local_nc(const DevOpts &o, int qlen, int p, EmaSeq query, int tlen, EmaSeq target,
                                             int minsc, int endsc, uint64_t *bsc)
{
	const int lane = (int)ema_lane();
	const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins, e_del = o.e_del, e_ins = o.e_ins;
	const int qpad = (qlen + p - 1) / p * p;
	const int maxsc = o.a > 0 ? o.a : 0;       // largest entry of the scoring matrix
	int hh[NC], ee[NC], qb[NC], hmax[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int j = lane * NC + c;
		hh[c] = 0; ee[c] = 0; hmax[c] = 0;          // hh[c] = H(i-1, j)
		qb[c] = j < qlen ? query.at(j) : 5;         // 5 = padding column
	}
	int gmax = 0, te = -1;
	// sub-optimal bookkeeping: ksw's b[] list of (row maximum, row); the exclusion window needs te, so the
	// list is kept (wave-uniform writes) and scanned at the end.
	int n_b = 0;
	int last_sc = 0, last_row = -2;                 // copy of b[n_b-1]
	EmaRowBases rows;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 255) == 0) rows.load(target, i, tlen);
		const int tb = rows.get(i);
		const int up = ema_wave_shr1(hh[NC - 1], 0);  // H(i-1, 4*lane-1)
		int Hd[NC], g[NC];
		int run = EMA_NEG_BIG;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j < qpad;
			const int diag = c == 0 ? (lane == 0 ? 0 : up) : hh[c - 1];
			const int s = qb[c] == 5 ? 0 : ema_score(o, tb, qb[c]);
			int hv = diag + s; hv = hv > 0 ? hv : 0;
			hv = hv > ee[c] ? hv : ee[c];
			Hd[c] = hv;                                  // before the F term
			int t = hv - oe_ins;
			g[c] = in ? t + j * e_ins : EMA_NEG_BIG;
			run = max(run, g[c]);
		}
		int pre = ema_wave_exscan_max(run);
		int rowmax = 0;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			const bool in = j < qpad;
			int f = pre == EMA_NEG_BIG ? 0 : pre - (j - 1) * e_ins;
			f = f > 0 ? f : 0;
			const int hv = Hd[c] > f ? Hd[c] : f;
			if (in) {
				hh[c] = hv;
				rowmax = rowmax > hv ? rowmax : hv;
				int e = ee[c] - e_del; e = e > 0 ? e : 0;
				int t = hv - oe_del; t = t > 0 ? t : 0;
				ee[c] = e > t ? e : t;
			}
			pre = max(pre, g[c]);
		}
		const int imax = ema_wave_max(rowmax);
		if (imax >= minsc) {
			if (n_b == 0 || last_row + 1 != i) {
				bsc[n_b++] = (uint64_t)imax << 32 | (uint32_t)i;
				last_sc = imax; last_row = i;
			} else if (last_sc < imax) {
				bsc[n_b - 1] = (uint64_t)imax << 32 | (uint32_t)i;
				last_sc = imax; last_row = i;
			}
		}
		if (imax > gmax) {
			gmax = imax; te = i;
#pragma unroll
			for (int c = 0; c < NC; ++c) hmax[c] = hh[c];
			if (gmax >= endsc) break;
		}
	}
	EmaLocalRes r;
	r.score = gmax; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
	// qe: smallest query index holding the maximum of row te
	{
		int best = -1;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = lane * NC + c;
			if (j < qpad && hmax[c] == gmax && best < 0) best = j;
		}
		const unsigned long long b = __ballot(best >= 0);
		if (b) r.qe = ema_uni(__shfl(best, __ffsll((long long)b) - 1));
		if (te < 0) r.qe = -1;
	}
	if (n_b > 0) {
		// score2 = largest b[] score outside [te - d, te + d], d = ceil(score / max match score); first entry on ties
		const int d = (r.score + maxsc - 1) / maxsc;
		const int low = te - d, high = te + d;
		long long best = -1;                           // score << 32 | (0x7fffffff - index)
		for (int k = lane; k < n_b; k += EMA_WAVE) {
			const uint64_t b = bsc[k];
			const int row = (int)(uint32_t)b;
			if (row < low || row > high) {
				const long long key = (long long)(b >> 32) << 32 | (long long)(0x7fffffff - k);
				best = best > key ? best : key;
			}
		}
		{
			long long t;
			t = __shfl_xor(best, 1); best = best > t ? best : t;
			t = __shfl_xor(best, 2); best = best > t ? best : t;
			t = __shfl_xor(best, 4); best = best > t ? best : t;
			t = __shfl_xor(best, 8); best = best > t ? best : t;
			t = __shfl_xor(best, 16); best = best > t ? best : t;
			t = __shfl_xor(best, 32); best = best > t ? best : t;
			best = (long long)ema_uni((int64_t)best);
		}
		if (best >= 0) {
			const int k = 0x7fffffff - (int)(best & 0xffffffffLL);
			r.score2 = (int)(best >> 32);
			r.te2 = ema_uni((int)(uint32_t)bsc[k]);
		}
	}
	return r;
}

// one pass of ksw_u8 / ksw_i16 with the narrowest column layout that holds the padded query (qpad <= 64 NC)
__device__ inline EmaLocalRes ema_wave_local(const DevOpts &o, int qlen, int p, EmaSeq query, int tlen, EmaSeq target,
                                             int minsc, int endsc, uint64_t *bsc)
{
	const int qpad = (qlen + p - 1) / p * p;
	if (qpad <= 64) return ema_wave_local_nc<1>(o, qlen, p, query, tlen, target, minsc, endsc, bsc);
	if (qpad <= 128) return ema_wave_local_nc<2>(o, qlen, p, query, tlen, target, minsc, endsc, bsc);
	if (qpad <= 192) return ema_wave_local_nc<3>(o, qlen, p, query, tlen, target, minsc, endsc, bsc);
	return ema_wave_local_nc<4>(o, qlen, p, query, tlen, target, minsc, endsc, bsc);
}

#endif
