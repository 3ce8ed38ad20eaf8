// ema_amd/csrc/dev_ref.hpp -- reference-coordinate helpers on the HBM index (bwa's bntseq.c):
// contig lookup, forward-reverse coordinates, window fetch from the 2-bit pac.
#ifndef EMA_DEV_REF_HPP
#define EMA_DEV_REF_HPP

#include "dev_common.hpp"

// bwa's cal_max_gap (bwamem.c): the longest gap an extension over qlen query bases can pay for, capped at twice the band
__device__ __forceinline__ int ema_cal_max_gap(const DevOpts &o, int qlen)
{
	const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
	const int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < o.w << 1 ? l : o.w << 1;
}

// bns_pos2rid: contig holding forward position pos_f -- the last contig that starts at or below it.  bwa searches the contig
// offsets by bisection: five to twelve DEPENDENT loads per call, two calls per seed occurrence (ema_intv2rid), one per window.
// The coarse table (dev_types.h, ctg_tab) names the contigs of the 2^ctg_shift-base block the position lies in; what is left to
// bisect is the contigs that start inside that block -- none, for all but a few positions.
__device__ __forceinline__ int ema_pos2rid(const DevIndex &ix, int64_t pos_f)
{
	if (pos_f >= ix.l_pac) return -1;
	const int64_t b = pos_f >> ix.ctg_shift;
	int lo = ix.ctg_tab[b], hi = ix.ctg_tab[b + 1];
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (pos_f >= ix.ctg_off[mid]) lo = mid; else hi = mid - 1;
	}
	return lo;
}
// bntann1_t.is_alt of contig rid (<prefix>.alt; reference src/bwabridge.c:371 reads the flag bwa carries through to mem_aln_t)
__device__ __forceinline__ int ema_ctg_alt(const DevIndex &ix, int rid) { return ix.ctg_alt && rid >= 0 ? (int)ix.ctg_alt[rid] : 0; }
__device__ __forceinline__ int64_t ema_depos(const DevIndex &ix, int64_t pos, int &is_rev)
{
	is_rev = pos >= ix.l_pac;
	return is_rev ? (ix.l_pac << 1) - 1 - pos : pos;
}
// bns_intv2rid: contig of [rb, re), -1 if it spans two contigs, -2 if it spans the strand junction
__device__ __forceinline__ int ema_intv2rid(const DevIndex &ix, int64_t rb, int64_t re)
{
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	int r;
	const int64_t fb = ema_depos(ix, rb, r);
	const int rid_b = ema_pos2rid(ix, fb);
	if (!(rb < re) || rid_b < 0) return rid_b;
	// the other end lies in the same contig or the interval spans two (bwa looks its contig up as well and compares)
	const int64_t fe = ema_depos(ix, re - 1, r);
	if (fe >= ix.l_pac) return -1;
	return fe >= ix.ctg_off[rid_b] && fe < ix.ctg_off[rid_b + 1] ? rid_b : -1;
}
// base at forward-reverse coordinate p (reverse strand = complement read from the far end)
__device__ __forceinline__ int ema_ref_base(const DevIndex &ix, int64_t p)
{
	const bool rev = p >= ix.l_pac;
	const int64_t f = rev ? (ix.l_pac << 1) - 1 - p : p;
	const int b = ix.pac[f >> 2] >> ((~f & 3) << 1) & 3;
	return rev ? 3 - b : b;
}
// the same when the contig is known (a chain's rid IS the contig of its first seed): no search, two independent loads
__device__ __forceinline__ void ema_clamp_window_rid(const DevIndex &ix, int64_t &beg, int rid, bool is_rev, int64_t &end)
{
	if (end < beg) { const int64_t t = beg; beg = end; end = t; }
	int64_t far_beg = ix.ctg_off[rid], far_end = ix.ctg_off[rid + 1];
	if (is_rev) {
		const int64_t t = far_beg;
		far_beg = (ix.l_pac << 1) - far_end;
		far_end = (ix.l_pac << 1) - t;
	}
	beg = beg > far_beg ? beg : far_beg;
	end = end < far_end ? end : far_end;
}
// bns_fetch_seq's clamping: [beg, end) cut to the contig (and strand) of `mid`; returns its rid
__device__ __forceinline__ int ema_clamp_window(const DevIndex &ix, int64_t &beg, int64_t mid, int64_t &end)
{
	if (end < beg) { const int64_t t = beg; beg = end; end = t; }
	int is_rev;
	const int rid = ema_pos2rid(ix, ema_depos(ix, mid, is_rev));
	int64_t far_beg = ix.ctg_off[rid], far_end = ix.ctg_off[rid + 1];
	if (is_rev) {
		const int64_t t = far_beg;
		far_beg = (ix.l_pac << 1) - far_end;
		far_end = (ix.l_pac << 1) - t;
	}
	beg = beg > far_beg ? beg : far_beg;
	end = end < far_end ? end : far_end;
	return rid;
}
// wave-cooperative copy of reference [beg, end) (one strand, already clamped) into dst as nt4 bytes.
// Every lane takes whole 4-byte words of the packed reference (16 bases each): ONE round trip to memory per 1024 bases of window.
// (Until round 3 every lane fetched its bases one byte-load at a time, l = lane, lane + 64, ...: a 450-base extension window was
// seven dependent round trips, a 700-base rescue window eleven -- at loaded-memory latency that was the largest single wait of
// K2's per-chain set-up and of K3.)
struct EmaWin { int64_t w0; int shift, n, n_dw; bool rev; };      // a window as words of the packed reference (wave-uniform)
__device__ __forceinline__ EmaWin ema_win(const DevIndex &ix, int64_t beg, int64_t end)
{
	EmaWin w;
	w.n = (int)(end - beg);
	w.rev = beg >= ix.l_pac;
	const int64_t f_lo = w.rev ? (ix.l_pac << 1) - end : beg;      // the window on the forward strand: [f_lo, f_lo + n)
	w.w0 = (f_lo >> 2) & ~(int64_t)3;                              // its first byte, rounded down to a word
	w.shift = (int)(f_lo - (w.w0 << 2));                           // 0..15: where f_lo sits in that word
	w.n_dw = w.n > 0 ? (w.shift + w.n + 15) >> 4 : 0;
	return w;
}
// word d of the window (d < n_dw); the packed array is padded by 8 bytes
__device__ __forceinline__ uint32_t ema_win_load(const DevIndex &ix, const EmaWin &w, int d) { return *reinterpret_cast<const uint32_t *>(ix.pac + w.w0 + 4 * (int64_t)d); }
// the 16 bases of word d into dst (nt4 bytes, window order)
__device__ __forceinline__ void ema_win_unpack(const EmaWin &w, int d, uint32_t word, uint8_t *dst)
{
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const int o = 16 * d + i - w.shift;       // index along the forward strand
		if (o >= 0 && o < w.n) {
			const int code = (int)(word >> (8 * (i >> 2) + ((~i & 3) << 1))) & 3;      // byte i / 4 of the word, first base in the high bits
			dst[w.rev ? w.n - 1 - o : o] = (uint8_t)(w.rev ? 3 - code : code);
		}
	}
}
__device__ __forceinline__ void ema_wave_fetch(const DevIndex &ix, int64_t beg, int64_t end, uint8_t *dst)
{
	const EmaWin w = ema_win(ix, beg, end);
	for (int d = (int)ema_lane(); d < w.n_dw; d += EMA_WAVE) ema_win_unpack(w, d, ema_win_load(ix, w, d), dst);
	ema_wave_sync();
}

#endif
