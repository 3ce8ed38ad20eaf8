// ema_amd/csrc/k_sam.hip -- SAM text on the device (the writer part; include/ema_sam.h, ema_sam_dev_*).
//
// print_sam_record() (reference src/samrecord.c:104-284) for a whole bucket's selected records: ONE LANE RENDERS ONE LINE, twice --
//   ema_k_sam_len    the line's length (the same rendering routine over a writer that only counts), with the wave's exclusive
//                    prefix sum taken at once: local[i] = offset of line i within its chunk of 64 lines, chunk_tot[i / 64];
//   ema_k_sam_tops   one wave: chunk totals -> chunk bases, grand total;
//   ema_k_sam_write  the text, each lane at chunk_base[i / 64] + local[i].
// What a line is made of is already here: the bucket's names / bases / qualities / barcodes as the reader laid them out (uploaded
// as they are), the batch's CIGAR operations, the contig names, and 52 bytes per selected record from the cloud stage
// (ema_sam_desc: indices, the printed MAPQ and the "%.5g" text of gamma, which need libm / libc).  Integer and byte work only.
// Lines are ~450 bytes of which 300 are the read and its qualities: the writer gathers four bytes into a register and stores
// aligned words (the bytes before the line's first word boundary and after its last one go one at a time: those words are
// shared with the neighbouring lines, i.e. other lanes); plain copies load words at any alignment.  A lane's loads and stores
// walk its own line, so a wave touches 64 lines at a time -- ~7 cache lines per lane, all consumed before they are evicted.
// Byte for byte the text of host_sam.cpp's put_line(): tests/test_sam_format.py (interpreter, CPU) and the golden SAM cases (GPU).
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "dev_sam.h"

namespace {

enum { kPaired = 1, kProper = 2, kUnmapped = 4, kMateUnmapped = 8, kReversed = 16, kMateReversed = 32, k1st = 64, k2nd = 128,
       kDup = 1024 };      // reference include/samrecord.h:73-81

__device__ __forceinline__ uint32_t load_u32(const char *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// complement of A C G T N by the low four bits of the byte (1 3 7 4 14: all different), 0 elsewhere (rc(), src/samrecord.c:86-102)
__device__ __forceinline__ uint32_t comp_of(uint32_t c)
{
	const uint64_t lo = (uint64_t)'T' << 8 | (uint64_t)'G' << 24 | (uint64_t)'A' << 32 | (uint64_t)'C' << 56;      // nibbles 1, 3, 4, 7
	const uint32_t t = c & 15u;
	return t < 8 ? (uint32_t)(lo >> (8 * t)) & 255u : (t == 14 ? (uint32_t)'N' : 0u);
}

// ---- the two writers -------------------------------------------------------------------------------------------------------------
struct CountW {
	uint32_t n = 0;
	__device__ __forceinline__ void ch(char) { ++n; }
	__device__ __forceinline__ void copy(const char *, int k) { n += (uint32_t)k; }
	__device__ __forceinline__ bool rcopy(const char *, int k, bool) { n += (uint32_t)k; return true; }
	__device__ __forceinline__ void fin() {}
};

struct TextW {
	char *p;            // the next byte (before the first word boundary) or the next word
	uint32_t acc = 0;   // nb bytes gathered for the word at p
	int nb = 0;
	bool al;
	__device__ __forceinline__ explicit TextW(char *q) : p(q), al(((uintptr_t)q & 3) == 0) {}
	__device__ __forceinline__ void ch(char c)
	{
		if (al) {
			acc |= (uint32_t)(uint8_t)c << (8 * nb);
			if (++nb == 4) { *(uint32_t *)p = acc; p += 4; acc = 0; nb = 0; }
		} else {
			*p++ = c;
			al = ((uintptr_t)p & 3) == 0;
		}
	}
	__device__ __forceinline__ void word(uint32_t v)      // four more bytes, al set
	{
		if (nb == 0) { *(uint32_t *)p = v; p += 4; return; }
		*(uint32_t *)p = acc | v << (8 * nb);
		p += 4;
		acc = v >> (32 - 8 * nb);
	}
	__device__ __forceinline__ void copy(const char *s, int k)
	{
		int i = 0;
		while (!al && i < k) ch(s[i++]);
		for (; i + 4 <= k; i += 4) word(load_u32(s + i));
		for (; i < k; ++i) ch(s[i]);
	}
	// s[k-1] .. s[0], complemented or not; false: a byte without a complement
	__device__ __forceinline__ bool rcopy(const char *s, int k, bool complement)
	{
		bool ok = true;
		auto one = [&](uint32_t c) -> uint32_t {
			if (!complement) return c;
			const uint32_t y = comp_of(c);
			if (y == 0 || comp_of(y) != c) ok = false;      // (the round trip: only the five valid bytes survive it)
			return y;
		};
		int i = k;
		while (!al && i > 0) ch((char)one((uint8_t)s[--i]));
		for (; i >= 4; i -= 4) {
			const uint32_t v = load_u32(s + i - 4);
			word(one(v >> 24) | one((v >> 16) & 255u) << 8 | one((v >> 8) & 255u) << 16 | one(v & 255u) << 24);
		}
		while (i > 0) ch((char)one((uint8_t)s[--i]));
		return ok;
	}
	__device__ __forceinline__ void fin() { for (int k = 0; k < nb; ++k) p[k] = (char)(acc >> (8 * k)); }
};

// decimal digits from the highest power down (no digit buffer: a lane's array would live in scratch memory)
template <class W> __device__ __forceinline__ void put_u(W &w, uint32_t v)
{
	bool started = false;
#pragma unroll
	for (uint32_t p = 1000000000u; p >= 10u; p /= 10u) {
		if (v >= p || started) { const uint32_t d = v / p; v -= d * p; w.ch((char)('0' + d)); started = true; }
	}
	w.ch((char)('0' + v));
}
template <class W> __device__ __forceinline__ void put_i(W &w, int32_t v)
{
	if (v < 0) { w.ch('-'); put_u(w, 0u - (uint32_t)v); } else put_u(w, (uint32_t)v);
}
template <class W> __device__ __forceinline__ void put_u9(W &w, uint32_t v)      // exactly nine digits
{
#pragma unroll
	for (uint32_t p = 100000000u; p >= 10u; p /= 10u) { const uint32_t d = v / p; v -= d * p; w.ch((char)('0' + d)); }
	w.ch((char)('0' + v));
}
template <class W> __device__ __forceinline__ void put_i64(W &w, int64_t v)
{
	uint64_t u = (uint64_t)v;
	if (v < 0) { w.ch('-'); u = 0 - u; }
	if (u < 1000000000ull) { put_u(w, (uint32_t)u); return; }      // (a template length on a real contig ends here)
	const uint64_t hi = u / 1000000000ull;
	const uint32_t lo = (uint32_t)(u - hi * 1000000000ull);
	if (hi < 1000000000ull) put_u(w, (uint32_t)hi);
	else { const uint64_t top = hi / 1000000000ull; put_u(w, (uint32_t)top); put_u9(w, (uint32_t)(hi - top * 1000000000ull)); }
	put_u9(w, lo);
}
template <class W, int N> __device__ __forceinline__ void put_s(W &w, const char (&s)[N]) { for (int i = 0; i + 1 < N; ++i) w.ch(s[i]); }

template <class W> __device__ __forceinline__ void put_cigar(W &w, const SamJob &J, uint32_t off, int n)      // hard clips shown as soft: "MIDSS"
{
	const uint32_t *c = J.cigar + (off - J.cigar_lo);
	for (int i = 0; i < n; ++i) {
		const uint32_t op = c[i] & 15u;
		put_u(w, c[i] >> 4);
		w.ch(op == 0 ? 'M' : op == 1 ? 'I' : op == 2 ? 'D' : 'S');
	}
}
__device__ __forceinline__ int ref_len(const SamJob &J, uint32_t off, int n)      // get_rlen, src/samrecord.c:75-84
{
	const uint32_t *c = J.cigar + (off - J.cigar_lo);
	int l = 0;
	for (int i = 0; i < n; ++i) { const uint32_t op = c[i] & 15u; if (op == 0 || op == 2) l += (int)(c[i] >> 4); }
	return l;
}
template <class W> __device__ __forceinline__ void put_name(W &w, const SamJob &J, int32_t rid) { w.copy(J.names + J.name_off[rid], (int)(J.name_off[rid + 1] - J.name_off[rid])); }

template <class W> __device__ __forceinline__ void put_bc(W &w, const SamJob &J, uint64_t bc)      // decode_bc, src/util.c:78-95
{
	if (J.is_haplotag) {      // "A%02uC%02uB%02uD%02u"
		const char tag[4] = {'A', 'C', 'B', 'D'};
		for (int k = 0; k < 4; ++k) {
			const uint32_t v = (uint32_t)(bc >> (24 - 8 * k)) & 127u;
			w.ch(tag[k]);
			if (v < 10) w.ch('0');
			put_u(w, v);
		}
		return;
	}
	for (int i = 0; i < J.bc_len; ++i) { const uint32_t b = (uint32_t)bc & 3u; w.ch(b == 0 ? 'A' : b == 1 ? 'C' : b == 2 ? 'G' : 'T'); bc >>= 2; }
}

// print_sam_record(rec, mate) for line `line` of the bucket's output; false: a base outside ACGTN in a reversed read
template <class W> __device__ __forceinline__ bool render(W &w, const SamJob &J, uint32_t line)
{
	const uint32_t d0 = J.sel_at[line >> 1];
	const ema_sam_desc *first = J.desc + d0, *second = first->has_mate ? first + 1 : nullptr;
	const ema_sam_desc *rec = (line & 1) ? second : first, *mate = (line & 1) ? first : second;
	const uint32_t pair = first->pair;
	int flag = kPaired, mapq = 0;
	uint32_t pos = 0, r;      // r: the read this line prints
	if (rec) {
		pos = rec->pos; mapq = rec->mapq;
		if (rec->rev) flag |= kReversed;
		if (rec->duplicate) flag |= kDup;
		flag |= rec->mate == 0 ? k1st : k2nd;
		r = 2 * pair + rec->mate;
	} else {      // the line of an unaligned read: its mate's record stands in
		flag |= kUnmapped;
		flag |= mate->mate == 0 ? k2nd : k1st;
		r = 2 * pair + (1u - mate->mate);
	}
	if (mate) {
		if (rec && rec->rev != mate->rev && rec->rid == mate->rid) {      // is_pair, src/align.c:27-40
			const ema_sam_desc *r1 = rec, *r2 = mate;
			if (r2->rev) { r1 = mate; r2 = rec; }
			const int64_t d = (int64_t)(uint32_t)(r1->pos - r2->pos);      // the reference subtracts two uint32_t: never negative
			if ((int64_t)J.insert_min <= d && d <= (int64_t)J.insert_max) flag |= kProper;
		}
		if (mate->rev) flag |= kMateReversed;
	} else flag |= kMateUnmapped;
	{
		const uint32_t ib = J.id_off[pair], ie = J.id_off[pair + 1];
		if (ie > ib + 1) w.copy(J.ids + ib + 1, (int)(ie - ib - 1));
	}
	w.ch('\t'); put_i(w, flag); w.ch('\t');
	if (rec) put_name(w, J, rec->rid); else w.ch('*');
	w.ch('\t'); put_u(w, pos); w.ch('\t'); put_i(w, mapq); w.ch('\t');
	if (rec) put_cigar(w, J, rec->cigar_off, rec->n_cigar); else w.ch('*');
	if (mate) {
		const bool same_chrom = rec && mate->rid == rec->rid;
		w.ch('\t');
		if (same_chrom) w.ch('='); else put_name(w, J, mate->rid);
		w.ch('\t'); put_i(w, (int32_t)mate->pos);      // "%d" of a uint32_t
		if (same_chrom) {
			const int64_t p0 = (int64_t)rec->pos - 1 + (rec->rev ? ref_len(J, rec->cigar_off, rec->n_cigar) - 1 : 0);
			const int64_t p1 = (int64_t)mate->pos - 1 + (mate->rev ? ref_len(J, mate->cigar_off, mate->n_cigar) - 1 : 0);
			w.ch('\t');
			if (mate->n_cigar == 0 || rec->n_cigar == 0) w.ch('0');
			else put_i64(w, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else put_s(w, "\t0");
	} else put_s(w, "\t*\t0\t0");
	w.ch('\t');
	bool ok = true;
	{
		const uint32_t b = J.off[r];
		const int len = (int)(J.off[r + 1] - b);
		if (rec && rec->rev) {
			ok = w.rcopy(J.bases + b, len, true);
			w.ch('\t');
			(void)w.rcopy(J.quals + b, len, false);
		} else {
			w.copy(J.bases + b, len); w.ch('\t'); w.copy(J.quals + b, len);
		}
	}
	const uint64_t bc = J.bc[pair];
	if (rec) {
		put_s(w, "\tNM:i:"); put_i(w, rec->edit_dist); put_s(w, "\tBX:Z:"); put_bc(w, J, bc);
		if (!J.is_haplotag) { w.ch('-'); w.copy(J.bx, J.bx_len); }
		put_s(w, "\tXG:f:");
		for (int i = 0; i < (int)rec->gamma_len; ++i) w.ch(rec->gamma[i]);
		put_s(w, "\tMI:i:"); put_i(w, rec->cloud_id); put_s(w, "\tXF:i:"); put_i(w, (int32_t)rec->cloud_bad);
	} else {
		put_s(w, "\tBX:Z:"); put_bc(w, J, bc);
		if (!J.is_haplotag) put_s(w, "-1");      // the literal suffix, not bx_index (src/samrecord.c:255)
	}
	if (J.has_rg) { put_s(w, "\tRG:Z:"); w.copy(J.rg, J.rg_len); }
	if (rec && rec->xa >= 0) {
		const ema_sam_xa &a = J.xa[rec->xa];
		put_s(w, "\tXA:Z:");
		put_name(w, J, a.rid); w.ch(','); w.ch(a.rev ? '-' : '+'); put_i(w, (int32_t)a.pos); w.ch(',');
		put_cigar(w, J, a.cigar_off, a.n_cigar);
		w.ch(','); put_i(w, a.edit_dist); w.ch(';');
	}
	w.ch('\n');
	w.fin();
	return ok;
}

}  // namespace

__global__ void __launch_bounds__(256)
ema_k_sam_len(SamJob J, uint32_t *__restrict__ local, uint32_t *__restrict__ chunk_tot)
{
	const uint32_t line = blockIdx.x * 256u + threadIdx.x;
	CountW w;
	if (line < J.n_lines) (void)render(w, J, line);
	const int incl = ema_wave_incl_scan_add((int)w.n);      // (every lane of the wave is here: the grid is whole waves)
	if (line < J.n_lines) local[line] = (uint32_t)incl - w.n;
	const int tot = ema_lane_val(incl, 63);
	if (ema_lane() == 0 && line < J.n_lines) chunk_tot[line >> 6] = (uint32_t)tot;
}

// one wave: chunk totals -> exclusive chunk bases (64-bit: a bucket's text may pass 4 GB), the grand total to *total
__global__ void __launch_bounds__(64)
ema_k_sam_tops(uint32_t n_chunks, const uint32_t *__restrict__ chunk_tot, uint64_t *__restrict__ chunk_base, uint64_t *__restrict__ total)
{
	const uint32_t lane = ema_lane();
	const uint32_t per = (n_chunks + 63u) / 64u, lo = lane * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
	uint64_t mine = 0;
	for (uint32_t i = lo; i < hi; ++i) mine += chunk_tot[i];
	// exclusive prefix over the lanes of a 64-bit sum: two 32-bit scans would drop the carry, so the lanes' sums go through lane 0
	uint64_t before = 0, all = 0;
	for (int l = 0; l < 64; ++l) {
		const uint64_t v = (uint64_t)ema_lane_val((int64_t)mine, l);
		if ((uint32_t)l < lane) before += v;
		all += v;
	}
	uint64_t run = before;
	for (uint32_t i = lo; i < hi; ++i) { chunk_base[i] = run; run += chunk_tot[i]; }
	if (lane == 0) *total = all;
}

__global__ void __launch_bounds__(256)
ema_k_sam_write(SamJob J, const uint32_t *__restrict__ local, const uint64_t *__restrict__ chunk_base, char *__restrict__ text, int *__restrict__ bad)
{
	const uint32_t line = blockIdx.x * 256u + threadIdx.x;
	if (line >= J.n_lines) return;
	TextW w(text + chunk_base[line >> 6] + local[line]);
	if (!render(w, J, line)) atomicOr(bad, 1);
}

void ema_launch_sam_len(const SamJob &j, uint32_t *local, uint32_t *chunk_tot, hipStream_t st)
{
	if (!j.n_lines) return;
	hipLaunchKernelGGL(ema_k_sam_len, dim3((j.n_lines + 255u) / 256u), dim3(256), 0, st, j, local, chunk_tot);
}
void ema_launch_sam_tops(uint32_t n_chunks, const uint32_t *chunk_tot, uint64_t *chunk_base, uint64_t *total, hipStream_t st)
{
	if (!n_chunks) return;
	hipLaunchKernelGGL(ema_k_sam_tops, dim3(1), dim3(64), 0, st, n_chunks, chunk_tot, chunk_base, total);
}
void ema_launch_sam_write(const SamJob &j, const uint32_t *local, const uint64_t *chunk_base, char *text, int *bad, hipStream_t st)
{
	if (!j.n_lines) return;
	hipLaunchKernelGGL(ema_k_sam_write, dim3((j.n_lines + 255u) / 256u), dim3(256), 0, st, j, local, chunk_base, text, bad);
}
