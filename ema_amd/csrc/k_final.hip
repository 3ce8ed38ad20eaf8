// ema_amd/csrc/k_final.hip -- K4: final alignment of every candidate region (CIGAR, NM, position), one
// wavefront per read.
//
// Replaces the reference's bwa_smith_waterman() (reference src/bwabridge.c:301-311), which src/align.c calls
// for every region of both mates (:1013, :1038), i.e. bwa's mem_reg2aln: band inference from the local score,
// bwa_gen_cigar2 (gap-free fast path, or banded global alignment with traceback, retried with a doubled
// band up to three times), NM over the CIGAR, squeezing of a terminal deletion, soft clips, and the
// forward-strand position.  The read and the reference window sit in LDS; the direction matrix of the
// global DP lives in the wave's slab of HBM scratch.
#include <hip/hip_runtime.h>
#include "dev_regions.hpp"
#include "dev_prof.hpp"
#ifdef EMA_K34_PROF
__device__ unsigned long long ema_k4_lp[3][12];
extern "C" void ema_k4_prof_read(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ema_k4_lp), sizeof(ema_k4_lp)); static unsigned long long z[36]; (void)hipMemcpyToSymbol(HIP_SYMBOL(ema_k4_lp), z, sizeof(z)); }
#endif
#include "ema_engine.h"
#include <algorithm>
#include <cstring>

#define EMA_Z_BYTES ((size_t)256 * (EMA_RSEQ_CAP + 8))
#define EMA_FINAL_SLAB_BYTES (EMA_Z_BYTES + 4096 * 4 + 1024)
#define EMA_CIG_TMP 4096
#define EMA_FINAL_SLOT_OPS 32      // K4t: CIGAR operations of a region that go to the task's own slot of the arena
#define EMA_Z_LDS 6144      // bytes of LDS per wavefront for a staged direction matrix (a band of 21 columns x 290 rows)

struct DevAln {           // per candidate: what interpret_single_read_alignment reads (reference src/bwabridge.c:359-379)
	int64_t pos;
	int32_t is_rev, NM, n_cigar;
	uint32_t cigar_off;   // into the read's cigar pool
};

namespace {

__device__ __forceinline__ int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	const int d = l1 > l2 ? l1 - l2 : l2 - l1;
	return w < d ? d : w;
}

}  // namespace

// K4a: the gap-free regions, ONE LANE PER READ.  Most regions of most reads take bwa_gen_cigar2's gap-free path
// (query and reference span of equal length, inferred band 0 -- or few enough mismatches that no gapped alignment can
// beat the diagonal): the CIGAR is a single M run plus clips and NM is a
// mismatch count -- a few hundred scalar operations, which a whole wavefront per read (K4b below) issues 64 wide.
// Here every lane walks its own read's regions in order for as long as they are gap-free, comparing the packed
// query with the packed reference directly in HBM; at the first region that needs the dynamic program it stops,
// leaves `k_done[read]` = that region's index and the CIGAR pool's fill in cig_n[read], and puts the read on the
// todo list of K4b, which finishes it.
// qpack: K1's packed reads (16 words of 2-bit codes + 8 words of N mask per read).
__global__ void __launch_bounds__(256)
ema_k_final_simple(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
                   const int *__restrict__ n_pairs_dev, const int *__restrict__ map,
                   const DevReg *__restrict__ regs, const int *__restrict__ n_regs, DevAln *__restrict__ alns,
                   uint32_t *__restrict__ cigars, int *__restrict__ cig_n, int cig_cap, int *__restrict__ status,
                   int *__restrict__ k_done, int *__restrict__ todo, int *__restrict__ n_todo)
{
	const int read = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (read >= ema_work_count(n_reads, n_pairs_dev, 2)) return;
	const int nr = (status[read] | status[read ^ 1]) ? 0 : n_regs[read];      // flagged pairs are redone by the full-capacity tier
	int pool_n = 0, st = 0, k = 0;
	if (nr > 0) {
		const int in_read = ema_in_read(map, read);
		const int l_query = (int)(off[in_read + 1] - off[in_read]);
		const uint32_t *qp = qpack + (size_t)in_read * 24;
		const int64_t l_pac = ix.l_pac;
		uint32_t *pool = cigars + (size_t)read * cig_cap;
		for (; k < nr; ++k) {
			const DevReg ar = regs[(size_t)read * opt.reg_cap + k];
			DevAln out;
			out.pos = -1; out.is_rev = 0; out.NM = -1; out.n_cigar = 0; out.cigar_off = (uint32_t)pool_n;
			const int qb = ar.qb, qe = ar.qe, lq = qe - qb;
			const int64_t rb = ar.rb, re = ar.re;
			const int rlen = (int)(re - rb);
			const bool ok = lq > 0 && rb < re && !(rb < l_pac && re > l_pac) && rb >= 0 && re <= l_pac << 1 && rlen <= EMA_RSEQ_CAP;
			if (!ok) { if (rlen > EMA_RSEQ_CAP) st |= EMA_ST_RSEQ_OVERFLOW; alns[(size_t)read * opt.reg_cap + k] = out; continue; }
			int w2 = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_del, opt.e_del);
			{
				const int t = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_ins, opt.e_ins);
				w2 = w2 > t ? w2 : t;
			}
			if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
			w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
			if (lq != rlen) break;      // needs the dynamic program: K4b takes over from here
			// NM = positions where the read differs from the reference (an ambiguous read base always differs).  On the
			// reverse strand the reference is the complement of the forward strand read from the far end.
			int nm = 0, pen = 0;      // pen: what the differing positions cost the all-M alignment (a+b each, a+1 for an N)
			{
				const bool rs = rb >= l_pac;
				int64_t f = rs ? (l_pac << 1) - 1 - rb : rb;      // forward-strand coordinate of reference base j = 0
				uint32_t qw = 0, nw = 0, pw = 0;
				int64_t pw_at = -1;
				int qw_at = -1, nw_at = -1;
				for (int j = 0; j < lq; ++j, f += rs ? -1 : 1) {
					const int i = qb + j;
					if ((i >> 4) != qw_at) { qw_at = i >> 4; qw = qp[qw_at]; }
					if ((i >> 5) != nw_at) { nw_at = i >> 5; nw = qp[16 + nw_at]; }
					if ((f >> 4) != pw_at) { pw_at = f >> 4; pw = *reinterpret_cast<const uint32_t *>(ix.pac + (pw_at << 2)); }
					const uint32_t code = (qw >> ((i & 15) << 1)) & 3;
					const uint32_t b = (pw >> ((((uint32_t)f >> 2) & 3) << 3) >> ((~(uint32_t)f & 3) << 1)) & 3;
					const bool amb = (nw >> (i & 31)) & 1, diff = code != (rs ? 3 - b : b);
					nm += (int)(amb | diff);
					pen += amb ? opt.a + 1 : diff ? opt.a + opt.b : 0;
				}
			}
			// w2 == 0 is bwa_gen_cigar2's own gap-free path.  Otherwise it runs the banded global alignment -- whose result
			// is still the single M run when no gapped path can score more: between spans of equal length such a path has
			// an insertion and a deletion and at most lq - 1 aligned pairs, i.e. at most (lq-1)a - (o_ins+e_ins+o_del+e_del),
			// and ties go to M in ksw_global2's traceback.
			if (!(w2 == 0 || pen <= opt.a + opt.o_ins + opt.e_ins + opt.o_del + opt.e_del)) break;
			out.NM = nm;
			const int is_rev = (rb < l_pac ? rb : re - 1) >= l_pac;
			const int64_t pos = is_rev ? (l_pac << 1) - 1 - (re - 1) : rb;
			out.is_rev = is_rev;
			const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
			const int n_final = (clip5 ? 1 : 0) + 1 + (clip3 ? 1 : 0);
			if (pool_n + n_final > cig_cap) st |= EMA_ST_CIGAR_OVERFLOW;
			else {
				uint32_t *dst = pool + pool_n;
				int o = 0;
				if (clip5) dst[o++] = (uint32_t)clip5 << 4 | 3;
				dst[o++] = (uint32_t)lq << 4;
				if (clip3) dst[o++] = (uint32_t)clip3 << 4 | 3;
				out.n_cigar = n_final;
				pool_n += n_final;
			}
			const int rid = ema_pos2rid(ix, pos);
			out.pos = rid >= 0 ? pos - ix.ctg_off[rid] : pos;
			alns[(size_t)read * opt.reg_cap + k] = out;
		}
	}
	cig_n[read] = pool_n;
	k_done[read] = k;
	if (st) atomicOr(status + read, st);
	if (k < nr) todo[atomicAdd(n_todo, 1)] = read;
}

// Reads with many regions left for K4b (a read inside a repeat family in the full-capacity tier: hundreds of global alignments)
// are one long serial job for the wavefront that owns them and set the length of the launch (full tier, r02: K4 43 of 105 ms).
// Their regions are independent up to WHERE in the read's CIGAR pool each one's operations go, so K4b sets such a read aside:
//   K4t (ema_k_final_t<1>): one wavefront per (read, region) task runs the region's alignment and leaves position, strand, NM and
//        the finished CIGAR (squeezed, clipped) in a result record + operations taken from an arena;
//   K4r (ema_k_final_t<2>): one wavefront per read lays the results into the pool in region order -- the same pool offsets, the
//        same capacity flags as the in-order loop.
// The lists and the arena are K2's (dev_types.h, HeavyCtl: idle while K4 runs); whatever does not fit is done in place.
struct FinalHeavy {
	uint8_t *arena;                    // null: nothing is set aside.  [0, tasks_cap x 48): result records; then CIGAR operations
	unsigned long long arena_bytes;
	unsigned long long *arena_used;    // bump allocator of the operations beyond a slot (zero on entry), relative to ovf_base
	unsigned long long ovf_base;       // = ops_base + tasks_cap slots
	unsigned long long ops_base;       // = tasks_cap x 48
	unsigned long long *reads;         // read | first task << 32
	unsigned long long *tasks;         // read << 32 | region
	int *n_reads, *n_tasks;
	int reads_cap, tasks_cap;
	int min_regions;                   // a read with at least this many regions left is set aside
};
struct FinalRes { DevAln aln; int32_t st, n_final; unsigned long long ops_at; int64_t pad_; };      // 48 bytes

// K4t: LDS per wavefront for its staged direction matrix, and the resident blocks per CU its registers must allow.  Round 4: 12 KB and
// two blocks per CU (64 KB of LDS per block, 150 registers).  [r5] 9 KB and three blocks per CU = three waves per SIMD:
// K4 per slice with the chip to itself 5.37 -> 5.16 ms, the steady state unchanged (profiles/r05_ab.txt); a matrix over 9 KB -- a
// band of 31 columns over 290 rows -- is walked in the slab as before.
#ifndef EMA_K4T_ZLDS
#define EMA_K4T_ZLDS 9216
#endif
#ifndef EMA_K4T_MIN_BLOCKS
#define EMA_K4T_MIN_BLOCKS 3
#endif
#ifndef EMA_K4_MIN_BLOCKS
#define EMA_K4_MIN_BLOCKS 4      // K4b fits 128 registers without a spill (143 when left alone)
#endif
// K4b: the regions K4a left (one wavefront per read of its todo list).  MODE 0: that; 1: K4t; 2: K4r (above).
// alns: n_reads x opt.reg_cap; cigars: n_reads x cig_cap ops (pool per read, regions in order)
template <int MODE>
__global__ void __launch_bounds__(256, MODE == 1 ? EMA_K4T_MIN_BLOCKS : EMA_K4_MIN_BLOCKS)
ema_k_final_t(DevIndex ix, DevOpts opt, const uint8_t *__restrict__ bases, const uint32_t *__restrict__ off, int n_reads,
            const int *__restrict__ n_pairs_dev, const int *__restrict__ map,
            const DevReg *__restrict__ regs, const int *__restrict__ n_regs, DevAln *__restrict__ alns,
            uint32_t *__restrict__ cigars, int *__restrict__ cig_n, int cig_cap, int *__restrict__ status,
            const int *__restrict__ k_done, const int *__restrict__ todo, const int *__restrict__ n_todo,
            uint8_t *__restrict__ slabs, int *__restrict__ counter, int *dbg, FinalHeavy fh)
{
#define EMA_DBG(stage, val) do { if (dbg && lane == 0) { __hip_atomic_store(dbg + slot * 4 + 1, (stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(dbg + slot * 4 + 2, (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } } while (0)
	__shared__ uint8_t lds_q[4][256];
	__shared__ uint8_t lds_r[4][EMA_RSEQ_CAP];
	constexpr int Z_LDS = MODE == 1 ? EMA_K4T_ZLDS : EMA_Z_LDS;      // (K4t's regions are the gapped ones of repeat-rich reads: wider bands)
	__shared__ __attribute__((aligned(16))) uint8_t lds_z[MODE == 2 ? 1 : 4][MODE == 2 ? 16 : Z_LDS];      // the direction matrix, staged for the traceback
	const int lane = (int)ema_lane();
	const int wib = ema_uni((int)(threadIdx.x >> 6));      // scalar: slab and LDS pointers derived from it stay in SGPRs
	const int slot = (int)(blockIdx.x * (blockDim.x >> 6)) + wib;
	uint8_t *z = slabs + (size_t)slot * EMA_FINAL_SLAB_BYTES;
	uint32_t *ctmp = (uint32_t *)(z + EMA_Z_BYTES);
	uint8_t *query = lds_q[wib], *rseq = lds_r[wib];
	const int64_t l_pac = ix.l_pac;
	FinalRes *res = reinterpret_cast<FinalRes *>(fh.arena);

	EmaClaim claim;      // work items four at a time, with their list entries (dev_common.hpp)
	// (make prof-lib: phase clocks, dev_prof.hpp) 0 claim, 1 the read in / setting aside, 2 region record, window, band, 3 global DP, 4 traceback,
	// 5 NM, squeeze, clips, operations out, 6 K4r: a result placed, 7 the read's totals out
	EMA_LP_DECL(lp);
	for (;;) {
		EMA_LP_UPTO(lp, 7);
		int read = 0;
		unsigned long long list_entry = 0;
		if (MODE == 1) { const int n = *fh.n_tasks; read = ema_claim_next(claim, counter, n < fh.tasks_cap ? n : fh.tasks_cap, fh.tasks, list_entry); }
		else if (MODE == 2) { const int n = *fh.n_reads; read = ema_claim_next(claim, counter, n < fh.reads_cap ? n : fh.reads_cap, fh.reads, list_entry); }
		else { int t = 0; read = ema_claim_next(claim, counter, *n_todo, todo, t); list_entry = (unsigned long long)(unsigned)t; }
		if (read < 0) break;
		EMA_LP_UPTO(lp, 0); EMA_LP_ITEM(lp);
		int task = -1, task_k = 0, first_task = 0;
		if (MODE == 1) {
			task = read;
			const unsigned long long t = list_entry;
			if (t == ~0ULL) continue;      // a claim K4b gave back
			read = (int)(t >> 32); task_k = (int)(uint32_t)t;
		} else if (MODE == 2) {
			const unsigned long long t = list_entry;
			if (t == ~0ULL) continue;
			read = (int)(uint32_t)t; first_task = (int)(t >> 32);
		} else read = (int)(unsigned)list_entry;
		if (dbg && lane == 0) __hip_atomic_store(dbg + slot * 4, read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		EMA_DBG(1, 0);
		const int in_read = ema_uni(ema_in_read(map, read));
		const int l_query = (int)(off[in_read + 1] - off[in_read]);
		const int nr = ema_uni(n_regs[read]);
		const int k0 = ema_uni(k_done[read]);
		if (MODE == 0 && fh.arena && nr - k0 >= fh.min_regions) {      // set aside: a place on the read list, tasks for its regions -- or done here after all
			long long ri = -1, tb = -1;
			if (lane == 0) {
				ri = atomicAdd(fh.n_reads, 1);
				if (ri >= fh.reads_cap) ri = -1;
				if (ri >= 0) {
					tb = atomicAdd(fh.n_tasks, nr - k0);
					if (tb + (nr - k0) > fh.tasks_cap) {
						for (long long j = tb; j < fh.tasks_cap && j < tb + (nr - k0); ++j) fh.tasks[j] = ~0ULL;
						tb = -1;
					}
					fh.reads[ri] = tb >= 0 ? ((unsigned long long)(uint32_t)read | (unsigned long long)tb << 32) : ~0ULL;
				}
			}
			tb = (long long)ema_uni((int64_t)__shfl(tb, 0));
			if (tb >= 0) {
				for (int k = k0 + lane; k < nr; k += EMA_WAVE) fh.tasks[tb + (k - k0)] = (unsigned long long)(uint32_t)read << 32 | (uint32_t)k;
				EMA_DBG(9, -(nr - k0));
				continue;
			}
		}
		if (MODE != 2) {
			for (int i = lane; i < l_query; i += EMA_WAVE) query[i] = bases[off[in_read] + i];
			ema_wave_sync();
		}
		uint32_t *pool = cigars + (size_t)read * cig_cap;
		int pool_n = MODE == 1 ? 0 : ema_uni(cig_n[read]), st = 0;
		EMA_LP_UPTO(lp, 1);
		for (int k = MODE == 1 ? task_k : k0; k < (MODE == 1 ? task_k + 1 : nr); ++k) {
			EMA_DBG(2, k);
			if (MODE == 2) {      // K4r: the region's result as K4t left it, placed at the pool's fill
				const FinalRes *rp = res + (first_task + (k - k0));
				DevAln out = rp->aln;
				out.pos = ema_uni(out.pos); out.is_rev = ema_uni(out.is_rev); out.NM = ema_uni(out.NM);
				const int n_final = ema_uni(rp->n_final);
				const unsigned long long at = ema_uni((uint64_t)rp->ops_at);
				st |= ema_uni(rp->st);
				out.n_cigar = 0; out.cigar_off = (uint32_t)pool_n;
				if (n_final > 0) {
					if (pool_n + n_final > cig_cap) st |= EMA_ST_CIGAR_OVERFLOW;
					else {
						const uint32_t *src = reinterpret_cast<const uint32_t *>(fh.arena + at);
						for (int i = lane; i < n_final; i += EMA_WAVE) pool[pool_n + i] = src[i];
						out.n_cigar = n_final;
						pool_n += n_final;
					}
				}
				if (lane == 0) alns[(size_t)read * opt.reg_cap + k] = out;
				EMA_LP_UPTO(lp, 6);
				continue;
			}
			const DevReg ar = ema_uni(regs[(size_t)read * opt.reg_cap + k]);
			DevAln out;
			out.pos = -1; out.is_rev = 0; out.NM = -1; out.n_cigar = 0; out.cigar_off = (uint32_t)pool_n;
			const int qb = ar.qb, qe = ar.qe, lq = qe - qb;
			const int64_t rb = ar.rb, re = ar.re;
			const bool rev = rb >= l_pac;
			const int rlen = (int)(re - rb);
			const bool ok = lq > 0 && rb < re && !(rb < l_pac && re > l_pac) && rb >= 0 && re <= l_pac << 1 && rlen <= EMA_RSEQ_CAP;
			if (!ok) {
				if (rlen > EMA_RSEQ_CAP) st |= EMA_ST_RSEQ_OVERFLOW;
				if (MODE == 1) { if (lane == 0) { FinalRes r; r.aln = out; r.st = st; r.n_final = 0; r.ops_at = 0; r.pad_ = 0; res[task] = r; } }
				else if (lane == 0) alns[(size_t)read * opt.reg_cap + k] = out;
				continue;
			}
			ema_wave_fetch(ix, rb, re, rseq);
			// reversed views so that indels are left-aligned on the forward strand (bwa_gen_cigar2)
			const EmaSeq qs{rev ? query + qe - 1 : query + qb, rev ? -1 : 1};
			const EmaSeq ts{rev ? rseq + rlen - 1 : rseq, rev ? -1 : 1};
			int w2 = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_del, opt.e_del);
			{
				const int t = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_ins, opt.e_ins);
				w2 = w2 > t ? w2 : t;
			}
			if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
			int score = 0, last_sc = -(1 << 30), first = EMA_CIG_TMP, n_cig = 0;
			for (int it = 0;;) {
				w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
				if (lq == rlen && w2 == 0) {      // gap-free
					int part = 0;
					for (int i = lane; i < lq; i += EMA_WAVE) part += ema_score(opt, ts.at(i), qs.at(i));
					score = ema_wave_sum(part);
					first = EMA_CIG_TMP - 1; n_cig = 1;
					ema_wave_sync();
					if (lane == 0) ctmp[first] = (uint32_t)lq << 4;
					ema_wave_sync();
				} else {
					const int w = ema_cigar_band(opt, lq, rlen, w2);
					EMA_LP_UPTO(lp, 2);
					score = ema_wave_global(opt, lq, qs, rlen, ts, w, z);
					ema_wave_sync();
					EMA_LP_UPTO(lp, 3);
					// The traceback is ~lq + rlen DEPENDENT one-byte reads by one lane: from the slab each is a round trip to L2 -- a third
					// of K4b's and three fifths of K4t's wavefront lifetimes (r03, make prof-lib) -- so a matrix that fits is first copied
					// to LDS by the whole wavefront (one round trip).  Writing it to LDS in the first place makes the DP's rows wait for
					// their byte stores (r02: measured, slower); to the slab they are fire-and-forget.
					const int n_col = lq < 2 * w + 1 ? lq : 2 * w + 1, zb = n_col * rlen;
					int f = 0;
					if (MODE != 2 && zb <= Z_LDS) {
						const uint32_t *src = reinterpret_cast<const uint32_t *>(z);
						uint32_t *dst = reinterpret_cast<uint32_t *>(lds_z[MODE == 2 ? 0 : wib]);
						for (int i = lane; i < (zb + 3) >> 2; i += EMA_WAVE) dst[i] = src[i];
						ema_wave_sync();
						f = ema_traceback((const EMA_LDS uint8_t *)lds_z[MODE == 2 ? 0 : wib], lq, rlen, w, ctmp, EMA_CIG_TMP);
					} else f = ema_traceback((const uint8_t *)z, lq, rlen, w, ctmp, EMA_CIG_TMP);
					ema_wave_sync();
					EMA_LP_UPTO(lp, 4);
					if (f < 0) { st |= EMA_ST_CIGAR_OVERFLOW; first = EMA_CIG_TMP; n_cig = 0; }
					else { first = f; n_cig = EMA_CIG_TMP - f; }
				}
				if (score == last_sc || w2 == opt.w << 2) break;
				last_sc = score;
				w2 <<= 1;
				if (!(++it < 3 && score < ar.truesc - opt.a)) break;
			}
			// NM: mismatches inside M + inserted bases + deleted bases of interior D ops
			int nm = 0;
			{
				int x = 0, y = 0, n_gap = 0, n_mm = 0;
				for (int c = 0; c < n_cig; ++c) {
					const uint32_t cw = (uint32_t)ema_uni((int)ctmp[first + c]);
					const uint32_t op = cw & 0xf, len = cw >> 4;
					if (op == 0) {
						int part = 0;
						for (int i = lane; i < (int)len; i += EMA_WAVE) part += qs.at(x + i) != ts.at(y + i);
						n_mm += part;
						x += len; y += len;
					} else if (op == 2) {
						if (c > 0 && c < n_cig - 1) n_gap += len;
						y += len;
					} else { x += len; n_gap += len; }
				}
				nm = ema_wave_sum(n_mm) + n_gap;
			}
			out.NM = n_cig > 0 ? nm : -1;
			const int is_rev = (rb < l_pac ? rb : re - 1) >= l_pac;
			int64_t pos = is_rev ? (l_pac << 1) - 1 - (re - 1) : rb;
			out.is_rev = is_rev;
			// squeeze out a leading or trailing deletion, then add the soft clips
			int lo = first, hi = first + n_cig;
			if (n_cig > 0) {
				const uint32_t c0 = (uint32_t)ema_uni((int)ctmp[lo]), c1 = (uint32_t)ema_uni((int)ctmp[hi - 1]);
				if ((c0 & 0xf) == 2) { pos += c0 >> 4; ++lo; }
				else if ((c1 & 0xf) == 2) --hi;
			}
			const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
			const int n_final = (clip5 ? 1 : 0) + (hi - lo) + (clip3 ? 1 : 0);
			uint32_t *dst = nullptr;
			unsigned long long ops_at = 0;
			if (MODE == 1) {      // K4t: room for the operations from the arena; K4r applies the pool's capacity
				// a task's own slot when the operations fit it (nearly always), else from the arena's cursor: one atomic per region on one
				// address, with every wavefront of the chip behind it, was a sixth of K4t's lifetimes (r03, make prof-lib)
				long long at = -1;
				if (n_final <= EMA_FINAL_SLOT_OPS) at = (long long)(fh.ops_base + (unsigned long long)task * (EMA_FINAL_SLOT_OPS * 4));
				else {
					if (lane == 0) {
						at = (long long)(fh.ovf_base + atomicAdd(fh.arena_used, (unsigned long long)n_final * 4));
						if ((unsigned long long)at + (unsigned long long)n_final * 4 > fh.arena_bytes) at = -1;
					}
					at = (long long)ema_lane_val((int64_t)at, 0);
				}
				if (at < 0) st |= EMA_ST_CIGAR_OVERFLOW;
				else { dst = reinterpret_cast<uint32_t *>(fh.arena + at); ops_at = (unsigned long long)at; }
			} else if (pool_n + n_final > cig_cap) { st |= EMA_ST_CIGAR_OVERFLOW; }
			else dst = pool + pool_n;
			if (dst) {
				ema_wave_sync();
				if (lane == 0 && clip5) dst[0] = (uint32_t)clip5 << 4 | 3;
				const int o5 = clip5 ? 1 : 0;
				for (int i = lane; i < hi - lo; i += EMA_WAVE) dst[o5 + i] = ctmp[lo + i];
				if (lane == 0 && clip3) dst[o5 + hi - lo] = (uint32_t)clip3 << 4 | 3;
				if (MODE != 1) { out.n_cigar = n_final; pool_n += n_final; }
			}
			const int rid = ema_pos2rid(ix, pos);
			out.pos = rid >= 0 ? pos - ix.ctg_off[rid] : pos;
			if (MODE == 1) { if (lane == 0) { FinalRes r; r.aln = out; r.st = st; r.n_final = dst ? n_final : 0; r.ops_at = ops_at; r.pad_ = 0; res[task] = r; } }
			else if (lane == 0) alns[(size_t)read * opt.reg_cap + k] = out;
			ema_wave_sync();
			EMA_LP_UPTO(lp, 5);
		}
		if (MODE != 1 && lane == 0) { cig_n[read] = pool_n; if (st) atomicOr(status + read, st); }
		EMA_DBG(9, 0);
	}
#ifdef EMA_K34_PROF
	EMA_LP_FLUSH(lp, &ema_k4_lp[MODE][0]);
#endif
#undef EMA_DBG
}

// Packs the per-read slots into the contiguous arrays handed to the host: cand[cand_off[r] + k] and the
// read's CIGAR ops at cigar[cig_off[r] ..).  One wave per read.
__global__ void __launch_bounds__(256)
ema_k_pack(int n_reads, const int *__restrict__ n_pairs_dev, const int *__restrict__ status, int reg_cap, const DevReg *__restrict__ regs, const int *__restrict__ n_regs, const DevAln *__restrict__ alns,
           const uint32_t *__restrict__ cigars, const int *__restrict__ cig_n, int cig_cap,
           const uint64_t *__restrict__ cand_off, const uint64_t *__restrict__ cig_off, uint64_t cig_base,
           ema_cand_t *__restrict__ cand, uint32_t *__restrict__ cigar_out, uint64_t cand_cap, uint64_t cigar_cap)
{
	const int lane = (int)ema_lane();
	const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * blockDim.x) >> 6);
	n_reads = ema_work_count(n_reads, n_pairs_dev, 2);
	for (int r = wave; r < n_reads; r += n_waves) {
		if (status[r]) continue;      // redone by the full-capacity tier; the host fills this read's slots from there
		const int nr = n_regs[r];
		const uint64_t co = cand_off[r], go = cig_off[r];
		if (co + (uint64_t)nr > cand_cap || go + (uint64_t)cig_n[r] > cigar_cap) continue;      // (the host sees the totals and reports it)
		for (int k = lane; k < nr; k += EMA_WAVE) {
			const DevReg g = regs[(size_t)r * reg_cap + k];
			const DevAln a = alns[(size_t)r * reg_cap + k];
			ema_cand_t c;
			c.rb = g.rb; c.re = g.re; c.qb = g.qb; c.qe = g.qe; c.rid = g.rid; c.score = g.score; c.truesc = g.truesc;
			c.sub = g.sub; c.alt_sc = 0; c.csub = g.csub; c.sub_n = 0; c.w = g.w; c.seedcov = g.seedcov;
			c.secondary = g.secondary; c.secondary_all = 0; c.seedlen0 = g.seedlen0; c.n_comp = g.n_comp; c.is_alt = g.is_alt;
			c.frac_rep = g.frac_rep;
			c.pos = a.pos; c.is_rev = a.is_rev; c.NM = a.NM; c.n_cigar = a.n_cigar;
			c.cigar_off = (uint32_t)(cig_base + go + a.cigar_off);
			c.aln_score = g.score; c.aln_sub = g.sub > g.csub ? g.sub : g.csub;
			cand[co + k] = c;
		}
		const int nc = cig_n[r];
		for (int k = lane; k < nc; k += EMA_WAVE) cigar_out[go + k] = cigars[(size_t)r * cig_cap + k];
	}
}

// Lean tier -> full-capacity tier: every pair with a capacity flag on either read is marked (both reads, so that the
// result assembly skips them) and appended to the full tier's work list.  Pairs beyond the list's capacity stay marked
// and make ema_engine_fetch fail.
__global__ void __launch_bounds__(256)
ema_k_collect(int n_pairs, int first_pair, int *__restrict__ status, int *__restrict__ count, int *__restrict__ map, int cap)
{
	const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (p >= n_pairs) return;
	const int a = status[2 * p], b = status[2 * p + 1];
	if (!(a | b)) return;
	status[2 * p] = a | EMA_ST_REDO; status[2 * p + 1] = b | EMA_ST_REDO;
	const int at = atomicAdd(count, 1);
	if (at < cap) map[at] = first_pair + p;
}

extern "C" void ema_launch_collect(int n_pairs, int first_pair, int *status, int *count, int *map, int cap, hipStream_t stream)
{
	if (n_pairs <= 0) return;
	hipLaunchKernelGGL(ema_k_collect, dim3((n_pairs + 255) / 256), dim3(256), 0, stream, n_pairs, first_pair, status, count, map, cap);
}

extern "C" size_t ema_final_slab_bytes() { return EMA_FINAL_SLAB_BYTES; }

// K4 = K4a (gap-free regions, one lane per read) then K4b (the rest, one wavefront per read).  kdone / todo: n_reads ints
// each; n_todo: one int, zero on entry.
// heavy (may be null: nothing is set aside): K2's lists and arena (dev_types.h, HeavyCtl), idle while K4 runs; heavy_counters: five
// ints, zero on entry {reads set aside, their tasks, work queues of K4t and K4r} and arena_used: the operations' cursor (u64, zero on entry)
extern "C" void ema_launch_final(const DevIndex *ix, const DevOpts *opt, const uint8_t *bases, const uint32_t *qpack, const uint32_t *off,
                                 int n_reads, const int *n_pairs_dev, const int *map, const DevReg *regs, const int *n_regs,
                                 DevAln *alns, uint32_t *cigars, int *cig_n, int cig_cap, int *status, int *kdone, int *todo, int *n_todo,
                                 uint8_t *slabs, int *counter, int n_blocks, hipStream_t stream, int *dbg,
                                 const HeavyCtl *heavy, int *heavy_counters, unsigned long long *arena_used, int min_regions)
{
	if (n_reads <= 0) return;
	FinalHeavy fh;
	memset(&fh, 0, sizeof(fh));
	fh.min_regions = 1 << 30;
	if (heavy && heavy->arena && min_regions > 0) {
		fh.arena = heavy->arena; fh.arena_bytes = heavy->arena_bytes; fh.arena_used = arena_used;
		fh.reads = heavy->reads; fh.tasks = heavy->tasks; fh.reads_cap = heavy->reads_cap;
		fh.tasks_cap = (int)std::min<unsigned long long>((unsigned long long)heavy->tasks_cap, heavy->arena_bytes / 2 / (sizeof(FinalRes) + EMA_FINAL_SLOT_OPS * 4));
		fh.n_reads = heavy_counters; fh.n_tasks = heavy_counters + 1;
		fh.min_regions = min_regions;
		fh.ops_base = (unsigned long long)fh.tasks_cap * sizeof(FinalRes);      // the operations follow the result records: a slot per task, then the cursor's room
		fh.ovf_base = fh.ops_base + (unsigned long long)fh.tasks_cap * (EMA_FINAL_SLOT_OPS * 4);
	}
	hipLaunchKernelGGL(ema_k_final_simple, dim3((n_reads + 255) / 256), dim3(256), 0, stream, *ix, *opt, qpack, off, n_reads, n_pairs_dev, map,
	                   regs, n_regs, alns, cigars, cig_n, cig_cap, status, kdone, todo, n_todo);
	hipLaunchKernelGGL(ema_k_final_t<0>, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, bases, off, n_reads, n_pairs_dev, map, regs, n_regs, alns,
	                   cigars, cig_n, cig_cap, status, kdone, todo, n_todo, slabs, counter, dbg, fh);
	if (fh.arena) {
		hipLaunchKernelGGL(ema_k_final_t<1>, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, bases, off, n_reads, n_pairs_dev, map, regs, n_regs, alns,
		                   cigars, cig_n, cig_cap, status, kdone, todo, n_todo, slabs, heavy_counters + 2, dbg, fh);
		hipLaunchKernelGGL(ema_k_final_t<2>, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, bases, off, n_reads, n_pairs_dev, map, regs, n_regs, alns,
		                   cigars, cig_n, cig_cap, status, kdone, todo, n_todo, slabs, heavy_counters + 3, dbg, fh);
	}
}

extern "C" size_t ema_sizeof_aln() { return sizeof(DevAln); }

extern "C" void ema_launch_pack(int n_reads, const int *n_pairs_dev, const int *status, int reg_cap, const DevReg *regs, const int *n_regs, const DevAln *alns, const uint32_t *cigars,
                                const int *cig_n, int cig_cap, const uint64_t *cand_off, const uint64_t *cig_off,
                                uint64_t cig_base, ema_cand_t *cand, uint32_t *cigar_out, uint64_t cand_cap, uint64_t cigar_cap, int n_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL(ema_k_pack, dim3(n_blocks), dim3(256), 0, stream, n_reads, n_pairs_dev, status, reg_cap, regs, n_regs, alns, cigars, cig_n, cig_cap,
	                   cand_off, cig_off, cig_base, cand, cigar_out, cand_cap, cigar_cap);
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_final_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_final_t<0>, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
