// ema_amd/csrc/k_seed_bwd.hip -- K1b: the backward phases of bwt_smem1 as tasks of their own machine, one lane per task.
//
// Replaces (un-vendored bwa, reached from reference src/bwabridge.c:236-237 -> mem_align1_core -> mem_collect_intv -> bwt_smem1):
// the second half of bwt_smem1 -- for i = x - 1 down to -1, every interval of the previous row is extended to the left by base
// q[i]; an interval that falls under min_intv occurrences dies, and the first one of a row to die is an SMEM unless a longer one
// reported before contains it; survivors whose size differs from the survivor before them make the next row -- for the searches
// of passes 1 and 2 whose forward phase K1 (k_seed.hip, the split form) ended with a SeedTask.
//
// Why apart from K1.  A wavefront of lane machines executes the code of every state one of its 64 lanes is in, and K1's machine has
// twenty states: ~850 vector instructions per tick for one extend per lane, 60 % of the ticks in backward rows (r05: the chip is
// short of instruction issue, not of gathers).  A backward row needs three states -- post the extend of the row's next entry,
// its result, the row's end -- one working list, no text, no window test, no forward coordinate: this machine's tick is a third of
// K1's, and a read's backward phases, independent of each other once their forward lists exist, run on as many lanes as there are
// searches instead of one after the other on the read's lane.  Results go to the read's interval list through an atomic counter
// (the consumer orders a read's list by (start, end): the order of discovery is not part of the result, k_seed.hip), the extends a
// task used are added to the read's count (DevOpts::seed_ext), so the lean budget gives up exactly the reads it gave up as one
// machine: those whose passes need more extends in all.
// Requests per tick: the two 32-byte rank blocks of k - 1 and k - 1 + size, or one k-mer table entry when the extended string is
// at most kmer_k bases (table mode only: the machine carries the string's 2-bit code, not k'), and the row's next entry -- from
// the forward list in the pool for the first row, walked from its end, from the lane's own list B (lane-interleaved slab, compacted
// in place as in K1) afterwards.  Bound: instruction issue + two dependent gathers per extend.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

#ifndef EMA_SEED_BWD_WPS
#define EMA_SEED_BWD_WPS 5
#endif

namespace { enum { B_IDLE = 0, B_ROW, B_RES, B_N }; }

// tasks / n_task / caps: SeedSplit (dev_types.h); pool: the forward lists; lists: gridDim.x * 256 lanes x EMA_LIST_CAP entries (list B,
// entry e of a lane at [(e << 6) + lane] of its wavefront's slab); counter: zero on entry
__global__ void __launch_bounds__(256, EMA_SEED_BWD_WPS)
ema_k_seed_bwd(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const int *__restrict__ map, const SeedTask *__restrict__ tasks,
               const int *__restrict__ n_task, int cap_heavy, int cap_light, const Intv *__restrict__ pool, Intv *__restrict__ intv,
               int *__restrict__ n_intv, int *__restrict__ status, Intv *__restrict__ lists, int *__restrict__ counter, unsigned long long *prof)
{
	__shared__ uint32_t lds_q[4][16 * 64];      // 2-bit read codes, 16 words per lane, lane-interleaved
	__shared__ uint32_t lds_n[4][8 * 64];       // N mask, 8 words per lane
	__shared__ uint64_t lds_rt[4][EMA_RANK_TABLE];
	const int lane = (int)(threadIdx.x & 63), wib = ema_uni((int)(threadIdx.x >> 6));
	const uint32_t *qw = lds_q[wib] + lane, *nm = lds_n[wib] + lane;
	const uint64_t *rt = lds_rt[wib];
	ema_rank_table_init(ix, lds_rt[wib]);
	Intv *wl = lists + ((size_t)blockIdx.x * (blockDim.x >> 6) + wib) * (EMA_LIST_CAP * 64) + lane;
	const int n_heavy = n_task[0] < cap_heavy ? n_task[0] : cap_heavy, n_light = n_task[1] < cap_light ? n_task[1] : cap_light;
	const int n_total = n_heavy + n_light, kk = ix.kmer_k;
	auto q = [&](int p_) -> int {
		const int code = (qw[(p_ >> 4) << 6] >> ((p_ & 15) << 1)) & 3;
		return ((nm[(p_ >> 5) << 6] >> (p_ & 31)) & 1) ? 4 : code;
	};
	int pc = B_IDLE, read = -1, i = 0, j = 0, n_prev = 0, n_curr = 0, rev = 0, min_intv = 1, n_mem_call = 0, last_mem_start = 0;
	int n_ext = 0, n_ext0 = 0, st = 0, req_c = 0, has_req = 0;
	uint64_t c0 = 0, c2 = 0, f0 = 0, f2 = 0, r0 = 0, r2 = 0, last_size = 0;      // interval being extended / first one pushed in this row / the extend's result
	uint32_t c_code = 0, c_end = 0, f_code = 0, f_end = 0, r_code = 0, req_code = 0, req_len = 0;
	const Intv *fl = nullptr, *ld_p = nullptr;      // the task's forward list; the entry to fetch with this tick's extend
	Intv ent; ent.x0 = ent.x1 = ent.x2 = ent.info = 0;
	size_t out_base = 0;
	bool exhausted = false;
	unsigned long long n_tick = 0, n_busy = 0;      // (prof)
	int n_first = 0;
	for (;;) {
		// ---- phase A: one pass of the control program (registers and LDS; the task fetch and an interval's slot wait for memory)
		if (!has_req && !exhausted) {
			int ev = 0;      // 2: report v as an SMEM, 3: push v onto the next row
			uint64_t v0 = 0, v2 = 0;
			uint32_t v_code = 0, v_start = 0, v_end = 0;
			bool nxt = false;
			if (pc == B_RES) {      // bwt_smem1, backward loop body for the row's entry c at read position i
				if (r2 < (uint64_t)min_intv) {
					if (n_curr == 0 && (n_mem_call == 0 || i + 1 < last_mem_start)) {
						++n_mem_call; last_mem_start = i + 1;
						if ((int)c_end - (i + 1) >= opt.min_seed_len) { ev = 2; v0 = c0; v2 = c2; v_start = (uint32_t)(i + 1); v_end = c_end; }
					}
				} else if (n_curr == 0 || r2 != last_size) { ev = 3; v0 = r0; v_code = r_code; v2 = r2; v_end = c_end; last_size = r2; }
				if (j + 1 < n_prev) { c0 = ent.x0; c_code = (uint32_t)ent.x1; c2 = ent.x2; c_end = (uint32_t)ent.info; }      // fetched with the extend
				nxt = true; pc = B_ROW;
			} else if (pc == B_N) {   // the read's start or an ambiguous base: every entry of the row dies, only the first can be reported
				if (n_curr == 0 && (n_mem_call == 0 || i + 1 < last_mem_start)) {
					++n_mem_call; last_mem_start = i + 1;
					if ((int)c_end - (i + 1) >= opt.min_seed_len) { ev = 2; v0 = c0; v2 = c2; v_start = (uint32_t)(i + 1); v_end = c_end; }
				}
				pc = B_IDLE;
			}
			if (ev == 3) {
				if (n_curr >= EMA_LIST_CAP) st |= EMA_ST_LIST_OVERFLOW;
				else {
					Intv e; e.x0 = v0; e.x1 = v_code; e.x2 = v2; e.info = v_end;
					wl[(size_t)n_curr << 6] = e;
					if (n_curr == 0) { f0 = v0; f2 = v2; f_code = v_code; f_end = v_end; }
					++n_curr;
				}
			} else if (ev == 2) {
				const int at = atomicAdd(n_intv + read, 1);
				if (at >= opt.intv_cap) { atomicSub(n_intv + read, 1); st |= EMA_ST_INTV_OVERFLOW; }
				else { Intv e; e.x0 = v0; e.x1 = 0; e.x2 = v2; e.info = (uint64_t)v_start << 32 | v_end; intv[out_base + at] = e; }
			}
			if (nxt && ++j == n_prev) {      // the row is done: the next one (its first entry is f), or the search is over
				if (n_curr == 0) pc = B_IDLE;
				else { n_prev = n_curr; n_curr = 0; rev = 0; j = 0; --i; c0 = f0; c2 = f2; c_code = f_code; c_end = f_end; }
			}
			if (pc == B_IDLE) {              // the task's totals out, the next task in
				if (read >= 0) {
					if (n_ext) atomicAdd(opt.seed_ext + read, n_ext);
					if (st) atomicOr(status + read, st);
					if (prof) {
						const int b = n_first <= 2 ? 0 : n_first <= 4 ? 1 : n_first <= 8 ? 2 : n_first <= 16 ? 3 : n_first <= 32 ? 4 : n_first <= 64 ? 5 : n_first <= 128 ? 6 : 7;
						atomicAdd(prof, 1ULL); atomicAdd(prof + 1, (unsigned long long)n_ext); atomicMax(prof + 4, (unsigned long long)n_ext);
						atomicAdd(prof + 8 + b, 1ULL); atomicAdd(prof + 16 + b, (unsigned long long)n_ext);
					}
					read = -1;
				}
				const int t = atomicAdd(counter, 1);
				if (t >= n_total) exhausted = true;
				else {
					const uint4 *tp = reinterpret_cast<const uint4 *>(tasks + (t < n_heavy ? t : cap_heavy + (t - n_heavy)));
					const uint4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
					c0 = (uint64_t)t0.y << 32 | t0.x; c2 = (uint64_t)t0.w << 32 | t0.z;
					c_code = t1.x; read = (int)t1.y; fl = pool + t1.z; min_intv = (int)t1.w;
					n_ext0 = (int)t2.x; n_prev = (int)(t2.y & 0xffff); i = (int)((t2.y >> 16) & 0xff) - 1; c_end = t2.y >> 24;
					n_ext = 0; st = 0; n_curr = 0; rev = 1; j = 0; n_mem_call = 0; last_mem_start = 0; n_first = n_prev;
					out_base = (size_t)read * opt.intv_cap;
					const int in_read = ema_in_read(map, read);
					const uint4 *pw = reinterpret_cast<const uint4 *>(qpack + (size_t)in_read * 24);
					const uint4 a = pw[0], b = pw[1], c = pw[2], d = pw[3], m0 = pw[4], m1 = pw[5];
					uint32_t *qd = lds_q[wib] + lane, *nd = lds_n[wib] + lane;
					qd[0 << 6] = a.x; qd[1 << 6] = a.y; qd[2 << 6] = a.z; qd[3 << 6] = a.w;
					qd[4 << 6] = b.x; qd[5 << 6] = b.y; qd[6 << 6] = b.z; qd[7 << 6] = b.w;
					qd[8 << 6] = c.x; qd[9 << 6] = c.y; qd[10 << 6] = c.z; qd[11 << 6] = c.w;
					qd[12 << 6] = d.x; qd[13 << 6] = d.y; qd[14 << 6] = d.z; qd[15 << 6] = d.w;
					nd[0 << 6] = m0.x; nd[1 << 6] = m0.y; nd[2 << 6] = m0.z; nd[3 << 6] = m0.w;
					nd[4 << 6] = m1.x; nd[5 << 6] = m1.y; nd[6 << 6] = m1.z; nd[7 << 6] = m1.w;
					pc = B_ROW;
				}
			}
			if (pc == B_ROW) {               // the one place that looks up the row's base and posts the extend
				const int b = i >= 0 ? q(i) : 4;
				if (b > 3) pc = B_N;
				else if (n_ext0 + ++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = B_IDLE; }      // too long for this tier
				else {
					has_req = 1; req_c = b;
					if (j + 1 < n_prev) ld_p = rev ? fl + (n_prev - 2 - j) : wl + ((size_t)(j + 1) << 6);      // first row: the forward list from its end
					const int rl = (int)c_end - i;      // the extended string is q[i .. c_end): short enough for the table?
					if (rl <= kk) { has_req = 2; req_len = (uint32_t)rl; req_code = ((uint32_t)b << (2 * (rl - 1))) | c_code; }
					pc = B_RES;
				}
			}
		}
		const unsigned long long busy = __ballot(!exhausted);
		if (!busy) break;
		if (prof) { ++n_tick; n_busy += (unsigned long long)__popcll(__ballot(has_req != 0)); }
		// ---- phase B: the tick's loads, issued together
		if (has_req) {
			if (j + 1 < n_prev) {
				const ulong2 *src = reinterpret_cast<const ulong2 *>(ld_p);
				const ulong2 lo = src[0], hi = src[1];
				ent.x0 = lo.x; ent.x1 = lo.y; ent.x2 = hi.x; ent.info = hi.y;
			}
			const bool tab = has_req == 2;
			const uint64_t pk = c0 - 1, pl = c0 - 1 + c2;
			const uint64_t qk = pk - (pk >= ix.primary ? 1 : 0), ql = pl - (pl >= ix.primary ? 1 : 0);      // '$' is not stored
			const uint4 *p0, *p2;
			if (tab) {
				const int L = (int)req_len;
				p0 = L <= EMA_KMER_WIDE ? reinterpret_cast<const uint4 *>(ix.kmer_wide + 2 * ((size_t)ema_kmer_base_wide(L) + req_code))
				                        : reinterpret_cast<const uint4 *>(ix.kmer_narrow + (size_t)ema_kmer_base_narrow(L) + req_code);
				p2 = p0;
			} else {
				p0 = reinterpret_cast<const uint4 *>(ix.occ + (qk >> 6));
				p2 = reinterpret_cast<const uint4 *>(ix.occ + (ql >> 6));
			}
			const int o1 = tab ? 0 : 1;      // (a table entry is 16 or 8 bytes: the second halves repeat the first)
			const uint4 a0 = p0[0], a1 = p0[o1], b0 = p2[0], b1 = p2[o1];
			if (tab) {
				const uint64_t ea = (uint64_t)a0.y << 32 | a0.x;
				if ((int)req_len <= EMA_KMER_WIDE) { r0 = ea; r2 = (uint64_t)a0.w << 32 | a0.z; }
				else { r0 = ea & 0xFFFFFFFFFFULL; r2 = ea >> 40; }
				r_code = req_code;
			} else {
				uint32_t o_size, n_gt;
				ema_extend_blocks(ix, rt, qk, ql, a0, a1, b0, b1, req_c & 3, r0, o_size, n_gt);
				r2 = o_size; r_code = 0;
			}
			has_req = 0;
		}
	}
	if (prof && lane == 0) { atomicAdd(prof + 2, n_tick); atomicAdd(prof + 3, n_busy); atomicMax(prof + 5, n_tick); }
}

extern "C" void ema_launch_seed_bwd(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const int *map, const SeedSplit *sp, Intv *intv,
                                    int *n_intv, int *status, Intv *lists, int *counter, int n_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL(ema_k_seed_bwd, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, qpack, map, sp->tasks, sp->n_task, sp->cap_heavy, sp->cap_light,
	                   sp->pool, intv, n_intv, status, lists, counter, sp->prof);
}
// resident 256-thread blocks per CU (sizes the grid and the list slabs)
extern "C" int ema_seed_bwd_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_seed_bwd, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
