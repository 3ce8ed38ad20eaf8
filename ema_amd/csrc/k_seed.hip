// ema_amd/csrc/k_seed.hip -- K1: SMEM / seed-interval collection on the FM-index (gfx950).
//
// Replaces, for a whole batch of reads, the seeding stage the reference reaches at
// src/bwabridge.c:236-237 (mem_align1_core -> mem_chain -> mem_collect_intv in the un-vendored
// bwa: bwt_smem1 x3 passes + bwt_seed_strategy1).  Output per read: the interval list
// (start, end, k, k', size) sorted by (start, end) -- exactly what mem_collect_intv leaves in
// aux->mem.
//
// Mapping to the hardware.  One bwt_extend is two dependent-free 64-byte reads (the occ blocks
// of k-1 and l) followed by a few popcounts; successive extends of one search are strictly
// dependent, so a search is latency-bound and the only way to load HBM is to keep many
// searches in flight.  A wavefront therefore carries EIGHT reads, one per group of 8 lanes:
// lanes 0-3 of a group fetch the block of k-1 (one 16-byte global_load_dwordx4 each = one
// 64-byte line), lanes 4-7 the block of l, and the group reduces the popcounts with DPP-range
// shuffles.  Every read is driven by a small state machine (forward phase / backward phase of
// an SMEM search, re-seeding pass, LAST-like pass) that yields each time it needs an extend; all
// eight machines of a wave then issue their loads together, so a wave keeps 16 cache lines in
// flight and a CU at 16-32 waves several hundred.  The per-search working lists (bwa's
// prev/curr vectors) live in a per-group slab of HBM scratch that stays L2-resident.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

namespace {

enum { PC_DONE = 0, PC_P1_NEXT, PC_P2_INIT, PC_P2_NEXT, PC_P3_NEXT, PC_FWD, PC_BWD, PC_S3, PC_FINISH };

struct SeedSM {
	// read
	const uint8_t *q;
	int len;
	Intv *out;           // EMA_INTV_CAP entries
	Intv *la, *lb;       // working lists, EMA_LIST_CAP entries each
	int status;
	// control
	int pc, pass, x, sm_x, min_intv, i, j;
	int n_prev, n_curr, rev, prev_is_a;
	int n_mem_call, last_mem_start, ret, seg_start, n_out, old_n, k2;
	uint64_t last_curr_size;
	// current interval (forward phases) and the list entry being extended (backward phase)
	uint64_t ik0, ik1, ik2; uint32_t ik_end;
	uint64_t p0, p1, p2; uint32_t p_end;
	// pending request
	bool has_req; int req_c, req_back;

	__device__ __forceinline__ Intv *prev() { return prev_is_a ? la : lb; }
	__device__ __forceinline__ Intv *curr() { return prev_is_a ? lb : la; }

	__device__ __forceinline__ void set_intv(const DevIndex &ix, int c)
	{
		ik0 = ix.L2[c] + 1; ik2 = ix.L2[c + 1] - ix.L2[c]; ik1 = ix.L2[3 - c] + 1;
	}
	__device__ __forceinline__ void push_curr(uint64_t a0, uint64_t a1, uint64_t a2, uint32_t end)
	{
		if (n_curr >= EMA_LIST_CAP) { status |= EMA_ST_LIST_OVERFLOW; return; }
		Intv e; e.x0 = a0; e.x1 = a1; e.x2 = a2; e.info = end;
		curr()[n_curr++] = e;
	}
	__device__ __forceinline__ void emit(uint64_t a0, uint64_t a1, uint64_t a2, int start, int end)
	{
		if (n_out >= EMA_INTV_CAP) { status |= EMA_ST_INTV_OVERFLOW; return; }
		Intv e; e.x0 = a0; e.x1 = a1; e.x2 = a2; e.info = (uint64_t)(uint32_t)start << 32 | (uint32_t)end;
		out[n_out++] = e;
	}
	__device__ __forceinline__ void after_forward()
	{
		ret = (int)curr()[n_curr - 1].info;      // longest match = last pushed
		prev_is_a ^= 1;                           // curr becomes prev, read in reverse order
		n_prev = n_curr; n_curr = 0; rev = 1;
		i = sm_x - 1; j = 0;
		pc = PC_BWD;
	}
	__device__ __forceinline__ void start_smem(const DevIndex &ix, int x_, int min_)
	{
		sm_x = x_; min_intv = min_;
		set_intv(ix, q[x_]); ik_end = (uint32_t)(x_ + 1);
		n_curr = 0; n_mem_call = 0; seg_start = n_out;
		i = x_ + 1;
		if (i >= len) { push_curr(ik0, ik1, ik2, ik_end); after_forward(); }
		else pc = PC_FWD;
	}
	__device__ __forceinline__ void end_smem()
	{
		// bwa reverses the MEMs of this call here; the list is fully sorted at PC_FINISH and entries with equal
		// (start, end) are identical, so the intermediate order has no effect on the result
		if (pass == 1) { x = ret; pc = PC_P1_NEXT; }
		else pc = PC_P2_NEXT;
	}
	// backward step bookkeeping for list entry p at query position i; `dead` = cannot be extended
	__device__ __forceinline__ void bwd_consume(bool dead, uint64_t o0, uint64_t o1, uint64_t o2, const DevOpts &opt)
	{
		if (dead) {
			if (n_curr == 0 && (n_mem_call == 0 || i + 1 < last_mem_start)) {
				++n_mem_call; last_mem_start = i + 1;
				if ((int)p_end - (i + 1) >= opt.min_seed_len) emit(p0, p1, p2, i + 1, (int)p_end);
			}
		} else if (n_curr == 0 || o2 != last_curr_size) {
			push_curr(o0, o1, o2, p_end);
			last_curr_size = o2;
		}
		if (++j == n_prev) {
			if (n_curr == 0) end_smem();
			else { prev_is_a ^= 1; n_prev = n_curr; n_curr = 0; rev = 0; j = 0; --i; }
		}
	}
	// run the control program until it needs an extend (has_req) or the read is finished (PC_DONE)
	__device__ void advance(const DevIndex &ix, const DevOpts &opt)
	{
		has_req = false;
		for (;;) {
			switch (pc) {
			case PC_P1_NEXT:
				while (x < len && q[x] > 3) ++x;
				if (x >= len) { pc = PC_P2_INIT; break; }
				start_smem(ix, x, 1);
				break;
			case PC_P2_INIT:
				pass = 2; old_n = n_out; k2 = 0; pc = PC_P2_NEXT;
				break;
			case PC_P2_NEXT: {
				bool found = false;
				while (k2 < old_n) {
					const Intv p = out[k2++];
					const int s = (int)(p.info >> 32), e = (int)(uint32_t)p.info;
					if (e - s < opt.split_len || p.x2 > (uint64_t)opt.split_width) continue;
					start_smem(ix, (s + e) >> 1, (int)p.x2 + 1);
					found = true;
					break;
				}
				if (!found) { pass = 3; x = 0; pc = opt.max_mem_intv > 0 ? PC_P3_NEXT : PC_FINISH; }
				break;
			}
			case PC_P3_NEXT:
				while (x < len && q[x] > 3) ++x;
				if (x >= len) { pc = PC_FINISH; break; }
				set_intv(ix, q[x]);
				i = x + 1;
				if (i >= len) { x = len; pc = PC_FINISH; }
				else pc = PC_S3;
				break;
			case PC_FWD:
				if (q[i] < 4) { has_req = true; req_back = 0; req_c = 3 - q[i]; return; }
				push_curr(ik0, ik1, ik2, ik_end);
				after_forward();
				break;
			case PC_BWD: {
				const Intv p = prev()[rev ? n_prev - 1 - j : j];
				p0 = p.x0; p1 = p.x1; p2 = p.x2; p_end = (uint32_t)p.info;
				const int c = (i < 0 || q[i] > 3) ? -1 : q[i];
				if (c >= 0) { has_req = true; req_back = 1; req_c = c; return; }
				bwd_consume(true, 0, 0, 0, opt);
				break;
			}
			case PC_S3:
				if (q[i] < 4) { has_req = true; req_back = 0; req_c = 3 - q[i]; return; }
				x = i + 1; pc = PC_P3_NEXT;
				break;
			case PC_FINISH:
				// order by info (start, end).  Entries with equal info describe the same substring and
				// are identical, so any correct sort reproduces ks_introsort(mem_intv)'s result.
				for (int a = 1; a < n_out; ++a) {
					const Intv t = out[a];
					int b = a - 1;
					while (b >= 0 && out[b].info > t.info) { out[b + 1] = out[b]; --b; }
					out[b + 1] = t;
				}
				pc = PC_DONE;
				return;
			default:
				return;
			}
		}
	}
	// apply the result ok[c] of the pending extend
	__device__ __forceinline__ void consume(uint64_t o_nb, uint64_t o_b, uint64_t o_size, const DevOpts &opt)
	{
		// forward extension works on x[1] (nb = 1), backward on x[0] (nb = 0)
		const uint64_t o0 = req_back ? o_nb : o_b, o1 = req_back ? o_b : o_nb;
		if (pc == PC_FWD) {
			if (o_size != ik2) {
				push_curr(ik0, ik1, ik2, ik_end);
				if (o_size < (uint64_t)min_intv) { after_forward(); return; }
			}
			ik0 = o0; ik1 = o1; ik2 = o_size; ik_end = (uint32_t)(i + 1);
			if (++i == len) { push_curr(ik0, ik1, ik2, ik_end); after_forward(); }
		} else if (pc == PC_BWD) {
			bwd_consume(o_size < (uint64_t)min_intv, o0, o1, o_size, opt);
		} else {   // PC_S3
			if (o_size < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
				if (o_size > 0) emit(o0, o1, o_size, x, i + 1);
				x = i + 1; pc = PC_P3_NEXT;
			} else {
				ik0 = o0; ik1 = o1; ik2 = o_size;
				if (++i == len) { x = len; pc = PC_FINISH; }
			}
		}
	}
};

}  // namespace

// reads: nt4 codes (0..3, 4 = N), read r at bases[off[r] .. off[r+1])
// intv : n_reads x EMA_INTV_CAP, n_intv / status : n_reads
// lists: (gridDim.x * blockDim.x / 8) x 2 x EMA_LIST_CAP scratch entries
__global__ void __launch_bounds__(256)
ema_k_seed(DevIndex ix, DevOpts opt, const uint8_t *__restrict__ bases, const uint32_t *__restrict__ off, int n_reads,
           Intv *__restrict__ intv, int *__restrict__ n_intv, int *__restrict__ status, Intv *__restrict__ lists)
{
	const int group = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
	const int n_groups = (int)((gridDim.x * blockDim.x) >> 3);
	SeedSM sm;
	sm.la = lists + (size_t)group * 2 * EMA_LIST_CAP;
	sm.lb = sm.la + EMA_LIST_CAP;
	sm.pc = PC_DONE; sm.has_req = false;
	sm.ik0 = sm.ik1 = sm.ik2 = 0; sm.p0 = sm.p1 = sm.p2 = 0; sm.req_c = 0; sm.req_back = 0;
	int read = group - n_groups;
	bool exhausted = false;
	for (;;) {
		// drive every machine to its next extend request; start the next read when one finishes
		while (!sm.has_req && !exhausted) {
			if (sm.pc == PC_DONE) {
				if (read >= 0) { n_intv[read] = sm.n_out; status[read] = sm.status; }
				read += n_groups;
				if (read >= n_reads) { exhausted = true; break; }
				sm.q = bases + off[read];
				sm.len = (int)(off[read + 1] - off[read]);
				sm.out = intv + (size_t)read * EMA_INTV_CAP;
				sm.status = 0; sm.n_out = 0; sm.pass = 1; sm.x = 0; sm.prev_is_a = 1; sm.n_curr = 0;
				if (sm.len < opt.min_seed_len) continue;      // mem_chain: no seeds for a read shorter than min_seed_len
				sm.pc = PC_P1_NEXT;
			}
			sm.advance(ix, opt);
		}
		if (!__any(sm.has_req)) break;
		uint64_t o_nb, o_b, o_size;
		const uint64_t x_nb = sm.pc == PC_BWD ? sm.p0 : sm.ik1;
		const uint64_t x_b = sm.pc == PC_BWD ? sm.p1 : sm.ik0;
		const uint64_t size = sm.pc == PC_BWD ? sm.p2 : sm.ik2;
		ema_group8_extend(ix, x_nb, x_b, size, sm.req_c, sm.has_req, o_nb, o_b, o_size);
		if (sm.has_req) { sm.has_req = false; sm.consume(o_nb, o_b, o_size, opt); }
	}
}

extern "C" void ema_launch_seed(const DevIndex *ix, const DevOpts *opt, const uint8_t *bases, const uint32_t *off,
                                int n_reads, Intv *intv, int *n_intv, int *status, Intv *lists, int n_blocks,
                                hipStream_t stream)
{
	hipLaunchKernelGGL(ema_k_seed, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, bases, off, n_reads, intv, n_intv,
	                   status, lists);
}
