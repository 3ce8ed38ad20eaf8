// ema_amd/csrc/k_seed.hip -- K1: SMEM / seed-interval collection on the FM-index (gfx950).
//
// Replaces, for a whole batch of reads, the seeding stage the reference reaches at
// src/bwabridge.c:236-237 (mem_align1_core -> mem_chain -> mem_collect_intv in the un-vendored
// bwa: bwt_smem1 x3 passes + bwt_seed_strategy1).  Output per read: the interval list
// (start, end, k, k', size) sorted by (start, end) -- exactly what mem_collect_intv leaves in
// aux->mem.
//
// Mapping to the hardware.  One bwt_extend is two independent 64-byte reads (the occ blocks of k-1 and l) followed
// by a few popcounts; successive extends of one search are strictly dependent, so a search is latency-bound and
// bandwidth only comes from many searches in flight.  Every LANE therefore carries its own read: a small state
// machine (forward / backward phase of an SMEM search, re-seeding pass, LAST-like pass) that yields each time it
// needs an extend; the 64 machines of a wave then issue their block loads together (16-byte loads, four per
// block: one 64-byte line per occ4 query, 128 lines in flight per wave) and resume.  Between two extends a
// machine only runs cheap bookkeeping, and lanes that are in the same state share its instruction stream, so the
// control program costs a few instructions per read and tick.  (Round 1 first mapped one read to a group of 8
// lanes that split each block load; the 8 lanes had to replay the whole control program and the kernel was
// issue-bound at ~10 M reads/s.)  The per-search working lists (bwa's prev/curr vectors) live in a per-lane slab
// of HBM scratch.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

namespace {

enum { PC_DONE = 0, PC_P1_NEXT, PC_P2_INIT, PC_P2_NEXT, PC_P3_NEXT, PC_FWD, PC_BWD, PC_S3, PC_FINISH };

struct SeedSM {
	// read: 2-bit codes + N mask, staged in LDS (word k of this lane at qw[k * 64], nm[k * 64])
	const uint32_t *qw, *nm;
	int len;
	__device__ __forceinline__ int q(int i) const
	{
		const int code = (qw[(i >> 4) << 6] >> ((i & 15) << 1)) & 3;
		return ((nm[(i >> 5) << 6] >> (i & 31)) & 1) ? 4 : code;
	}
	Intv *out;           // EMA_INTV_CAP entries
	Intv *la, *lb;       // working lists, EMA_LIST_CAP entries each, interleaved over the lanes of the wave: entry e at [e * 64]
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
		curr()[(n_curr++) << 6] = e;
	}
	__device__ __forceinline__ void emit(uint64_t a0, uint64_t a1, uint64_t a2, int start, int end)
	{
		if (n_out >= EMA_INTV_CAP) { status |= EMA_ST_INTV_OVERFLOW; return; }
		Intv e; e.x0 = a0; e.x1 = a1; e.x2 = a2; e.info = (uint64_t)(uint32_t)start << 32 | (uint32_t)end;
		out[n_out++] = e;
	}
	__device__ __forceinline__ void after_forward()
	{
		ret = (int)curr()[(n_curr - 1) << 6].info;      // longest match = last pushed
		prev_is_a ^= 1;                           // curr becomes prev, read in reverse order
		n_prev = n_curr; n_curr = 0; rev = 1;
		i = sm_x - 1; j = 0;
		pc = PC_BWD;
	}
	__device__ __forceinline__ void start_smem(const DevIndex &ix, int x_, int min_)
	{
		sm_x = x_; min_intv = min_;
		set_intv(ix, q(x_)); ik_end = (uint32_t)(x_ + 1);
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
				while (x < len && q(x) > 3) ++x;
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
				while (x < len && q(x) > 3) ++x;
				if (x >= len) { pc = PC_FINISH; break; }
				set_intv(ix, q(x));
				i = x + 1;
				if (i >= len) { x = len; pc = PC_FINISH; }
				else pc = PC_S3;
				break;
			case PC_FWD:
				if (q(i) < 4) { has_req = true; req_back = 0; req_c = 3 - q(i); return; }
				push_curr(ik0, ik1, ik2, ik_end);
				after_forward();
				break;
			case PC_BWD: {
				const Intv p = prev()[(rev ? n_prev - 1 - j : j) << 6];
				p0 = p.x0; p1 = p.x1; p2 = p.x2; p_end = (uint32_t)p.info;
				const int c = (i < 0 || q(i) > 3) ? -1 : q(i);
				if (c >= 0) { has_req = true; req_back = 1; req_c = c; return; }
				bwd_consume(true, 0, 0, 0, opt);
				break;
			}
			case PC_S3:
				if (q(i) < 4) { has_req = true; req_back = 0; req_c = 3 - q(i); return; }
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

// reads: qpack[r * 24 ..]: 16 words of 2-bit codes (base i at bits 2(i%16) of word i/16, N stored as 0) followed by
//        8 words of N flags (bit i%32 of word i/32); read lengths from off[]
// intv : n_reads x EMA_INTV_CAP, n_intv / status : n_reads
// lists: (gridDim.x * blockDim.x) x 2 x EMA_LIST_CAP scratch entries (one pair of working lists per lane, interleaved
//        over the 64 lanes of a wave so that lanes at the same list index touch one contiguous 2 KB run)
// counter: zero on entry; reads are handed out one by one
__global__ void __launch_bounds__(256)
ema_k_seed(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
           Intv *__restrict__ intv, int *__restrict__ n_intv, int *__restrict__ status, Intv *__restrict__ lists,
           int *__restrict__ counter)
{
	__shared__ uint32_t lds_q[4][16 * 64];      // 2-bit read codes, 16 words per lane, lane-interleaved
	__shared__ uint32_t lds_n[4][8 * 64];       // N mask, 8 words per lane
	const int lane = (int)(threadIdx.x & 63), wib = (int)(threadIdx.x >> 6);
	const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wib;
	uint32_t *qw = lds_q[wib] + lane, *nm = lds_n[wib] + lane;
	SeedSM sm;
	sm.la = lists + wave * (2 * EMA_LIST_CAP * 64) + lane;
	sm.lb = sm.la + EMA_LIST_CAP * 64;
	sm.qw = qw; sm.nm = nm;
	sm.pc = PC_DONE; sm.has_req = false;
	sm.ik0 = sm.ik1 = sm.ik2 = 0; sm.p0 = sm.p1 = sm.p2 = 0; sm.req_c = 0; sm.req_back = 0;
	sm.n_out = 0; sm.status = 0;
	int read = -1;
	bool exhausted = false;
	for (;;) {
		// drive every machine to its next extend request; take the next read from the queue when one finishes
		while (!sm.has_req && !exhausted) {
			if (sm.pc == PC_DONE) {
				if (read >= 0) { n_intv[read] = sm.n_out; status[read] = sm.status; }
				read = atomicAdd(counter, 1);
				if (read >= n_reads) { exhausted = true; break; }
				sm.len = (int)(off[read + 1] - off[read]);
				{      // the read, packed by the host: 16 code words (2 bit/base) + 8 mask words (N positions)
					const uint4 *pw = reinterpret_cast<const uint4 *>(qpack + (size_t)read * 24);
					const uint4 a = pw[0], b = pw[1], c = pw[2], d = pw[3], m0 = pw[4], m1 = pw[5];
					qw[0 << 6] = a.x; qw[1 << 6] = a.y; qw[2 << 6] = a.z; qw[3 << 6] = a.w;
					qw[4 << 6] = b.x; qw[5 << 6] = b.y; qw[6 << 6] = b.z; qw[7 << 6] = b.w;
					qw[8 << 6] = c.x; qw[9 << 6] = c.y; qw[10 << 6] = c.z; qw[11 << 6] = c.w;
					qw[12 << 6] = d.x; qw[13 << 6] = d.y; qw[14 << 6] = d.z; qw[15 << 6] = d.w;
					nm[0 << 6] = m0.x; nm[1 << 6] = m0.y; nm[2 << 6] = m0.z; nm[3 << 6] = m0.w;
					nm[4 << 6] = m1.x; nm[5 << 6] = m1.y; nm[6 << 6] = m1.z; nm[7 << 6] = m1.w;
				}
				sm.out = intv + (size_t)read * EMA_INTV_CAP;
				sm.status = 0; sm.n_out = 0; sm.pass = 1; sm.x = 0; sm.prev_is_a = 1; sm.n_curr = 0;
				if (sm.len < opt.min_seed_len) continue;      // mem_chain: no seeds for a read shorter than min_seed_len
				sm.pc = PC_P1_NEXT;
			}
			sm.advance(ix, opt);
		}
		if (!__any(sm.has_req)) break;
		uint64_t o_nb = 0, o_b = 0, o_size = 0;
		const uint64_t x_nb = sm.pc == PC_BWD ? sm.p0 : sm.ik1;
		const uint64_t x_b = sm.pc == PC_BWD ? sm.p1 : sm.ik0;
		const uint64_t size = sm.pc == PC_BWD ? sm.p2 : sm.ik2;
		if (sm.has_req) {
			ema_lane_extend(ix, x_nb, x_b, size, sm.req_c, o_nb, o_b, o_size);
			sm.has_req = false;
			sm.consume(o_nb, o_b, o_size, opt);
		}
	}
}

extern "C" void ema_launch_seed(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off,
                                int n_reads, Intv *intv, int *n_intv, int *status, Intv *lists, int *counter, int n_blocks,
                                hipStream_t stream)
{
	hipLaunchKernelGGL(ema_k_seed, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, qpack, off, n_reads, intv, n_intv,
	                   status, lists, counter);
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_seed_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_seed, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
