// ema_amd/csrc/k_seed.hip -- K1: SMEM / seed-interval collection on the FM-index (gfx950).
//
// Replaces, for a whole batch of reads, the seeding stage the reference reaches at
// src/bwabridge.c:236-237 (mem_align1_core -> mem_chain -> mem_collect_intv in the un-vendored
// bwa: bwt_smem1 x3 passes + bwt_seed_strategy1).  Output per read: the seed intervals
// (start, end, k, k', size) that mem_collect_intv leaves in aux->mem, in discovery order; the
// consumer (K2, or the host for the debug entry point) orders them by (start, end) -- entries with
// equal (start, end) describe the same substring and are identical, so that order is unique.
//
// Mapping to the hardware.  One bwt_extend is two independent 64-byte reads (the occ blocks of k-1 and l) followed
// by a few popcounts; successive extends of one search are strictly dependent, so a search is latency-bound and
// bandwidth only comes from many searches in flight.  Every LANE therefore carries its own read: a small state
// machine (forward / backward phase of an SMEM search, re-seeding pass, LAST-like pass).  A tick of the wave is
//   A  every lane runs its control program on registers and LDS only, until it needs an extend or a list entry;
//   B  all lanes issue their global loads TOGETHER at one place in the code -- the two occ blocks (four 16-byte
//      loads each: one 64-byte line per occ4 query) and the prefetch of the working-list entry the lane will need
//      next -- so the wave waits for memory once per tick, with up to 128 + 64 lines in flight;
//   C  every lane applies its result.
// Keeping loads out of the divergent control program matters: lanes in different states execute their code one
// state after the other, and a load inside it costs a memory round trip per state instead of one per tick.
// The read sits in LDS as 2-bit codes + N mask (packed by the host); the working lists (bwa's prev/curr vectors)
// live in a lane-interleaved slab of HBM scratch, written fire-and-forget and read back one tick ahead of use.
// (Round 1 first mapped one read to a group of 8 lanes sharing each block load; the 8 lanes replayed the whole
// control program and the kernel was issue-bound at ~10 M reads/s.)
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

namespace {

enum { PC_DONE = 0, PC_P1_NEXT, PC_P2_INIT, PC_P2_NEXT, PC_P3_NEXT, PC_FWD, PC_BWD, PC_S3 };

struct SeedSM {
	// read: 2-bit codes + N mask, staged in LDS (word k of this lane at qw[k * 64], nm[k * 64])
	const uint32_t *qw, *nm;
	int len;
	Intv *out;           // out_cap entries
	int out_cap;
	Intv *la, *lb;       // working lists, EMA_LIST_CAP entries each, interleaved over the lanes of the wave: entry e at [e * 64]
	int status;
	// control
	int pc, pass, x, sm_x, min_intv, i, j;
	int n_prev, n_curr, rev, prev_is_a;
	int n_mem_call, last_mem_start, n_out, old_n, k2;
	uint64_t last_curr_size;
	uint64_t ik0, ik1, ik2; uint32_t ik_end;      // current interval of a forward phase
	uint64_t p0, p1, p2; uint32_t p_end;          // list entry being extended in the backward phase
	uint64_t f0, f1, f2; uint32_t f_end;          // first entry pushed into curr in this row (the next row starts with it)
	uint64_t l0, l1, l2; uint32_t l_end;          // last entry pushed (the backward phase starts with it)
	// requests handed to the uniform part of the tick
	bool has_req; int req_c, req_back;            // one bwt_extend
	const Intv *ld_ptr; int ld_kind;              // one entry load: 1 = next list entry (lands in p), 2 = out[k2] for pass 2

	__device__ __forceinline__ int q(int i_) const
	{
		const int code = (qw[(i_ >> 4) << 6] >> ((i_ & 15) << 1)) & 3;
		return ((nm[(i_ >> 5) << 6] >> (i_ & 31)) & 1) ? 4 : code;
	}
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
		curr()[(size_t)n_curr << 6] = e;
		if (n_curr == 0) { f0 = a0; f1 = a1; f2 = a2; f_end = end; }
		l0 = a0; l1 = a1; l2 = a2; l_end = end;
		++n_curr;
	}
	__device__ __forceinline__ void emit(uint64_t a0, uint64_t a1, uint64_t a2, int start, int end)
	{
		if (n_out >= out_cap) { status |= EMA_ST_INTV_OVERFLOW; return; }
		Intv e; e.x0 = a0; e.x1 = a1; e.x2 = a2; e.info = (uint64_t)(uint32_t)start << 32 | (uint32_t)end;
		out[n_out++] = e;
	}
	// forward phase over: its list (longest match = last pushed) becomes prev and is walked in reverse order
	__device__ __forceinline__ void after_forward()
	{
		if (pass == 1) x = (int)l_end;      // bwt_smem1's return value: where the forward extension stopped
		prev_is_a ^= 1;
		n_prev = n_curr; n_curr = 0; rev = 1;
		i = sm_x - 1; j = 0;
		p0 = l0; p1 = l1; p2 = l2; p_end = l_end;
		pc = PC_BWD;
	}
	__device__ __forceinline__ void start_smem(const DevIndex &ix, int x_, int min_)
	{
		sm_x = x_; min_intv = min_;
		set_intv(ix, q(x_)); ik_end = (uint32_t)(x_ + 1);
		n_curr = 0; n_mem_call = 0;
		i = x_ + 1;
		if (i >= len) { push_curr(ik0, ik1, ik2, ik_end); after_forward(); }
		else pc = PC_FWD;
	}
	__device__ __forceinline__ void end_smem() { pc = pass == 1 ? PC_P1_NEXT : PC_P2_NEXT; }
	// entry p (row position j) is done: next entry of the row (already prefetched into p), next row, or end of search
	__device__ __forceinline__ void bwd_next()
	{
		if (++j == n_prev) {
			if (n_curr == 0) { end_smem(); return; }
			prev_is_a ^= 1; n_prev = n_curr; n_curr = 0; rev = 0; j = 0; --i;
			p0 = f0; p1 = f1; p2 = f2; p_end = f_end;
		}
	}
	// backward bookkeeping for entry p at query position i; `dead` = it cannot be extended by q[i]
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
	}
	// Phase A: runs on registers/LDS until the lane needs an extend (has_req), an entry load (ld_kind) or is done
	__device__ void advance(const DevIndex &ix, const DevOpts &opt)
	{
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
			case PC_P2_NEXT:      // fetch the next pass-1 SMEM; the tick examines it once loaded
				if (k2 >= old_n) { pass = 3; x = 0; pc = opt.max_mem_intv > 0 ? PC_P3_NEXT : PC_DONE; break; }
				ld_ptr = out + k2; ld_kind = 2; ++k2;
				return;
			case PC_P3_NEXT:
				while (x < len && q(x) > 3) ++x;
				if (x >= len) { pc = PC_DONE; break; }
				set_intv(ix, q(x));
				i = x + 1;
				if (i >= len) { x = len; pc = PC_DONE; }
				else pc = PC_S3;
				break;
			case PC_FWD: {
				const int b = q(i);
				if (b < 4) { has_req = true; req_back = 0; req_c = 3 - b; return; }
				push_curr(ik0, ik1, ik2, ik_end);
				after_forward();
				break;
			}
			case PC_BWD: {
				const int c = i < 0 ? -1 : q(i);
				if (c >= 0 && c < 4) {
					has_req = true; req_back = 1; req_c = c;
					if (j + 1 < n_prev) { ld_ptr = prev() + ((size_t)(rev ? n_prev - 2 - j : j + 1) << 6); ld_kind = 1; }
					return;
				}
				// start of the read or an ambiguous base: every entry of the row dies here and only the first can be
				// emitted (the others fail the `start < last emitted start` test), then the search is over
				bwd_consume(true, 0, 0, 0, opt);
				end_smem();
				break;
			}
			case PC_S3: {
				const int b = q(i);
				if (b < 4) { has_req = true; req_back = 0; req_c = 3 - b; return; }
				x = i + 1; pc = PC_P3_NEXT;
				break;
			}
			default:
				return;
			}
		}
	}
	// Phase C for the forward phases: the result ok[c] of the extend requested in phase A (o0 = x[0], o1 = x[1])
	__device__ __forceinline__ void consume_fwd(uint64_t o0, uint64_t o1, uint64_t o_size, const DevOpts &opt)
	{
		if (pc == PC_FWD) {
			if (o_size != ik2) {
				push_curr(ik0, ik1, ik2, ik_end);
				if (o_size < (uint64_t)min_intv) { after_forward(); return; }
			}
			ik0 = o0; ik1 = o1; ik2 = o_size; ik_end = (uint32_t)(i + 1);
			if (++i == len) { push_curr(ik0, ik1, ik2, ik_end); after_forward(); }
		} else {      // PC_S3
			if (o_size < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
				if (o_size > 0) emit(o0, o1, o_size, x, i + 1);
				x = i + 1; pc = PC_P3_NEXT;
			} else {
				ik0 = o0; ik1 = o1; ik2 = o_size;
				if (++i == len) { x = len; pc = PC_DONE; }
			}
		}
	}
};

}  // namespace

// reads: qpack[r * 24 ..]: 16 words of 2-bit codes (base i at bits 2(i%16) of word i/16, N stored as 0) followed by
//        8 words of N flags (bit i%32 of word i/32); read lengths from off[]
// intv : n_reads x opt.intv_cap (in discovery order, see the header), n_intv / status : n_reads
// lists: (gridDim.x * blockDim.x) x 2 x EMA_LIST_CAP scratch entries (one pair of working lists per lane, interleaved
//        over the 64 lanes of a wave so that lanes at the same list index touch one contiguous 2 KB run)
// counter: zero on entry; reads are handed out one by one
__global__ void __launch_bounds__(256)
ema_k_seed(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
           const int *__restrict__ n_pairs_dev, const int *__restrict__ map, Intv *__restrict__ intv, int *__restrict__ n_intv, int *__restrict__ status, Intv *__restrict__ lists,
           int *__restrict__ counter, unsigned long long *prof)
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
	sm.pc = PC_DONE; sm.has_req = false; sm.ld_kind = 0; sm.ld_ptr = sm.la;
	sm.ik0 = sm.ik1 = sm.ik2 = 0; sm.p0 = sm.p1 = sm.p2 = 0; sm.req_c = 0; sm.req_back = 0;
	sm.f0 = sm.f1 = sm.f2 = 0; sm.l0 = sm.l1 = sm.l2 = 0; sm.f_end = sm.l_end = sm.p_end = sm.ik_end = 0;
	sm.n_out = 0; sm.status = 0; sm.len = 0;
	int read = -1;
	bool exhausted = false;
	n_reads = ema_work_count(n_reads, n_pairs_dev, 2);
	// diagnostic (prof != null): ticks, active lane-ticks and shader clocks of this wave
	unsigned long long n_tick = 0, n_active = 0, t_start = prof ? __builtin_amdgcn_s_memtime() : 0;
	for (;;) {
		// ---- phase A: control programs, registers and LDS only
		while (!sm.has_req && !sm.ld_kind && !exhausted) {
			if (sm.pc == PC_DONE) {
				if (read >= 0) { n_intv[read] = sm.n_out; status[read] = sm.status; }
				read = atomicAdd(counter, 1);
				if (read >= n_reads) { exhausted = true; break; }
				const int in_read = ema_in_read(map, read);
				sm.len = (int)(off[in_read + 1] - off[in_read]);
				{      // the read, packed by the host: 16 code words (2 bit/base) + 8 mask words (N positions)
					const uint4 *pw = reinterpret_cast<const uint4 *>(qpack + (size_t)in_read * 24);
					const uint4 a = pw[0], b = pw[1], c = pw[2], d = pw[3], m0 = pw[4], m1 = pw[5];
					qw[0 << 6] = a.x; qw[1 << 6] = a.y; qw[2 << 6] = a.z; qw[3 << 6] = a.w;
					qw[4 << 6] = b.x; qw[5 << 6] = b.y; qw[6 << 6] = b.z; qw[7 << 6] = b.w;
					qw[8 << 6] = c.x; qw[9 << 6] = c.y; qw[10 << 6] = c.z; qw[11 << 6] = c.w;
					qw[12 << 6] = d.x; qw[13 << 6] = d.y; qw[14 << 6] = d.z; qw[15 << 6] = d.w;
					nm[0 << 6] = m0.x; nm[1 << 6] = m0.y; nm[2 << 6] = m0.z; nm[3 << 6] = m0.w;
					nm[4 << 6] = m1.x; nm[5 << 6] = m1.y; nm[6 << 6] = m1.z; nm[7 << 6] = m1.w;
				}
				sm.out = intv + (size_t)read * opt.intv_cap; sm.out_cap = opt.intv_cap;
				sm.status = 0; sm.n_out = 0; sm.pass = 1; sm.x = 0; sm.prev_is_a = 1; sm.n_curr = 0;
				if (sm.len < opt.min_seed_len) continue;      // mem_chain: no seeds for a read shorter than min_seed_len
				sm.pc = PC_P1_NEXT;
			}
			sm.advance(ix, opt);
		}
		if (!__any(sm.has_req || sm.ld_kind)) break;
		if (prof) { ++n_tick; n_active += __popcll(__ballot(sm.has_req)); }
		// ---- phase B: every global load of the tick, issued together
		Intv ent; ent.x0 = ent.x1 = ent.x2 = ent.info = 0;
		if (sm.ld_kind) {
			const ulong2 *src = reinterpret_cast<const ulong2 *>(sm.ld_ptr);
			const ulong2 lo = src[0], hi = src[1];
			ent.x0 = lo.x; ent.x1 = lo.y; ent.x2 = hi.x; ent.info = hi.y;
		}
		uint64_t o_nb = 0, o_b = 0, o_size = 0;
		const bool back = sm.pc == PC_BWD;
		if (sm.has_req)
			ema_lane_extend(ix, back ? sm.p0 : sm.ik1, back ? sm.p1 : sm.ik0, back ? sm.p2 : sm.ik2, sm.req_c, o_nb, o_b, o_size);
		// ---- phase C: apply
		if (sm.has_req) {
			sm.has_req = false;
			if (back) {      // backward extension works on x[0]: o_nb = x[0], o_b = x[1]
				sm.bwd_consume(o_size < (uint64_t)sm.min_intv, o_nb, o_b, o_size, opt);
				if (sm.ld_kind == 1) { sm.p0 = ent.x0; sm.p1 = ent.x1; sm.p2 = ent.x2; sm.p_end = (uint32_t)ent.info; }
				sm.bwd_next();
			} else sm.consume_fwd(o_b, o_nb, o_size, opt);      // forward extension works on x[1]
		} else if (sm.ld_kind == 2) {
			// pass 2 (re-seeding): a pass-1 SMEM of length >= split_len with at most split_width occurrences
			const int s = (int)(ent.info >> 32), e = (int)(uint32_t)ent.info;
			if (!(e - s < opt.split_len || ent.x2 > (uint64_t)opt.split_width)) sm.start_smem(ix, (s + e) >> 1, (int)ent.x2 + 1);
		}
		sm.ld_kind = 0;
	}
	if (prof && lane == 0) {
		atomicAdd(prof + 8, n_tick); atomicAdd(prof + 9, n_active);
		atomicAdd(prof + 10, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_start)); atomicMax(prof + 11, n_tick);
	}
}

extern "C" void ema_launch_seed(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off,
                                int n_reads, const int *n_pairs_dev, const int *map, Intv *intv, int *n_intv, int *status, Intv *lists, int *counter, int n_blocks,
                                hipStream_t stream, unsigned long long *prof)
{
	hipLaunchKernelGGL(ema_k_seed, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, qpack, off, n_reads, n_pairs_dev, map, intv, n_intv,
	                   status, lists, counter, prof);
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_seed_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_seed, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
