// ema_amd/csrc/k_seed_wave.hip -- K1w: SMEM / seed-interval collection, ONE WAVEFRONT PER READ, for long reads.
//
// Same stage and same output as k_seed.hip (mem_collect_intv of the un-vendored bwa, reached from reference
// src/bwabridge.c:236-237).  k_seed.hip gives every lane its own read, which is the right shape for the bulk: but a read
// inside a high-copy repeat needs ten thousand dependent extends there (one per entry of every backward row), a 30 ms
// latency chain that sets the duration of its launch.  The entries of one backward row of bwt_smem1 are independent
// extends by the same base, followed by a compaction ("keep an entry if it is still frequent enough and its size differs
// from the previous kept one; the first entry of the row, if it just died, is the SMEM to report"), so a wave can do a
// whole row of up to 64 entries per step: lane t extends entry t, ballots and prefix counts do the compaction.  The
// forward phase (one dependent extend per base) and the LAST-like pass stay sequential, computed redundantly by all
// lanes.  The full-capacity tier, which takes the reads that exceeded the lean tier's extend budget, uses this kernel.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

// entries of a backward row taken per step (a test build of the host interpreter lowers it to exercise rows that span
// several steps on ordinary reads)
#ifndef EMA_SEED_ROW_CHUNK
#define EMA_SEED_ROW_CHUNK EMA_WAVE
#endif

namespace {

struct WaveSeed {
	const DevIndex *ix;
	const DevOpts *opt;
	const uint32_t *qw, *nm;      // the read in LDS: 16 words of 2-bit codes, 8 words of N flags
	Intv *la, *lb;                // working lists in LDS, EMA_LIST_CAP entries each
	uint64_t *tmp;                // 64 words: sizes of the surviving entries of a row chunk, compacted
	Intv *out;                    // opt->intv_cap entries
	int len, n_out, status;

	__device__ __forceinline__ int q(int i) const
	{
		const int code = (qw[i >> 4] >> ((i & 15) << 1)) & 3;
		return ((nm[i >> 5] >> (i & 31)) & 1) ? 4 : code;
	}
	__device__ __forceinline__ void emit(uint64_t x0, uint64_t x1, uint64_t x2, int start, int end)
	{
		if (n_out >= opt->intv_cap) { status |= EMA_ST_INTV_OVERFLOW; return; }
		if (ema_lane() == 0) { Intv e; e.x0 = x0; e.x1 = x1; e.x2 = x2; e.info = (uint64_t)(uint32_t)start << 32 | (uint32_t)end; out[n_out] = e; }
		++n_out;
	}
	// bwt_smem1(x, min_intv): SMEMs covering position x are reported through found(); returns where the forward
	// extension stopped.  All arguments and the control flow are wave-uniform.
	__device__ inline int smem1(int x, int min_intv)
	{
		const int lane = (int)ema_lane();
		Intv *curr = la, *prev = lb;
		int n_curr = 0;
		uint64_t k0, k1, k2;
		{
			const int b = q(x);
			k0 = ix->L2[b] + 1; k2 = ix->L2[b + 1] - ix->L2[b]; k1 = ix->L2[3 - b] + 1;
		}
		int k_end = x + 1;
		auto push = [&](uint64_t a0, uint64_t a1, uint64_t a2, int end) {
			if (n_curr >= EMA_LIST_CAP) { status |= EMA_ST_LIST_OVERFLOW; return; }
			if (lane == 0) { Intv e; e.x0 = a0; e.x1 = a1; e.x2 = a2; e.info = (uint64_t)(uint32_t)end; curr[n_curr] = e; }
			++n_curr;
		};
		// ---- forward: one dependent extend per base
		int i;
		for (i = x + 1; i < len; ++i) {
			const int b = q(i);
			if (b > 3) { push(k0, k1, k2, k_end); break; }
			uint64_t o_nb, o_b, o_size;
			ema_lane_extend(*ix, k1, k0, k2, 3 - b, o_nb, o_b, o_size);      // forward extension works on x[1]
			o_nb = ema_uni(o_nb); o_b = ema_uni(o_b); o_size = ema_uni(o_size);
			if (o_size != k2) {
				push(k0, k1, k2, k_end);
				if (o_size < (uint64_t)min_intv) break;
			}
			k0 = o_b; k1 = o_nb; k2 = o_size; k_end = i + 1;
		}
		if (i == len) push(k0, k1, k2, k_end);
		const int ret = k_end;      // end of the longest match = the last interval pushed
		ema_wave_sync();
		// ---- backward: the list (longest match first, i.e. walked from its end) shrinks row by row
		{ Intv *t = curr; curr = prev; prev = t; }
		int n_prev = n_curr;
		bool rev = true;
		int n_mem_call = 0, last_start = 0;
		for (i = x - 1; i >= -1; --i) {
			const int c = i < 0 ? -1 : q(i) < 4 ? q(i) : -1;
			int n_new = 0;
			uint64_t carry = 0;      // size of the last surviving entry of the previous chunks
			for (int base = 0; base < n_prev; base += EMA_SEED_ROW_CHUNK) {
				const int t = base + lane;
				const bool valid = lane < EMA_SEED_ROW_CHUNK && t < n_prev;
				Intv p; p.x0 = p.x1 = p.x2 = p.info = 0;
				if (valid) p = prev[rev ? n_prev - 1 - t : t];
				uint64_t o_nb = 0, o_b = 0, o_size = 0;
				bool alive = false;
				if (valid && c >= 0) {
					ema_lane_extend(*ix, p.x0, p.x1, p.x2, c, o_nb, o_b, o_size);      // backward extension works on x[0]
					alive = o_size >= (uint64_t)min_intv;
				}
				const unsigned long long am = __ballot(alive);
				if (base == 0 && !(am & 1ull)) {
					// the first entry of the row cannot be extended: it is an SMEM unless a longer one reported earlier contains it
					if (n_mem_call == 0 || i + 1 < last_start) {
						++n_mem_call; last_start = i + 1;
						Intv p0;      // lane 0 holds the row's first entry
						p0.x0 = ema_uni((uint64_t)__shfl((long long)p.x0, 0)); p0.x1 = ema_uni((uint64_t)__shfl((long long)p.x1, 0));
						p0.x2 = ema_uni((uint64_t)__shfl((long long)p.x2, 0)); p0.info = ema_uni((uint64_t)__shfl((long long)p.info, 0));
						if ((int)(uint32_t)p0.info - (i + 1) >= opt->min_seed_len) emit(p0.x0, p0.x1, p0.x2, i + 1, (int)(uint32_t)p0.info);
					}
				}
				if (am) {
					// survivors in list order; one is kept unless its size equals that of the survivor before it
					const int idx = __popcll(am & ((1ull << lane) - 1));
					if (alive) tmp[idx] = o_size;
					ema_wave_sync();
					bool keep = false;
					if (alive) {
						if (idx == 0) keep = n_new == 0 || o_size != carry;
						else keep = o_size != tmp[idx - 1];
					}
					const unsigned long long km = __ballot(keep);
					const int n_alive = __popcll(am);
					carry = ema_uni(tmp[n_alive - 1]);
					if (keep) {
						const int at = n_new + __popcll(km & ((1ull << lane) - 1));
						if (at < EMA_LIST_CAP) { Intv e; e.x0 = o_nb; e.x1 = o_b; e.x2 = o_size; e.info = p.info; curr[at] = e; }
					}
					n_new += __popcll(km);
					if (n_new > EMA_LIST_CAP) { status |= EMA_ST_LIST_OVERFLOW; n_new = EMA_LIST_CAP; }
					ema_wave_sync();
				}
			}
			if (n_new == 0) break;
			{ Intv *t = curr; curr = prev; prev = t; }
			n_prev = n_new; rev = false;
			ema_wave_sync();
		}
		return ret;
	}
	// mem_collect_intv for the staged read: the three passes
	__device__ void collect()
	{
	if (len >= opt->min_seed_len) {      // mem_chain: no seeds for a read shorter than min_seed_len
		// pass 1: SMEMs from left to right
		int x = 0;
		while (x < len) {
			if (q(x) < 4) x = smem1(x, 1);
			else ++x;
		}
		// pass 2 (re-seeding): a pass-1 SMEM of length >= split_len with at most split_width occurrences
		const int old_n = n_out;
		for (int k = 0; k < old_n; ++k) {
			ema_wave_sync();
			const Intv p = ema_uni(out[k]);
			const int s = (int)(p.info >> 32), e = (int)(uint32_t)p.info;
			if (e - s < opt->split_len || p.x2 > (uint64_t)opt->split_width) continue;
			smem1((s + e) >> 1, (int)p.x2 + 1);
		}
		// pass 3: LAST-like seeds (bwt_seed_strategy1)
		if (opt->max_mem_intv > 0) {
			x = 0;
			while (x < len) {
				if (q(x) > 3) { ++x; continue; }
				const int b0 = q(x);
				uint64_t k0 = ix->L2[b0] + 1, k2 = ix->L2[b0 + 1] - ix->L2[b0], k1 = ix->L2[3 - b0] + 1;
				int i;
				bool found = false;
				for (i = x + 1; i < len; ++i) {
					const int b = q(i);
					if (b > 3) break;
					uint64_t o_nb, o_b, o_size;
					ema_lane_extend(*ix, k1, k0, k2, 3 - b, o_nb, o_b, o_size);
					o_nb = ema_uni(o_nb); o_b = ema_uni(o_b); o_size = ema_uni(o_size);
					if (o_size < (uint64_t)opt->max_mem_intv && i - x >= opt->min_seed_len) {
						if (o_size > 0) emit(o_b, o_nb, o_size, x, i + 1);
						found = true;
						break;
					}
					k0 = o_b; k1 = o_nb; k2 = o_size;
				}
				x = (found || i < len) ? i + 1 : len;
			}
		}
	}
	}
};

}  // namespace

// Same arguments as ema_k_seed where they apply; no scratch lists (they are in LDS).  The index and option records
// are read through pointers to device memory rather than passed by value: held in scalar registers next to this kernel's
// many wave-uniform variables they made the compiler spill scalar registers, and that build faulted on the GPU.
// (two waves per block: a block's LDS is limited to 64 KB and each wave keeps 16 KB of lists there)
__global__ void __launch_bounds__(128)
ema_k_seed_wave(const DevIndex *__restrict__ ixp, const DevOpts *__restrict__ optp, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
                const int *__restrict__ n_pairs_dev, const int *__restrict__ map, const int *__restrict__ read_list, Intv *__restrict__ intv,
                int *__restrict__ n_intv, int *__restrict__ status, int *__restrict__ counter)
{
	__shared__ uint32_t lds_q[2][24];
	__shared__ Intv lds_lists[2][2][EMA_LIST_CAP];
	__shared__ uint64_t lds_tmp[2][EMA_WAVE];
	const int lane = (int)ema_lane(), wib = ema_uni((int)(threadIdx.x >> 6));
	// read_list (a lean slice's long reads, listed by K1): item i is read read_list[i] of the slice -- input and results at that
	// number -- and *n_pairs_dev counts the reads K1 wanted to list, n_reads is the list's room.  Otherwise the full-capacity tier's
	// pairs (dev_common.hpp, ema_work_count / ema_in_read).
	const int n_total = read_list ? (*n_pairs_dev < n_reads ? *n_pairs_dev : n_reads) : ema_work_count(n_reads, n_pairs_dev, 2);
	const DevIndex &ix = *ixp;
	const DevOpts &opt = *optp;
	WaveSeed w;
	w.ix = ixp; w.opt = optp;
	w.qw = lds_q[wib]; w.nm = lds_q[wib] + 16;
	w.la = lds_lists[wib][0]; w.lb = lds_lists[wib][1];
	w.tmp = lds_tmp[wib];
	auto next_read = [&]() -> int {      // one read per wave from the shared counter
		int r = 0;
		if (lane == 0) r = atomicAdd(counter, 1);
		return __builtin_amdgcn_readlane(r, 0);
	};
	for (int item = next_read(); item < n_total; item = next_read()) {
		const int read = read_list ? ema_uni(read_list[item]) : item;
		const int in_read = read_list ? read : ema_uni(ema_in_read(map, read));
		w.len = ema_uni((int)(off[in_read + 1] - off[in_read]));
		ema_wave_sync();
		if (lane < 24) lds_q[wib][lane] = qpack[(size_t)in_read * 24 + lane];
		ema_wave_sync();
		w.out = intv + (size_t)read * opt.intv_cap;
		w.n_out = 0; w.status = 0;
		w.collect();
		ema_wave_sync();
		if (lane == 0) { n_intv[read] = w.n_out; status[read] = w.status; }
	}
}

extern "C" int ema_seed_wave_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_seed_wave, 128, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}

// ix / opt: DEVICE pointers
extern "C" void ema_launch_seed_wave(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off, int n_reads,
                                     const int *n_pairs_dev, const int *map, const int *read_list, Intv *intv, int *n_intv, int *status, int *counter,
                                     int n_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL(ema_k_seed_wave, dim3(n_blocks), dim3(128), 0, stream, ix, opt, qpack, off, n_reads, n_pairs_dev, map, read_list, intv, n_intv,
	                   status, counter);
}
