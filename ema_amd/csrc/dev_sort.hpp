// ema_amd/csrc/dev_sort.hpp -- klib's ks_introsort / ks_combsort, wave-uniform.
//
// bwa sorts chains by weight (mem_flt), regions by end (mem_ars2) and by (score, rb, qb) (mem_ars) with
// klib's unstable introsort, and the order it leaves ties in is visible downstream (chain filter,
// region dedup), so the engine runs the same sequence of comparisons and swaps: n == 2 special case,
// depth budget 2*ceil(log2 n), median-of-(first, middle+1, last) pivot moved to the end, Hoare scan,
// sub-ranges <= 16 left to one final insertion sort, comb sort on depth exhaustion.
// Executed redundantly by every lane of the wave (all lanes hold the same scalars and issue the same
// loads/stores), so it needs no cross-lane traffic.
#ifndef EMA_DEV_SORT_HPP
#define EMA_DEV_SORT_HPP

#include "dev_common.hpp"
#include <type_traits>
#include <utility>

// The sorts take any array-like `A` with a[i] and a + k: a plain pointer, or a lane-interleaved view (k_align_lane.hip).
template <typename A> __host__ __device__ A &ema_declref();      // declaration only: names an lvalue of type A inside decltype
#define EMA_ELEM(A) typename std::remove_cv<typename std::remove_reference<decltype(ema_declref<A>()[0])>::type>::type

template <typename A, typename LT>
__device__ inline void ema_insertsort(A a, int s, int t, LT lt)   // [s, t)
{
	using T = EMA_ELEM(A);
	for (int i = s + 1; i < t; ++i)
		for (int j = i; j > s; --j) {
			const T x = a[j], y = a[j - 1];
			if (!lt(x, y)) break;
			a[j] = y; a[j - 1] = x;
		}
}

template <typename A, typename LT>
__device__ inline void ema_combsort(A a, int n, LT lt)
{
	using T = EMA_ELEM(A);
	const double shrink = 1.2473309501039786540366528676643;
	int gap = n;
	bool swapped;
	do {
		if (gap > 2) {
			gap = (int)(gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = false;
		for (int i = 0; i < n - gap; ++i) {
			const int j = i + gap;
			const T x = a[j], y = a[i];
			if (lt(x, y)) { a[i] = x; a[j] = y; swapped = true; }
		}
	} while (swapped || gap > 2);
	if (gap != 1) ema_insertsort(a, 0, n, lt);
}

// stack: 3 ints per frame, at least 3 * (2*32 + 2) ints
template <typename A, typename LT>
__device__ inline void ema_introsort(A a, int n, LT lt, int *stack)
{
	using T = EMA_ELEM(A);
	if (n < 1) return;
	if (n == 2) {
		const T x = a[1], y = a[0];
		if (lt(x, y)) { a[0] = x; a[1] = y; }
		return;
	}
	int d;
	for (d = 2; (1u << d) < (unsigned)n; ++d) {}
	int top = 0, s = 0, t = n - 1;
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) {
				ema_combsort(a + s, t - s + 1, lt);
				t = s;
				continue;
			}
			int i = s, j = t, k = i + ((j - i) >> 1) + 1;
			{
				const T vk = a[k], vi = a[i], vj = a[j];
				if (lt(vk, vi)) { if (lt(vk, vj)) k = j; }
				else k = lt(vj, vi) ? i : j;
			}
			const T piv = a[k];
			if (k != t) { const T tmp = a[t]; a[k] = tmp; a[t] = piv; }
			for (;;) {
				do ++i; while (lt(a[i], piv));
				do --j; while (i <= j && lt(piv, a[j]));
				if (j <= i) break;
				const T x = a[i], y = a[j];
				a[i] = y; a[j] = x;
			}
			{ const T x = a[i], y = a[t]; a[i] = y; a[t] = x; }
			if (i - s > t - i) {
				if (i - s > 16) { stack[top * 3] = s; stack[top * 3 + 1] = i - 1; stack[top * 3 + 2] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack[top * 3] = i + 1; stack[top * 3 + 1] = t; stack[top * 3 + 2] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) {
				ema_insertsort(a, 0, n, lt);
				return;
			}
			--top; s = stack[top * 3]; t = stack[top * 3 + 1]; d = stack[top * 3 + 2];
		}
	}
}


// [r5] ks_introsort by the WHOLE wavefront, for n <= 256 keys in LDS (or any memory all lanes see): the same result as
// ema_introsort above, permutation for permutation -- ties included -- at a fraction of its latency.  The single-lane sort pays a
// ~100-clock LDS round trip per comparison, and the chain filter's sort of a repeat-rich read's 50-250 chain weights (ties are the
// rule there: weights are seed lengths) was a third of K2b mode 0's "chaining" (profiles/r05_k2_profile.txt: 34-37 Gclk per batch).
// What makes a parallel form exact:
//   * the control flow (ranges, depth budget, explicit stack, median of three) is scalar and is kept as it is;
//   * one Hoare scan is a pairing: i stops at the r-th element of (s, t] that is not below the pivot, j at the r-th element of
//     [s, t) from the right that is not above it (or at i - 1 when there is none left above i), and the pair is swapped while
//     j > i.  Elements the scan has passed are never looked at again, so both sequences can be read off the array as it stands:
//     two ballots per 64 elements, ranks by population counts, S = the number of pairs with y_r > x_r, S swaps at once, and i
//     ends on the (S + 1)-th stopper from the left or on y_S, whichever comes first (y_S holds a swapped-in stopper by then);
//   * the closing insertion sort over the whole array moves an element left only past strictly greater ones: it is THE stable
//     sort of the arrangement the partitions left, i.e. every element's rank is (elements below it) + (equal ones before it).
// The depth-exhaustion fallback (comb sort: never seen on chain weights) runs on lane 0 as before.
// scratch: 2 x 256 uint16_t all lanes see (positions of the stoppers by rank).  stack: as for ema_introsort.
template <typename LT>
__device__ inline void ema_introsort_wave(uint64_t *a, int n, LT lt, int *stack, uint16_t *scratch)
{
	const int lane = (int)ema_lane();
	if (n < 1) return;
	if (n == 2) {
		ema_wave_sync();
		if (lane == 0) { const uint64_t x = a[1], y = a[0]; if (lt(x, y)) { a[0] = x; a[1] = y; } }
		ema_wave_sync();
		return;
	}
	uint16_t *const sl_ = scratch, *const sr_ = scratch + 256;
	int d;
	for (d = 2; (1u << d) < (unsigned)n; ++d) {}
	int top = 0, s = 0, t = n - 1;
	d <<= 1;
	ema_wave_sync();
	for (;;) {
		if (s < t) {
			if (--d == 0) {
				if (lane == 0) ema_combsort(a + s, t - s + 1, lt);
				ema_wave_sync();
				t = s;
				continue;
			}
			int k = s + ((t - s) >> 1) + 1;
			const uint64_t vk = ema_uni(a[k]), vi = ema_uni(a[s]), vj = ema_uni(a[t]);
			if (lt(vk, vi)) { if (lt(vk, vj)) k = t; }
			else k = lt(vj, vi) ? s : t;
			const uint64_t piv = k == s ? vi : k == t ? vj : vk;
			ema_wave_sync();
			if (k != t && lane == 0) { a[k] = vj; a[t] = piv; }
			ema_wave_sync();
			// the two stopper sequences of the scan over [s, t], 64 elements per round of ballots
			const int m = t - s + 1, n_ch = (m + 63) >> 6;
			unsigned long long lm[4] = {0, 0, 0, 0}, rm[4] = {0, 0, 0, 0};
#pragma unroll
			for (int c = 0; c < 4; ++c) if (c < n_ch) {
				const int x = s + c * 64 + lane;
				const bool in = x <= t;
				const uint64_t v = in ? a[x] : 0;
				lm[c] = __ballot(in && x > s && !lt(v, piv));
				rm[c] = __ballot(in && x < t && !lt(piv, v));
			}
			int n_l = 0, n_r = 0;
#pragma unroll
			for (int c = 0; c < 4; ++c) { n_l += __popcll(lm[c]); n_r += __popcll(rm[c]); }
			{
				int base_l = 0, above_r = n_r;
#pragma unroll
				for (int c = 0; c < 4; ++c) if (c < n_ch) {
					const int x = s + c * 64 + lane;
					above_r -= __popcll(rm[c]);      // stoppers in the chunks above this one
					if ((lm[c] >> lane) & 1) sl_[base_l + __popcll(lm[c] & ((2ULL << lane) - 1)) - 1] = (uint16_t)x;      // rank from the left, 0-based
					if ((rm[c] >> lane) & 1) sr_[above_r + __popcll(rm[c] >> lane) - 1] = (uint16_t)x;                    // rank from the right, 0-based
					base_l += __popcll(lm[c]);
				}
			}
			ema_wave_sync();
			const int n_p = n_l < n_r ? n_l : n_r;
			int n_swap = 0;      // pairs with y_r > x_r: a prefix of the pairs (x grows, y falls)
			for (int r0 = 0; r0 < n_p; r0 += EMA_WAVE) {
				const int r = r0 + lane;
				n_swap += __popcll(__ballot(r < n_p && sr_[r < n_p ? r : 0] > sl_[r < n_p ? r : 0]));
			}
			for (int r0 = 0; r0 < n_swap; r0 += EMA_WAVE) {      // (the positions of different pairs are distinct: every lane reads, then every lane writes)
				const int r = r0 + lane;
				const bool on = r < n_swap;
				const int x = on ? (int)sl_[r] : 0, y = on ? (int)sr_[r] : 0;
				const uint64_t vx = on ? a[x] : 0, vy = on ? a[y] : 0;
				ema_wave_sync();
				if (on) { a[x] = vy; a[y] = vx; }
			}
			ema_wave_sync();
			// where i stops last: at the next stopper from the left -- or, sooner, at the place of the last swap's right element, which
			// now holds a value that is not below the pivot (n_swap < n_l: position t is the last stopper and pairs with nothing)
			int i = ema_uni((int)sl_[n_swap]);
			if (n_swap > 0) { const int y_last = ema_uni((int)sr_[n_swap - 1]); i = i < y_last ? i : y_last; }
			if (lane == 0) { const uint64_t x = a[i], y = a[t]; a[i] = y; a[t] = x; }
			ema_wave_sync();
			if (i - s > t - i) {
				if (i - s > 16) { stack[top * 3] = s; stack[top * 3 + 1] = i - 1; stack[top * 3 + 2] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack[top * 3] = i + 1; stack[top * 3 + 1] = t; stack[top * 3 + 2] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) break;
			--top; s = stack[top * 3]; t = stack[top * 3 + 1]; d = stack[top * 3 + 2];
		}
	}
	// the insertion sort over the whole array = the stable sort of what the partitions left
	uint64_t mine[4] = {0, 0, 0, 0};
	int rank[4] = {0, 0, 0, 0};
#pragma unroll
	for (int c = 0; c < 4; ++c) { const int p = c * 64 + lane; if (p < n) mine[c] = a[p]; }
	for (int q = 0; q < n; ++q) {
		const uint64_t v = a[q];      // (one address for the whole wavefront: a broadcast read)
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const int p = c * 64 + lane;
			rank[c] += (lt(v, mine[c]) || (q < p && !lt(mine[c], v))) ? 1 : 0;
		}
	}
	ema_wave_sync();
#pragma unroll
	for (int c = 0; c < 4; ++c) { const int p = c * 64 + lane; if (p < n) a[rank[c]] = mine[c]; }
	ema_wave_sync();
}

// Ascending sort of n <= 64 DISTINCT 64-bit keys by the whole wavefront: lane i ranks key i against the others (values read
// lane to lane, no memory) and stores it at its rank.  With distinct keys every correct sort gives the same array, so this stands
// in for ks_introsort where bwa's keys cannot tie (mem_chain2aln's seed order: score << 32 | index) -- the single-lane sort
// on LDS costs a hundred-clock round trip per comparison.  The caller synchronises before (keys written) and after.
__device__ __forceinline__ void ema_rank_sort_distinct(uint64_t *keys, int n)
{
	const int lane = (int)ema_lane();
	const uint64_t mine = lane < n ? keys[lane] : ~0ULL;
	int rank = 0;
	for (int j = 0; j < n; ++j) rank += ema_lane_val(mine, j) < mine ? 1 : 0;
	ema_wave_sync();
	if (lane < n) keys[rank] = mine;
}

#endif
