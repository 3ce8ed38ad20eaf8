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
