/*
 * oracle/introsort.h -- restatement of klib's ks_introsort / ks_combsort
 * (ksort.h, used by bwa for mem_intv, mem_flt, mem_ars2, mem_ars and 64-bit
 * keys).  TEST INFRASTRUCTURE (see oracle.h).  The sort is unstable, and the
 * order it leaves ties in feeds order-dependent logic (chain filter, region
 * dedup), so the exact sequence of comparisons and swaps is restated:
 *   - n == 2: one compare-and-swap;
 *   - depth budget 2*ceil(log2 n) (d starts at 2), explicit stack;
 *   - pivot = median of (first, middle+1, last), moved to the end;
 *   - Hoare scan `do ++i while a[i]<p; do --j while i<=j && p<a[j]`;
 *   - sub-ranges of <= 16 elements are left for one final insertion sort;
 *   - on depth exhaustion: comb sort (shrink 1.2473309501039787, gaps 9/10 -> 11).
 */
#ifndef ORC_INTROSORT_H
#define ORC_INTROSORT_H

#include <stdlib.h>
#include <stddef.h>

#define ORC_SORT_INIT(name, type_t, lt)                                             \
	static inline void orc_insertsort_##name(type_t *s, type_t *t)                  \
	{                                                                               \
		type_t *i, *j, tmp;                                                         \
		for (i = s + 1; i < t; ++i)                                                 \
			for (j = i; j > s && lt(*j, *(j - 1)); --j) {                           \
				tmp = *j; *j = *(j - 1); *(j - 1) = tmp;                            \
			}                                                                       \
	}                                                                               \
	static void orc_combsort_##name(size_t n, type_t a[])                           \
	{                                                                               \
		const double shrink = 1.2473309501039786540366528676643;                    \
		int swapped;                                                                \
		size_t gap = n;                                                             \
		type_t tmp, *i, *j;                                                         \
		do {                                                                        \
			if (gap > 2) {                                                          \
				gap = (size_t)(gap / shrink);                                       \
				if (gap == 9 || gap == 10) gap = 11;                                \
			}                                                                       \
			swapped = 0;                                                            \
			for (i = a; i < a + n - gap; ++i) {                                     \
				j = i + gap;                                                        \
				if (lt(*j, *i)) { tmp = *i; *i = *j; *j = tmp; swapped = 1; }       \
			}                                                                       \
		} while (swapped || gap > 2);                                               \
		if (gap != 1) orc_insertsort_##name(a, a + n);                              \
	}                                                                               \
	static void orc_introsort_##name(size_t n, type_t a[])                          \
	{                                                                               \
		struct frame { type_t *left, *right; int depth; } *stack, *top;             \
		int d;                                                                      \
		type_t piv, tmp, *s, *t, *i, *j, *k;                                        \
		if (n < 1) return;                                                          \
		if (n == 2) {                                                               \
			if (lt(a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; }            \
			return;                                                                 \
		}                                                                           \
		for (d = 2; (1ul << d) < n; ++d) {}                                         \
		stack = (struct frame *)malloc(sizeof(struct frame) * (sizeof(size_t) * d + 2)); \
		top = stack; s = a; t = a + (n - 1); d <<= 1;                               \
		for (;;) {                                                                  \
			if (s < t) {                                                            \
				if (--d == 0) {                                                     \
					orc_combsort_##name((size_t)(t - s) + 1, s);                    \
					t = s;                                                          \
					continue;                                                       \
				}                                                                   \
				i = s; j = t; k = i + ((j - i) >> 1) + 1;                           \
				if (lt(*k, *i)) {                                                   \
					if (lt(*k, *j)) k = j;                                          \
				} else k = lt(*j, *i) ? i : j;                                      \
				piv = *k;                                                           \
				if (k != t) { tmp = *k; *k = *t; *t = tmp; }                        \
				for (;;) {                                                          \
					do ++i; while (lt(*i, piv));                                    \
					do --j; while (i <= j && lt(piv, *j));                          \
					if (j <= i) break;                                              \
					tmp = *i; *i = *j; *j = tmp;                                    \
				}                                                                   \
				tmp = *i; *i = *t; *t = tmp;                                        \
				if (i - s > t - i) {                                                \
					if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; } \
					s = t - i > 16 ? i + 1 : t;                                     \
				} else {                                                            \
					if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; } \
					t = i - s > 16 ? i - 1 : s;                                     \
				}                                                                   \
			} else {                                                                \
				if (top == stack) {                                                 \
					free(stack);                                                    \
					orc_insertsort_##name(a, a + n);                                \
					return;                                                         \
				}                                                                   \
				--top; s = top->left; t = top->right; d = top->depth;               \
			}                                                                       \
		}                                                                           \
	}

#endif
