// ema_amd/csrc/k_kmer.hip -- builds the k-mer interval table of the index (dev_types.h) on the device when the engine opens:
// level L from level L-1, one lane per (L-1)-mer S: the two rank queries of a backward extension give the intervals of all
// four c.S at once (the same arithmetic bwt_extend performs step by step, ema_lane_occ4 on the same rank structure), so the
// table holds exactly what a chain of rank queries would have produced for every string of up to kmer_k bases.
// 4^(k-1) * 4/3 lanes in all: a few milliseconds for k = 14.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

__global__ void __launch_bounds__(256)
ema_k_kmer_level(DevIndex ix, int L, uint64_t *__restrict__ wide, uint64_t *__restrict__ narrow, int *__restrict__ overflow)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t n_prev = L == 1 ? 1 : (size_t)1 << (2 * (L - 1));
	if (t >= n_prev) return;
	uint64_t k[4], s[4];
	if (L == 1) {
		for (int c = 0; c < 4; ++c) { k[c] = ix.L2[c] + 1; s[c] = ix.L2[c + 1] - ix.L2[c]; }
	} else {
		uint64_t x0, x2, tk[4], tl[4];
		ema_kmer_lookup(ix, L - 1, (uint32_t)t, x0, x2);
		ema_lane_occ4(ix, x0 - 1, tk);
		ema_lane_occ4(ix, x0 - 1 + x2, tl);
		for (int c = 0; c < 4; ++c) { k[c] = ix.L2[c] + 1 + tk[c]; s[c] = tl[c] - tk[c]; }
	}
	for (int c = 0; c < 4; ++c) {
		const size_t code = ((size_t)c << (2 * (L - 1))) | t;
		if (L <= EMA_KMER_WIDE) {
			uint64_t *e = wide + 2 * ((((size_t)1 << (2 * L)) - 4) / 3 + code);
			e[0] = k[c]; e[1] = s[c];
		} else {
			if (s[c] >> 24 || k[c] >> 40) atomicOr(overflow, 1);      // does not fit the packed entry: the engine falls back to no table
			narrow[(((size_t)1 << (2 * L)) - ((size_t)1 << (2 * (EMA_KMER_WIDE + 1)))) / 3 + code] = (s[c] << 40) | (k[c] & 0xFFFFFFFFFFULL);
		}
	}
}

extern "C" void ema_launch_kmer_level(const DevIndex *ix, int L, uint64_t *wide, uint64_t *narrow, int *overflow, hipStream_t stream)
{
	const size_t n_prev = L == 1 ? 1 : (size_t)1 << (2 * (L - 1));
	hipLaunchKernelGGL(ema_k_kmer_level, dim3((unsigned)((n_prev + 255) / 256)), dim3(256), 0, stream, *ix, L, wide, narrow, overflow);
}

// DevIndex::text2: the indexed text (forward strand + reverse complement, 2 * l_pac bases) as 2-bit codes, 32 to a word, base j
// at bits 2(j%32) of word j/32; words beyond the text are zero.  One lane per word.
__global__ void __launch_bounds__(256)
ema_k_text2(const uint8_t *__restrict__ pac, int64_t l_pac, uint64_t *__restrict__ text2, size_t n_words)
{
	const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words) return;
	uint64_t v = 0;
	for (int t = 0; t < 32; ++t) {
		const int64_t j = (int64_t)(w << 5) + t;
		if (j >= 2 * l_pac) break;
		const int64_t f = j < l_pac ? j : 2 * l_pac - 1 - j;
		unsigned b = (pac[f >> 2] >> ((~f & 3) << 1)) & 3u;
		if (j >= l_pac) b = 3u - b;
		v |= (uint64_t)b << (t << 1);
	}
	text2[w] = v;
}

extern "C" size_t ema_text2_words(int64_t l_pac) { return (size_t)((2 * l_pac + 31) >> 5) + 8; }
extern "C" void ema_launch_text2(const uint8_t *pac, int64_t l_pac, uint64_t *text2, hipStream_t stream)
{
	const size_t n = ema_text2_words(l_pac);
	hipLaunchKernelGGL(ema_k_text2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pac, l_pac, text2, n);
}
