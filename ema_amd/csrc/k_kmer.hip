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

// The flat suffix array of a stock bwa index (no <prefix>.fsa): one lane per row walks bwa's bwt_sa() -- LF-mapping (bwt_invPsi: the
// row's BWT symbol c and its rank, L2[c] + occ(c, row)) until a row of bwa's sample, counting the steps -- on the rank structure
// already in HBM: at most sa_intv - 1 dependent 32-byte gathers per row, 6.2 G rows x 15.5 on average at the scale of a human genome.
// sampled[j] = SA[j << shift], sampled[0] = -1 (bwa's bwt_restore_sa); row 0 (the sentinel's suffix) gets seq_len as ema_index_build
// writes it.  Reference path: src/bwabridge.c:79 (bwa_idx_load) and every bwt_sa() behind mem_chain.
__global__ void __launch_bounds__(256)
ema_k_sa_expand(DevIndex ix, const uint64_t *__restrict__ sampled, int shift, void *__restrict__ sa_out, int width, uint64_t row0, uint64_t n_rows)
{
	const uint64_t r = row0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rows) return;
	const uint64_t mask = ((uint64_t)1 << shift) - 1;
	uint64_t k = r, steps = 0;
	while (k & mask) {
		if (k == ix.primary) k = 0;
		else {
			const uint64_t p = k - (k > ix.primary ? 1 : 0);
			const OccBlock *blk = ix.occ + (p >> 6);
			const uint4 head = *reinterpret_cast<const uint4 *>(blk);
			const ulong2 sym = *(reinterpret_cast<const ulong2 *>(blk) + 1);
			const int q = (int)(p & 63);
			// the symbol at position q and how many of positions 0..q hold it: the two bit planes matched against it
			const unsigned c = (unsigned)((sym.x >> q) & 1u) | (unsigned)((sym.y >> q) & 1u) << 1;
			const uint64_t m = (2ULL << q) - 1;
			const unsigned cnt = (unsigned)__popcll((c & 1 ? sym.x : ~sym.x) & (c & 2 ? sym.y : ~sym.y) & m);
			uint64_t total = (uint64_t)(c == 0 ? head.x : c == 1 ? head.y : c == 2 ? head.z : head.w) + cnt;
			if (ix.n_super > 1) {
				const int sb = (int)(p >> EMA_OCC_SUPER_SHIFT);
				total += sb == 0 ? 0 : sb == 1 ? ix.occ_super[0][c] : sb == 2 ? ix.occ_super[1][c] : ix.occ_super[2][c];
			}
			k = ix.L2[c] + total;
		}
		++steps;
	}
	uint64_t v = steps + sampled[k >> shift];
	if (r == 0) v = ix.seq_len;
	if (width == 4) reinterpret_cast<uint32_t *>(sa_out)[r] = (uint32_t)v;
	else reinterpret_cast<uint64_t *>(sa_out)[r] = v;
}

// rows [row0, row0 + n) of the flat suffix array (the engine launches it in pieces of at most 2^31 rows)
extern "C" void ema_launch_sa_expand(const DevIndex *ix, const uint64_t *sampled, int shift, void *sa_out, int width, uint64_t row0, uint64_t n,
                                     hipStream_t stream)
{
	hipLaunchKernelGGL(ema_k_sa_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *ix, sampled, shift, sa_out, width, row0, row0 + n);
}
