// ema_amd/csrc/k_pack.hip -- result layout on the device: where every read's candidates and CIGAR operations go in the packed
// arrays handed to the host (exclusive prefix sums over the per-read counts of one slice), computed in the slice's own stream
// right behind K4 so that the packing kernel (ema_k_pack, k_final.hip) can follow at once and the NEXT pass over the slice's
// result slots may be queued without the host in between (engine.hip, ema_engine_run_async).  Reads with a capacity flag
// count as empty: their pair is redone by the full-capacity tier.
// Three small launches: per-block totals (1024 reads per block), one block turning them into block offsets, per-read offsets.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "dev_merge.h"

namespace {

__device__ __forceinline__ void counts_of(int r, int n, const int *status, const int *n_regs, const int *cig_n, unsigned &c, unsigned &g)
{
	c = g = 0;
	if (r < n && (!status || status[r] == 0)) { c = (unsigned)n_regs[r]; g = (unsigned)cig_n[r]; }      // (status null: every read counts)
}

// exclusive prefix (within the block) of this thread's 4-read run; returns the block totals through LDS
__device__ __forceinline__ void block_scan(unsigned c4, unsigned g4, unsigned &c_excl, unsigned &g_excl, unsigned &c_tot, unsigned &g_tot)
{
	__shared__ unsigned wsum[2][4];
	const int lane = (int)ema_lane(), w = (int)(threadIdx.x >> 6);
	const unsigned ci = (unsigned)ema_wave_incl_scan_add((int)c4), gi = (unsigned)ema_wave_incl_scan_add((int)g4);
	if (lane == 63) { wsum[0][w] = ci; wsum[1][w] = gi; }
	__syncthreads();
	unsigned cb = 0, gb = 0, ct = 0, gt = 0;
	for (int k = 0; k < 4; ++k) { if (k < w) { cb += wsum[0][k]; gb += wsum[1][k]; } ct += wsum[0][k]; gt += wsum[1][k]; }
	c_excl = cb + ci - c4; g_excl = gb + gi - g4; c_tot = ct; g_tot = gt;
	__syncthreads();
}

}  // namespace

__global__ void __launch_bounds__(256)
ema_k_scan_blocks(int n_reads, const int *__restrict__ n_pairs_dev, const int *__restrict__ status, const int *__restrict__ n_regs,
                  const int *__restrict__ cig_n, uint2 *__restrict__ block_tot)
{
	const int n = ema_work_count(n_reads, n_pairs_dev, 2);
	const int r0 = (int)(blockIdx.x * 1024 + threadIdx.x * 4);
	unsigned c4 = 0, g4 = 0;
	for (int k = 0; k < 4; ++k) { unsigned c, g; counts_of(r0 + k, n, status, n_regs, cig_n, c, g); c4 += c; g4 += g; }
	unsigned ce, ge, ct, gt;
	block_scan(c4, g4, ce, ge, ct, gt);
	if (threadIdx.x == 0) block_tot[blockIdx.x] = make_uint2(ct, gt);
}

// one block: block totals -> exclusive block offsets (in place); grand totals to tot[0] (candidates), tot[1] (CIGAR operations)
__global__ void __launch_bounds__(256)
ema_k_scan_tops(int n_blocks, uint2 *__restrict__ block_tot, uint64_t *__restrict__ tot)
{
	unsigned long long c_run = 0, g_run = 0;
	for (int base = 0; base < n_blocks; base += 1024) {
		const int b0 = base + (int)threadIdx.x * 4;
		uint2 v[4];
		unsigned c4 = 0, g4 = 0;
		for (int k = 0; k < 4; ++k) { v[k] = b0 + k < n_blocks ? block_tot[b0 + k] : make_uint2(0, 0); c4 += v[k].x; g4 += v[k].y; }
		unsigned ce, ge, ct, gt;
		block_scan(c4, g4, ce, ge, ct, gt);
		unsigned long long c = c_run + ce, g = g_run + ge;
		for (int k = 0; k < 4; ++k) {
			// offsets stay below 2^32 (the host checks the totals): two 32-bit halves of the running sums
			if (b0 + k < n_blocks) block_tot[b0 + k] = make_uint2((unsigned)c, (unsigned)g);
			c += v[k].x; g += v[k].y;
		}
		c_run += ct; g_run += gt;
	}
	if (threadIdx.x == 0) { tot[0] = c_run; tot[1] = g_run; }
}

__global__ void __launch_bounds__(256)
ema_k_scan_write(int n_reads, const int *__restrict__ n_pairs_dev, const int *__restrict__ status, const int *__restrict__ n_regs,
                 const int *__restrict__ cig_n, const uint2 *__restrict__ block_off, uint64_t *__restrict__ cand_off, uint64_t *__restrict__ cig_off)
{
	const int n = ema_work_count(n_reads, n_pairs_dev, 2);
	const int r0 = (int)(blockIdx.x * 1024 + threadIdx.x * 4);
	unsigned c[4], g[4], c4 = 0, g4 = 0;
	for (int k = 0; k < 4; ++k) { counts_of(r0 + k, n, status, n_regs, cig_n, c[k], g[k]); c4 += c[k]; g4 += g[k]; }
	unsigned ce, ge, ct, gt;
	block_scan(c4, g4, ce, ge, ct, gt);
	const uint2 bo = block_off[blockIdx.x];
	uint64_t co = (uint64_t)bo.x + ce, go = (uint64_t)bo.y + ge;
	for (int k = 0; k < 4; ++k) {
		if (r0 + k <= n) { cand_off[r0 + k] = co; cig_off[r0 + k] = go; }      // entry n = the totals
		co += c[k]; go += g[k];
	}
}

// cand_off / cig_off: n + 1 entries each; block_tot: (n_reads + 1023) / 1024 + 1 entries; tot: 2 entries
extern "C" void ema_launch_scan(int n_reads, const int *n_pairs_dev, const int *status, const int *n_regs, const int *cig_n, uint2 *block_tot,
                                uint64_t *tot, uint64_t *cand_off, uint64_t *cig_off, hipStream_t stream)
{
	if (n_reads <= 0) return;
	const int nb = n_reads / 1024 + 1;      // covers entry n_reads too
	hipLaunchKernelGGL(ema_k_scan_blocks, dim3(nb), dim3(256), 0, stream, n_reads, n_pairs_dev, status, n_regs, cig_n, block_tot);
	hipLaunchKernelGGL(ema_k_scan_tops, dim3(1), dim3(256), 0, stream, nb, block_tot, tot);
	hipLaunchKernelGGL(ema_k_scan_write, dim3(nb), dim3(256), 0, stream, n_reads, n_pairs_dev, status, n_regs, cig_n, block_tot, cand_off, cig_off);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The batch's final layout on the device (ema_engine_fetch_ticket).  A pass leaves one packed result set per slice and one for the
// full-capacity tier; the batch the caller gets is ONE set in read order, a pair on the full tier's list taking its results from
// there.  Round 3 cut the sets apart on the host (0.6 CPU-seconds per million pairs: what an 8-GPU node with few CPUs is short
// of); now three small launches behind the last pack do it -- where every read's results are and how many, the batch-wide
// offsets (the scan above), the copy with CIGAR offsets rebased -- and the host downloads five arrays into a page-locked buffer
// that IS the batch.
// redo[0] = pairs listed, redo[1..] = their batch pair numbers; redo_idx[pair] = place on the list (filled with -1 before)
__global__ void __launch_bounds__(256)
ema_k_merge_redo(const int *__restrict__ redo, int cap, int *__restrict__ redo_idx)
{
	const int i = (int)(blockIdx.x * 256 + threadIdx.x);
	const int n = redo[0] < cap ? redo[0] : cap;
	if (i < n) redo_idx[redo[1 + i]] = i;
}

__global__ void __launch_bounds__(256)
ema_k_merge_src(MergeParts P, int n_reads, const int *__restrict__ redo_idx, uint32_t *__restrict__ src, int *__restrict__ m_c, int *__restrict__ m_g,
                int *__restrict__ status_out)
{
	const int r = (int)(blockIdx.x * 256 + threadIdx.x);
	if (r >= n_reads) return;
	int part = 0, idx = 0;
	const int ri = redo_idx[r >> 1];
	if (ri >= 0) { part = P.n_parts - 1; idx = 2 * ri + (r & 1); }
	else {
		for (int k = 0; k < P.n_parts - 1; ++k)
			if (r >= P.first_read[k] && r < P.first_read[k] + P.n_reads[k]) { part = k; idx = r - P.first_read[k]; }
	}
	src[r] = (uint32_t)part << 27 | (uint32_t)idx;
	m_c[r] = (int)(P.c_off[part][idx + 1] - P.c_off[part][idx]);
	m_g[r] = (int)(P.g_off[part][idx + 1] - P.g_off[part][idx]);
	status_out[r] = P.status[part][idx];
}

__global__ void __launch_bounds__(256)
ema_k_merge_copy(MergeParts P, int n_reads, const uint32_t *__restrict__ src, const uint64_t *__restrict__ cand_off, const uint64_t *__restrict__ cig_off,
                 ema_cand_t *__restrict__ cand, uint32_t *__restrict__ cigar, uint64_t cand_cap, uint64_t cigar_cap)
{
	const int r = (int)(blockIdx.x * 256 + threadIdx.x);
	if (r >= n_reads) return;
	const int part = (int)(src[r] >> 27), idx = (int)(src[r] & 0x7ffffffu);
	const uint64_t c0 = P.c_off[part][idx], nc = P.c_off[part][idx + 1] - c0, g0 = P.g_off[part][idx], ng = P.g_off[part][idx + 1] - g0;
	const uint64_t co = cand_off[r], go = cig_off[r];
	if (co + nc > cand_cap || go + ng > cigar_cap) return;      // (the host checks the totals against the same capacities and falls back to its own assembly)
	// a part whose own pack overflowed: its offsets count every candidate, its arrays end at their capacity -- the host
	// sees the part's totals after this launch and reports EMA_ELIMIT; nothing is read past the allocation meanwhile
	if (c0 + nc > P.cand_cap[part] || g0 + ng > P.cig_cap[part]) return;
	const ema_cand_t *sc = P.cand[part] + c0;
	for (uint64_t k = 0; k < nc; ++k) {
		ema_cand_t c = sc[k];
		c.cigar_off = (uint32_t)((uint64_t)c.cigar_off - g0 + go);
		cand[co + k] = c;
	}
	const uint32_t *sg = P.cig[part] + g0;
	for (uint64_t k = 0; k < ng; ++k) cigar[go + k] = sg[k];
}

// redo_idx: n_reads / 2 ints; src, m_c, m_g: n_reads words each; status_out, cand_off (n_reads + 1), cig_off (n_reads + 1), tot (2),
// cand, cigar: the merged set.  block_tot as for ema_launch_scan.
extern "C" void ema_launch_merge(const MergeParts *P, int n_reads, const int *redo, int redo_cap, int *redo_idx, uint32_t *src, int *m_c, int *m_g,
                                 uint2 *block_tot, uint64_t *tot, int *status_out, uint64_t *cand_off, uint64_t *cig_off, ema_cand_t *cand,
                                 uint32_t *cigar, uint64_t cand_cap, uint64_t cigar_cap, hipStream_t stream)
{
	if (n_reads <= 0) return;
	(void)hipMemsetAsync(redo_idx, 0xff, (size_t)(n_reads / 2) * 4, stream);
	hipLaunchKernelGGL(ema_k_merge_redo, dim3((unsigned)((redo_cap + 255) / 256)), dim3(256), 0, stream, redo, redo_cap, redo_idx);
	hipLaunchKernelGGL(ema_k_merge_src, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, *P, n_reads, redo_idx, src, m_c, m_g, status_out);
	ema_launch_scan(n_reads, nullptr, nullptr, m_c, m_g, block_tot, tot, cand_off, cig_off, stream);
	hipLaunchKernelGGL(ema_k_merge_copy, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, *P, n_reads, src, cand_off, cig_off, cand, cigar,
	                   cand_cap, cigar_cap);
}

// Input staging on the device.  A batch arrives as the caller's ASCII bases; one thread per 32 bases of a read turns them into nt4
// codes in place (seq_convert, reference src/bwabridge.c:151-157 with bwa's nst_nt4_table: ACGT / acgt -> 0..3, any other
// byte 4) and writes its share of the read's packed form (24 words per read: 16 of 2-bit codes, first base lowest, and 8 of
// "ambiguous base" flags) -- byte work that took 0.46 CPU-seconds per million pairs on the host (r03) and is nothing here.
__global__ void __launch_bounds__(256)
ema_k_stage_reads(const uint32_t *__restrict__ off, int n_reads, uint8_t *__restrict__ bases, uint32_t *__restrict__ qpack)
{
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t r = t >> 3;
	const int c = (int)(t & 7);
	if (r >= (size_t)n_reads) return;
	const uint32_t b0 = off[r], len = off[r + 1] - b0;
	uint32_t w[2] = {0, 0}, amb = 0;
#pragma unroll
	for (int i = 0; i < 32; ++i) {
		const uint32_t pos = (uint32_t)c * 32 + (uint32_t)i;
		if (pos < len) {
			const uint32_t x = bases[b0 + pos] | 0x20u;
			const uint32_t code = x == 'a' ? 0u : x == 'c' ? 1u : x == 'g' ? 2u : x == 't' ? 3u : 4u;
			bases[b0 + pos] = (uint8_t)code;
			w[i >> 4] |= (code & 3u) << ((i & 15) << 1);
			if (code > 3u) amb |= 1u << i;
		}
	}
	uint32_t *q = qpack + r * 24;
	q[2 * c] = w[0]; q[2 * c + 1] = w[1]; q[16 + c] = amb;
}
extern "C" void ema_launch_stage_reads(const uint32_t *off, int n_reads, uint8_t *bases, uint32_t *qpack, hipStream_t stream)
{
	if (n_reads <= 0) return;
	hipLaunchKernelGGL(ema_k_stage_reads, dim3((unsigned)(((size_t)n_reads * 8 + 255) / 256)), dim3(256), 0, stream, off, n_reads, bases, qpack);
}

// offsets of one bucket laid behind others in a shared input slot (ema_engine_stage_async_dev): dst[r] = src[r] + add, where src[r] is
// the END of the bucket's read r (src points at its off[1]).  *too_long is set when a read is longer than max_len or the offsets
// run backwards: the kernels behind (ema_k_stage_reads, K1's packed reads) hold EMA_MAX_READ bases per read and nothing more
__global__ void __launch_bounds__(256)
ema_k_rebase_off(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t n, uint32_t add, uint32_t max_len, int *__restrict__ too_long)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r >= n) return;
	const uint32_t end = src[r], beg = src[(int)r - 1];
	if (end < beg || end - beg > max_len) *too_long = 1;
	dst[r] = end + add;
}
extern "C" void ema_launch_rebase_off(uint32_t *dst, const uint32_t *src, uint32_t n, uint32_t add, uint32_t max_len, int *too_long, hipStream_t stream)
{
	if (!n) return;
	hipLaunchKernelGGL(ema_k_rebase_off, dim3((n + 255u) / 256u), dim3(256), 0, stream, dst, src, n, add, max_len, too_long);
}
