// ema_amd/csrc/dev_common.hpp -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#ifndef EMA_DEV_COMMON_HPP
#define EMA_DEV_COMMON_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"

#define EMA_WAVE 64

__device__ __forceinline__ unsigned ema_lane() { return threadIdx.x & 63u; }

// Wave-level ordering point.  The lanes of a wavefront execute in lock-step, so this emits no instruction; it
// stops the compiler from moving memory operations across it and marks the places where one lane's stores are
// about to be read by the others (or where old values must have been read before the leader lane overwrites them).
__device__ __forceinline__ void ema_wave_sync() { __builtin_amdgcn_wave_barrier(); }

// ---------------------------------------------------------------------------------------------
// occ4 on the HBM block layout of dev_types.h, computed by FOUR adjacent lanes.
// Lane j (= lane & 3) of the quad loads slot j of the block that holds BWT position `pos`
// (one 16-byte load; the quad's four loads are one contiguous 64-byte line) and the quad
// returns, in lane j, occ(pos, symbol j) = number of symbol j in B[0..pos] (bwa's bwt_occ4
// semantics, reference path: src/bwabridge.c:236 -> mem_align1_core -> bwt_extend).
// `pos` is in the with-sentinel row space; quad-uniform.  All 64 lanes must call this.
__device__ __forceinline__ uint64_t ema_quad_occ4(const DevIndex &ix, uint64_t pos, bool active)
{
	const unsigned j = ema_lane() & 3u;
	uint64_t cnt = 0, w = 0;
	unsigned r = 0;
	if (active) {
		const uint64_t p = pos - (pos >= ix.primary ? 1 : 0);   // '$' is not stored
		const OccSlot *s = ix.occ + ((p >> 7) << 2) + j;
		const ulong2 v = *reinterpret_cast<const ulong2 *>(s);
		cnt = v.x; w = v.y;
		r = (unsigned)(p & 127);
	}
	int nvalid = (int)r + 1 - (int)(j << 5);
	nvalid = nvalid < 0 ? 0 : (nvalid > 32 ? 32 : nvalid);
	const uint64_t m55 = nvalid == 32 ? 0x5555555555555555ULL : (((1ULL << (2 * nvalid)) - 1) & 0x5555555555555555ULL);
	const uint64_t lo = w & m55, hi = (w >> 1) & m55;
	const unsigned p3 = __popcll(hi & lo), p2 = __popcll(hi & ~lo), p1 = __popcll(~hi & lo);
	const unsigned p0 = (unsigned)nvalid - p1 - p2 - p3;
	unsigned packed = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);   // each partial <= 32, sums <= 128
	packed += __shfl_xor(packed, 1);
	packed += __shfl_xor(packed, 2);
	return cnt + ((packed >> (j << 3)) & 0xffu);
}

// ---------------------------------------------------------------------------------------------
// bwt_extend for ONE symbol, computed by a group of EIGHT adjacent lanes: lanes 0-3 of the group
// resolve occ4(k), lanes 4-7 occ4(l).  Inputs and outputs are group-uniform.
//   x_nb = ik.x[!is_back], x_b = ik.x[is_back], size = ik.x[2]; c = symbol index into ok[]
//   returns ok[c]: o_nb = ok[c].x[!is_back], o_b = ok[c].x[is_back], o_size = ok[c].x[2]
__device__ __forceinline__ void ema_group8_extend(const DevIndex &ix, uint64_t x_nb, uint64_t x_b, uint64_t size,
                                                  int c, bool active, uint64_t &o_nb, uint64_t &o_b, uint64_t &o_size)
{
	const unsigned sub = ema_lane() & 7u, j = sub & 3u, half = sub >> 2;
	const uint64_t pos = half ? x_nb - 1 + size : x_nb - 1;
	const uint64_t mine = ema_quad_occ4(ix, pos, active);
	const uint64_t other = __shfl_xor(mine, 4);
	const uint64_t tk = half ? other : mine, tl = half ? mine : other;
	const uint64_t s = tl - tk;                       // ok[j].x[2]
	const uint64_t nb = ix.L2[j] + 1 + tk;            // ok[j].x[!is_back]
	const uint64_t s1 = __shfl(s, 1, 8), s2 = __shfl(s, 2, 8), s3 = __shfl(s, 3, 8);
	const uint64_t b3 = x_b + ((x_nb <= ix.primary && x_nb + size - 1 >= ix.primary) ? 1 : 0);
	const uint64_t b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
	const int cc = c & 3;
	o_b = cc == 3 ? b3 : cc == 2 ? b2 : cc == 1 ? b1 : b0;
	o_nb = __shfl(nb, cc, 8);
	o_size = __shfl(s, cc, 8);
}

// suffix array row -> text position (the whole SA is resident)
__device__ __forceinline__ uint64_t ema_sa(const DevIndex &ix, uint64_t row)
{
	return ix.sa_width == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(ix.sa)[row]
	                        : reinterpret_cast<const uint64_t *>(ix.sa)[row];
}

#endif
