// ema_amd/csrc/dev_common.hpp -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#ifndef EMA_DEV_COMMON_HPP
#define EMA_DEV_COMMON_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"

#define EMA_WAVE 64
// LDS addressed as LDS (ds_read / ds_write): through a generic pointer every access is a FLAT instruction, which takes the longer
// way and waits for the global stores in flight
#if defined(__HIP_DEVICE_COMPILE__)
#define EMA_LDS __attribute__((address_space(3)))
#else
#define EMA_LDS
#endif

__device__ __forceinline__ unsigned ema_lane() { return threadIdx.x & 63u; }

// Wave-level ordering point, used wherever lanes of one wavefront communicate through memory (one lane's stores
// are about to be read by the others, or old values must have been read before the leader lane overwrites them).
// The lanes execute in lock-step and share one path to memory, so no instruction is needed -- but the compiler
// must be told: to it every lane is an independent thread whose earlier loads stay valid until that thread
// itself stores.  The wavefront-scope fence (no code at this scope) makes it drop values cached in registers
// and keeps loads/stores from moving across this point.
// Work list of a launch.  Lean tier: units 0..n-1 of the slice.  Full-capacity tier: the pairs the lean tier flagged,
// listed on the device -- *n_pairs_dev of them (clamped to the tier's capacity) -- and unit i reads its input (bases,
// offsets, packed reads) at batch pair map[i] while all per-read results are indexed by i.
__device__ __forceinline__ int ema_work_count(int n, const int *n_pairs_dev, int per_pair)
{
	if (!n_pairs_dev) return n;
	const int m = per_pair * *n_pairs_dev;
	return m < n ? m : n;
}
__device__ __forceinline__ int ema_in_read(const int *map, int read) { return map ? ((map[read >> 1] << 1) | (read & 1)) : read; }

__device__ __forceinline__ void ema_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Marks a value that is the same in every lane of the wavefront ("wave-uniform"): it is read from lane 0 into a
// scalar register, so loops and branches on it become scalar branches (no exec-mask bookkeeping, fewer VGPRs) and
// every lane provably follows the same path into the cross-lane operations that follow.
__device__ __forceinline__ int ema_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool ema_uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ int64_t ema_uni(int64_t v)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
	return (int64_t)((uint64_t)hi << 32 | lo);
}
__device__ __forceinline__ uint64_t ema_uni(uint64_t v) { return (uint64_t)ema_uni((int64_t)v); }
// lane `l`'s value (l wave-uniform) as a scalar
__device__ __forceinline__ int ema_lane_val(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int64_t ema_lane_val(int64_t v, int l)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, l);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
	return (int64_t)((uint64_t)hi << 32 | lo);
}
__device__ __forceinline__ uint64_t ema_lane_val(uint64_t v, int l) { return (uint64_t)ema_lane_val((int64_t)v, l); }
__device__ __forceinline__ float ema_uni(float v) { return __int_as_float(ema_uni(__float_as_int(v))); }
__device__ __forceinline__ Intv ema_uni(const Intv &v)
{
	Intv r; r.x0 = ema_uni(v.x0); r.x1 = ema_uni(v.x1); r.x2 = ema_uni(v.x2); r.info = ema_uni(v.info);
	return r;
}
__device__ __forceinline__ SeedRec ema_uni(const SeedRec &s)
{
	SeedRec r; r.rbeg = ema_uni(s.rbeg); r.qbeg = ema_uni(s.qbeg); r.len = ema_uni(s.len); r.next = ema_uni(s.next); r.pad = 0;
	return r;
}
__device__ __forceinline__ ChainRec ema_uni(const ChainRec &c)
{
	ChainRec r;
	r.pos = ema_uni(c.pos); r.f_rbeg = ema_uni(c.f_rbeg); r.l_rbeg = ema_uni(c.l_rbeg);
	r.f_qbeg = ema_uni(c.f_qbeg); r.l_qbeg = ema_uni(c.l_qbeg); r.l_len = ema_uni(c.l_len);
	r.rid = ema_uni(c.rid); r.n = ema_uni(c.n); r.first_seed = ema_uni(c.first_seed); r.last_seed = ema_uni(c.last_seed);
	r.w = ema_uni(c.w); r.kept = ema_uni(c.kept); r.first = ema_uni(c.first);
	return r;
}
__device__ __forceinline__ DevReg ema_uni(const DevReg &g)
{
	DevReg r;
	r.rb = ema_uni(g.rb); r.re = ema_uni(g.re); r.qb = ema_uni(g.qb); r.qe = ema_uni(g.qe); r.rid = ema_uni(g.rid);
	r.score = ema_uni(g.score); r.truesc = ema_uni(g.truesc); r.sub = ema_uni(g.sub); r.csub = ema_uni(g.csub);
	r.w = ema_uni(g.w); r.seedcov = ema_uni(g.seedcov); r.secondary = ema_uni(g.secondary); r.seedlen0 = ema_uni(g.seedlen0);
	r.n_comp = ema_uni(g.n_comp); r.is_alt = ema_uni(g.is_alt); r.frac_rep = ema_uni(g.frac_rep);
	return r;
}

// ---------------------------------------------------------------------------------------------
// Work items of a launch, taken from its shared counter EMA_CLAIM at a time.  One atomic per item -- every wavefront of the chip
// on one address, a dependent look-up of the item's list entry behind it -- cost the wave-per-read kernels 30-50 K clocks per
// item (r03, product-build profile of K2b: the L2 serialises the updates of one address); a claim of four, with the four list
// entries fetched together by four lanes, costs 2 K.  Near the end of the queue -- fewer than two claims of four per resident
// wavefront left, judged by this wave's previous claim -- items are taken singly again: their costs have long tails, and a
// wavefront working through four heavy ones while the others have run out of work set the length of K3 (r03q: 5.2 -> 9.5 ms).
#define EMA_CLAIM 4
struct EmaClaim { int base = 0, n = 0, i = 0; unsigned long long mine = 0; };
__device__ __forceinline__ int ema_claim_step(int last_base, int total)
{
	const int waves = (int)(gridDim.x * (blockDim.x >> 6));
	return (long long)last_base + 2LL * EMA_CLAIM * waves < (long long)total ? EMA_CLAIM : 1;
}
// The next item's index, or -1 when none is left.  total: items in all (final before the launch).  list (may be null): one entry
// per item; `entry` receives this item's.
template <typename T>
__device__ __forceinline__ int ema_claim_next(EmaClaim &c, int *counter, int total, const T *list, T &entry)
{
	const int lane = (int)ema_lane();
	if (c.i == c.n) {
		const int step = ema_claim_step(c.base, total);
		int b = 0;
		if (lane == 0) b = atomicAdd(counter, step);
		b = __builtin_amdgcn_readlane(b, 0);
		if (b >= total) return -1;
		c.base = b; c.n = total - b < step ? total - b : step; c.i = 0;
		if (list && lane < c.n) c.mine = (unsigned long long)list[b + lane];
	}
	const int k = c.i++;
	if (list) entry = (T)ema_lane_val((uint64_t)c.mine, k);
	return c.base + k;
}

// ---------------------------------------------------------------------------------------------
// Wave-wide scans and reductions on DPP (data-parallel primitives: operands are read from a neighbouring lane by
// the VALU itself -- a few cycles -- instead of going through the LDS crossbar like ds_bpermute/__shfl, which
// costs >100 cycles per dependent step and dominated the row-parallel DPs).  gfx950 keeps the GFX9 controls
// row_shr:n (within a row of 16 lanes), row_bcast:15 / row_bcast:31 (carry a row's last lane into the next
// row / next two rows) and wave_shr:1.  Lanes without a source receive `ident`.
#define EMA_DPP(old, v, ctrl, row_mask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (row_mask), 0xf, false)

// One instruction per step: v_max_i32_dpp / v_add_u32_dpp with the lane's own register as destination and both sources -- a
// lane without a source, or in a row the step's row mask leaves out, is not written and keeps its value, which is what the
// identity would have given it.  Through __builtin_amdgcn_update_dpp the compiler emits three instructions per step (reload the
// identity, v_mov_b32_dpp, operate) and does not fold them (r03: the two scans of every DP row were a fifth of the row's
// instructions).  The s_nop covers the two wait states a DPP read needs after a VALU write of the same register.
#if defined(__HIP_DEVICE_COMPILE__)
#define EMA_DPP_STEP(op, v, ctrl) asm volatile("s_nop 1\n\t" op " %0, %0, %0 " ctrl : "+v"(v))
// first step of a scan: 5 wait states, which also cover a VALU write of EXEC (v_cmpx, v_readlane-style) that the compiler may have
// scheduled right before the block -- its hazard recognizer does not look inside the asm
#define EMA_DPP_STEP0(op, v, ctrl) asm volatile("s_nop 4\n\t" op " %0, %0, %0 " ctrl : "+v"(v))
#endif
__device__ __forceinline__ int ema_wave_incl_scan_max(int v, int ident)
{
#if defined(__HIP_DEVICE_COMPILE__)
	(void)ident;
	EMA_DPP_STEP0("v_max_i32_dpp", v, "row_shr:1 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_max_i32_dpp", v, "row_shr:2 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_max_i32_dpp", v, "row_shr:4 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_max_i32_dpp", v, "row_shr:8 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_max_i32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");      // into rows 1 and 3
	EMA_DPP_STEP("v_max_i32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");      // into rows 2 and 3
	return v;
#else      // (the host interpreter of tests/emu)
	int t;
	t = EMA_DPP(ident, v, 0x111, 0xf); v = max(v, t);     // row_shr:1
	t = EMA_DPP(ident, v, 0x112, 0xf); v = max(v, t);     // row_shr:2
	t = EMA_DPP(ident, v, 0x114, 0xf); v = max(v, t);     // row_shr:4
	t = EMA_DPP(ident, v, 0x118, 0xf); v = max(v, t);     // row_shr:8
	t = EMA_DPP(ident, v, 0x142, 0xa); v = max(v, t);     // row_bcast:15 into rows 1 and 3
	t = EMA_DPP(ident, v, 0x143, 0xc); v = max(v, t);     // row_bcast:31 into rows 2 and 3
	return v;
#endif
}
__device__ __forceinline__ int ema_wave_incl_scan_add(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	EMA_DPP_STEP0("v_add_u32_dpp", v, "row_shr:1 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_add_u32_dpp", v, "row_shr:2 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_add_u32_dpp", v, "row_shr:4 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_add_u32_dpp", v, "row_shr:8 row_mask:0xf bank_mask:0xf");
	EMA_DPP_STEP("v_add_u32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
	EMA_DPP_STEP("v_add_u32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
	return v;
#else
	int t;
	t = EMA_DPP(0, v, 0x111, 0xf); v += t;
	t = EMA_DPP(0, v, 0x112, 0xf); v += t;
	t = EMA_DPP(0, v, 0x114, 0xf); v += t;
	t = EMA_DPP(0, v, 0x118, 0xf); v += t;
	t = EMA_DPP(0, v, 0x142, 0xa); v += t;
	t = EMA_DPP(0, v, 0x143, 0xc); v += t;
	return v;
#endif
}
// value of the lane below (lane 0 receives `ident`)
__device__ __forceinline__ int ema_wave_shr1(int v, int ident) { return EMA_DPP(ident, v, 0x138, 0xf); }
// value of the lane above (lane 63 receives `ident`): wave_shl:1
__device__ __forceinline__ int ema_wave_shl1(int v, int ident) { return EMA_DPP(ident, v, 0x130, 0xf); }
// wave-uniform results in scalar registers
__device__ __forceinline__ int ema_wave_sum(int v) { return __builtin_amdgcn_readlane(ema_wave_incl_scan_add(v), 63); }

// ---------------------------------------------------------------------------------------------
// occ4 by ONE lane on the rank structure of dev_types.h: cnt[c] = number of symbol c in B[0..pos] (bwa's bwt_occ4
// semantics; reference path: src/bwabridge.c:236 -> mem_align1_core -> bwt_extend), `pos` in the with-sentinel row
// space.  The lane fetches the 32-byte block with two 16-byte loads and does the popcounts itself, so a wavefront
// keeps 64 independent searches in flight and needs no cross-lane traffic.
//
// [r6] The block's 64 symbols are two BIT PLANES (dev_types.h): word 0 = the symbols' low bits, word 1 = their high bits, symbol t at
// bit t of both.  Counting the symbols 0..r is one mask and three population counts -- low bits, high bits, both -- from which the
// four counts follow by subtraction: with the symbols as 2-bit codes side by side (rounds 1-5) the same query took a 64-bit mask
// per word and nine masked population counts, ~95 vector instructions per block against ~20.
__device__ __forceinline__ void ema_occ_planes(uint64_t lo, uint64_t hi, int r, unsigned &pl, unsigned &ph, unsigned &p3)
{
	const uint64_t m = (2ULL << r) - 1;      // positions 0..r (r = 63: 2 << 63 wraps to 0, minus one = all of them)
	lo &= m; hi &= m;
	pl = (unsigned)__popcll(lo); ph = (unsigned)__popcll(hi); p3 = (unsigned)__popcll(lo & hi);
}
// the arithmetic of ema_lane_occ4 on a block that is already in registers (head = the four counts, sym4 = the two planes):
// p = the row with the sentinel taken out
__device__ __forceinline__ void ema_occ4_decode(const DevIndex &ix, uint64_t p, const uint4 &head, const uint4 &sym4, uint64_t cnt[4])
{
	const int r = (int)(p & 63);
	unsigned pl, ph, p3;
	ema_occ_planes((uint64_t)sym4.y << 32 | sym4.x, (uint64_t)sym4.w << 32 | sym4.z, r, pl, ph, p3);
	cnt[0] = (uint64_t)head.x + ((unsigned)(r + 1) - pl - ph + p3); cnt[1] = (uint64_t)head.y + (pl - p3);
	cnt[2] = (uint64_t)head.z + (ph - p3); cnt[3] = (uint64_t)head.w + p3;
	if (ix.n_super > 1) {      // wave-uniform: only references beyond 2^31 BWT symbols have more than one superblock
		const int sb = (int)(p >> EMA_OCC_SUPER_SHIFT);
#pragma unroll
		for (int c = 0; c < 4; ++c)
			cnt[c] += sb == 0 ? 0 : sb == 1 ? ix.occ_super[0][c] : sb == 2 ? ix.occ_super[1][c] : ix.occ_super[2][c];
	}
}
__device__ __forceinline__ void ema_lane_occ4(const DevIndex &ix, uint64_t pos, uint64_t cnt[4])
{
	const uint64_t p = pos - (pos >= ix.primary ? 1 : 0);   // '$' is not stored
	const OccBlock *blk = ix.occ + (p >> 6);
	const uint4 head = *reinterpret_cast<const uint4 *>(blk);
	const uint4 sym = *(reinterpret_cast<const uint4 *>(blk) + 1);
	ema_occ4_decode(ix, p, head, sym, cnt);
}

// bwt_extend for one symbol by one lane: x_nb = the coordinate the rank queries are made on (k for a backward extension, k' for a
// forward one), x_b = the other, c = the symbol (complemented by the caller for a forward extension)
__device__ __forceinline__ void ema_lane_extend(const DevIndex &ix, uint64_t x_nb, uint64_t x_b, uint64_t size, int c,
                                                uint64_t &o_nb, uint64_t &o_b, uint64_t &o_size)
{
	uint64_t tk[4], tl[4];
	ema_lane_occ4(ix, x_nb - 1, tk);
	ema_lane_occ4(ix, x_nb - 1 + size, tl);
	const uint64_t s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
	const uint64_t b3 = x_b + ((x_nb <= ix.primary && x_nb + size - 1 >= ix.primary) ? 1 : 0);
	const uint64_t b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
	const int cc = c & 3;
	o_b = cc == 3 ? b3 : cc == 2 ? b2 : cc == 1 ? b1 : b0;
	o_size = cc == 3 ? s3 : cc == 2 ? s2 : cc == 1 ? s1 : s0;
	o_nb = ix.L2[cc] + 1 + (cc == 3 ? tk[3] : cc == 2 ? tk[2] : cc == 1 ? tk[1] : tk[0]);
}

// [r6] bwt_extend for ONE symbol from the two rank blocks in registers -- the lane machines' form (k_seed.hip,
// k_seed_p3.hip), where a tick's instruction count is what the chip is short of.  bwt_extend derives all four symbols' intervals;
// the caller keeps one: its size s[cc], the rows of the larger symbols before it (n_gt = s[cc + 1] + .. + s[3], what the untouched
// coordinate advances by) and its start L2[cc] + 1 + occ(cc, k - 1).  So the sizes stay 32-bit DIFFERENCES of the two blocks'
// counts (an interval that needs a rank query is far below 2^32 rows, and the superblock bases cancel unless the interval
// straddles a superblock boundary: a branch no wavefront takes in practice), and the one 64-bit quantity, the start, takes
// its base -- L2[cc] + 1 + the superblock's count -- from a 16-entry table in LDS (ema_rank_table_init) instead of two chains of selects.
#define EMA_RANK_TABLE 16
__device__ __forceinline__ void ema_rank_table_init(const DevIndex &ix, uint64_t *T)      // T: this WAVEFRONT's 16 words of LDS
{
	const int t = (int)(threadIdx.x & 63);
#pragma unroll
	for (int sb = 0; sb < 4; ++sb)
#pragma unroll
		for (int c = 0; c < 4; ++c)
			if (t == sb * 4 + c) T[t] = ix.L2[c] + 1 + (sb ? ix.occ_super[sb ? sb - 1 : 0][c] : 0);
	ema_wave_sync();
}
__device__ __forceinline__ void ema_extend_blocks(const DevIndex &ix, const uint64_t *T, uint64_t qk, uint64_t ql, const uint4 &hk, const uint4 &sk,
                                                  const uint4 &hl, const uint4 &sl, int cc, uint64_t &o_start, uint32_t &o_size, uint32_t &n_gt)
{
	const int rk = (int)(qk & 63), rl = (int)(ql & 63);
	unsigned plk, phk, p3k, pll, phl, p3l;
	ema_occ_planes((uint64_t)sk.y << 32 | sk.x, (uint64_t)sk.w << 32 | sk.z, rk, plk, phk, p3k);
	ema_occ_planes((uint64_t)sl.y << 32 | sl.x, (uint64_t)sl.w << 32 | sl.z, rl, pll, phl, p3l);
	const uint32_t d3 = p3l - p3k, dh = phl - phk, dl = pll - plk;
	uint32_t s3 = hl.w - hk.w + d3, s2 = hl.z - hk.z + dh - d3, s1 = hl.y - hk.y + dl - d3, s0 = hl.x - hk.x + (uint32_t)(rl - rk) - dh - dl + d3;
	int sbk = 0;
	if (ix.n_super > 1) {      // wave-uniform
		sbk = (int)(qk >> EMA_OCC_SUPER_SHIFT);
		const int sbl = (int)(ql >> EMA_OCC_SUPER_SHIFT);
		if (sbk != sbl) {      // the interval straddles a superblock boundary
			s0 += (uint32_t)(T[sbl * 4 + 0] - T[sbk * 4 + 0]); s1 += (uint32_t)(T[sbl * 4 + 1] - T[sbk * 4 + 1]);
			s2 += (uint32_t)(T[sbl * 4 + 2] - T[sbk * 4 + 2]); s3 += (uint32_t)(T[sbl * 4 + 3] - T[sbk * 4 + 3]);
		}
	}
	// (selects on two shared conditions, spelt with masks: left to itself the compiler turns the four-way choices into branches,
	// and a divergent branch costs a wavefront more than the three selects)
	const uint32_t odd = 0u - (uint32_t)(cc & 1), top = 0u - (uint32_t)((cc >> 1) & 1);
	auto sel4 = [&](uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) -> uint32_t {
		const uint32_t a = (v0 & ~odd) | (v1 & odd), b = (v2 & ~odd) | (v3 & odd);
		return (a & ~top) | (b & top);
	};
	const uint32_t ck = sel4(hk.x + ((uint32_t)(rk + 1) - plk - phk + p3k), hk.y + (plk - p3k), hk.z + (phk - p3k), hk.w + p3k);
	o_size = sel4(s0, s1, s2, s3);
	n_gt = sel4(s3 + s2 + s1, s3 + s2, s3, 0u);
	o_start = T[sbk * 4 + cc] + ck;
}

// k-mer interval table (dev_types.h): where level L begins, in entries.  (4^L - 1) / 3 is the bit pattern 0101..01 with L ones, so
// the offsets (4^L - 4) / 3 and (4^L - 4^(EMA_KMER_WIDE + 1)) / 3 are a mask and a subtraction (written as a division the compiler
// emits a 64-bit multiply-high sequence per look-up, in every tick of the lane machines).  L <= EMA_KMER_MAX = 15: 32 bits hold them.
__device__ __forceinline__ uint32_t ema_kmer_base_wide(int L) { return (0x55555555u & ((1u << (2 * L)) - 1u)) - 1u; }
__device__ __forceinline__ uint32_t ema_kmer_base_narrow(int L)
{
	return (0x55555555u & ((1u << (2 * L)) - 1u)) - (0x55555555u & ((1u << (2 * (EMA_KMER_WIDE + 1))) - 1u));
}
// suffix-array interval of the string with 2-bit code `code` (first base in the high bits) and length L <= ix.kmer_k
__device__ __forceinline__ void ema_kmer_lookup(const DevIndex &ix, int L, uint32_t code, uint64_t &x0, uint64_t &x2)
{
	if (L <= EMA_KMER_WIDE) {
		const ulong2 e = reinterpret_cast<const ulong2 *>(ix.kmer_wide)[(size_t)ema_kmer_base_wide(L) + code];
		x0 = e.x; x2 = e.y;
	} else {
		const uint64_t e = ix.kmer_narrow[(size_t)ema_kmer_base_narrow(L) + code];
		x0 = e & 0xFFFFFFFFFFULL; x2 = e >> 40;
	}
}
// code of the reverse complement of a string of length L
__device__ __forceinline__ uint32_t ema_kmer_revcomp(uint32_t code, int L)
{
	uint32_t v = ~code;                                   // complement: 3 - base
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);      // reverse the order of the 16 two-bit groups
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	v = (v >> 16) | (v << 16);
	return v >> (32 - 2 * L);
}

// suffix array row -> text position (the whole SA is resident)
__device__ __forceinline__ uint64_t ema_sa(const DevIndex &ix, uint64_t row)
{
	return ix.sa_width == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(ix.sa)[row]
	                        : reinterpret_cast<const uint64_t *>(ix.sa)[row];
}

#endif
