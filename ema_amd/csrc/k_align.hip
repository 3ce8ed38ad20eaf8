// ema_amd/csrc/k_align.hip -- K2b: seeds -> chains -> extended, de-duplicated regions, one wavefront per read, for the
// reads K2a (k_align_lane.hip, one lane per read) left on its todo list: the ones with many seed occurrences or regions.
//
// Replaces, for a batch, everything mem_align1_core does after seeding (un-vendored bwa, reached from
// reference src/bwabridge.c:236-237): bwt_sa for every seed occurrence, mem_chain (chaining through a sorted
// set of open chains), mem_chain_flt, mem_chain2aln (banded extension of the chain's seeds) and
// mem_sort_dedup_patch.
//
// Why one wave per read: which seeds get extended depends on the regions produced so far for the same read,
// chain filtering depends on an unstable sort, and dedup/patch walks pairs of regions in order -- the control
// flow is inherently sequential per read, while its heavy pieces are data-parallel: the suffix-array loads of
// one seed interval (coalesced: consecutive rows), the DP rows (dev_dp.hpp), the containment tests against all
// previous regions, the chain-overlap tests against all kept chains, window fetches.  So the wave runs the
// sequential control program with wave-uniform scalars and spreads each of those pieces over its 64 lanes.
// Per-read working sets (seed pool, chain table, sort keys, region list) live in a private slab of HBM scratch
// per resident wave -- with 288 GB there is room for generous worst-case slabs (~3.5 MB each) and no need for
// inter-kernel compaction; the read and the reference window sit in LDS.
#include <hip/hip_runtime.h>
#include "dev_regions.hpp"

struct AlignSlab {           // per resident wave
	SeedRec *seeds;          // EMA_SEED_CAP     seed pool, chains are linked lists through it
	ChainRec *chains;        // EMA_CHAIN_CAP    in creation order
	int64_t *cpos;           // EMA_CHAIN_CAP    chain positions, ascending (the "tree")
	int32_t *cord;           // EMA_CHAIN_CAP    chain ids in the same order
	uint64_t *skey;          // EMA_CHAIN_CAP    weight << 32 | id, for the filter's sort
	int32_t *kept;           // EMA_CHAIN_CAP    kept chains (positions in sorted order)
	SeedRec *cs;             // EMA_SEED_CAP     seeds of the chain being extended, contiguous
	uint64_t *srt;           // EMA_SEED_CAP
	DevReg *av, *av_tmp;     // EMA_AV_CAP
	uint64_t *rkeys;         // EMA_AV_CAP
	Intv *ivs;               // EMA_INTV_CAP      the read's seed intervals ordered by (start, end)
};

#define EMA_ALIGN_SLAB_BYTES                                                                                       \
	((size_t)EMA_SEED_CAP * (2 * sizeof(SeedRec) + 8) + (size_t)EMA_CHAIN_CAP * (sizeof(ChainRec) + 8 + 4 + 8 + 4) +  \
	 (size_t)EMA_AV_CAP * (2 * sizeof(DevReg) + 8) + (size_t)EMA_INTV_CAP * sizeof(Intv) + 1024)

// Most reads have a few dozen seed occurrences at most.  Their chaining tables then sit in LDS instead of the HBM slab:
// chaining is a sequence of dependent small look-ups and edits (find the neighbouring chain, test, link the seed), and
// each one costs a memory round trip when the tables are in HBM.
// The build is ema_k_align_t<32, 8, 4> (template parameters SMALL = seed occurrences whose tables sit in LDS, AVL = regions before
// de-duplication kept in LDS, blocks per SIMD): 16 waves per CU.
// Above SMALL occurrences ( at most EMA_MED_CHAINS chains -- a tenth of the reads and, while everything of
// theirs sat in the slab, two fifths of this kernel's wave clocks): the chain and seed RECORDS stay in the slab, but the
// structures every step searches or walks sit in LDS that is idle in that phase --
//   chaining:  the sorted chain positions and their ids (cpos in the reference-window buffer, cord in the small-table area):
//              the look-up, the neighbour test and the insertion shift cost no memory round trip; what remains per occurrence
//              is the one load of the neighbouring chain's record;
//   filter:    the sort keys (small-table area), a summary {weight, query begin, end, kept flag} per chain (window buffer) and the
//              kept list with its first-shadowed entries (region-list area): the single-lane sort and the kept-chain loop run on
//              LDS alone, and the kept flags go back to the chain records in one pass;
//   extension: the sort keys stay where they are; the seeds of the chain being extended and their order (cs, srt) use the head
//              of the small-table area when the chain has no more than SMALL seeds.
// A read whose chain count outgrows EMA_MED_CHAINS moves its position table to the slab and carries on there.
#define EMA_MED_CHAINS 256
#ifndef EMA_SORT_WAVE
#define EMA_SORT_WAVE 1      // [r5] the chain filter's ks_introsort (medium layout: up to 256 weights, ties the rule) by the whole wavefront (dev_sort.hpp, ema_introsort_wave); 0: by one lane
#endif
#ifndef EMA_CHAIN_REGS
#define EMA_CHAIN_REGS 1      // [r5] the medium layout's tables in registers while a read has at most 64 chains (chain_insert_reg); 0: round 4's LDS tables from the start
#endif
#define EMA_SMALL_BYTES(SMALL) ((SMALL) * (2 * sizeof(SeedRec) + sizeof(ChainRec) + 3 * 8 + 2 * 4))
#define EMA_AVL_BYTES(AVL) (((AVL) > 0 ? (AVL) : 1) * (2 * sizeof(DevReg) + 8))

template <int SMALL>
__device__ __forceinline__ void ema_small_tables(AlignSlab &s, uint8_t *lds)
{
	size_t o = 0;
	auto take = [&](size_t bytes) { uint8_t *p = lds + o; o += bytes; return p; };
	s.chains = (ChainRec *)take(SMALL * sizeof(ChainRec));
	s.seeds = (SeedRec *)take(SMALL * sizeof(SeedRec));
	s.cs = (SeedRec *)take(SMALL * sizeof(SeedRec));
	s.cpos = (int64_t *)take(SMALL * 8);
	s.skey = (uint64_t *)take(SMALL * 8);
	s.srt = (uint64_t *)take(SMALL * 8);
	s.cord = (int32_t *)take(SMALL * 4);
	s.kept = (int32_t *)take(SMALL * 4);
}

__device__ __forceinline__ AlignSlab ema_carve_slab(uint8_t *base)
{
	AlignSlab s;
	size_t o = 0;
	auto take = [&](size_t bytes) { uint8_t *p = base + o; o += (bytes + 63) & ~(size_t)63; return p; };
	s.seeds = (SeedRec *)take((size_t)EMA_SEED_CAP * sizeof(SeedRec));
	s.cs = (SeedRec *)take((size_t)EMA_SEED_CAP * sizeof(SeedRec));
	s.srt = (uint64_t *)take((size_t)EMA_SEED_CAP * 8);
	s.chains = (ChainRec *)take((size_t)EMA_CHAIN_CAP * sizeof(ChainRec));
	s.cpos = (int64_t *)take((size_t)EMA_CHAIN_CAP * 8);
	s.cord = (int32_t *)take((size_t)EMA_CHAIN_CAP * 4);
	s.skey = (uint64_t *)take((size_t)EMA_CHAIN_CAP * 8);
	s.kept = (int32_t *)take((size_t)EMA_CHAIN_CAP * 4);
	s.av = (DevReg *)take((size_t)EMA_AV_CAP * sizeof(DevReg));
	s.av_tmp = (DevReg *)take((size_t)EMA_AV_CAP * sizeof(DevReg));
	s.rkeys = (uint64_t *)take((size_t)EMA_AV_CAP * 8);
	s.ivs = (Intv *)take((size_t)EMA_INTV_CAP * sizeof(Intv));
	return s;
}

namespace {

__device__ __forceinline__ int cal_max_gap(const DevOpts &o, int qlen)
{
	const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
	const int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < o.w << 1 ? l : o.w << 1;
}

// first index in cpos[0..n) with cpos >= key; 64-ary search, one global load per lane and round
__device__ __forceinline__ int lower_bound_pos(const int64_t *cpos, int n, int64_t key)
{
	const int lane = (int)ema_lane();
	int lo = 0, len = n;
	while (len > EMA_WAVE) {
		const int stride = (len + EMA_WAVE - 1) >> 6;
		int pos = (lane + 1) * stride - 1;
		if (pos > len - 1) pos = len - 1;
		const int cnt = __popcll(__ballot(cpos[lo + pos] < key));
		if (cnt == EMA_WAVE) return lo + len;
		const int start = cnt * stride;
		const int rest = len - start;
		lo += start;
		len = stride < rest ? stride : rest;
	}
	return lo + __popcll(__ballot(lane < len && cpos[lo + (lane < len ? lane : 0)] < key));
}

// Everything this wavefront has in flight -- LDS instructions, FLAT instructions that end in LDS, global loads and stores --
// completes before anything that follows is issued.  The kernel touches the same LDS bytes through typed LDS pointers and
// through generic pointers kept in its table structs, and those two kinds of instruction take different routes to LDS: between
// the phases of a read (whose tables overlay one another) the routes are drained.
__device__ __forceinline__ void ema_phase_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_wave_barrier();
}

struct ChainBuild {
	AlignSlab sl;
	int n_chain, n_seed, status;
};

// mem_chain's loop body for one seed: find the closest chain at or below rbeg, merge into it or open a new one.
// All lanes take the same decisions from the same reads; lane 0 alone edits the tables, after an ema_wave_sync().
__device__ inline void chain_insert(const DevOpts &o, int64_t l_pac, ChainBuild &cb, int64_t rbeg, int qbeg, int len, int rid)
{
	const int lane = (int)ema_lane();
	const bool leader = lane == 0;
	int at = 0, lower = -1;
	if (cb.n_chain) {
		const int lo = lower_bound_pos(cb.sl.cpos, cb.n_chain, rbeg);
		if (lo < cb.n_chain && ema_uni(cb.sl.cpos[lo]) == rbeg) { lower = ema_uni(cb.sl.cord[lo]); at = lo + 1; }
		else if (lo > 0) { lower = ema_uni(cb.sl.cord[lo - 1]); at = lo; }
	}
	if (lower >= 0) {   // test_and_merge
		ChainRec c = ema_uni(cb.sl.chains[lower]);
		const int64_t qend = c.l_qbeg + c.l_len, rend = c.l_rbeg + c.l_len;
		bool open_new = false;
		if (rid != c.rid) open_new = true;
		else if (qbeg >= c.f_qbeg && qbeg + len <= qend && rbeg >= c.f_rbeg && rbeg + len <= rend) return;      // contained: absorbed
		else if ((c.l_rbeg < l_pac || c.f_rbeg < l_pac) && rbeg >= l_pac) open_new = true;                      // other strand
		else {
			const int64_t x = qbeg - c.l_qbeg, y = rbeg - c.l_rbeg;
			if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.l_len < o.max_chain_gap && y - c.l_len < o.max_chain_gap) {
				if (cb.n_seed >= EMA_SEED_CAP) { cb.status |= EMA_ST_SEED_OVERFLOW; return; }
				const int id = cb.n_seed++;
				ema_wave_sync();
				if (leader) {
					SeedRec s; s.rbeg = rbeg; s.qbeg = qbeg; s.len = len; s.next = -1; s.pad = 0;
					cb.sl.seeds[id] = s;
					cb.sl.seeds[c.last_seed].next = id;
					c.last_seed = id; c.l_rbeg = rbeg; c.l_qbeg = qbeg; c.l_len = len; ++c.n;
					cb.sl.chains[lower] = c;
				}
				return;
			}
			open_new = true;
		}
		(void)open_new;
	}
	// open a new chain right after the element the lookup returned
	if (cb.n_chain >= EMA_CHAIN_CAP) { cb.status |= EMA_ST_CHAIN_OVERFLOW; return; }
	if (cb.n_seed >= EMA_SEED_CAP) { cb.status |= EMA_ST_SEED_OVERFLOW; return; }
	for (int hi = cb.n_chain; hi > at; hi -= EMA_WAVE) {      // shift [at, n) up by one, top chunk first
		const int idx = hi - 1 - lane;
		int64_t v = 0; int32_t id = 0;
		if (idx >= at) { v = cb.sl.cpos[idx]; id = cb.sl.cord[idx]; }
		ema_wave_sync();
		if (idx >= at) { cb.sl.cpos[idx + 1] = v; cb.sl.cord[idx + 1] = id; }
	}
	const int sid = cb.n_seed++, cid = cb.n_chain++;
	ema_wave_sync();
	if (leader) {
		SeedRec s; s.rbeg = rbeg; s.qbeg = qbeg; s.len = len; s.next = -1; s.pad = 0;
		cb.sl.seeds[sid] = s;
		ChainRec c;
		c.pos = rbeg; c.f_rbeg = c.l_rbeg = rbeg; c.f_qbeg = c.l_qbeg = qbeg; c.l_len = len;
		c.rid = rid; c.n = 1; c.first_seed = c.last_seed = sid; c.w = 0; c.kept = 0; c.first = -1;
		cb.sl.chains[cid] = c;
		cb.sl.cpos[at] = rbeg; cb.sl.cord[at] = cid;
	}
}

// ---- the medium layout's chaining (see the head of the file): every decision of an insertion is taken from LDS --
// positions, ids and a 12-byte summary of each chain's mutable end -- and what goes to the slab are stores nobody waits for
// (the seed, the link from the chain's last seed, a new chain's record).  LDS is addressed as LDS (ds_read / ds_write): a
// generic-pointer access would be a FLAT instruction, which has to wait for every global store before it.
// Summary of a chain, three words by chain id: what test_and_merge looks at, relative to the chain's position (= rbeg of its
// first seed):  [0] rbeg of the last seed - position   [1] rid | last_seed << 16   [2] f_qbeg | l_qbeg << 8 | l_len << 16
struct MedTables {
	EMA_LDS int64_t *cpos;   // EMA_MED_CHAINS, ascending
	EMA_LDS int32_t *cord;   // chain ids in the same order
	EMA_LDS uint32_t *csm;   // 3 x EMA_MED_CHAINS
};

__device__ __forceinline__ int med_lower_bound(EMA_LDS const int64_t *cpos, int n, int64_t key)
{
	const int lane = (int)ema_lane();
	int lo = 0, len = n;
	while (len > EMA_WAVE) {
		const int stride = (len + EMA_WAVE - 1) >> 6;
		int pos = (lane + 1) * stride - 1;
		if (pos > len - 1) pos = len - 1;
		const int cnt = __popcll(__ballot(cpos[lo + pos] < key));
		if (cnt == EMA_WAVE) return lo + len;
		const int start = cnt * stride;
		const int rest = len - start;
		lo += start;
		len = stride < rest ? stride : rest;
	}
	return lo + __popcll(__ballot(lane < len && cpos[lo + (lane < len ? lane : 0)] < key));
}

// chain_insert on the medium layout.  Returns false, having done nothing, when a new chain is needed and the tables are full
// (the caller moves the read to the slab layout and repeats the insertion there).
__device__ inline bool chain_insert_med(const DevOpts &o, int64_t l_pac, ChainBuild &cb, const MedTables &mt, int64_t rbeg, int qbeg, int len, int rid)
{
	const int lane = (int)ema_lane();
	const bool leader = lane == 0;
	int at = 0, lower = -1;
	int64_t pos = 0;
	if (cb.n_chain) {
		// the two table entries around the insertion point in ONE LDS round trip (lane 0: entry lo, lane 1: entry lo - 1), not a
		// dependent read for each of position, id, and position again
		const int lo = med_lower_bound(mt.cpos, cb.n_chain, rbeg);
		const int mine = lo - lane;      // lanes 0 and 1
		int64_t p2 = 0; int id2 = -1;
		if (lane < 2 && mine >= 0 && mine < cb.n_chain) { p2 = mt.cpos[mine]; id2 = mt.cord[mine]; }
		const int64_t p_lo = ema_lane_val(p2, 0), p_lm = ema_lane_val(p2, 1);
		const int id_lo = ema_lane_val(id2, 0), id_lm = ema_lane_val(id2, 1);
		if (lo < cb.n_chain && p_lo == rbeg) { lower = id_lo; at = lo + 1; pos = p_lo; }
		else if (lo > 0) { lower = id_lm; at = lo; pos = p_lm; }
	}
	if (lower >= 0) {   // test_and_merge
		uint32_t mv = 0;      // the chain's three summary words, one per lane
		if (lane < 3) mv = mt.csm[3 * lower + lane];
		const uint32_t m0 = (uint32_t)ema_lane_val((int)mv, 0), m1 = (uint32_t)ema_lane_val((int)mv, 1), m2 = (uint32_t)ema_lane_val((int)mv, 2);
		const int l_delta = (int)m0, c_rid = (int)(m1 & 0xffff), last_seed = (int)(m1 >> 16);
		const int f_qbeg = (int)(m2 & 0xff), l_qbeg = (int)(m2 >> 8 & 0xff), l_len = (int)(m2 >> 16 & 0xff);
		const int64_t f_rbeg = pos, l_rbeg = pos + l_delta;
		const int64_t qend = l_qbeg + l_len, rend = l_rbeg + l_len;
		if (rid != c_rid) { }
		else if (qbeg >= f_qbeg && qbeg + len <= qend && rbeg >= f_rbeg && rbeg + len <= rend) return true;      // contained: absorbed
		else if ((l_rbeg < l_pac || f_rbeg < l_pac) && rbeg >= l_pac) { }                                      // other strand
		else {
			const int64_t x = qbeg - l_qbeg, y = rbeg - l_rbeg;
			if (y >= 0 && x - y <= o.w && y - x <= o.w && x - l_len < o.max_chain_gap && y - l_len < o.max_chain_gap) {
				if (cb.n_seed >= EMA_SEED_CAP) { cb.status |= EMA_ST_SEED_OVERFLOW; return true; }
				const int id = cb.n_seed++;
				if (leader) {
					SeedRec s; s.rbeg = rbeg; s.qbeg = qbeg; s.len = len; s.next = -1; s.pad = 0;
					cb.sl.seeds[id] = s;
					cb.sl.seeds[last_seed].next = id;
					mt.csm[3 * lower] = (uint32_t)(int32_t)(rbeg - pos);
					mt.csm[3 * lower + 1] = (uint32_t)c_rid | (uint32_t)id << 16;
					mt.csm[3 * lower + 2] = (uint32_t)f_qbeg | (uint32_t)qbeg << 8 | (uint32_t)len << 16;
				}
				ema_wave_sync();
				return true;
			}
		}
	}
	// a new chain right after the element the lookup returned
	if (cb.n_chain >= EMA_MED_CHAINS) return false;
	if (cb.n_seed >= EMA_SEED_CAP) { cb.status |= EMA_ST_SEED_OVERFLOW; return true; }
	for (int hi = cb.n_chain; hi > at; hi -= EMA_WAVE) {      // shift [at, n) up by one, top chunk first
		const int idx = hi - 1 - lane;
		int64_t v = 0; int32_t id = 0;
		if (idx >= at) { v = mt.cpos[idx]; id = mt.cord[idx]; }
		ema_wave_sync();
		if (idx >= at) { mt.cpos[idx + 1] = v; mt.cord[idx + 1] = id; }
	}
	const int sid = cb.n_seed++, cid = cb.n_chain++;
	ema_wave_sync();
	if (leader) {
		SeedRec s; s.rbeg = rbeg; s.qbeg = qbeg; s.len = len; s.next = -1; s.pad = 0;
		cb.sl.seeds[sid] = s;
		ChainRec c;
		c.pos = rbeg; c.f_rbeg = c.l_rbeg = rbeg; c.f_qbeg = c.l_qbeg = qbeg; c.l_len = len;
		c.rid = rid; c.n = 1; c.first_seed = c.last_seed = sid; c.w = 0; c.kept = 0; c.first = -1;
		cb.sl.chains[cid] = c;      // l_rbeg, l_qbeg, l_len, last_seed and n are brought up to date by med_settle
		mt.cpos[at] = rbeg; mt.cord[at] = cid;
		mt.csm[3 * cid] = 0;
		mt.csm[3 * cid + 1] = (uint32_t)rid | (uint32_t)sid << 16;
		mt.csm[3 * cid + 2] = (uint32_t)qbeg | (uint32_t)qbeg << 8 | (uint32_t)len << 16;
	}
	ema_wave_sync();
	return true;
}

// ---- [r5] the medium layout's chaining while the read has at most 64 chains: the tables in REGISTERS.  Slot t of the sorted
// position table (position, chain id) lives in lane t, the summary of chain id c in lane c.  The look-up is one ballot, the two
// entries around the insertion point and the chain's summary are lane reads, a merge rewrites one lane's summary, a new chain
// shifts the upper lanes by one with a DPP move: no LDS round trip, no wave-level ordering point -- an insertion was four dependent
// LDS round trips and three of those (profiles/r04_k2_profile.txt: mode 0 spends two fifths of its lifetimes chaining, ~1,500
// clocks per seed occurrence).  Same decisions from the same values as chain_insert_med.  Returns false, having done nothing,
// when a 65th chain is needed: the caller writes the tables to LDS (reg_chains_to_lds) and carries on with chain_insert_med.
struct RegChains { int64_t pos; int32_t id; uint32_t m0, m1, m2; };

__device__ inline bool chain_insert_reg(const DevOpts &o, int64_t l_pac, ChainBuild &cb, RegChains &rc, int64_t rbeg, int qbeg, int len, int rid)
{
	const int lane = (int)ema_lane();
	const bool leader = lane == 0;
	const int n = cb.n_chain;
	int at = 0, lower = -1;
	int64_t pos = 0;
	if (n) {
		const int lo = __popcll(__ballot(lane < n && rc.pos < rbeg));      // first slot with position >= rbeg
		const int64_t p_lo = ema_lane_val(rc.pos, lo < n ? lo : 0), p_lm = ema_lane_val(rc.pos, lo > 0 ? lo - 1 : 0);
		const int id_lo = ema_lane_val(rc.id, lo < n ? lo : 0), id_lm = ema_lane_val(rc.id, lo > 0 ? lo - 1 : 0);
		if (lo < n && p_lo == rbeg) { lower = id_lo; at = lo + 1; pos = p_lo; }
		else if (lo > 0) { lower = id_lm; at = lo; pos = p_lm; }
	}
	if (lower >= 0) {   // test_and_merge
		const uint32_t m0 = (uint32_t)ema_lane_val((int)rc.m0, lower), m1 = (uint32_t)ema_lane_val((int)rc.m1, lower), m2 = (uint32_t)ema_lane_val((int)rc.m2, lower);
		const int l_delta = (int)m0, c_rid = (int)(m1 & 0xffff), last_seed = (int)(m1 >> 16);
		const int f_qbeg = (int)(m2 & 0xff), l_qbeg = (int)(m2 >> 8 & 0xff), l_len = (int)(m2 >> 16 & 0xff);
		const int64_t f_rbeg = pos, l_rbeg = pos + l_delta;
		const int64_t qend = l_qbeg + l_len, rend = l_rbeg + l_len;
		if (rid != c_rid) { }
		else if (qbeg >= f_qbeg && qbeg + len <= qend && rbeg >= f_rbeg && rbeg + len <= rend) return true;      // contained: absorbed
		else if ((l_rbeg < l_pac || f_rbeg < l_pac) && rbeg >= l_pac) { }                                      // other strand
		else {
			const int64_t x = qbeg - l_qbeg, y = rbeg - l_rbeg;
			if (y >= 0 && x - y <= o.w && y - x <= o.w && x - l_len < o.max_chain_gap && y - l_len < o.max_chain_gap) {
				if (cb.n_seed >= EMA_SEED_CAP) { cb.status |= EMA_ST_SEED_OVERFLOW; return true; }
				const int id = cb.n_seed++;
				if (leader) {
					SeedRec s; s.rbeg = rbeg; s.qbeg = qbeg; s.len = len; s.next = -1; s.pad = 0;
					cb.sl.seeds[id] = s;
					cb.sl.seeds[last_seed].next = id;
				}
				if (lane == lower) {
					rc.m0 = (uint32_t)(int32_t)(rbeg - pos);
					rc.m1 = (uint32_t)c_rid | (uint32_t)id << 16;
					rc.m2 = (uint32_t)f_qbeg | (uint32_t)qbeg << 8 | (uint32_t)len << 16;
				}
				return true;
			}
		}
	}
	// a new chain right after the element the look-up returned
	if (n >= EMA_WAVE) return false;
	if (cb.n_seed >= EMA_SEED_CAP) { cb.status |= EMA_ST_SEED_OVERFLOW; return true; }
	{
		const int lo32 = ema_wave_shr1((int)(uint32_t)(uint64_t)rc.pos, 0), hi32 = ema_wave_shr1((int)(uint32_t)((uint64_t)rc.pos >> 32), 0);
		const int up_id = ema_wave_shr1(rc.id, 0);
		const int64_t up_pos = (int64_t)((uint64_t)(uint32_t)hi32 << 32 | (uint32_t)lo32);
		const int cid_new = cb.n_chain;
		if (lane > at) { rc.pos = up_pos; rc.id = up_id; }
		else if (lane == at) { rc.pos = rbeg; rc.id = cid_new; }
	}
	const int sid = cb.n_seed++, cid = cb.n_chain++;
	if (leader) {
		SeedRec s; s.rbeg = rbeg; s.qbeg = qbeg; s.len = len; s.next = -1; s.pad = 0;
		cb.sl.seeds[sid] = s;
		ChainRec c;
		c.pos = rbeg; c.f_rbeg = c.l_rbeg = rbeg; c.f_qbeg = c.l_qbeg = qbeg; c.l_len = len;
		c.rid = rid; c.n = 1; c.first_seed = c.last_seed = sid; c.w = 0; c.kept = 0; c.first = -1;
		cb.sl.chains[cid] = c;      // l_rbeg, l_qbeg, l_len, last_seed and n are brought up to date from the summaries (as for chain_insert_med)
	}
	if (lane == cid) { rc.m0 = 0; rc.m1 = (uint32_t)rid | (uint32_t)sid << 16; rc.m2 = (uint32_t)qbeg | (uint32_t)qbeg << 8 | (uint32_t)len << 16; }
	return true;
}

// the register tables written out as chain_insert_med's LDS tables (the filter, and chaining beyond 64 chains, read those)
__device__ inline void reg_chains_to_lds(const ChainBuild &cb, const RegChains &rc, const MedTables &mt)
{
	const int lane = (int)ema_lane();
	if (lane < cb.n_chain) {
		mt.cpos[lane] = rc.pos; mt.cord[lane] = rc.id;
		mt.csm[3 * lane] = rc.m0; mt.csm[3 * lane + 1] = rc.m1; mt.csm[3 * lane + 2] = rc.m2;
	}
	ema_wave_sync();
}

// mem_chain_weight for one chain (walks its seed list); run lane-parallel over chains
__device__ __forceinline__ int chain_weight(const SeedRec *seeds, int first, int *n_seeds = nullptr)
{
	int64_t end = 0;
	int w = 0, cnt = 0;
	for (int k = first; k >= 0; k = seeds[k].next) {
		const SeedRec s = seeds[k];
		++cnt;
		if (s.qbeg >= end) w += s.len;
		else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
		end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
	}
	const int tmp = w;
	w = 0; end = 0;
	for (int k = first; k >= 0; k = seeds[k].next) {
		const SeedRec s = seeds[k];
		if (s.rbeg >= end) w += s.len;
		else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
		end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
	}
	w = w < tmp ? w : tmp;
	if (n_seeds) *n_seeds = cnt;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

}  // namespace

// one wavefront = one read at a time, reads taken from a shared counter
// intv/n_intv: K1's output (stride opt.intv_cap).  regs: n_reads x opt.reg_cap, n_regs: n_reads.  status is OR-ed.
// MODE 0: the kernel described above, over K2a's todo list (or every read when todo is null); it sets chain-rich reads aside when
// hv.arena is given; 1: K2c, 2: K2d (dev_types.h, HeavyCtl);
// 3: the reads K2a handed over with their chains ready (dev_types.h, HandHdr: *n_todo dense records in `hand`) -- five sixths of
// the reads that reach this kernel.  Without the chaining, the filter and the setting-aside in the same function the register
// allocator has a far easier job, and the path is laid out for few DEPENDENT round trips to memory, which is what a read costs
// here while K1 of another slice keeps the memory system saturated (a round trip then takes several microseconds): records are
// claimed four at a time with their headers, a record's tables and the read's bases arrive together, the windows of all its
// chains are planned at once (lane per chain) and fetched up to four at a time.
// PROF 1: the diagnostic build (phase clocks, per-read log); the product build (0) carries none of its registers.  PROF 2: the
// product build with a handful of clocks kept in scalar registers (EMA_PHASE_PROFILE=3) -- per mode: the wavefronts' lifetimes
// (= the slot-time the launch costs the chip), the clocks inside the extension DPs, record + window set-up, dedup + output --
// because the diagnostic build spills eight times as much as the product and its split says little about the product's.
template <int SMALL, int AVL, int WPS, int MODE, int PROF>
__global__ void __launch_bounds__(256, WPS)
ema_k_align_t(DevIndex ix, DevOpts opt, const uint8_t *__restrict__ bases, const uint32_t *__restrict__ off, int n_reads,
            const int *__restrict__ n_pairs_dev, const int *__restrict__ map,
            const Intv *__restrict__ intv, const int *__restrict__ n_intv, DevReg *__restrict__ regs, int *__restrict__ n_regs,
            int *__restrict__ status, const int *__restrict__ todo, const int *__restrict__ n_todo,
            const uint8_t *__restrict__ hand, uint8_t *__restrict__ slabs, int *__restrict__ counter, int *dbg_arg, unsigned long long *prof_arg,
            HeavyCtl hv)
{
	// The phase clocks and the per-read log exist in the diagnostic build only.  (The watchdog's progress words stay a run-time
	// pointer: with both folded away this kernel faulted on the device -- r02: every other combination passes the suite, the
	// interpreter and AddressSanitizer find nothing -- and until that is understood the build that is tested is the one shipped.)
	unsigned long long *const prof = PROF == 1 ? prof_arg : nullptr;
	// PROF 2: the clocks since the previous mark go to phase k -- 0 claiming a work item, 1 the read and its record (mode 0: the
	// intervals, suffix-array rows, chaining, filter, setting aside), 2 a chain's head and seeds, 3 its window (bounds, fetch, seed
	// order), 4 the per-seed control (cover tests, region records), 5 the extension DPs, 6 dedup / patch and output
	unsigned long long lp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [r5] 7..11: mode 0's phase 1 split -- intervals in order + repetitive fraction, suffix-array rows + contigs, insertions, chain filter, setting aside
	int lp_n_dp = 0, lp_reads = 0;
	const unsigned long long lp_t0 = PROF == 2 ? __builtin_amdgcn_s_memtime() : 0;
	unsigned long long lp_mark = lp_t0;
#define EMA_LP(k) do { if (PROF == 2) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); lp[k] += t_ - lp_mark; lp_mark = t_; } } while (0)
	int *const dbg = dbg_arg;
	// diagnostic phase timing (prof != null): shader-clock ticks per phase, summed over all waves
	unsigned long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
	int phase = 0;
	const unsigned long long t_launch = t_prev;
	// EMA_PHASE_PROFILE=2: one record per read {read, intervals (-1: chains handed over by K2a), seed occurrences, chains, seeds,
	// regions before dedup, extension DPs run, shader clocks / 16} appended to the log whose address sits in prof[31]
	int *rlog = prof ? reinterpret_cast<int *>(prof[31]) : nullptr;
	int n_dp = 0;
#define EMA_PHASE(idx) do { if (prof) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); acc[phase] += t_now - t_prev; t_prev = t_now; phase = (idx); } } while (0)
#define EMA_DBG(stage, val) do { if (dbg && lane == 0) { __hip_atomic_store(dbg + slot * 4 + 1, (stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(dbg + slot * 4 + 2, (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } } while (0)
	__shared__ uint8_t lds_q[4][256];
	__shared__ uint8_t lds_r[4][EMA_RSEQ_CAP];
	__shared__ int lds_stack[4][3 * 70];
	__shared__ __attribute__((aligned(16))) uint8_t lds_small[4][EMA_SMALL_BYTES(SMALL)];
	__shared__ __attribute__((aligned(16))) uint8_t lds_av[4][EMA_AVL_BYTES(AVL)];      // regions of the read while there are few
	// the medium layout needs 4 KB of small-table area, the 2 KB window buffer and 1 KB of region-list area
	constexpr bool MED = EMA_SMALL_BYTES(SMALL) >= 4096 && EMA_RSEQ_CAP >= EMA_MED_CHAINS * 8 && (AVL > 0 && EMA_AVL_BYTES(AVL) >= EMA_MED_CHAINS * 4);
	const int lane = (int)ema_lane();
	const int wib = ema_uni((int)(threadIdx.x >> 6));      // scalar: everything derived from it (slab, LDS table pointers) stays in SGPRs
	const int slot = (int)(blockIdx.x * (blockDim.x >> 6)) + wib;
	uint8_t *query = lds_q[wib];
	uint8_t *rseq = lds_r[wib];
	MedTables mt;      // medium layout while chaining: positions in the window buffer, summaries and ids in the small-table area
	mt.cpos = (EMA_LDS int64_t *)(&lds_r[wib][0]);      // (address-space casts: C style)
	mt.csm = (EMA_LDS uint32_t *)(&lds_small[wib][0]);
	mt.cord = (EMA_LDS int32_t *)(&lds_small[wib][3072]);
	ChainBuild cb;
	const AlignSlab slab = ema_carve_slab(slabs + (size_t)slot * EMA_ALIGN_SLAB_BYTES);
	cb.sl = slab;
	const AlignSlab &sl = cb.sl;
	const int64_t l_pac = ix.l_pac;

	// MODE 3: records claimed four at a time; lane k < 4 keeps record k's header until its turn
	int cl_base = 0, cl_n = 0, cl_i = 0, win_end = 0;
	EmaClaim claim;                        // the other modes
	unsigned long long list_entry = 0;     // this item's entry of the launch's list (todo / tasks / reads)
	HandHdr pf;
	pf.read = pf.n_chn = pf.n_seed = pf.l_query = pf.chain_from = pf.n_av = 0; pf.base_off = 0;
	for (;;) {
		int read = 0;
		constexpr bool handed = MODE == 3;      // K2a already chained and filtered this read (dev_types.h, HandHdr)
		int rec_i = 0;                          // MODE 3: this record's number (its slots in K2x's result table)
		HandHdr hd;                             // MODE 3: this record's header, wave-uniform
		hd.read = hd.n_chn = hd.n_seed = hd.l_query = hd.chain_from = hd.n_av = 0; hd.base_off = 0;
		const uint8_t *hrec = nullptr;
		if (MODE == 3) {
			if (cl_i == cl_n) {
				const int total = *n_todo;
				const int step = ema_claim_step(cl_base, total);      // (dev_common.hpp: singly near the end of the queue)
				int base = 0;
				if (lane == 0) base = atomicAdd(counter, step);
				base = __builtin_amdgcn_readlane(base, 0);
				if (base >= total) break;
				cl_base = base; cl_n = total - base < step ? total - base : step; cl_i = 0;
				if (lane < cl_n) pf = *reinterpret_cast<const HandHdr *>(hand + (size_t)(base + lane) * EMA_HAND_BYTES);
			}
			hd.read = __builtin_amdgcn_readlane(pf.read, cl_i); hd.n_chn = __builtin_amdgcn_readlane(pf.n_chn, cl_i);
			hd.n_seed = __builtin_amdgcn_readlane(pf.n_seed, cl_i); hd.l_query = __builtin_amdgcn_readlane(pf.l_query, cl_i);
			hd.base_off = (uint32_t)__builtin_amdgcn_readlane((int)pf.base_off, cl_i);
			hd.chain_from = __builtin_amdgcn_readlane(pf.chain_from, cl_i); hd.n_av = __builtin_amdgcn_readlane(pf.n_av, cl_i);
			hrec = hand + (size_t)(cl_base + cl_i) * EMA_HAND_BYTES;
			rec_i = cl_base + cl_i;
			++cl_i;
			read = hd.read;
		} else {      // (dev_common.hpp, EmaClaim: four items and their list entries at a time)
			if (MODE == 1) { const int cap = hv.tasks_cap, n = *hv.n_tasks; read = ema_claim_next(claim, counter, n < cap ? n : cap, hv.tasks, list_entry); }
			else if (MODE == 2) { const int cap = hv.reads_cap, n = *hv.n_reads; read = ema_claim_next(claim, counter, n < cap ? n : cap, hv.reads, list_entry); }
			else {
				int t = 0;
				read = ema_claim_next(claim, counter, todo ? *n_todo : ema_work_count(n_reads, n_pairs_dev, 2), todo, t);
				if (todo) list_entry = (unsigned long long)(unsigned)t;
			}
			if (read < 0) {
				if (MODE == 0 && prof && lane == 0 && opt.reg_cap <= EMA_REG_LEAN) atomicMin(prof + 27, (unsigned long long)__builtin_amdgcn_s_memtime());      // the queue ran dry (lean tier)
				break;
			}
		}
		EMA_LP(0);
		if (PROF == 2) ++lp_reads;
		const uint8_t *rec = nullptr;      // MODE 1, 2: the record of the read set aside
		int task_chain = -1;               // MODE 1: the chain of this task (index in filtered order)
		if (MODE == 1) {
			const unsigned long long t = list_entry;
			if (t == ~0ULL) continue;      // a claim K2b gave back
			rec = hv.arena + (size_t)(t >> 32) * 64; task_chain = (int)(uint32_t)t;
			read = ema_uni(reinterpret_cast<const HeavyHdr *>(rec)->read);
		} else if (MODE == 2) {
			const unsigned long long t = list_entry;
			if (t == ~0ULL) continue;
			rec = hv.arena + (size_t)t;
			read = ema_uni(reinterpret_cast<const HeavyHdr *>(rec)->read);
		} else if (MODE == 0 && todo) read = (int)(unsigned)list_entry;
		if (dbg && lane == 0) __hip_atomic_store(dbg + slot * 4, read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		const unsigned long long t_read = rlog ? __builtin_amdgcn_s_memtime() : 0;
		int log_iv = -1, log_occ = 0;
		n_dp = 0;
		EMA_DBG(1, 0);
		EMA_PHASE(6);      // 6: the read and (handed reads) K2a's record arrive
		if (MODE == 0 && ema_uni(status[read])) {      // over a capacity in K1: the pair is redone by the full-capacity tier
			if (lane == 0) n_regs[read] = 0;      // (K2a hands over only reads whose status is clear)
			EMA_DBG(9, 0);
			continue;
		}
		int l_query = 0;
		uint8_t qv[4] = {0, 0, 0, 0};      // MODE 3: the read's bases on their way while the record's tables are
		if (MODE == 3) {
			l_query = hd.l_query;
			const uint8_t *src = bases + hd.base_off;
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int i = lane + k * EMA_WAVE; if (i < l_query) qv[k] = src[i]; }
		} else {
			ema_phase_fence();      // the previous read's stores have landed
			const int in_read = ema_uni(ema_in_read(map, read));
			l_query = ema_uni((int)(off[in_read + 1] - off[in_read]));
			for (int i = lane; i < l_query; i += EMA_WAVE) query[i] = bases[off[in_read] + i];
			ema_wave_sync();
		}
		float frac_rep = 0.f;
		int n_chn = 0, n_keep = 0;
		cb.n_chain = 0; cb.n_seed = 0; cb.status = 0;
		bool med = false;         // the medium layout (see the head of the file) is in force
		bool reg_mode = false;    // ... with its tables in registers: at most 64 chains so far (chain_insert_reg)
		RegChains rc; rc.pos = 0; rc.id = 0; rc.m0 = rc.m1 = rc.m2 = 0;
		const int32_t *hv_first = nullptr;      // MODE 1, 2: per chain (filtered order) the first slot of its seeds in the tables below
		SeedRec *hv_cs = nullptr; DevReg *hv_res = nullptr; uint8_t *hv_valid = nullptr;
		if (MODE == 1 || MODE == 2) {
			// the tables as K2b left them: read where they are
			const HeavyHdr *h = reinterpret_cast<const HeavyHdr *>(rec);
			cb.sl = slab;
			cb.sl.skey = reinterpret_cast<uint64_t *>(const_cast<uint8_t *>(rec) + ema_uni(h->off_skey));
			cb.sl.chains = reinterpret_cast<ChainRec *>(const_cast<uint8_t *>(rec) + ema_uni(h->off_chains));
			cb.sl.seeds = reinterpret_cast<SeedRec *>(const_cast<uint8_t *>(rec) + ema_uni(h->off_seeds));
			hv_first = reinterpret_cast<const int32_t *>(rec + ema_uni(h->off_first));
			hv_cs = reinterpret_cast<SeedRec *>(const_cast<uint8_t *>(rec) + ema_uni(h->off_cs));
			hv_res = reinterpret_cast<DevReg *>(const_cast<uint8_t *>(rec) + ema_uni(h->off_res));
			hv_valid = const_cast<uint8_t *>(rec) + ema_uni(h->off_valid);
			n_chn = ema_uni(h->n_chn); n_keep = n_chn;
			cb.n_chain = ema_uni(h->n_chain); cb.n_seed = ema_uni(h->n_seed);
			cb.status = MODE == 2 ? ema_uni(h->status) : 0;
			frac_rep = ema_uni(h->frac_rep);
		} else if (MODE == 3) {
			// small tables in LDS, filled from K2a's record: one entry per lane (EMA_HAND_SEEDS <= 64), all loads in flight together
			// with the read's bases; then the point where the previous read's stores must have landed; then LDS
			cb.sl = slab;
			ema_small_tables<SMALL>(cb.sl, lds_small[wib]);
			n_chn = hd.n_chn;
			const int n_sd = hd.n_seed;
			const uint64_t *hk = reinterpret_cast<const uint64_t *>(hrec + sizeof(HandHdr));
			const ChainRec *hc = reinterpret_cast<const ChainRec *>(hrec + sizeof(HandHdr) + EMA_HAND_SEEDS * 8);
			const SeedRec *hs = reinterpret_cast<const SeedRec *>(hrec + sizeof(HandHdr) + EMA_HAND_SEEDS * (8 + sizeof(ChainRec)));
			const DevReg *hr = reinterpret_cast<const DevReg *>(hrec + sizeof(HandHdr) + EMA_HAND_SEEDS * (8 + sizeof(ChainRec) + sizeof(SeedRec)));
			uint64_t kv = 0;
			ChainRec cv; SeedRec sv; DevReg rv;
			if (lane < n_chn) { kv = hk[lane]; cv = hc[lane]; }
			if (lane < n_sd) sv = hs[lane];
			if (lane < hd.n_av) rv = hr[lane];      // the regions of the chains K2a finished (below: into the region list)
			ema_phase_fence();
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int i = lane + k * EMA_WAVE; if (i < l_query) query[i] = qv[k]; }
			if (lane < n_chn) { sl.skey[lane] = kv; sl.chains[lane] = cv; }
			if (lane < n_sd) sl.seeds[lane] = sv;
			if (lane < hd.n_av) { if (AVL > 0 && hd.n_av <= AVL) reinterpret_cast<DevReg *>(lds_av[wib])[lane] = rv; else slab.av[lane] = rv; }
			cb.n_chain = n_chn; cb.n_seed = n_sd;
			n_keep = n_chn;
			ema_wave_sync();
			// The windows of ALL chains, planned at once, one lane per chain (filtered order): the bounds over the chain's seeds, the
			// strand junction, the contig's ends (the chain's rid IS the contig bns_fetch_seq would look up from its first seed).
			// Start, length and -- once fetched -- place in the window buffer sit in the tables chaining no longer needs.
			if (lane < n_chn) {
				const ChainRec c = sl.chains[(int)(uint32_t)kv];
				int64_t r0 = 0;
				int wl = 0;
				if (c.kept != 0 && lane >= hd.chain_from) {      // (the chains before chain_from are K2a's: done)
					int64_t rmax0 = l_pac << 1, rmax1 = 0;
					int k = c.first_seed;
					for (int t = 0; t < c.n; ++t) {
						const SeedRec s = sl.seeds[k];
						const int64_t b = s.rbeg - (s.qbeg + cal_max_gap(opt, s.qbeg));
						const int tail = l_query - s.qbeg - s.len;
						const int64_t e = s.rbeg + s.len + (tail + cal_max_gap(opt, tail));
						rmax0 = rmax0 < b ? rmax0 : b;
						rmax1 = rmax1 > e ? rmax1 : e;
						k = s.next;
					}
					rmax0 = rmax0 > 0 ? rmax0 : 0;
					rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
					if (rmax0 < l_pac && l_pac < rmax1) {
						if (c.f_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
					}
					ema_clamp_window_rid(ix, rmax0, c.rid, c.f_rbeg >= l_pac, rmax1);
					const int64_t d = rmax1 - rmax0;
					r0 = rmax0; wl = d > (1 << 20) ? 1 << 20 : d < -(1 << 20) ? -(1 << 20) : (int)d;
				}
				sl.cpos[lane] = r0; sl.cord[lane] = wl;
			}
			win_end = 0;
			ema_wave_sync();
		} else {
		EMA_PHASE(10);      // 10: the intervals in order, the repetitive fraction
		const int n_iv = ema_uni(n_intv[read]);
		const Intv *iv = slab.ivs;
		{   // K1 delivers the intervals in discovery order; mem_collect_intv ends with a sort on (start, end).
			// Entries with equal keys are identical, so ranking each entry (ties by position) gives THE order.
			const Intv *raw = intv + (size_t)read * opt.intv_cap;
			for (int i = lane; i < n_iv; i += EMA_WAVE) {
				const Intv mine = raw[i];
				int rank = 0;
				for (int k = 0; k < n_iv; ++k) {
					const uint64_t other = raw[k].info;
					rank += other < mine.info || (other == mine.info && k < i);
				}
				slab.ivs[rank] = mine;
			}
			ema_wave_sync();
		}

		// ---------------- mem_chain: frac_rep, seed occurrences, chaining ----------------
		int l_rep = 0;
		{
			// (64 intervals at a time, one per lane: the occurrence total is a wave sum, and the merge of the repetitive intervals'
			// query spans -- the only part that is sequential -- walks the few lanes that hold one.  As a loop over the intervals every
			// iteration waited for its own load from the slab.)
			int b = 0, e = 0;
			int64_t tot_occ = 0;      // seed occurrences the chaining loop below will look up
			for (int base = 0; base < n_iv; base += EMA_WAVE) {
				const int i = base + lane;
				int occ = 0, sb_v = 0, se_v = 0;
				bool rep = false;
				if (i < n_iv) {
					const Intv p = iv[i];
					sb_v = (int)(p.info >> 32); se_v = (int)(uint32_t)p.info;
					rep = p.x2 > (uint64_t)opt.max_occ;
					occ = rep ? opt.max_occ : (int)p.x2;
				}
				tot_occ += ema_wave_sum(occ);
				for (unsigned long long todo_ = __ballot(rep); todo_; todo_ &= todo_ - 1) {
					const int l = __ffsll((long long)todo_) - 1;
					const int sb = ema_lane_val(sb_v, l), se = ema_lane_val(se_v, l);
					if (sb > e) { l_rep += e - b; b = sb; e = se; }
					else e = e > se ? e : se;
				}
			}
			l_rep += e - b;
			cb.sl = slab;
			if (ema_uni(tot_occ <= SMALL)) ema_small_tables<SMALL>(cb.sl, lds_small[wib]);
			else if (MED && ix.n_seqs <= 0xffff) {      // (the summaries keep a contig id in 16 bits)
				med = true;
				reg_mode = EMA_CHAIN_REGS != 0;      // the first 64 chains' tables in registers (chain_insert_reg)
				// The medium tables are written with LDS instructions; the same bytes were last written -- by the previous read's
				// extension phase: window buffer, seed copies, region list -- through generic pointers, i.e. FLAT instructions, which
				// take the longer way to LDS.  An older FLAT store must not land on top of a younger LDS store: everything in flight
				// completes first.
				ema_phase_fence();
			}
			log_iv = n_iv; log_occ = (int)(tot_occ < (1 << 30) ? tot_occ : (1 << 30));
		}
		frac_rep = (float)l_rep / (float)l_query;
		EMA_LP(7);
		// The seed occurrences in mem_chain's order -- interval by interval, k = 0, step, ... within one -- taken SIXTY-FOUR AT A TIME
		// ACROSS intervals: one round trip for the suffix-array rows of a whole batch and one for their contigs, where a read from a
		// repeat family (dozens of intervals with a handful of occurrences each) paid those round trips per interval.  Lane L keeps
		// interval L of the current chunk of 64 intervals in registers; a batch is assembled from them with scalar reads.
		for (int iv_base = 0; iv_base < n_iv; iv_base += EMA_WAVE) {
			const int chunk_n = n_iv - iv_base < EMA_WAVE ? n_iv - iv_base : EMA_WAVE;
			uint64_t my_x0 = 0;
			int64_t my_step = 1;
			int my_nocc = 0, my_qbeg = 0, my_slen = 0;
			if (lane < chunk_n) {
				const Intv p = iv[iv_base + lane];
				my_qbeg = (int)(p.info >> 32); my_slen = (int)((uint32_t)p.info - (uint32_t)(p.info >> 32));
				my_step = p.x2 > (uint64_t)opt.max_occ ? (int64_t)(p.x2 / (uint64_t)opt.max_occ) : 1;
				int64_t n_occ = ((int64_t)p.x2 + my_step - 1) / my_step;        // k = 0, step, ... < size
				if (n_occ > opt.max_occ) n_occ = opt.max_occ;
				my_nocc = (int)n_occ; my_x0 = p.x0 | (p.x1 == EMA_INTV_BYPOS ? EMA_INTV_BYPOS : 0);      // (an interval K1 hands over by position: its one occurrence's place, marked)
			}
			int ci = 0, ck = 0;      // next occurrence: number ck of interval ci of the chunk
			while (ci < chunk_n) {
				EMA_DBG(2, iv_base + ci);
				EMA_PHASE(11);      // 11: suffix-array rows and contig ids of up to 64 occurrences
				int filled = 0, a_q = 0, a_l = 0;
				uint64_t a_row = 0;
				while (filled < EMA_WAVE && ci < chunk_n) {
					const int n = ema_lane_val(my_nocc, ci);
					const int take = n - ck < EMA_WAVE - filled ? n - ck : EMA_WAVE - filled;
					const uint64_t x0 = ema_lane_val(my_x0, ci);
					const int64_t step = ema_lane_val(my_step, ci);
					const int q = ema_lane_val(my_qbeg, ci), l = ema_lane_val(my_slen, ci);
					if (lane >= filled && lane < filled + take) { a_row = x0 + (uint64_t)((int64_t)(ck + lane - filled) * step); a_q = q; a_l = l; }
					filled += take; ck += take;
					if (ck >= n) { ++ci; ck = 0; }
				}
				int64_t rbeg = 0; int rid = -1;
				if (lane < filled) {      // consecutive suffix-array rows within an interval (step 1) -> coalesced loads
					rbeg = (a_row & EMA_INTV_BYPOS) ? (int64_t)(a_row & ~EMA_INTV_BYPOS) : (int64_t)ema_sa(ix, a_row);
					rid = ema_intv2rid(ix, rbeg, rbeg + a_l);
				}
				EMA_PHASE(1);      // 1: the insertions
				EMA_LP(8);
				for (int t = 0; t < filled; ++t) {
					const int64_t rb = ema_lane_val(rbeg, t);
					const int rd = ema_lane_val(rid, t);
					if (rd < 0) continue;
					const int qbeg = ema_lane_val(a_q, t), slen = ema_lane_val(a_l, t);
					if (MED && med) {
						if (reg_mode) {
							if (chain_insert_reg(opt, l_pac, cb, rc, rb, qbeg, slen, rd)) continue;
							reg_chains_to_lds(cb, rc, mt);      // a 65th chain: the tables move to LDS
							reg_mode = false;
						}
						if (chain_insert_med(opt, l_pac, cb, mt, rb, qbeg, slen, rd)) continue;
						// EMA_MED_CHAINS chains and one more to open: the chain records are brought up to date, the position table moves to
						// the slab, and the read carries on in the slab layout
						ema_wave_sync();
#ifdef EMA_EMU_TRACE
						if (lane == 0) fprintf(stderr, "K2b medium layout outgrown at %d chains\n", cb.n_chain);
#endif
						for (int k = lane; k < cb.n_chain; k += EMA_WAVE) {
							const int id = mt.cord[k];
							const int64_t pos = mt.cpos[k];
							const uint32_t m0 = mt.csm[3 * id], m1 = mt.csm[3 * id + 1], m2 = mt.csm[3 * id + 2];
							int cnt = 0;
							(void)chain_weight(sl.seeds, sl.chains[id].first_seed, &cnt);
							sl.chains[id].l_rbeg = pos + (int32_t)m0; sl.chains[id].l_qbeg = (int)(m2 >> 8 & 0xff); sl.chains[id].l_len = (int)(m2 >> 16 & 0xff);
							sl.chains[id].last_seed = (int)(m1 >> 16); sl.chains[id].n = cnt;
							slab.cpos[k] = pos; slab.cord[k] = id;
						}
						med = false;
						ema_wave_sync();
					}
					chain_insert(opt, l_pac, cb, rb, qbeg, slen, rd);
				}
				EMA_LP(9);
			}
		}

		if (MED && med && reg_mode) { reg_chains_to_lds(cb, rc, mt); reg_mode = false; }      // (the filter reads the LDS tables)
		// ---------------- mem_chain_flt ----------------
		n_chn = cb.n_chain; n_keep = 0;
		EMA_DBG(3, n_chn);
		EMA_PHASE(2);
		if (MED && med && n_chn > 0) {
#ifdef EMA_EMU_TRACE
			if (lane == 0) fprintf(stderr, "K2b medium layout: read %d, %d chains, %d seeds\n", read, n_chn, cb.n_seed);
#endif
			// the filter on LDS: keys, per-chain summaries (indexed by chain id), kept list.  These overlay the chaining tables, so
			// every lane first takes its (up to four) chains' entries into registers.
			uint64_t *skey = reinterpret_cast<uint64_t *>(lds_small[wib] + 1024);
			uint64_t *csum = reinterpret_cast<uint64_t *>(rseq);      // w << 32 | ALT contig << 31 | end << 16 | beg << 8 | kept flag
			uint16_t *kept = reinterpret_cast<uint16_t *>(lds_av[wib]);
			int16_t *firstk = reinterpret_cast<int16_t *>(lds_av[wib] + 2 * EMA_MED_CHAINS);
			int64_t my_pos[EMA_MED_CHAINS / EMA_WAVE];
			int my_id[EMA_MED_CHAINS / EMA_WAVE];
			uint32_t my_m0[EMA_MED_CHAINS / EMA_WAVE], my_m1[EMA_MED_CHAINS / EMA_WAVE], my_m2[EMA_MED_CHAINS / EMA_WAVE];
#pragma unroll
			for (int r = 0; r < EMA_MED_CHAINS / EMA_WAVE; ++r) {
				const int i = lane + r * EMA_WAVE;
				my_id[r] = 0; my_pos[r] = 0; my_m0[r] = my_m1[r] = my_m2[r] = 0;
				if (i < n_chn) {
					const int id = mt.cord[i];
					my_id[r] = id; my_pos[r] = mt.cpos[i];
					my_m0[r] = mt.csm[3 * id]; my_m1[r] = mt.csm[3 * id + 1]; my_m2[r] = mt.csm[3 * id + 2];
				}
			}
			ema_wave_sync();
#pragma unroll
			for (int r = 0; r < EMA_MED_CHAINS / EMA_WAVE; ++r) {      // weights, chains taken in position order; records brought up to date
				const int i = lane + r * EMA_WAVE;
				if (i < n_chn) {
					const int id = my_id[r];
					int cnt = 0;
					const int w = chain_weight(sl.seeds, sl.chains[id].first_seed, &cnt);
					const int f_qbeg = (int)(my_m2[r] & 0xff), l_qbeg = (int)(my_m2[r] >> 8 & 0xff), l_len = (int)(my_m2[r] >> 16 & 0xff);
					sl.chains[id].l_rbeg = my_pos[r] + (int32_t)my_m0[r]; sl.chains[id].l_qbeg = l_qbeg; sl.chains[id].l_len = l_len;
					sl.chains[id].last_seed = (int)(my_m1[r] >> 16); sl.chains[id].n = cnt; sl.chains[id].w = w;
					skey[i] = (uint64_t)(uint32_t)w << 32 | (uint32_t)id;
					csum[id] = (uint64_t)(uint32_t)w << 32 | (uint64_t)(uint32_t)ema_ctg_alt(ix, (int)(my_m1[r] & 0xffff)) << 31 |
					           (uint64_t)(uint32_t)(l_qbeg + l_len) << 16 | (uint64_t)(uint32_t)f_qbeg << 8;
				}
			}
			ema_wave_sync();
			// (ks_introsort by the whole wavefront, dev_sort.hpp: the same array, ties included; the stoppers' ranks go to the first
			// kilobyte of the small-table area, whose summaries every lane took into registers above)
			if (EMA_SORT_WAVE) ema_introsort_wave(skey, n_chn, [](uint64_t x, uint64_t y) { return (x >> 32) > (y >> 32); }, lds_stack[wib], reinterpret_cast<uint16_t *>(lds_small[wib]));
			else if (lane == 0) ema_introsort(skey, n_chn, [](uint64_t x, uint64_t y) { return (x >> 32) > (y >> 32); }, lds_stack[wib]);
			ema_wave_sync();
			EMA_DBG(4, n_chn);
			uint8_t *flag = reinterpret_cast<uint8_t *>(csum);      // byte 0 of a summary
			int n_kept = 1;
			if (lane == 0) { flag[(size_t)(uint32_t)skey[0] << 3] = 3; kept[0] = 0; firstk[0] = -1; }
			ema_wave_sync();
			for (int i = 1; i < n_chn; ++i) {
				const int ci = ema_uni((int)(uint32_t)skey[i]);
				const uint64_t s_i = ema_uni(csum[ci]);
				const int beg_i = (int)(s_i >> 8 & 0xff), end_i = (int)(s_i >> 16 & 0x7fff), w_i = (int)(s_i >> 32), alt_i = (int)(s_i >> 31 & 1);
				bool large_ovlp = false, dropped = false;
				for (int base = 0; base < n_kept && !dropped; base += EMA_WAVE) {
					const int k = base + lane;
					const bool valid = k < n_kept;
					const uint64_t s_j = valid ? csum[(uint32_t)skey[kept[k]]] : s_i;
					const int beg_j = (int)(s_j >> 8 & 0xff), end_j = (int)(s_j >> 16 & 0x7fff), w_j = (int)(s_j >> 32), alt_j = (int)(s_j >> 31 & 1);
					const int b_max = beg_j > beg_i ? beg_j : beg_i;
					const int e_min = end_j < end_i ? end_j : end_i;
					bool ovlp = false, drop = false;
					if (valid && e_min > b_max && (!alt_j || alt_i)) {      // a kept ALT chain does not shadow a primary one
						const int li = end_i - beg_i, lj = end_j - beg_j;
						const int min_l = li < lj ? li : lj;
						if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
							ovlp = true;
							if ((float)w_i < (float)w_j * opt.drop_ratio && w_j - w_i >= opt.min_seed_len << 1) drop = true;
						}
					}
					const unsigned long long bd = __ballot(drop);
					const int first_drop = bd ? __ffsll((long long)bd) - 1 : EMA_WAVE;
					const bool counted = ovlp && lane <= first_drop;
					if (counted && firstk[k] < 0) firstk[k] = (int16_t)i;
					if (__ballot(counted)) large_ovlp = true;
					if (bd) dropped = true;
				}
				if (!dropped) {
					if (lane == 0) { kept[n_kept] = (uint16_t)i; firstk[n_kept] = -1; flag[(size_t)ci << 3] = large_ovlp ? 2 : 3; }
					++n_kept;
					ema_wave_sync();
				}
			}
			ema_wave_sync();
			for (int k = lane; k < n_kept; k += EMA_WAVE) {
				const int f = firstk[k];
				if (f >= 0) flag[(size_t)(uint32_t)skey[f] << 3] = 1;
			}
			ema_wave_sync();
			for (int id = lane; id < n_chn; id += EMA_WAVE) sl.chains[id].kept = flag[(size_t)id << 3];      // back to the records
			cb.sl.skey = skey;
			n_keep = n_chn;
			ema_wave_sync();
		} else if (n_chn > 0) {
			for (int i = lane; i < n_chn; i += EMA_WAVE) {      // weights, chains taken in position order
				const int id = sl.cord[i];
				const int w = chain_weight(sl.seeds, sl.chains[id].first_seed);
				sl.chains[id].w = w;
				sl.skey[i] = (uint64_t)(uint32_t)w << 32 | (uint32_t)id;
			}
			// min_chain_weight is 0 on this path: nothing is dropped before the sort
			ema_wave_sync();
			if (lane == 0) ema_introsort(sl.skey, n_chn, [](uint64_t x, uint64_t y) { return (x >> 32) > (y >> 32); }, lds_stack[wib]);
			ema_wave_sync();
			EMA_DBG(4, n_chn);
			int n_kept = 0;
			{
				const int id0 = ema_uni((int)(uint32_t)sl.skey[0]);
				if (lane == 0) { sl.chains[id0].kept = 3; sl.kept[0] = 0; }
				n_kept = 1;
			}
			for (int i = 1; i < n_chn; ++i) {
				const int ci = ema_uni((int)(uint32_t)sl.skey[i]);
				const ChainRec a_i = ema_uni(sl.chains[ci]);
				const int beg_i = a_i.f_qbeg, end_i = a_i.l_qbeg + a_i.l_len;
				bool large_ovlp = false, dropped = false;
				for (int base = 0; base < n_kept && !dropped; base += EMA_WAVE) {
					const int k = base + lane;
					const bool valid = k < n_kept;
					const int cj = valid ? (int)(uint32_t)sl.skey[sl.kept[k]] : ci;
					const ChainRec a_j = sl.chains[cj];
					const int beg_j = a_j.f_qbeg, end_j = a_j.l_qbeg + a_j.l_len;
					const int b_max = beg_j > beg_i ? beg_j : beg_i;
					const int e_min = end_j < end_i ? end_j : end_i;
					bool ovlp = false, drop = false;
					if (valid && e_min > b_max && (!ema_ctg_alt(ix, a_j.rid) || ema_ctg_alt(ix, a_i.rid))) {     // a kept ALT chain does not shadow a primary one
						const int li = end_i - beg_i, lj = end_j - beg_j;
						const int min_l = li < lj ? li : lj;
						if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
							ovlp = true;
							if ((float)a_i.w < (float)a_j.w * opt.drop_ratio && a_j.w - a_i.w >= opt.min_seed_len << 1) drop = true;
						}
					}
					const unsigned long long bd = __ballot(drop);
					const int first_drop = bd ? __ffsll((long long)bd) - 1 : EMA_WAVE;
					const bool counted = ovlp && lane <= first_drop;
					if (counted && a_j.first < 0) sl.chains[cj].first = i;     // first shadowed chain of a kept chain
					if (__ballot(counted)) large_ovlp = true;
					if (bd) dropped = true;
				}
				if (!dropped) {
					if (lane == 0) { sl.kept[n_kept] = i; sl.chains[ci].kept = large_ovlp ? 2 : 3; }
					++n_kept;
				}
			}
			ema_wave_sync();
			for (int k = lane; k < n_kept; k += EMA_WAVE) {
				const int cj = (int)(uint32_t)sl.skey[sl.kept[k]];
				const int f = sl.chains[cj].first;
				if (f >= 0) sl.chains[(int)(uint32_t)sl.skey[f]].kept = 1;
			}
			ema_wave_sync();
			// max_chain_extend is 1<<30 on this path: the cap on kept=1/2 chains never triggers
			n_keep = n_chn;
		}

		}

		if (MODE == 0) EMA_LP(10);
		// ---------------- a chain-rich read is set aside for K2c / K2d (dev_types.h, HeavyCtl) ----------------
		if (MODE == 0 && hv.arena && n_keep >= hv.min_chains) {
			int n_ext = 0, tot = 0;      // chains to extend, their seeds
			for (int base = 0; base < n_keep; base += EMA_WAVE) {
				const int i = base + lane;
				int v = 0;
				if (i < n_keep) { const ChainRec &c = sl.chains[(int)(uint32_t)sl.skey[i]]; if (c.kept != 0) v = c.n; }
				n_ext += __popcll(__ballot(v > 0));
				tot += ema_wave_sum(v);
			}
			if (n_ext >= hv.min_chains) {
				auto up64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
				HeavyHdr h;
				h.read = read; h.n_chn = n_chn; h.n_chain = cb.n_chain; h.n_seed = cb.n_seed; h.status = cb.status; h.n_ext = n_ext;
				h.frac_rep = frac_rep; h.pad = 0;
				h.off_skey = up64(sizeof(HeavyHdr));
				h.off_chains = h.off_skey + up64((size_t)n_chn * 8);
				h.off_seeds = h.off_chains + up64((size_t)cb.n_chain * sizeof(ChainRec));
				h.off_first = h.off_seeds + up64((size_t)cb.n_seed * sizeof(SeedRec));
				h.off_cs = h.off_first + up64((size_t)(n_chn + 1) * 4);
				h.off_res = h.off_cs + up64((size_t)tot * sizeof(SeedRec));
				h.off_valid = h.off_res + up64((size_t)tot * sizeof(DevReg));
				h.bytes = h.off_valid + up64((size_t)tot);
				// a place on the read list, room on the task list, room in the arena -- or the read is extended here after all
				// (entries already claimed are marked void for their consumers)
				long long ri = -1, tb = -1, at = -1;
				if (lane == 0) {
					ri = atomicAdd(hv.n_reads, 1);
					if (ri >= hv.reads_cap) ri = -1;
					if (ri >= 0) {
						tb = atomicAdd(hv.n_tasks, n_ext);
						if (tb + n_ext > hv.tasks_cap) {
							for (long long j = tb; j < hv.tasks_cap && j < tb + n_ext; ++j) hv.tasks[j] = ~0ULL;
							tb = -1;
						}
					}
					if (tb >= 0) {
						at = (long long)atomicAdd(hv.arena_used, (unsigned long long)h.bytes);
						if ((unsigned long long)at + h.bytes > hv.arena_bytes) {
							for (long long j = tb; j < tb + n_ext; ++j) hv.tasks[j] = ~0ULL;
							at = -1;
						}
					}
					if (ri >= 0) hv.reads[ri] = at >= 0 ? (unsigned long long)at : ~0ULL;
				}
				at = (long long)ema_uni((int64_t)__shfl(at, 0)); tb = (long long)ema_uni((int64_t)__shfl(tb, 0));
				if (at >= 0) {
					uint8_t *rc = hv.arena + at;
					if (lane == 0) *reinterpret_cast<HeavyHdr *>(rc) = h;
					uint64_t *d_skey = reinterpret_cast<uint64_t *>(rc + h.off_skey);
					ChainRec *d_chains = reinterpret_cast<ChainRec *>(rc + h.off_chains);
					SeedRec *d_seeds = reinterpret_cast<SeedRec *>(rc + h.off_seeds);
					int32_t *d_first = reinterpret_cast<int32_t *>(rc + h.off_first);
					uint8_t *d_valid = rc + h.off_valid;
					for (int i = lane; i < cb.n_chain; i += EMA_WAVE) d_chains[i] = sl.chains[i];
					for (int i = lane; i < cb.n_seed; i += EMA_WAVE) d_seeds[i] = sl.seeds[i];
					for (int i = lane; i < tot; i += EMA_WAVE) d_valid[i] = 0;
					int run = 0, run_ext = 0;
					for (int base = 0; base < n_keep; base += EMA_WAVE) {
						const int i = base + lane;
						int v = 0;
						uint64_t key = 0;
						if (i < n_keep) { key = sl.skey[i]; const ChainRec &c = sl.chains[(int)(uint32_t)key]; if (c.kept != 0) v = c.n; }
						const int incl = ema_wave_incl_scan_add(v);
						const unsigned long long ext = __ballot(v > 0);
						if (i < n_keep) {
							d_skey[i] = key;
							d_first[i] = run + incl - v;
							if (v > 0) hv.tasks[tb + run_ext + __popcll(ext & ((1ULL << lane) - 1))] = (unsigned long long)(at >> 6) << 32 | (uint32_t)i;
						}
						run += ema_uni(__builtin_amdgcn_readlane(incl, 63));
						run_ext += __popcll(ext);
					}
					if (lane == 0) d_first[n_keep] = run;
					EMA_DBG(9, -n_ext);
					EMA_PHASE(0);
					EMA_LP(11);
					continue;
				}
			}
		}

		// regions start out in LDS (sorting and de-duplicating a handful of them in the HBM slab is dozens of dependent round
		// trips); the list moves to the slab when it outgrows AVL
		bool av_lds = AVL > 0 && !(MODE == 3 && hd.n_av > AVL);      // (mode 3 starts with the regions K2a handed over)
		if (av_lds) {
			uint8_t *b = lds_av[wib];
			cb.sl.av = (DevReg *)b; cb.sl.av_tmp = (DevReg *)b + AVL; cb.sl.rkeys = (uint64_t *)((DevReg *)b + 2 * AVL);
		}
		// ---------------- mem_chain2aln for every surviving chain, in filtered order ----------------
		EMA_LP(1);
		int n_av = MODE == 3 ? hd.n_av : 0;
		EMA_DBG(5, n_keep);
		EMA_PHASE(3);
		for (int ci_sorted = MODE == 1 ? task_chain : MODE == 3 ? hd.chain_from : 0; ci_sorted < (MODE == 1 ? task_chain + 1 : n_keep); ++ci_sorted) {
			EMA_DBG(6, ci_sorted);
			const int cid = ema_uni((int)(uint32_t)sl.skey[ci_sorted]);
			const ChainRec c = ema_uni(sl.chains[cid]);
			if (c.kept == 0) continue;
			const int cn = c.n;
			EMA_PHASE(7);      // 7: per chain -- its seeds, the window bounds, the window fetch, the seed order
			const int hv_base = (MODE == 1 || MODE == 2) ? ema_uni(hv_first[ci_sorted]) : 0;      // MODE 1, 2: the chain's slots in cs / res / valid
			if ((MED && med) || MODE == 1 || MODE == 2) {      // the chain's seed copies and their order in LDS when they fit the small-table area's 1 KB head
				const bool fits = cn <= 32;
				cb.sl.cs = fits ? reinterpret_cast<SeedRec *>(lds_small[wib]) : slab.cs;
				cb.sl.srt = fits ? reinterpret_cast<uint64_t *>(lds_small[wib] + 32 * sizeof(SeedRec)) : slab.srt;
			}
			if (MODE == 2) {
				// K2c left the seeds in processing order: entry p is the seed the loop below visits as k = cn - 1 - p
				cb.sl.cs = hv_cs + hv_base;
				for (int t = lane; t < cn; t += EMA_WAVE) sl.srt[t] = (uint64_t)1 << 32 | (uint32_t)(cn - 1 - t);
				ema_wave_sync();
			} else {   // gather the chain's seeds
				int k = c.first_seed;
				for (int t = 0; t < cn; ++t) { const SeedRec s = ema_uni(sl.seeds[k]); if (lane == 0) sl.cs[t] = s; k = s.next; }
				ema_wave_sync();
			}
			EMA_LP(2);
			int64_t rmax0 = l_pac << 1, rmax1 = 0;
			uint8_t *rs = rseq;      // where this chain's window sits
			if (MODE == 3) {         // planned with the record (above)
				for (int t = lane; t < cn; t += EMA_WAVE) sl.srt[t] = (uint64_t)(uint32_t)sl.cs[t].len << 32 | (uint32_t)t;      // score == len
				rmax0 = ema_uni((int64_t)sl.cpos[ci_sorted]);
				rmax1 = rmax0 + ema_uni((int)sl.cord[ci_sorted]);
			} else {
				for (int t = lane; t < cn; t += EMA_WAVE) {
					const SeedRec s = sl.cs[t];
					const int64_t b = s.rbeg - (s.qbeg + cal_max_gap(opt, s.qbeg));
					const int tail = l_query - s.qbeg - s.len;
					const int64_t e = s.rbeg + s.len + (tail + cal_max_gap(opt, tail));
					rmax0 = rmax0 < b ? rmax0 : b;
					rmax1 = rmax1 > e ? rmax1 : e;
					if (MODE != 2) sl.srt[t] = (uint64_t)(uint32_t)s.len << 32 | (uint32_t)t;      // score == len
				}
				for (int m = 1; m < EMA_WAVE; m <<= 1) {
					const int64_t o0 = __shfl_xor(rmax0, m), o1 = __shfl_xor(rmax1, m);
					rmax0 = rmax0 < o0 ? rmax0 : o0;
					rmax1 = rmax1 > o1 ? rmax1 : o1;
				}
				rmax0 = ema_uni(rmax0); rmax1 = ema_uni(rmax1);
				rmax0 = rmax0 > 0 ? rmax0 : 0;
				rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
				if (rmax0 < l_pac && l_pac < rmax1) {
					if (c.f_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
				}
				ema_clamp_window(ix, rmax0, c.f_rbeg, rmax1);
			}
			if (rmax1 - rmax0 > EMA_RSEQ_CAP) { cb.status |= EMA_ST_RSEQ_OVERFLOW; continue; }
			bool have_win = MODE != 2;      // K2d fetches the window only if it has to run a DP itself
			if (MODE == 3) {
				if (ci_sorted >= win_end) {
					// The windows of this chain and of the next ones, as many as fit the buffer (at most four, each of at most 63 words +
					// the odd bases: a longer one travels alone), in ONE round trip: every lane takes a word of each.
					const int b = ci_sorted;
					const int my_len = lane < n_keep ? (int)sl.cord[lane] : 0;
					const int64_t my_r0 = lane < n_keep ? (int64_t)sl.cpos[lane] : 0;
					if (rmax1 - rmax0 > 1008) {
						ema_wave_fetch(ix, rmax0, rmax1, rseq);
						if (lane == 0) sl.kept[b] = 0;
						win_end = b + 1;
					} else {
						const unsigned long long big = __ballot(lane > b && my_len > 1008);
						const int limit = big ? __ffsll((long long)big) - 1 : n_keep;
						const int wl = (lane >= b && lane < limit && my_len > 0) ? my_len : 0;
						const int incl = ema_wave_incl_scan_add(wl);
						const unsigned long long nz = __ballot(wl > 0);
						unsigned long long take = nz & __ballot(incl <= EMA_RSEQ_CAP), first4 = 0;
						int ct[4] = {-1, -1, -1, -1};
#pragma unroll
						for (int t = 0; t < 4; ++t) if (take) { ct[t] = __ffsll((long long)take) - 1; first4 |= 1ULL << ct[t]; take &= take - 1; }
						const unsigned long long blocked = nz & ~first4;
						win_end = blocked ? __ffsll((long long)blocked) - 1 : limit;
						if ((first4 >> lane) & 1) sl.kept[lane] = incl - wl;
						EmaWin wn[4];
						uint32_t word[4] = {0, 0, 0, 0};
						int at[4] = {0, 0, 0, 0};
#pragma unroll
						for (int t = 0; t < 4; ++t) if (ct[t] >= 0) {
							const int64_t r0 = ema_lane_val(my_r0, ct[t]);
							wn[t] = ema_win(ix, r0, r0 + ema_lane_val(wl, ct[t]));
							at[t] = ema_lane_val(incl - wl, ct[t]);
							if (lane < wn[t].n_dw) word[t] = ema_win_load(ix, wn[t], lane);
						}
#pragma unroll
						for (int t = 0; t < 4; ++t) if (ct[t] >= 0 && lane < wn[t].n_dw) ema_win_unpack(wn[t], lane, word[t], rseq + at[t]);
					}
					ema_wave_sync();
				}
				rs = rseq + ema_uni((int)sl.kept[ci_sorted]);
				// ks_introsort_64 on (score << 32 | index): keys are distinct, so the result is THE sorted order
				ema_wave_sync();
				if (cn > 1 && cn <= EMA_WAVE) ema_rank_sort_distinct(sl.srt, cn);      // (distinct keys: dev_sort.hpp)
				else if (cn > 1 && lane == 0) ema_introsort(sl.srt, cn, [](uint64_t x, uint64_t y) { return x < y; }, lds_stack[wib]);
				ema_wave_sync();
			} else if (MODE != 2) {
				ema_wave_fetch(ix, rmax0, rmax1, rseq);
				// ks_introsort_64 on (score << 32 | index): keys are distinct, so the result is THE sorted order
				ema_wave_sync();
				if (cn > 1 && cn <= EMA_WAVE) ema_rank_sort_distinct(sl.srt, cn);      // (distinct keys: dev_sort.hpp)
				else if (cn > 1 && lane == 0) ema_introsort(sl.srt, cn, [](uint64_t x, uint64_t y) { return x < y; }, lds_stack[wib]);
				ema_wave_sync();
			}

			EMA_LP(3);
			EMA_PHASE(3);
			for (int k = cn - 1; k >= 0; --k) {
				EMA_DBG(7, k);
				const SeedRec s = ema_uni(sl.cs[(int)(uint32_t)sl.srt[k]]);
				if (MODE == 1 && lane == 0) hv_cs[hv_base + (cn - 1 - k)] = s;
				// already covered by an earlier extension of this read?
				bool covered = false;
				for (int base = 0; base < n_av && !covered; base += EMA_WAVE) {
					bool hit = false;
					const int i = base + lane;
					if (i < n_av) {
						const DevReg p = sl.av[i];
						if (!(s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) &&
						    !((double)(s.len - p.seedlen0) > .1 * (double)l_query)) {
							int qd = s.qbeg - p.qb; int64_t rd = s.rbeg - p.rb;
							int max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
							int w = max_gap < p.w ? max_gap : p.w;
							if (qd - rd < w && rd - qd < w) hit = true;
							else {
								qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
								max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
								w = max_gap < p.w ? max_gap : p.w;
								if (qd - rd < w && rd - qd < w) hit = true;
							}
						}
					}
					if (__ballot(hit)) covered = true;
				}
				if (covered) {   // ... unless an already-extended, overlapping seed of the chain lies on another diagonal
					bool other = false;
					for (int base = k + 1; base < cn && !other; base += EMA_WAVE) {
						bool hit = false;
						const int i = base + lane;
						if (i < cn && sl.srt[i] != 0) {
							const SeedRec t = sl.cs[(int)(uint32_t)sl.srt[i]];
							if (!((double)t.len < (double)s.len * .95)) {
								if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) hit = true;
								if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) hit = true;
							}
						}
						if (__ballot(hit)) other = true;
					}
					if (!other) { if (lane == 0) sl.srt[k] = 0; continue; }
				}
				if (n_av >= EMA_AV_CAP) { cb.status |= EMA_ST_REG_OVERFLOW; continue; }
				if (av_lds && n_av >= AVL) {      // outgrown: continue in the slab
					ema_wave_sync();
					for (int i = lane; i < n_av; i += EMA_WAVE) slab.av[i] = sl.av[i];
					cb.sl.av = slab.av; cb.sl.av_tmp = slab.av_tmp; cb.sl.rkeys = slab.rkeys;
					av_lds = false;
					ema_wave_sync();
				}

				DevReg a;
				if (MODE == 2 && ema_uni((int)hv_valid[hv_base + (cn - 1 - k)]) != 0) {
					a = ema_uni(hv_res[hv_base + (cn - 1 - k)]);      // K2c extended this seed
				} else {
				if (MODE == 2 && !have_win) { ema_wave_fetch(ix, rmax0, rmax1, rseq); ema_wave_sync(); have_win = true; rs = rseq; }
				a.sub = a.csub = a.secondary = a.n_comp = a.is_alt = 0; a.seedcov = 0;
				int aw0 = opt.w, aw1 = opt.w;
				a.score = a.truesc = -1;
				a.rid = c.rid;
				EMA_PHASE(4);
				if (s.qbeg) {     // left extension, both sequences reversed
					const int tlen = (int)(s.rbeg - rmax0);
					EmaExtRes r; r.score = -1; r.qle = r.tle = r.gtle = 0; r.gscore = -1; r.max_off = 0;
					for (int i = 0; i < 2; ++i) {        // MAX_BAND_TRY
						const int prev = a.score;
						aw0 = opt.w << i;
						++n_dp;
						EMA_LP(4); if (PROF == 2) ++lp_n_dp;
						r = ema_wave_extend(opt, s.qbeg, EmaSeq{query + s.qbeg - 1, -1}, tlen, EmaSeq{rs + tlen - 1, -1}, aw0,
						                    opt.pen_clip5, opt.zdrop, s.len * opt.a);
						EMA_LP(5);
						a.score = r.score;
						if (a.score == prev || r.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
					}
					if (r.gscore <= 0 || r.gscore <= a.score - opt.pen_clip5) {
						a.qb = s.qbeg - r.qle; a.rb = s.rbeg - r.tle; a.truesc = a.score;
					} else {
						a.qb = 0; a.rb = s.rbeg - r.gtle; a.truesc = r.gscore;
					}
				} else { a.score = a.truesc = s.len * opt.a; a.qb = 0; a.rb = s.rbeg; }
				if (s.qbeg + s.len != l_query) {     // right extension
					const int sc0 = a.score, qe = s.qbeg + s.len, re = (int)(s.rbeg + s.len - rmax0);
					EmaExtRes r; r.score = -1; r.qle = r.tle = r.gtle = 0; r.gscore = -1; r.max_off = 0;
					for (int i = 0; i < 2; ++i) {
						const int prev = a.score;
						aw1 = opt.w << i;
						++n_dp;
						EMA_LP(4); if (PROF == 2) ++lp_n_dp;
						r = ema_wave_extend(opt, l_query - qe, EmaSeq{query + qe, 1}, (int)(rmax1 - rmax0 - re), EmaSeq{rs + re, 1},
						                    aw1, opt.pen_clip3, opt.zdrop, sc0);
						EMA_LP(5);
						a.score = r.score;
						if (a.score == prev || r.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
					}
					if (r.gscore <= 0 || r.gscore <= a.score - opt.pen_clip3) {
						a.qe = qe + r.qle; a.re = rmax0 + re + r.tle; a.truesc += a.score - sc0;
					} else {
						a.qe = l_query; a.re = rmax0 + re + r.gtle; a.truesc += r.gscore - sc0;
					}
				} else { a.qe = l_query; a.re = s.rbeg + s.len; }
				EMA_PHASE(8);      // 8: after the DPs -- seedcov, the region record
				{   // seedcov: seeds of the chain fully inside the region
					int cov = 0;
					for (int t = lane; t < cn; t += EMA_WAVE) {
						const SeedRec u = sl.cs[t];
						if (u.qbeg >= a.qb && u.qbeg + u.len <= a.qe && u.rbeg >= a.rb && u.rbeg + u.len <= a.re) cov += u.len;
					}
					a.seedcov = ema_wave_sum(cov);
				}
				a.w = aw0 > aw1 ? aw0 : aw1;
				a.seedlen0 = s.len;
				a.frac_rep = frac_rep;
				}
				if (MODE == 1 && lane == 0) { hv_res[hv_base + (cn - 1 - k)] = a; hv_valid[hv_base + (cn - 1 - k)] = 1; }
				if (lane == 0) sl.av[n_av] = a;
				++n_av;
				ema_wave_sync();
				EMA_PHASE(3);
			}
		}

		if (MODE == 1) { EMA_PHASE(0); EMA_LP(4); continue; }      // K2c: this chain's results are in the record
		// ---------------- mem_sort_dedup_patch ----------------
		EMA_DBG(8, n_av);
		EMA_LP(4);
		EMA_PHASE(5);
		EmaRegWork wk; wk.a = sl.av; wk.tmp = sl.av_tmp; wk.keys = sl.rkeys; wk.stack = lds_stack[wib]; wk.rseq = rseq; wk.mark = dbg ? dbg + slot * 4 : nullptr;
		int n_out = ema_sort_dedup_patch(ix, opt, query, n_av, wk, cb.status);
		if (n_out > opt.reg_cap) { cb.status |= EMA_ST_REG_OVERFLOW; n_out = opt.reg_cap; }
		ema_wave_sync();
		EMA_PHASE(9);      // 9: results out
		DevReg *dst = regs + (size_t)read * opt.reg_cap;
		for (int i = lane; i < n_out; i += EMA_WAVE) { DevReg r = sl.av[i]; r.is_alt = ema_ctg_alt(ix, r.rid); dst[i] = r; }      // mem_align1_core's last loop
		if (lane == 0) { n_regs[read] = n_out; if (cb.status) atomicOr(status + read, cb.status); }
		EMA_DBG(9, n_out);
		EMA_LP(6);
		EMA_PHASE(0);
		if (rlog && lane == 0) {
			const int at = atomicAdd(rlog, 1);
			if (at < rlog[1]) {
				int *r = rlog + 16 + (size_t)at * 8;
				r[0] = read; r[1] = log_iv; r[2] = log_occ; r[3] = cb.n_chain; r[4] = cb.n_seed; r[5] = n_av; r[6] = n_dp;
				r[7] = (int)((__builtin_amdgcn_s_memtime() - t_read) >> 4);
			}
		}
	}
	if (prof && lane == 0 && opt.reg_cap <= EMA_REG_LEAN) {      // lean-tier launch start / end as the waves saw them (slots 26, 28)
		atomicMin(prof + 26, t_launch); atomicMax(prof + 28, (unsigned long long)__builtin_amdgcn_s_memtime());
	}
	if (prof && lane == 0) for (int i = 0; i < 12; ++i) atomicAdd(prof + (i < 8 ? i : i + 4), acc[i]);      // 8.. -> slots 12.. (8..11 are K1's)
	if (PROF == 2 && lane == 0 && prof_arg) {      // twelve slots per mode: the seven phases, lifetimes, DP calls, work items, wavefronts
		unsigned long long *o = prof_arg + 16 * MODE;
		for (int k = 0; k < 7; ++k) atomicAdd(o + k, lp[k]);
		for (int k = 7; k < 12; ++k) atomicAdd(o + 4 + k, lp[k]);      // slots 11..15
		atomicAdd(o + 7, __builtin_amdgcn_s_memtime() - lp_t0); atomicAdd(o + 8, (unsigned long long)lp_n_dp); atomicAdd(o + 9, (unsigned long long)lp_reads);
		atomicAdd(o + 10, 1ULL);
	}
#undef EMA_LP
#undef EMA_DBG
#undef EMA_PHASE
}

extern "C" size_t ema_align_slab_bytes() { return EMA_ALIGN_SLAB_BYTES; }

// EMA_PHASE_PROFILE=3 (engine.hip): 64 device words, sixteen per mode -- the clocks of seven phases (see the kernel), wavefront
// lifetimes, DP calls, work items, wavefronts
static unsigned long long *ema_align_light_prof = nullptr;
extern "C" void ema_align_set_light_profile(unsigned long long *buf) { ema_align_light_prof = buf; }

// mode 0: K2b (reads K2a could not finish and has no chains for: chaining, filter, extension or setting aside); 1: K2c; 2: K2d;
// 3: the reads K2a handed over with their chains ready.  (Round 2's measurement builds -- everything in LDS at one or two blocks per
// CU, the region list in the slab, three blocks per CU -- were measured, lost (profiles/r02c_k2b_variants.txt) and are gone.)
extern "C" void ema_launch_align(const DevIndex *ix, const DevOpts *opt, const uint8_t *bases, const uint32_t *off,
                                 int n_reads, const int *n_pairs_dev, const int *map, const Intv *intv, const int *n_intv, DevReg *regs, int *n_regs, int *status,
                                 const int *todo, const int *n_todo, const uint8_t *hand, uint8_t *slabs, int *counter, int n_blocks,
                                 hipStream_t stream, int *dbg,
                                 unsigned long long *prof, const HeavyCtl *heavy, int mode)
{
	HeavyCtl hv;
	if (heavy) hv = *heavy;
	else { hv.arena = nullptr; hv.arena_bytes = 0; hv.arena_used = nullptr; hv.reads = nullptr; hv.tasks = nullptr; hv.n_reads = hv.n_tasks = nullptr; hv.reads_cap = hv.tasks_cap = 0; hv.min_chains = 1 << 30; }
#define EMA_ALIGN_LAUNCH(...) hipLaunchKernelGGL((ema_k_align_t<__VA_ARGS__>), dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, bases, off, n_reads, n_pairs_dev, map, intv, n_intv, regs, \
	                   n_regs, status, todo, n_todo, hand, slabs, counter, dbg, prof, hv)
	const int level = ema_align_light_prof ? 2 : prof != nullptr ? 1 : 0;
	if (level == 2) prof = ema_align_light_prof;
#define EMA_ALIGN_MODE(M) do { if (level == 2) EMA_ALIGN_LAUNCH(32, 8, 4, M, 2); else if (level == 1) EMA_ALIGN_LAUNCH(32, 8, 4, M, 1); else EMA_ALIGN_LAUNCH(32, 8, 4, M, 0); } while (0)
	if (mode == 1) EMA_ALIGN_MODE(1);            // K2c: one chain of a read set aside per wavefront
	else if (mode == 2) EMA_ALIGN_MODE(2);       // K2d: the replay of a read set aside
	else if (mode == 3) EMA_ALIGN_MODE(3);       // the reads handed over by K2a
	else EMA_ALIGN_MODE(0);
#undef EMA_ALIGN_MODE
#undef EMA_ALIGN_LAUNCH
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_align_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_align_t<32, 8, 4, 0, 0>, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
