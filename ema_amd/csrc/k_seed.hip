// ema_amd/csrc/k_seed.hip -- K1: SMEM / seed-interval collection on the FM-index (gfx950).
//
// Replaces, for a whole batch of reads, the seeding stage the reference reaches at
// src/bwabridge.c:236-237 (mem_align1_core -> mem_chain -> mem_collect_intv in the un-vendored
// bwa: bwt_smem1 x3 passes + bwt_seed_strategy1).  Output per read: the seed intervals
// (start, end, k, k', size) that mem_collect_intv leaves in aux->mem, in discovery order; the
// consumer (K2, or the host for the debug entry point) orders them by (start, end) -- entries with
// equal (start, end) describe the same substring and are identical, so that order is unique.
//
// Mapping to the hardware.  One bwt_extend is two independent 64-byte reads (the occ blocks of k-1 and l) followed
// by a few popcounts; successive extends of one search are strictly dependent, so a search is latency-bound and
// bandwidth only comes from many searches in flight.  Every LANE therefore carries its own read: a small state
// machine (forward / backward phase of an SMEM search, re-seeding pass, LAST-like pass).  A tick of the wave is
//   A  every lane runs its control program on registers and LDS only, until it needs an extend or a list entry;
//   B  all lanes issue their global loads TOGETHER at one place in the code -- the two occ blocks (four 16-byte
//      loads each: one 64-byte line per occ4 query) and the prefetch of the working-list entry the lane will need
//      next -- so the wave waits for memory once per tick, with up to 128 + 64 lines in flight.
// Lanes in different states execute their code one state after the other, so the kernel is bound by instruction
// issue (rocprof: ~45 % of all SIMD cycles issue VALU work, memory stalls are hidden), and the control program is
// laid out to keep the divergent part tiny: a step is a fixed sequence of guarded blocks
//   result handlers (per state, a few integer ops each; they only DECIDE: which interval to store, which follow-up)
//   -> the one place that stores an interval (working list or output) -> end of forward phase -> next row entry /
//   next row / end of search -> start of the next search -> the one place that looks up the next base and posts
//   the extend request,
// so that almost every tick is one pass over this sequence whatever mix of states the 64 lanes are in.
// The read sits in LDS as 2-bit codes + N mask (packed by the host).
// Working lists (bwa's prev/curr vectors).  The forward phase of a search pushes its intervals onto list F, in a
// lane-interleaved slab of HBM scratch, written fire-and-forget; the first backward row walks F from its end, one entry
// prefetched per tick.  Every backward row writes the intervals that survive it onto list B -- and B is ONE list, compacted in
// place: row r reads B[j] for j = 0, 1, ... and pushes at most one entry per entry read, so the push index never passes the
// read index and the next row's list grows over the current one (bwa swaps two vectors).  B is where the traffic is (a
// search's backward phase reads and writes ~100 entries, its forward phase ~17), and B's first EMA_SEED_LDS_LIST entries live in
// LDS as 16-byte entries {k: 40 bit, end: 8, size: 40, aux: 40 = the string's code in table mode, k' otherwise}: no memory
// request, no tick of latency; entries beyond that go to the slab.  EMA_SEED_LDS_LIST = 0 (the product build, see below why) keeps
// all of B in the slab.
// (Earlier forms: one read per 8-lane group sharing each block load -- the 8 lanes replayed the whole control
// program, ~10 M reads/s; one read per lane with per-state code that stored/pushed at a dozen inlined sites --
// every tick walked ~1500 vector instructions, 189 VGPRs.)
//
// k-mer interval table (DevIndex.kmer_k > 0, dev_types.h).  A rank query whose RESULT is a string of at most kmer_k bases is
// answered from the table of all short strings' intervals instead: one look-up (16 or 8 bytes, cache-resident for the short
// strings that every search starts with) in place of two dependent 32-byte gathers from the rank structure.  At the default
// scale that is the first 14 steps of every forward search, most of every backward phase (its rows are the short prefixes
// extended to the left) and three quarters of every LAST-like seed: half of all rank queries.  The machine carries the 2-bit
// code of the string it is extending (working-list entries keep it in the slot of the reverse-strand coordinate k', which
// nothing downstream reads: mem_chain uses k, size, start, end); k' itself is fetched from the table once, when a forward
// extension leaves the table's range and the next step needs it for its rank query.  Emitted intervals carry k' = 0 in this
// mode (the exact-k' build of this kernel is kmer_k = 0).
//
// Tails (DevIndex.text2 != null; needs the table mode).  Once a forward extension of pass 1 is down to ONE occurrence and past the
// table's range, every further base costs bwt_smem1 a rank query whose only possible outcomes are "still that occurrence" and
// "none" -- a hundred dependent pairs of gathers for the usual read, whose first SMEM runs to the first mismatch or to the end.
// The machine instead asks for the occurrence's suffix-array row (one tick), then for the text behind it (DevIndex.text2: both
// strands as 2-bit codes in the read's own packing; one tick, 64 bytes = up to 224 bases), and compares 32 bases per step with
// the read in LDS: the forward phase ends where the first difference, the read's end, an ambiguous base or the text's end is,
// exactly where the rank queries would have ended it (a forward extension of a one-row interval keeps k; k' is never read
// again: backward extensions pass it through, and in table mode it is not emitted).  15 % of all ticks, a quarter of the
// gathers that miss the caches.
// Pass 3 jumps (table mode): bwt_seed_strategy1 tests nothing before its match is min_seed_len + 1 bases long, so its first
// min(kmer_k, min_seed_len) bases are ONE table look-up instead of that many dependent ones, a start too close to the read's
// end for a seed to fit ends the pass, and an ambiguous base inside the first bases moves the start behind it without a look-up.
//
// The window test of pass 2 (table mode, DevOpts.seed_flags bit 0).  Re-seeding runs bwt_smem1 from the middle of every long, rare
// SMEM with min_intv = its occurrences + 1 -- 300 of a read's 740 extends on the benchmark mix -- and nearly always reports
// NOTHING: what it reports are matches of at least min_seed_len bases over the middle position x that occur min_intv times, and in
// unique sequence no 19-mer occurs twice.  Any such match contains a window of exactly min_seed_len bases over x that is at least
// as frequent, so: if no window [e - W, e), x < e <= x + W, W = min_seed_len, reaches min_intv occurrences, the search reports
// nothing and is skipped -- exactly.  A window is tested from its right end: its last kmer_k bases in one table look-up, then one
// base to the left per rank query, until the suffix is too rare (length L) or the window is whole (then the search runs as before).
// A suffix of L bases that is too rare condemns every window that holds it -- right ends e .. e + W - L -- so the next window
// tested ends at e + W - L + 1 (an ambiguous base condemns the windows that hold it likewise): about six windows of four
// dependent steps for the usual search instead of 150 extends.  When a window CAN be frequent enough the search runs, without the
// prefixes that end before that window's right end (every window ending there has been condemned: they can report nothing).  K1's lane-ticks per read 555 -> 475 on the benchmark mix (442 with
// the same test deciding about the backward phases of pass-1 searches that start on a mismatch, 413 with the anchors below).
//
// Anchors (tails + window test; DevOpts.seed_flags bit 1).  A pass-1 search whose forward match ended as ONE occurrence of at least
// min_seed_len bases has the usual read's shape: the match is the read's true place, and bwt_smem1's backward phase walks ~17
// shorter prefixes leftwards beside it, a triangle of ~150 extends, to report one thing -- the longest match, extended to the
// left as far as the text agrees.  The machine knows where that occurrence is (the tail's suffix-array row), so it compares the
// read leftwards against the text as well (DevIndex.text2 holds the reverse complement too: going left on one strand is going
// right on the other), which gives the SMEM's start p + 1 and its place in the text; the interval goes out BY POSITION
// (EMA_INTV_BYPOS: K2 wants the position, not the row).  The prefixes beside it can report something only after the anchor has
// died, i.e. as matches that span position p -- the window test over p decides: none can (the usual case) -> the search is over;
// one might -> the regular backward phase runs from its start, nothing having been reported.  Exact, like the test itself.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

namespace {

// states; the three that wait for an extend have their "result arrived" twin at +1
enum { PC_DONE = 0, PC_P1_NEXT, PC_P2_NEXT, PC_P2_RES, PC_P3_NEXT, PC_FWD_STOP, PC_BWD_N,
       PC_FWD = 8, PC_FWD_RES, PC_BWD, PC_BWD_RES, PC_S3, PC_S3_RES,
       PC_TSA_RES, PC_TXT, PC_TXT_RES,        // a tail (below): suffix-array row requested / text to request / text requested
       PC_WT_NEXT, PC_WT, PC_WT_RES,          // the window test of a pass-2 search (below): next window / one more base to the left / answer
       PC_AB, PC_AB_RES, PC_AB_EMIT };        // an anchor (below): text to the left to request / requested / its SMEM to report

// A machine taken off its lane (see "re-packing" below): everything phase B and the next step need.  The working
// lists stay where they are -- `wl` is the address of the lane's list slab -- and the read is re-staged from qpack.
// Entries of list B kept in LDS.  MEASURED (r03c, default workload): with 14 (80 KB of LDS per block, two blocks per CU) K1 on its own
// takes 48.2 ms per slice against 48.3 ms with every entry in the slab -- the list requests were never what the memory system was
// short of (they are lane-interleaved runs that hit the caches; the bound is the random 32-byte gathers from the 6 GB rank
// structure + table) -- while beside the other slices' kernels it takes 131 ms instead of 90: two such blocks own a CU's whole LDS,
// no K2-K4 block can share the CU, and the timed steps lose 10 %.  So the product keeps list B in the slab (0) and only the
// in-place compaction (one list where bwa swaps two: half the footprint); the LDS path stays compiled and is exercised through the
// host interpreter with 3 and 14 (tools/emu_seed.py).
#ifndef EMA_SEED_LDS_LIST
#define EMA_SEED_LDS_LIST 0
#endif
#define EMA_SEED_LDS_ROOM (EMA_SEED_LDS_LIST > 0 ? EMA_SEED_LDS_LIST : 1)
struct SeedPark {
	uint64_t last_curr_size, c0, c1, c2, f0, f1, f2, ld_at, wl;
	uint64_t b_lds[EMA_SEED_LDS_ROOM][2];      // list B's LDS entries
	int32_t pc, pass, len, read, x, sm_x, min_intv, i, j, n_prev, n_curr, rev, pad_, n_mem_call, last_mem_start;
	int32_t n_out, old_n, k2, st, req_c, ld_kind, has_req, n_ext;
	uint32_t c_end, f_end;
	uint32_t c_code, f_code, req_code, req_len;      // k-mer table mode (DevIndex.kmer_k > 0)
};

// bits [s, s + 64) of hi:lo, 0 <= s < 64
__device__ __forceinline__ uint64_t seed_funnel(uint64_t lo, uint64_t hi, int s) { return s ? (lo >> s) | (hi << (64 - s)) : lo; }
// the order of the 16 two-bit groups of v reversed
__device__ __forceinline__ uint32_t seed_rev_groups(uint32_t v)
{
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	return (v >> 16) | (v << 16);
}

// the 32 bits of v in reverse order
__device__ __forceinline__ uint32_t seed_rev_bits(uint32_t v)
{
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	return seed_rev_groups(v);
}

}  // namespace

// reads: qpack[r * 24 ..]: 16 words of 2-bit codes (base i at bits 2(i%16) of word i/16, N stored as 0) followed by
//        8 words of N flags (bit i%32 of word i/32); read lengths from off[]
// intv : n_reads x opt.intv_cap (in discovery order, see the header), n_intv / status : n_reads
// lists: (gridDim.x * blockDim.x) x 2 x EMA_LIST_CAP scratch entries (per lane list F, then the part of list B that is not in
//        LDS; interleaved over the 64 lanes of a wave so that lanes at the same list index touch one contiguous 2 KB run)
// counter: zero on entry; reads are handed out one by one
// Re-packing.  A read costs anything from a hundred to ten thousand ticks, so once the queue is empty a wave keeps
// ticking -- at full instruction cost -- for its last few long reads.  Instead, a wave that is down to park_max
// machines after the queue ran dry writes them to park_out (count in *n_park_out) and retires; the next launch of this
// kernel (park_in / n_park_in set, same grid) takes parked machines instead of fresh reads, 64 to a wave again.  The
// engine queues a fixed, short series of such launches, the last one with park_max = 0.  (Each launch shrinks the
// parked population about 64 / park_max-fold; a wave that starts with no more than park_max machines keeps them.)
#ifndef EMA_SEED_WPS
#define EMA_SEED_WPS 4      // [r4] the product build is held to four waves per SIMD (128 registers, nothing spilled); left alone it takes 135 since the anchors
#endif
// P3: pass 3 (bwt_seed_strategy1) is part of this machine.  [r5] The product runs it as a kernel of its own (k_seed_p3.hip: a forward-only
// machine of three states whose tick costs a fifth of this one's) and launches the <.., false> build, which carries none of its states.
template <bool PROF, bool P3>      // PROF: the diagnostic build (tick statistics); the product build carries none of its registers
__global__ void __launch_bounds__(256, PROF ? 1 : EMA_SEED_WPS)
ema_k_seed_t(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
           const int *__restrict__ n_pairs_dev, const int *__restrict__ map, Intv *__restrict__ intv,
           int *__restrict__ n_intv, int *__restrict__ status, Intv *__restrict__ lists,
           int *__restrict__ counter, const SeedPark *__restrict__ park_in, const int *__restrict__ n_park_in,
           SeedPark *__restrict__ park_out, int *__restrict__ n_park_out, int park_max, int *__restrict__ long_list,
           int *__restrict__ n_long, int long_cap, const int *__restrict__ order, unsigned long long *prof_arg)
{
	unsigned long long *const prof = PROF ? prof_arg : nullptr;
	__shared__ uint32_t lds_q[4][16 * 64];      // 2-bit read codes, 16 words per lane, lane-interleaved
	__shared__ uint32_t lds_n[4][8 * 64];       // N mask, 8 words per lane
	__shared__ uint4 lds_b[4][EMA_SEED_LDS_ROOM * 64];      // list B's first entries, entry e of a lane at [(e << 6) + lane]
	__shared__ uint64_t lds_rt[4][EMA_RANK_TABLE];          // ema_extend_blocks' bases, one copy per wavefront
	const int lane = (int)(threadIdx.x & 63), wib = ema_uni((int)(threadIdx.x >> 6));
	const uint64_t *rt = lds_rt[wib];
	ema_rank_table_init(ix, lds_rt[wib]);
	const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wib;
	const uint32_t *qw = lds_q[wib] + lane, *nm = lds_n[wib] + lane;
	// this wave's slab: list F then list B (entries from EMA_SEED_LDS_LIST on), entry e of this lane at [(e << 6) + lane]
	Intv *wl = lists + wave * (2 * EMA_LIST_CAP * 64) + lane;
	uint4 *bl = lds_b[wib] + lane;
	const int n_tasks = park_in ? *n_park_in : ema_work_count(n_reads, n_pairs_dev, 2);

	// ---- per-lane machine state
	int pc = PC_DONE, pass = 1, len = 0, read = -1;
	int x = 0, sm_x = 0, min_intv = 1, i = 0, j = 0;
	int n_prev = 0, n_curr = 0, rev = 0;      // rev: the row being walked is list F from its end (the first backward row)
	int n_mem_call = 0, last_mem_start = 0, n_out = 0, old_n = 0, k2 = 0, st = 0, n_ext = 0;
	uint64_t last_curr_size = 0;
	uint64_t c0 = 0, c1 = 0, c2 = 0; uint32_t c_end = 0;      // interval being extended: bwa's ik (forward) / *p (backward)
	uint64_t f0 = 0, f1 = 0, f2 = 0; uint32_t f_end = 0;      // first interval pushed in this backward row (the next row starts with it)
	uint64_t r0 = 0, r1 = 0, r2 = 0;                          // result of the extend posted in the previous tick
	Intv ent; ent.x0 = ent.x1 = ent.x2 = ent.info = 0;        // list entry / output entry loaded in the previous tick
	size_t out_base = 0;
	int has_req = 0;                 // 1: rank query posted, 2: table look-up posted
	bool exhausted = false;
	int req_c = 0, ld_kind = 0;      // ld_kind: 1 = next list entry of the backward row, 2 = out[k2 - 1] for pass 2
	size_t ld_at = 0;
	const int kk = ix.kmer_k;        // 0: no table
	const bool tails = kk > 0 && ix.text2 != nullptr;
	const int jump = P3 && kk > 0 && opt.max_mem_intv > 0 ? (kk < opt.min_seed_len ? kk : opt.min_seed_len) : 0;      // pass 3's first look-up
	const uint64_t n_text = (uint64_t)ix.l_pac << 1;
	const bool wtest = kk > 0 && (opt.seed_flags & 1) && opt.min_seed_len >= 2 && opt.min_seed_len <= 32;      // pass 2's window test
	const int wlen = opt.min_seed_len;
	const bool anchors = wtest && tails && (opt.seed_flags & 2);
	const bool one_pass = (opt.seed_flags & 4) != 0;
	uint32_t c_code = 0, f_code = 0, r_code = 0, req_code = 0, req_len = 0;      // 2-bit codes of the strings behind c, f, r

	auto q = [&](int p_) -> int {
		const int code = (qw[(p_ >> 4) << 6] >> ((p_ & 15) << 1)) & 3;
		return ((nm[(p_ >> 5) << 6] >> (p_ & 31)) & 1) ? 4 : code;
	};
	// diagnostic (prof != null): ticks, active lane-ticks and shader clocks of this wave
	int peak = 0;
	unsigned long long n_tick = 0, n_active = 0, t_start = prof ? __builtin_amdgcn_s_memtime() : 0;
	unsigned long long n_pass_lane = 0, n_pass_wave = 0, n_kind[4] = {0, 0, 0, 0};
	unsigned long long n_by_pass[4] = {0, 0, 0, 0}, n_wt[4] = {0, 0, 0, 0};      // (PROF) this lane's ticks by pass / in window tests; searches skipped / run: pass 2, pass 1      // (PROF) phase-A passes by lane / by wave; ticks with a rank / table / tail / list-only lane
	for (;;) {
		// ---- phase A: control programs, registers and LDS only
		// (one pass per tick: a machine whose pass ends between two states -- 2 % of the lane-ticks, but 61 % of the ticks have one -- waits
		// for the next tick's pass instead of making the whole wave run the pass again for it; EMA_SEED_ONEPASS=0: as many as it takes)
		bool again = true;
		while (!has_req && !ld_kind && !exhausted && again) {
			again = !one_pass;
			if (prof) { n_pass_lane += 1; }
			int ev = 0;                       // 1: push v onto the current list, 2: emit v as a seed interval
			uint64_t v0 = 0, v1 = 0, v2 = 0;
			uint32_t v_start = 0, v_end = 0;
			bool aft = false, nxt = false, start = false, wt_done = false;
			int prune = 0;      // pass 2, set with `start`: the right end of the first window that can be frequent enough
			// (1) handlers: the extend result / the loaded entry / a forward phase that cannot go on
			switch (pc) {
			case PC_FWD_RES:      // bwt_smem1, forward loop body after bwt_extend(ik, ok, 0)
				if (r2 != c2) {
					const bool last = r2 < (uint64_t)min_intv;
					// (rev == 3: a pass-2 search after its window test -- a prefix that ends before the first window that can be frequent
					// enough, n_prev, can report nothing, and neither can anything else because of it: it is not put on the list)
					if (last || !(rev == 3 && (int)c_end < n_prev)) { ev = 1; v0 = c0; v1 = kk ? c_code : c1; v2 = c2; v_end = c_end; }
					if (last) { aft = true; break; }
				}
				c0 = r0; c1 = r1; c2 = r2; c_end = (uint32_t)(i + 1); c_code = r_code;
				++i;
				pc = PC_FWD;
				break;
			case PC_FWD_STOP:     // end of the read or an ambiguous base: the current interval is the longest match
				ev = 1; v0 = c0; v1 = kk ? c_code : c1; v2 = c2; v_end = c_end;
				aft = true;
				break;
			case PC_BWD_RES:      // backward loop body for row entry c at query position i
				if (r2 < (uint64_t)min_intv) {
					if (n_curr == 0 && (n_mem_call == 0 || i + 1 < last_mem_start)) {
						++n_mem_call; last_mem_start = i + 1;
						if ((int)c_end - (i + 1) >= opt.min_seed_len) { ev = 2; v0 = c0; v1 = c1; v2 = c2; v_start = (uint32_t)(i + 1); v_end = c_end; }
					}
				} else if (n_curr == 0 || r2 != last_curr_size) {
					ev = 3; v0 = r0; v1 = kk ? r_code : r1; v2 = r2; v_end = c_end;
					last_curr_size = r2;
				}
				if (j + 1 < n_prev) {      // the row's next entry
					if (rev || j + 1 >= EMA_SEED_LDS_LIST) { c0 = ent.x0; c1 = ent.x1; c2 = ent.x2; c_end = (uint32_t)ent.info; c_code = (uint32_t)ent.x1; }      // prefetched with the extend
					else {
						const uint4 w = bl[(j + 1) << 6];      // {k: 40, end: 8, aux low: 16} {size: 40, aux high: 24}
						c0 = ((uint64_t)w.y << 32 | w.x) & 0xFFFFFFFFFFULL; c_end = (w.y >> 8) & 0xff;
						c2 = ((uint64_t)w.w << 32 | w.z) & 0xFFFFFFFFFFULL;
						c1 = (uint64_t)(w.y >> 16) | (uint64_t)(w.w >> 8) << 16; c_code = (uint32_t)c1;
					}
				}
				nxt = true;
				pc = PC_BWD;
				break;
			case PC_BWD_N:        // start of the read or an ambiguous base: every entry of the row dies; only the first can be
			                      // emitted (the others fail `start < last emitted start`), and the search is over
				if (n_curr == 0 && (n_mem_call == 0 || i + 1 < last_mem_start)) {
					++n_mem_call; last_mem_start = i + 1;
					if ((int)c_end - (i + 1) >= opt.min_seed_len) { ev = 2; v0 = c0; v1 = c1; v2 = c2; v_start = (uint32_t)(i + 1); v_end = c_end; }
				}
				pc = pass == 1 ? PC_P1_NEXT : PC_P2_NEXT;
				break;
			case PC_S3_RES:       // bwt_seed_strategy1, loop body after the extend
				if (!P3) break;
				if (r2 < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
					if (r2 > 0) { ev = 2; v0 = r0; v1 = r1; v2 = r2; v_start = (uint32_t)x; v_end = (uint32_t)(i + 1); }
					x = i + 1; pc = PC_P3_NEXT;
				} else {
					c0 = r0; c1 = r1; c2 = r2; c_code = r_code;
					if (++i == len) { x = len; pc = PC_DONE; } else pc = PC_S3;
				}
				break;
			case PC_P2_RES: {     // pass 2 (re-seeding): a pass-1 SMEM of length >= split_len with at most split_width occurrences
				const int s = (int)(ent.info >> 32), e = (int)(uint32_t)ent.info;
				if (e - s < opt.split_len || ent.x2 > (uint64_t)opt.split_width) pc = PC_P2_NEXT;
				else {
					sm_x = (s + e) >> 1; min_intv = (int)ent.x2 + 1;
					if (wtest) { n_prev = sm_x; i = sm_x + 1; pc = PC_WT_NEXT; } else start = true;
				}
				break;
			}
			case PC_WT_RES:       // window test: the suffix of j bases ending at i has r2 occurrences
				if (r2 < (uint64_t)min_intv) { i += wlen + 1 - j; pc = PC_WT_NEXT; }      // too rare: so is every window that holds it
				else if (j >= wlen) {      // a whole window is frequent enough: the search (pass 2) / its backward phase (pass 1) runs
					if (prof) n_wt[pass == 2 ? 1 : 3] += 1;
					if (pass == 2) { start = true; prune = i; } else { aft = true; wt_done = true; prune = i; }
				} else { f1 = r0; f2 = r2; pc = PC_WT; }      // (the test's interval lives in f1 / f2, idle outside backward rows, and is parked with them)
				break;
			case PC_TSA_RES:      // tail: the occurrence's place in the text is known (f0 = the text position of read base i)
				pc = PC_TXT;
				break;
			case PC_TXT_RES:      // tail: r2 bases agreed; r0 != 0: and the match ends there -- then what PC_FWD_STOP does
				i += (int)r2; f0 += r2;
				if (r0) {
					c_end = (uint32_t)i; ev = 1; v0 = c0; v1 = c_code; v2 = c2; v_end = c_end;
					if (anchors && pass == 1 && i - sm_x >= wlen) {      // an anchor: its SMEM from the text (f0 = the text position of read base i)
						x = i;
						f0 -= (uint64_t)(i - sm_x);                     // the match's place in the text
						if (sm_x == 0) { n_prev = -1; pc = PC_AB_EMIT; }      // nothing to the left of it
						else { f0 = n_text - f0; i = sm_x - 1; pc = PC_AB; }  // leftwards = rightwards on the other strand, from there
					} else aft = true;
				}
				else pc = PC_TXT;
				break;
			case PC_AB_RES:       // anchor: r2 more bases agree to the left; r0 != 0: and that is where the match ends
				i -= (int)r2; f0 += r2;
				if (!r0) { pc = PC_AB; break; }
				f0 = n_text - f0;      // the SMEM q[i + 1 .. c_end) begins here in the text
				n_prev = i;
				if (i < 0 || q(i) > 3) pc = PC_AB_EMIT;      // the read's start or an ambiguous base: every prefix beside the anchor dies there too
				else { rev = 2; i = n_prev + 1; pc = PC_WT_NEXT; }      // can a match that spans position n_prev exist at all?
				break;
			case PC_AB_EMIT:      // anchor: the search's one SMEM, by position
				ev = 2; v0 = f0; v1 = EMA_INTV_BYPOS; v2 = 1; v_start = (uint32_t)(n_prev + 1); v_end = c_end;
				pc = PC_P1_NEXT;
				break;
			default:
				break;
			}
			// (2) the one place that stores an interval
			if (ev == 1 && (int)v_end - sm_x < 64) last_curr_size |= 1ULL << ((int)v_end - sm_x);
			if (ev) {
				Intv e; e.x0 = v0; e.x1 = v1; e.x2 = v2;
				Intv *dst = nullptr;
				if (ev != 2) {
					if (n_curr >= EMA_LIST_CAP) st |= EMA_ST_LIST_OVERFLOW;
					else {
						e.info = v_end;
						if (ev == 3 && n_curr < EMA_SEED_LDS_LIST) {
							uint4 w;
							w.x = (uint32_t)v0; w.y = ((uint32_t)(v0 >> 32) & 0xff) | (v_end & 0xff) << 8 | (uint32_t)(v1 & 0xffff) << 16;
							w.z = (uint32_t)v2; w.w = ((uint32_t)(v2 >> 32) & 0xff) | (uint32_t)(v1 >> 16) << 8;
							bl[n_curr << 6] = w;
						} else dst = wl + ((size_t)((ev == 3 ? EMA_LIST_CAP : 0) + n_curr) << 6);
						if (n_curr == 0) { f0 = v0; f1 = v1; f2 = v2; f_end = v_end; f_code = (uint32_t)v1; }
						++n_curr;
					}
				} else {
					if (n_out >= opt.intv_cap) st |= EMA_ST_INTV_OVERFLOW;
					else { e.info = (uint64_t)v_start << 32 | v_end; if (kk && v1 != EMA_INTV_BYPOS) e.x1 = 0; dst = intv + out_base + n_out; ++n_out; }
				}
				if (dst) *dst = e;
			}
			// (3) forward phase over: its list (longest match = the interval just pushed = c) becomes prev, walked in reverse
			if (aft) {
				if (pass == 1) x = (int)c_end;      // bwt_smem1's return value: where the forward extension stopped
				if (wtest && pass == 1 && !wt_done && (int)c_end - sm_x < wlen) {
					// a forward match shorter than a seed (the search that starts ON a mismatch): its backward phase reports something only
					// if a window of min_seed_len bases over sm_x occurs at all -- the window test, right ends up to the match's end
					n_prev = sm_x; rev = 0; i = sm_x + 1; pc = PC_WT_NEXT;
				} else {
					// (after a window test of pass 1 that found a window ending at `prune`: the prefixes on list F that end before it can
					// report nothing -- they are the list's first entries, the last ones of the first backward row: left out)
					int n_skip = 0;
					if (wt_done && prune > sm_x) {
						const int d = prune - sm_x;
						n_skip = __popcll(last_curr_size & (d < 64 ? (1ULL << d) - 1 : ~0ULL));
						if (n_skip >= n_curr) n_skip = n_curr - 1;
					}
					if (pass == 1) k2 = n_skip;      // (k2 is pass 2's cursor: idle in pass 1)
					n_prev = n_curr - n_skip; n_curr = 0; rev = 1;
					i = sm_x - 1; j = 0;
					pc = PC_BWD;
				}
			}
			// (4) row entry done: next entry (already in c), next row, or end of the search
			if (nxt && ++j == n_prev) {
				if (n_curr == 0) pc = pass == 1 ? PC_P1_NEXT : PC_P2_NEXT;
				else {
					n_prev = n_curr; n_curr = 0; rev = 0; j = 0; --i;
					c0 = f0; c1 = f1; c2 = f2; c_end = f_end; c_code = f_code;
				}
			}
			// (5) between searches
			switch (pc) {
			case PC_DONE:
				if (read >= 0) {
					n_intv[read] = n_out; status[read] = st;
					if (!P3 && opt.seed_ext) opt.seed_ext[read] = n_ext;      // pass 3 follows in its own kernel: the budget runs on
					if ((st & EMA_ST_LONG) && long_list) {      // over the extend budget: on the list K1w (one wavefront per read) works through next
						const int at = atomicAdd(n_long, 1);
						if (at < long_cap) long_list[at] = read;
					}
				}
				read = atomicAdd(counter, 1);
				if (read >= n_tasks) { read = -1; exhausted = true; break; }
				if (order && !park_in) read = order[read];      // the reads expected to be long first (ema_k_seed_order below)
				if (park_in) {      // resume a parked machine: it waits for its extend / entry load
					const SeedPark &k = park_in[read];
					last_curr_size = k.last_curr_size; c0 = k.c0; c1 = k.c1; c2 = k.c2; f0 = k.f0; f1 = k.f1; f2 = k.f2;
					ld_at = (size_t)k.ld_at; wl = reinterpret_cast<Intv *>(k.wl);
					pc = k.pc; pass = k.pass; x = k.x; sm_x = k.sm_x; min_intv = k.min_intv; i = k.i; j = k.j;
					n_prev = k.n_prev; n_curr = k.n_curr; rev = k.rev; n_mem_call = k.n_mem_call;
#pragma unroll
					for (int t = 0; t < EMA_SEED_LDS_LIST; ++t) {
						uint4 w; w.x = (uint32_t)k.b_lds[t][0]; w.y = (uint32_t)(k.b_lds[t][0] >> 32); w.z = (uint32_t)k.b_lds[t][1]; w.w = (uint32_t)(k.b_lds[t][1] >> 32);
						bl[t << 6] = w;
					}
					last_mem_start = k.last_mem_start; n_out = k.n_out; old_n = k.old_n; k2 = k.k2; st = k.st;
					req_c = k.req_c; ld_kind = k.ld_kind; has_req = k.has_req; n_ext = k.n_ext; c_end = k.c_end; f_end = k.f_end;
					c_code = k.c_code; f_code = k.f_code; req_code = k.req_code; req_len = k.req_len;
					read = k.read;
				}
				{
					const int in_read = ema_in_read(map, read);
					len = (int)(off[in_read + 1] - off[in_read]);
					// the read, packed by the host: 16 code words (2 bit/base) + 8 mask words (N positions)
					const uint4 *pw = reinterpret_cast<const uint4 *>(qpack + (size_t)in_read * 24);
					const uint4 a = pw[0], b = pw[1], c = pw[2], d = pw[3], m0 = pw[4], m1 = pw[5];
					uint32_t *qd = lds_q[wib] + lane, *nd = lds_n[wib] + lane;
					qd[0 << 6] = a.x; qd[1 << 6] = a.y; qd[2 << 6] = a.z; qd[3 << 6] = a.w;
					qd[4 << 6] = b.x; qd[5 << 6] = b.y; qd[6 << 6] = b.z; qd[7 << 6] = b.w;
					qd[8 << 6] = c.x; qd[9 << 6] = c.y; qd[10 << 6] = c.z; qd[11 << 6] = c.w;
					qd[12 << 6] = d.x; qd[13 << 6] = d.y; qd[14 << 6] = d.z; qd[15 << 6] = d.w;
					nd[0 << 6] = m0.x; nd[1 << 6] = m0.y; nd[2 << 6] = m0.z; nd[3 << 6] = m0.w;
					nd[4 << 6] = m1.x; nd[5 << 6] = m1.y; nd[6 << 6] = m1.z; nd[7 << 6] = m1.w;
				}
				out_base = (size_t)read * opt.intv_cap;
				if (park_in) break;
				st = 0; n_out = 0; pass = 1; x = 0; n_curr = 0; n_ext = 0;
				if (len >= opt.min_seed_len) pc = PC_P1_NEXT;      // mem_chain: no seeds for a read shorter than min_seed_len
				break;
			case PC_WT_NEXT: {    // window test: the next window [i - wlen, i) over sm_x that is not condemned yet
				if (i < wlen) i = wlen;
				const int e_lim = pass == 1 ? (int)c_end : len;      // (pass 1: a window that reaches beyond the forward match holds a string that does not occur)
				const int e_hi = n_prev + wlen < e_lim ? n_prev + wlen : e_lim;      // (n_prev: the position the windows span)
				if (i > e_hi) {      // none can be frequent enough: the search (its backward phase; the prefixes beside an anchor) would report nothing
					if (prof) n_wt[pass == 2 ? 0 : 2] += 1;
					pc = pass == 2 ? PC_P2_NEXT : rev == 2 ? PC_AB_EMIT : PC_P1_NEXT;
					break;
				}
				const int p0 = i - wlen, wn = p0 >> 5;
				const uint64_t nn = (uint64_t)(wn < 7 ? nm[(wn + 1) << 6] : 0u) << 32 | nm[wn << 6];
				const uint32_t nbits = (uint32_t)(nn >> (p0 & 31)) & (wlen < 32 ? (1u << wlen) - 1u : ~0u);
				if (nbits) { i = p0 + (31 - __clz(nbits)) + wlen + 1; break; }      // an ambiguous base: no window that holds it matches anywhere
				if (++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = PC_DONE; break; }
				const int J = kk < wlen ? kk : wlen, pj = i - J, wq = pj >> 4;
				const uint64_t qq = (uint64_t)(wq < 15 ? qw[(wq + 1) << 6] : 0u) << 32 | qw[wq << 6];
				req_code = seed_rev_groups((uint32_t)(qq >> ((pj & 15) << 1))) >> (32 - 2 * J);
				req_len = (uint32_t)J; req_c = 0; has_req = 2;
				j = J;
				pc = PC_WT_RES;
				break;
			}
			case PC_P1_NEXT:      // pass 1: SMEMs from left to right
				while (x < len && q(x) > 3) ++x;
				if (x >= len) { pass = 2; old_n = n_out; k2 = 0; pc = PC_P2_NEXT; }
				else { sm_x = x; min_intv = 1; start = true; }
				break;
			case PC_P3_NEXT:      // pass 3: LAST-like seeds
				if (!P3) break;
				while (x < len && q(x) > 3) ++x;
				if (x >= len) { pc = PC_DONE; break; }
				if (jump > 0) {      // the first `jump` bases in one look-up (see the header)
					if (len - x <= opt.min_seed_len) { x = len; pc = PC_DONE; break; }      // no seed fits any more
					const int wn = x >> 5, wq = x >> 4;
					const uint64_t nn = (uint64_t)(wn < 7 ? nm[(wn + 1) << 6] : 0u) << 32 | nm[wn << 6];
					const uint32_t nbits = (uint32_t)(nn >> (x & 31)) & ((1u << jump) - 1u);
					if (nbits) { x += __ffs(nbits); break; }      // an ambiguous base ends the attempt; the next one starts behind it
					if (++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = PC_DONE; break; }
					const uint64_t qq = (uint64_t)(wq < 15 ? qw[(wq + 1) << 6] : 0u) << 32 | qw[wq << 6];
					req_code = seed_rev_groups((uint32_t)(qq >> ((x & 15) << 1))) >> (32 - 2 * jump);
					req_len = (uint32_t)jump; req_c = 0; has_req = 2;
					i = x + jump - 1;
					pc = PC_S3_RES;
					break;
				}
				{
					const int b = q(x);
					c0 = ix.L2[b] + 1; c2 = ix.L2[b + 1] - ix.L2[b]; c1 = ix.L2[3 - b] + 1; c_code = (uint32_t)b;
				}
				i = x + 1;
				if (i >= len) { x = len; pc = PC_DONE; } else pc = PC_S3;
				break;
			default:
				break;
			}
			if (pc == PC_P2_NEXT) {      // pass 2: fetch the next pass-1 SMEM; examined (PC_P2_RES) once loaded
				if (k2 >= old_n) { pass = 3; x = 0; pc = (P3 && opt.max_mem_intv > 0) ? PC_P3_NEXT : PC_DONE; }
				else { ld_at = out_base + k2; ld_kind = 2; ++k2; pc = PC_P2_RES; }
			}
			// (6) start of an SMEM search at sm_x (bwt_smem1 with min_intv)
			if (start) {
				const int b = q(sm_x);
				c0 = ix.L2[b] + 1; c2 = ix.L2[b + 1] - ix.L2[b]; c1 = ix.L2[3 - b] + 1; c_code = (uint32_t)b;
				c_end = (uint32_t)(sm_x + 1);
				n_curr = 0; n_mem_call = 0;
				i = sm_x + 1;
				rev = prune ? 3 : 0;
				if (prune) n_prev = prune;
				last_curr_size = 0;      // (forward phase: which prefix lengths are on list F, bit c_end - sm_x; a backward row's own use comes later)
				pc = PC_FWD;
			}
			// (7) the one place that looks up the next base and posts the extend
			if (pc == PC_WT) {      // window test: one more base to the left (the window holds no ambiguous base)
				if (++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = PC_DONE; }
				else { req_c = q(i - j - 1); ++j; has_req = 1; pc = PC_WT_RES; }
			}
			else if (pc == PC_TXT) { has_req = 4; pc = PC_TXT_RES; }
			else if (pc == PC_AB) { has_req = 5; pc = PC_AB_RES; }
			else if (pc == PC_FWD || pc == PC_BWD || (P3 && pc == PC_S3)) {
				const int b = (i >= 0 && i < len) ? q(i) : 4;
				if (b < 4 && ++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = PC_DONE; }      // too long for this tier
				else if (b < 4 && tails && pc == PC_FWD && c2 == 1 && i - sm_x > kk) { has_req = 3; pc = PC_TSA_RES; }      // a tail begins
				else if (b < 4) {
					has_req = 1;
					if (pc == PC_BWD) {
						req_c = b;
						if (j + 1 < n_prev) {      // the row's next entry comes from the slab: list F walked from its end, or list B beyond its LDS part
							if (rev) { ld_at = (size_t)(n_prev - 2 - j + (pass == 1 ? k2 : 0)) << 6; ld_kind = 1; }      // (+ the entries left out at the list's start)
							else if (j + 1 >= EMA_SEED_LDS_LIST) { ld_at = (size_t)(EMA_LIST_CAP + j + 1) << 6; ld_kind = 1; }
						}
						// the extended string is q[i .. c_end): short enough for the table?
						const int rl = (int)c_end - i;
						if (rl <= kk) { has_req = 2; req_len = (uint32_t)rl; req_code = ((uint32_t)b << (2 * (rl - 1))) | c_code; }
					} else {
						req_c = 3 - b;
						const int rl = i + 1 - (pc == PC_FWD ? sm_x : x);      // q[start .. i]
						if (rl <= kk) { has_req = 2; req_len = (uint32_t)rl; req_code = (c_code << 2) | (uint32_t)b; }
					}
					pc += 1;
				} else if (pc == PC_FWD) pc = PC_FWD_STOP;
				else if (pc == PC_BWD) pc = PC_BWD_N;
				else { x = i + 1; pc = PC_P3_NEXT; }
			}
		}
		const unsigned long long busy = __ballot(!exhausted);      // every machine with a request or an entry load, and those between two states
		if (!busy) break;
		const int n_busy = __popcll(busy);
		peak = peak > n_busy ? peak : n_busy;
		// the queue is empty and the wave, once fuller, nearly so (a wave that never held more runs its machines to the end)
		if (park_max > 0 && n_busy <= park_max && peak > park_max && __ballot(exhausted)) {
			if (!exhausted) {
				SeedPark k;
				k.last_curr_size = last_curr_size; k.c0 = c0; k.c1 = c1; k.c2 = c2; k.f0 = f0; k.f1 = f1; k.f2 = f2;
				k.ld_at = ld_at; k.wl = reinterpret_cast<uint64_t>(wl);
				k.pc = pc; k.pass = pass; k.len = len; k.read = read; k.x = x; k.sm_x = sm_x; k.min_intv = min_intv; k.i = i; k.j = j;
				k.n_prev = n_prev; k.n_curr = n_curr; k.rev = rev; k.pad_ = 0; k.n_mem_call = n_mem_call;
#pragma unroll
				for (int t = 0; t < EMA_SEED_LDS_LIST; ++t) {
					const uint4 w = bl[t << 6];
					k.b_lds[t][0] = (uint64_t)w.y << 32 | w.x; k.b_lds[t][1] = (uint64_t)w.w << 32 | w.z;
				}
				k.last_mem_start = last_mem_start; k.n_out = n_out; k.old_n = old_n; k.k2 = k2; k.st = st;
				k.req_c = req_c; k.ld_kind = ld_kind; k.has_req = has_req; k.n_ext = n_ext; k.c_end = c_end; k.f_end = f_end;
				k.c_code = c_code; k.f_code = f_code; k.req_code = req_code; k.req_len = req_len;
				park_out[atomicAdd(n_park_out, 1)] = k;
			}
			break;
		}
		if (prof) {
			++n_tick; n_active += __popcll(__ballot(has_req != 0));
			n_kind[0] += __ballot(has_req == 1) != 0; n_kind[1] += __ballot(has_req == 2) != 0; n_kind[2] += __ballot(has_req >= 3) != 0;
			n_kind[3] += __ballot(has_req >= 4) != 0;
			if (has_req) { n_by_pass[pass == 1 ? 0 : pass == 2 ? 1 : 2] += 1; if (pc == PC_WT_RES) n_by_pass[3] += 1; }
		}
		// ---- phase B: every global load of the tick, issued together
		if (ld_kind) {
			const ulong2 *src = reinterpret_cast<const ulong2 *>(ld_kind == 1 ? wl + ld_at : intv + ld_at);
			const ulong2 lo = src[0], hi = src[1];
			ent.x0 = lo.x; ent.x1 = lo.y; ent.x2 = hi.x; ent.info = hi.y;
			ld_kind = 0;
		}
		if (has_req) {
			// One place issues every load of the tick, whatever the lane asks for, so that the wave waits for memory once:
			//   rank query (has_req 1): the two 32-byte rank blocks of k - 1 and l;
			//   table look-up (has_req 2): the entry of the result string (its interval) and, when a forward extension reaches
			//   the table's last level, the entry of its reverse complement (the reverse-strand coordinate the next rank query
			//   needs) -- read as 32 bytes each like the blocks (the entries are 16 or 8 bytes; the tables are padded).
			const bool tab = has_req == 2, back = pc == PC_BWD_RES || pc == PC_WT_RES;
			const bool wt = pc == PC_WT_RES;      // (the window test keeps its interval apart, in f1 / f2: pass 1 still needs c)
			const uint64_t x_nb = wt ? f1 : back ? c0 : c1, x_b = back ? c1 : c0, sz = wt ? f2 : c2;
			const uint64_t pk = x_nb - 1, pl = x_nb - 1 + sz;                    // rows whose occ4 the extend needs
			const uint64_t qk = pk - (pk >= ix.primary ? 1 : 0), ql = pl - (pl >= ix.primary ? 1 : 0);      // '$' is not stored
			const bool want_rc = tab && (int)req_len == kk && !back;
			const uint4 *p0, *p1, *p2, *p3;      // the tick's four 16-byte loads
			if (has_req == 3) {      // tail: the suffix-array row of the one occurrence (the aligned 16 bytes that hold it)
				p0 = reinterpret_cast<const uint4 *>((reinterpret_cast<uintptr_t>(ix.sa) + c0 * (uint64_t)ix.sa_width) & ~(uintptr_t)15);
				p1 = p2 = p3 = p0;
			} else if (has_req >= 4) {      // tail / anchor: 8 words of text from the word that holds position f0
				p0 = reinterpret_cast<const uint4 *>(ix.text2 + (f0 >> 5));
				p1 = p0 + 1; p2 = p0 + 2; p3 = p0 + 3;
			} else if (tab) {
				const int L = (int)req_len;
				const uint32_t rcode = want_rc ? ema_kmer_revcomp(req_code, kk) : req_code;
				if (L <= EMA_KMER_WIDE) {
					const size_t base = ema_kmer_base_wide(L);
					p0 = reinterpret_cast<const uint4 *>(ix.kmer_wide + 2 * (base + req_code));
					p2 = reinterpret_cast<const uint4 *>(ix.kmer_wide + 2 * (base + rcode));
				} else {
					const size_t base = ema_kmer_base_narrow(L);
					p0 = reinterpret_cast<const uint4 *>(ix.kmer_narrow + base + req_code);
					p2 = reinterpret_cast<const uint4 *>(ix.kmer_narrow + base + rcode);
				}
				p1 = p0; p3 = p2;      // an entry is 16 or 8 bytes: the second halves repeat the first (no second cache line touched)
			} else {
				p0 = reinterpret_cast<const uint4 *>(ix.occ + (qk >> 6));
				p2 = reinterpret_cast<const uint4 *>(ix.occ + (ql >> 6));
				p1 = p0 + 1; p3 = p2 + 1;
			}
			const uint4 a0 = *p0, a1 = *p1, b0 = *p2, b1 = *p3;
			if (has_req == 3) {
				const unsigned o = (unsigned)((reinterpret_cast<uintptr_t>(ix.sa) + c0 * (uint64_t)ix.sa_width) & 15);
				uint64_t pos;
				if (ix.sa_width == 4) pos = o == 0 ? a0.x : o == 4 ? a0.y : o == 8 ? a0.z : a0.w;
				else pos = o == 0 ? ((uint64_t)a0.y << 32 | a0.x) : ((uint64_t)a0.w << 32 | a0.z);
				f0 = pos + (uint64_t)(i - sm_x);
			} else if (has_req >= 4) {
				// read base i stands against text position f0: count the bases that agree, 32 per step, up to the first difference,
				// the read's end, its next ambiguous base (stored as code 0: cut by the mask) or the text's end.  An anchor (5) walks the
				// read LEFTWARDS from i against the other strand's text rightwards: the read's bases reversed and complemented.
				const uint64_t T[8] = {(uint64_t)a0.y << 32 | a0.x, (uint64_t)a0.w << 32 | a0.z, (uint64_t)a1.y << 32 | a1.x, (uint64_t)a1.w << 32 | a1.z,
				                       (uint64_t)b0.y << 32 | b0.x, (uint64_t)b0.w << 32 | b0.z, (uint64_t)b1.y << 32 | b1.x, (uint64_t)b1.w << 32 | b1.z};
				const bool left = has_req == 5;
				const int st_ = (int)(f0 & 31) << 1;
				int lim = left ? i + 1 : len - i;
				if (f0 >= n_text) lim = 0;
				else if (n_text - f0 < (uint64_t)lim) lim = (int)(n_text - f0);
				int total = 0;
				bool stopped = false;
#pragma unroll
				for (int c = 0; c < 7; ++c) {
					if (!stopped && total < lim) {
						// 32 read bases from position s up (and their ambiguity flags), s = i + 32 c, or i - 32 c - 31 for an anchor
						const int s_ = left ? i - 32 * c - 31 : i + 32 * c;
						const int sp = s_ < 0 ? 0 : s_, w = sp >> 5, sh = sp & 31;
						const uint64_t q_lo = w < 8 ? ((uint64_t)qw[(2 * w + 1) << 6] << 32 | qw[(2 * w) << 6]) : 0;
						const uint64_t q_hi = w < 7 ? ((uint64_t)qw[(2 * w + 3) << 6] << 32 | qw[(2 * w + 2) << 6]) : 0;
						const uint64_t n_w = (uint64_t)(w < 7 ? nm[(w + 1) << 6] : 0u) << 32 | (w < 8 ? nm[w << 6] : 0u);
						uint64_t rd = seed_funnel(q_lo, q_hi, sh << 1);
						uint32_t nb = (uint32_t)(n_w >> sh);
						if (left) {
							if (s_ < 0) { rd <<= (-s_) << 1; nb <<= -s_; }      // (positions below 0: beyond the limit)
							rd = ~((uint64_t)seed_rev_groups((uint32_t)rd) << 32 | seed_rev_groups((uint32_t)(rd >> 32)));
							nb = seed_rev_bits(nb);
						}
						uint64_t d = seed_funnel(T[c], T[c + 1], st_) ^ rd;
						d = (d | d >> 1) & 0x5555555555555555ULL;
						const int m_d = d ? (__ffsll((unsigned long long)d) - 1) >> 1 : 32, m_n = nb ? __ffs(nb) - 1 : 32;
						const int m = m_d < m_n ? m_d : m_n;
						total += m;
						stopped = m < 32;
					}
				}
				if (total >= lim) { total = lim; stopped = true; }
				r2 = (uint64_t)total; r0 = stopped ? 1 : 0;
			} else if (tab) {
				uint64_t ea = (uint64_t)a0.y << 32 | a0.x, eb = (uint64_t)b0.y << 32 | b0.x;
				if ((int)req_len <= EMA_KMER_WIDE) { r0 = ea; r2 = (uint64_t)a0.w << 32 | a0.z; r1 = want_rc ? eb : 0; }
				else { r0 = ea & 0xFFFFFFFFFFULL; r2 = ea >> 40; r1 = want_rc ? (eb & 0xFFFFFFFFFFULL) : 0; }
				r_code = req_code;
			} else {
				uint64_t o_nb; uint32_t o_size, n_gt;
				ema_extend_blocks(ix, rt, qk, ql, a0, a1, b0, b1, req_c & 3, o_nb, o_size, n_gt);
				const uint64_t o_b = x_b + ((x_nb <= ix.primary && x_nb + sz - 1 >= ix.primary) ? 1 : 0) + n_gt;
				r0 = back ? o_nb : o_b; r1 = back ? o_b : o_nb; r2 = o_size;
			}
			has_req = 0;
		}
	}
	if (prof) {
		atomicAdd(prof + 30, n_pass_lane);
		atomicAdd(prof + 32, n_by_pass[0]); atomicAdd(prof + 33, n_by_pass[1]); atomicAdd(prof + 34, n_by_pass[2]); atomicAdd(prof + 35, n_by_pass[3]);
		atomicAdd(prof + 36, n_wt[0]); atomicAdd(prof + 37, n_wt[1]); atomicAdd(prof + 38, n_wt[2]); atomicAdd(prof + 39, n_wt[3]);
	}
	if (prof && lane == 0) {
		atomicAdd(prof + 8, n_tick); atomicAdd(prof + 9, n_active);
		atomicAdd(prof + 10, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_start)); atomicMax(prof + 11, n_tick);
		atomicAdd(prof + 23, n_kind[0]); atomicAdd(prof + 24, n_kind[1]); atomicAdd(prof + 25, n_kind[2]); atomicAdd(prof + 29, n_kind[3]);
	}
}

extern "C" size_t ema_seed_park_bytes() { return sizeof(SeedPark); }

// One launch of the series (see "re-packing"): park_in == null takes fresh reads, otherwise the machines parked by the
// previous launch; park_max == 0 never parks.  long_list (may be null): the reads given up with EMA_ST_LONG are appended there (*n_long counts
// them all, long_cap is the list's room) for K1w to seed next (engine.hip, run_seed).
extern "C" void ema_launch_seed(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off,
                                int n_reads, const int *n_pairs_dev, const int *map, Intv *intv, int *n_intv, int *status,
                                Intv *lists, int *counter, const void *park_in, const int *n_park_in, void *park_out,
                                int *n_park_out, int park_max, int *long_list, int *n_long, int long_cap, const int *order, int n_blocks,
                                hipStream_t stream, unsigned long long *prof)
{
	// (seed_flags bit 3 with a seed_ext array: pass 3 is k_seed_p3.hip's; the diagnostic build keeps it, and its statistics, in one machine)
	const bool split3 = (opt->seed_flags & 8) && opt->seed_ext != nullptr && !prof;
	DevOpts o = *opt;
	if (!split3) o.seed_ext = nullptr;
#define EMA_SEED_ARGS *ix, o, qpack, off, n_reads, n_pairs_dev, map, intv, n_intv, status, lists, counter, (const SeedPark *)park_in, n_park_in, \
	(SeedPark *)park_out, n_park_out, park_max, long_list, n_long, long_cap, order, prof
	if (prof) hipLaunchKernelGGL((ema_k_seed_t<true, true>), dim3(n_blocks), dim3(256), 0, stream, EMA_SEED_ARGS);
	else if (split3) hipLaunchKernelGGL((ema_k_seed_t<false, false>), dim3(n_blocks), dim3(256), 0, stream, EMA_SEED_ARGS);
	else hipLaunchKernelGGL((ema_k_seed_t<false, true>), dim3(n_blocks), dim3(256), 0, stream, EMA_SEED_ARGS);
#undef EMA_SEED_ARGS
}
// does this launch of the series leave pass 3 to k_seed_p3.hip?  (the engine asks, so that both decide alike)
extern "C" int ema_seed_splits_pass3(const DevOpts *opt, const unsigned long long *prof) { return (opt->seed_flags & 8) && opt->seed_ext != nullptr && !prof; }

// The order in which K1 takes a slice's reads: the ones expected to be LONG first.  A launch series is as long as its bulk plus
// the tail of the last long reads -- a read from a repeat family needs 2,000-4,000 dependent ticks, and one that comes up when the
// queue is nearly empty finishes on an empty chip 20 ms after everybody else (half of K1's isolated time, r04).  What makes a read
// long is repeats, and the k-mer table knows them: a read is taken first when one of six k-mers (k = min(kmer_k, 12): a level the
// caches hold) spread over it occurs more than four times as often as a random one would (on the benchmark mix that is a tenth of
// the reads holding more than nine tenths of those over 1,500 extends).  order[]: those reads from the front, the others from the
// back; cnt[0], cnt[1] = how many of each (zero on entry).  Results do not depend on the order (every read's slots are its own).
// (Measured, r04i-l: a SELECTIVE first class matters -- with the threshold at 2x or 1x the expected count a fifth to a third of the
// reads go first and the gain is gone (47-52 ms per series against 42); and six classes by the size of the count, longest expected
// first, do nothing for the launches (42.3 ms) while their per-lane atomics on one counter cost 12 ms: the two-class form stays,
// whose atomics the compiler folds into one per wavefront.)
__global__ void __launch_bounds__(256)
ema_k_seed_order(DevIndex ix, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads, int *__restrict__ order, int *__restrict__ cnt,
                 int n_samples, int mult4)
{
	const int r = (int)(blockIdx.x * 256 + threadIdx.x);
	if (r >= n_reads) return;
	const int K = ix.kmer_k < 12 ? ix.kmer_k : 12;
	const int len = (int)(off[r + 1] - off[r]);
	const uint32_t *q = qpack + (size_t)r * 24;
	uint64_t worst = 0;
	if (len >= K) {
		for (int j = 0; j < n_samples; ++j) {
			const int p = (int)((long)(len - K) * j / (n_samples > 1 ? n_samples - 1 : 1));      // spread over the read, the last one at its end
			const int wq = p >> 4, wn = p >> 5;
			const uint64_t nn = (uint64_t)(wn < 7 ? q[16 + wn + 1] : 0u) << 32 | q[16 + wn];
			if ((uint32_t)(nn >> (p & 31)) & ((1u << K) - 1u)) continue;      // an ambiguous base in the k-mer
			const uint64_t qq = (uint64_t)(wq < 15 ? q[wq + 1] : 0u) << 32 | q[wq];
			const uint32_t code = seed_rev_groups((uint32_t)(qq >> ((p & 15) << 1))) >> (32 - 2 * K);
			uint64_t x0, x2;
			ema_kmer_lookup(ix, K, code, x0, x2);
			worst = x2 > worst ? x2 : worst;
		}
	}
	const uint64_t expected = (ix.seq_len >> (2 * K)) + 1;
	if (4 * worst > (uint64_t)mult4 * expected) order[atomicAdd(cnt, 1)] = r;
	else order[n_reads - 1 - atomicAdd(cnt + 1, 1)] = r;
}
// n_samples k-mers per read (<= 16); a read goes first when one of them occurs more than mult4 / 4 times the expected count
extern "C" void ema_launch_seed_order(const DevIndex *ix, const uint32_t *qpack, const uint32_t *off, int n_reads, int *order, int *cnt, int n_samples,
                                      int mult4, hipStream_t stream)
{
	if (n_reads <= 0) return;
	hipLaunchKernelGGL(ema_k_seed_order, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, *ix, qpack, off, n_reads, order, cnt, n_samples, mult4);
}

// resident 256-thread blocks per CU for this kernel's register/LDS footprint (sizes the grid and the scratch slabs)
extern "C" int ema_seed_blocks_per_cu()
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ema_k_seed_t<false, true>, 256, 0) != hipSuccess || n < 1) n = 1;
	return n > 8 ? 8 : n;
}
