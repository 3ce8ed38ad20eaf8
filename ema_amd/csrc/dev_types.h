// ema_amd/csrc/dev_types.h -- plain structs shared by the host side and the HIP kernels.
//
// HBM layout of the index (one replica per GPU):
//   occ     : 32-byte blocks, one per 64 BWT symbols (OccBlock below): four counts + the symbols as two bit planes.
//   sa      : the whole suffix array (seq_len+1 rows, 4 or 8 bytes each): locating
//             an occurrence is one load.
//   pac     : forward strand, bwa's byte layout (4 bases/byte, first base in the
//             high bits); the reverse strand is read complemented from the far end.
//   contigs : offsets[n+1] (offsets[n] = l_pac).
#ifndef EMA_DEV_TYPES_H
#define EMA_DEV_TYPES_H

#include <stdint.h>

// FM-index rank structure in HBM: one 32-byte block per 64 BWT symbols -- the number of A/C/G/T before the block
// (relative to the block's superblock of 2^31 symbols, whose absolute counts ride in DevIndex) and the 64 symbols
// themselves as two bit planes: bit t of bases[0] = the low bit of symbol t's 2-bit code, bit t of bases[1] = its high
// bit ([r6]; rounds 1-5 kept the codes side by side, which cost the rank query a mask and three masked population counts
// per 32 symbols -- dev_common.hpp, ema_occ_planes).  One occ4 query = one block = two 16-byte loads by the querying lane.
struct OccBlock { uint32_t cnt[4]; uint64_t bases[2]; };
#ifndef EMA_OCC_SUPER_SHIFT
#define EMA_OCC_SUPER_SHIFT 31      // (a test build of the host interpreter lowers it to exercise several superblocks on a small reference)
#endif
#define EMA_OCC_MAX_SUPER 4            // up to 2^33 BWT symbols (a human genome and its reverse complement: 2^32.5)

struct DevIndex {
	const OccBlock *occ;
	const void *sa;           // uint32_t* or uint64_t* by sa_width
	const uint8_t *pac;
	const int64_t *ctg_off;   // n_seqs + 1
	const uint8_t *ctg_alt;   // n_seqs flags: the contig is named in <prefix>.alt (bwa's bntann1_t.is_alt); null when none is
	// bns_pos2rid without its binary search: ctg_tab[b] = the contig holding forward position b << ctg_shift (at most 2^16 + 1
	// entries, the last one for l_pac - 1), so a position's contig lies in [ctg_tab[b], ctg_tab[b + 1]] -- nearly always one candidate
	const int32_t *ctg_tab;
	int32_t ctg_shift;
	uint64_t primary, seq_len;
	uint64_t L2[5];
	uint64_t occ_super[EMA_OCC_MAX_SUPER - 1][4];   // absolute counts at the start of superblocks 1, 2, 3
	int64_t l_pac;
	int32_t n_seqs, sa_width, n_super, kmer_k;
	// k-mer interval table (kmer_k > 0): the suffix-array interval (start, size) of EVERY string of length 1..kmer_k, so that a
	// rank query whose result is that short -- most of K1's: the first bases of every forward search, nearly all of a backward
	// phase, four fifths of every LAST-like seed -- is ONE look-up, cache-resident up to length ~11, instead of two dependent
	// 32-byte gathers from the 3 GB rank structure.  Level L holds 4^L entries indexed by the string's 2-bit code (first base
	// in the high bits).  Levels 1..EMA_KMER_WIDE: {u64 start, u64 size}; above: one u64, start in bits 0..39, size in 40..63.
	const uint64_t *kmer_wide;    // levels 1..EMA_KMER_WIDE, level L at entry offset (4^L - 4) / 3
	const uint64_t *kmer_narrow;  // levels EMA_KMER_WIDE+1..kmer_k, level L at offset (4^L - 4^(EMA_KMER_WIDE+1)) / 3
	// The text the FM-index is built on -- forward strand, then its reverse complement: 2 * l_pac bases -- as 2-bit codes in the
	// order K1 keeps its reads in (base j at bits 2(j%32) of word j/32), followed by at least 8 zero words.  K1 follows a match
	// that has a single occurrence left along this text (one suffix-array row + one 64-byte load for up to 224 bases) instead
	// of through two rank gathers per base (k_seed.hip, "tails").  Null: no tails (then K1 works on the rank structure alone).
	const uint64_t *text2;
};
#define EMA_KMER_WIDE 9
#define EMA_KMER_MAX 15

// bwa's mem_opt_t subset used by the kernels (mem_opt_init(); max_occ=3000 at reference src/align.c:185)
struct DevOpts {
	int a, b, o_del, e_del, o_ins, e_ins;
	int pen_clip5, pen_clip3, w, zdrop;
	int min_seed_len, split_len, split_width, max_mem_intv, max_occ, max_chain_gap;
	int min_chain_weight, max_chain_extend;
	float mask_level, drop_ratio, mask_level_redun;
	int8_t mat[25];
	// per-read output capacities of this launch (the engine runs a lean tier and, for the few reads that exceed it,
	// a second tier with the full EMA_INTV_CAP / EMA_REG_CAP / EMA_CIG_CAP)
	int intv_cap, reg_cap, cig_cap;
	// K1: a read that needs more extends than this is given up with EMA_ST_LONG (lean tier: keeps a launch's tail short;
	// the full tier has no budget)
	int seed_budget;
	// K1: bit 0 = a pass-2 search is skipped when no min_seed_len-base window over its position can be frequent enough (k_seed.hip, "window test");
	// bit 1 = a pass-1 search whose forward match ended as a single occurrence finds its SMEM on the text ("anchors")
	// bit 2 = one control pass per tick ("one pass per tick")
	// bit 3 = pass 3 (the LAST-like seeds, bwt_seed_strategy1) runs as a kernel of its own behind K1 (k_seed_p3.hip); K1 then ends a
	//         read after pass 2 and leaves the extends it has used in seed_ext[read] (the lean budget runs on across the passes)
	int seed_flags;
	int32_t *seed_ext;
};

// SMEM / seed interval: bwa's bwtintv_t.  info = start<<32 | end.
struct Intv { uint64_t x0, x1, x2, info; };
// K1 may hand a single-occurrence interval over BY POSITION: x1 == EMA_INTV_BYPOS, x2 == 1 and x0 is the occurrence's place in
// the text (what bwt_sa() would return for the row) instead of the row -- K2 then skips the suffix-array look-up (k_seed.hip, "anchors")
#define EMA_INTV_BYPOS (1ULL << 63)

// per-read capacities of the seeding stage
#define EMA_INTV_CAP 512      // intervals kept per read
#define EMA_LIST_CAP 256      // entries of a forward/backward working list (<= read length)
#define EMA_INTV_LEAN 96      // lean tier (engine.hip).  [r4] 96 / 96 / 192: at the GRCh38 scale 0.5 % of pairs had a read over 48 intervals and 0.2 % one
#define EMA_REG_LEAN 96       // over 48 regions (tools/gpu_capdist.py); with 96 both are under 0.04 %, and the full tier's pass is a quarter shorter
#define EMA_CIG_LEAN 192
#define EMA_SEED_BUDGET_LEAN 6144      // [r6] 4,096 until K1's rank decode got cheaper: 135.1-135.3 ms per step against 136.7-137.1 (5,120: 135.9; 7,168: 135.2; 8,192: 135.7-137.7; profiles/r06_ab.txt)
#define EMA_SEED_BUDGET_LANE 2048      // ... when the reads over it are seeded by K1w in place (engine.hip, run_seed) instead of going to the full tier
#define EMA_MAX_READ 255      // longest read the engine accepts (reference MAX_READ_LEN is 200, include/align.h:61)

// bwa's mem_seed_t plus the link to the next seed of the same chain
struct SeedRec { int64_t rbeg; int32_t qbeg, len; int32_t next, pad; };

// bwa's mem_chain_t; the first and last seed are cached for test_and_merge / chn_beg / chn_end
struct ChainRec {
	int64_t pos;            // rbeg of the first seed
	int64_t f_rbeg, l_rbeg;
	int32_t f_qbeg, l_qbeg, l_len;
	int32_t rid, n, first_seed, last_seed;
	int32_t w, kept, first;
};

// K2a -> K2b hand-over: when the lane-per-read kernel gives a read up only at the extension (or later), its chaining and
// chain-filter results travel with the read, so the wave-per-read kernel (k_align.hip, mode 3) starts at mem_chain2aln.
// Records are written densely, in the order K2a gives reads up (record i = the i-th read handed over; their number is a device
// counter), and carry everything the consumer needs to start -- the read's number, its length and where its bases are -- so
// that a wavefront goes from "item i" to its first extension in two round trips to memory: the header, then {bases, tables}.
// K2a's finished work travels too: the chains it extended completely before the one it gave up in (chain_from: where the consumer
// resumes, in filtered order) and the regions they produced (n_av of them: mem_chain2aln's list as it stood at that chain's start).
// Layout: HandHdr, the filter's sorted keys (weight << 32 | chain), the chains, the seed pool, the regions.
#define EMA_HAND_SEEDS 32
#define EMA_HAND_REGS 12        // = K2a's EMA_LANE_REGS
struct HandHdr { int32_t read, n_chn, n_seed, l_query; uint32_t base_off; int32_t chain_from, n_av, pad; };
#define EMA_HAND_BYTES (sizeof(HandHdr) + (size_t)EMA_HAND_SEEDS * (8 + sizeof(ChainRec) + sizeof(SeedRec)) + (size_t)EMA_HAND_REGS * sizeof(DevReg))

// Chain-rich reads (hundreds of chains, nearly every one of them extended: a read from a young repeat family) are a long
// serial job for the one wavefront that owns them -- two extension DPs per chain, one after the other -- and they set the
// length of a K2b launch once the work queue is empty.  K2b therefore sets such a read aside after the chain filter: its
// tables (sorted keys, chains, seed pool) go to a record in an arena and one TASK per chain to extend goes on a list.
//   K2c (ema_k_align_t<.., 1>): one wavefront per task runs mem_chain2aln's body for that chain alone and leaves, per seed in
//        processing order, the region the extension produces (or "not computed" where it would skip the seed);
//   K2d (ema_k_align_t<.., 2>): one wavefront per read replays mem_chain2aln over all chains in order -- cover tests against
//        the regions accepted so far, exactly the sequential decisions -- taking the extension results from K2c's table (they
//        depend on the seed and its chain only) and running a DP itself only where K2c skipped one; then dedup and output.
struct HeavyCtl {
	uint8_t *arena;                    // null: nothing is set aside
	unsigned long long arena_bytes;
	unsigned long long *arena_used;    // bump allocator
	unsigned long long *reads;         // record offsets of the reads set aside
	unsigned long long *tasks;         // record offset / 64 << 32 | chain (index in filtered order)
	int *n_reads, *n_tasks;
	int reads_cap, tasks_cap;
	int min_chains;                    // a read with at least this many chains to extend is set aside
};
struct HeavyHdr {                      // head of a record; the arrays follow at the offsets given (bytes from the record's start)
	int32_t read, n_chn, n_chain, n_seed, status, n_ext;
	float frac_rep;
	int32_t pad;
	uint64_t off_skey, off_chains, off_seeds, off_first, off_cs, off_res, off_valid, bytes;
};

// bwa's mem_alnreg_t (fields used on this path)
struct DevReg {
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, csub, w, seedcov, secondary, seedlen0, n_comp, is_alt;
	float frac_rep;
};

// per-read capacities of the chaining / extension stage (per-wave scratch, see k_align.hip)
#define EMA_SEED_CAP 32768
#define EMA_CHAIN_CAP 16384
#define EMA_AV_CAP 2048       // regions of one read before dedup
#define EMA_REG_CAP 1024      // regions of one read handed to the next stage
#define EMA_RSEQ_CAP 2048     // reference window bytes staged in LDS
#define EMA_CIG_CAP 4096      // CIGAR ops of all candidates of one read

// read status bits
#define EMA_ST_INTV_OVERFLOW 1
#define EMA_ST_LIST_OVERFLOW 2
#define EMA_ST_SEED_OVERFLOW 4
#define EMA_ST_CHAIN_OVERFLOW 8
#define EMA_ST_REG_OVERFLOW 16
#define EMA_ST_RSEQ_OVERFLOW 32
#define EMA_ST_CIGAR_OVERFLOW 64
#define EMA_ST_LONG 256           // lean tier only: the read's seeding exceeded the lean extend budget
#define EMA_ST_REDO 128           // lean tier only: the pair is on the full-capacity tier's work list

#endif
