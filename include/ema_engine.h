/*
 * include/ema_engine.h -- C ABI of the MI355X engine for EMA's seed-and-extend hot path.
 *
 * This is the drop-in boundary: a batched form of the reference's bridge API
 * (reference include/bwabridge.h:92-106).  The reference calls, once per read pair,
 *
 *     EasyAlignmentPairs bwa_mem_mate_sw(ref, opts, r1, l1, r2, l2, 25);   src/align.c:1005  (src/bwabridge.c:204-299)
 *     bwa_smith_waterman(ref, opts, read, len, a->chained_hit, &r);        src/align.c:1013,1038 (src/bwabridge.c:301-311)
 *
 * for every candidate region of both mates.  Here the same work is done for a whole batch of
 * pairs by HIP kernels with the index resident in HBM; the host then walks the per-pair
 * candidate lists in the order the reference would have produced them.  INTEGRATION.md shows
 * the binding a reference maintainer would add to src/align.c.
 *
 * Plain C types only.  One engine per GPU; calls on one engine must be serialised by the
 * caller; several engines (one per GPU) may be used from different threads/processes.
 * Every function returns 0 on success or a negative EMA_E* code; ema_engine_strerror()
 * describes the last error of an engine.  There is no CPU fallback: opening an engine
 * without a usable GPU fails.
 */
#ifndef EMA_ENGINE_H
#define EMA_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMA_OK 0
#define EMA_EARG (-1)       /* bad argument */
#define EMA_EINDEX (-2)     /* index files missing or inconsistent */
#define EMA_EDEVICE (-3)    /* HIP error (no device, out of memory, launch failure) */
#define EMA_ELIMIT (-4)     /* a read exceeded an engine capacity (see ema_engine_status_of) */
#define EMA_ESTATE (-5)     /* call sequence error (e.g. fetch before run) */

typedef struct ema_engine ema_engine_t;

/* bwa's mem_opt_t fields used on this path: mem_opt_init() defaults, with max_occ = 3000 as the
 * reference sets at src/align.c:185.  ema_engine_opts_default() fills them. */
typedef struct {
	int a, b, o_del, e_del, o_ins, e_ins;
	int pen_clip5, pen_clip3, w, zdrop;
	int min_seed_len, split_width, max_mem_intv, max_occ, max_chain_gap;
	int min_chain_weight, max_chain_extend;
	float split_factor, mask_level, drop_ratio, mask_level_redun;
	int score_delta;            /* reference src/align.c:1005 passes 25 */
	int max_rescue;             /* reference src/bwabridge.c:264,278: 50 */
	int pes_low, pes_high;      /* reference src/bwabridge.c:222-223: -35, 500 (FR only) */
	int batch_pairs;            /* pairs per device batch (0 = engine default, 262144) */
	int n_streams;              /* slices of a batch run on their own HIP streams so that kernel tails overlap (0 = default, 3) */
	int full_tier_pairs;        /* pairs per batch the full-capacity tier can redo (0 = default: batch/16 within 4096..65536) */
	int lean_intervals, lean_regions, lean_cigar_ops;   /* per-read capacities of the lean tier (0 = defaults 48, 48, 192) */
	int mapq_coef_len, mapq_coef_fac;   /* bwa's mapQ_coef_len = 50 and mapQ_coef_fac = (int)log(50) = 3, read by the mapq formula (reference src/align.c:969-973) */
	int lean_seed_extends;      /* lean tier: FM-index extends one read's seeding may take (0 = default 6144, < 0 = no limit) */
} ema_engine_opts;

void ema_engine_opts_default(ema_engine_opts *o);
/* the options an engine was opened with (the append stage takes the same record) */
int ema_engine_get_opts(const ema_engine_t *e, ema_engine_opts *o);

/* replaces load_reference()/bwa_init() (reference src/bwabridge.c:77-96, src/align.c:180-186):
 * reads <index_prefix>.{bwt,fsa,pac,ann} and uploads the index to HBM of `device`. */
int ema_engine_open(const char *index_prefix, int device, const ema_engine_opts *opts, ema_engine_t **out);
void ema_engine_close(ema_engine_t *e);
/* A second engine on the same GPU sharing `first`'s index in HBM, with its own batch buffers and streams: lets a host
 * overlap staging, kernels and fetching of consecutive batches (stage batch i+1 into one engine while the other runs
 * batch i).  `first` must be closed last. */
int ema_engine_open_shared(const ema_engine_t *first, const ema_engine_opts *opts, ema_engine_t **out);
const char *ema_engine_strerror(const ema_engine_t *e);
/* The engine's own second set of batch buffers and streams (same index, same options), created on first call and closed
 * with the engine; NULL when device memory does not allow it.  ema_engine_align_pairs uses it for inputs beyond the batch
 * capacity, ema_stream_buckets (include/ema_stream.h) for alternate buckets. */
ema_engine_t *ema_engine_peer(ema_engine_t *e);

/* contig table (bns->anns[i].name/len/offset; reference src/align.c:199-200, src/bwabridge.c:86-91) */
int ema_engine_n_contigs(const ema_engine_t *e);
const char *ema_engine_contig_name(const ema_engine_t *e, int rid);
int64_t ema_engine_contig_len(const ema_engine_t *e, int rid);
int64_t ema_engine_contig_offset(const ema_engine_t *e, int rid);
/* 1 if <index_prefix>.alt names the contig (bwa's bntann1_t.is_alt; every non-'@' line's first field is a contig name):
 * a kept chain on such a contig does not shadow chains on primary contigs in the chain filter, and ema_cand_t.is_alt
 * carries the flag the reference reads at src/bwabridge.c:371 */
int ema_engine_contig_is_alt(const ema_engine_t *e, int rid);
int64_t ema_engine_l_pac(const ema_engine_t *e);
/* layout of the index in HBM: info[0] = rank superblocks, info[1] = log2 symbols per superblock, info[2] = bytes per
 * suffix-array row (4, or 8 beyond 2^32 rows), info[3] = k of the k-mer interval table (0 = none) */
int ema_engine_index_info(const ema_engine_t *e, int32_t info[4]);
/* development: the launch grids this engine settled on, in 256-thread blocks per compute unit -- grids[0..4] = K1 (lane-per-read
 * seeding), K2a, K2b, K3b, K4b.  What the tuning string's `grid=k2a:k2b:k3:k4` and `seed_blocks_per_cu` knobs came to; no counterpart
 * in the reference (its parallelism is OpenMP threads, src/align.c:261). */
int ema_engine_debug_grids(const ema_engine_t *e, int32_t grids[5]);

/* One candidate = one element of the reference's results.a (mem_alnreg_t, read through
 * interpret_align, src/bwabridge.c:313-339, and mem_approx_mapq_se_insist, src/align.c:959-984)
 * together with its final alignment (mem_aln_t as unpacked by interpret_single_read_alignment,
 * src/bwabridge.c:359-379). */
typedef struct {
	/* region (forward-reverse coordinates, as in mem_alnreg_t) */
	int64_t rb, re;
	int32_t qb, qe;
	int32_t rid;
	int32_t score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
	float frac_rep;
	/* final alignment */
	int64_t pos;            /* 0-based leftmost position on contig rid */
	int32_t is_rev, NM, n_cigar;
	uint32_t cigar_off;     /* first op in ema_batch_out.cigar (BAM packing: len<<4 | op, MIDSH = 0..4) */
	int32_t aln_score, aln_sub;
} ema_cand_t;

typedef struct {
	size_t n_pairs;
	uint64_t *cand_off;     /* 2*n_pairs + 1: candidates of mate m of pair i are cand[cand_off[2i+m] .. cand_off[2i+m+1]) */
	ema_cand_t *cand;       /* in the reference's order (results.a after rescue) */
	uint32_t *cigar;
	size_t n_cigar;
	size_t n_redone;        /* pairs that went through the full-capacity tier */
	int32_t *status;        /* per read (2*n_pairs): 0, or capacity bits (1 intervals, 2 lists, 4 seeds, 8 chains, 16 regions,
	                         * 32 reference window, 64 CIGAR ops, 128 not redone: full-capacity tier was full, 256 seeding budget); a flagged read has no candidates */
	uint32_t *redone;       /* n_redone pair indices: the pairs whose results come from the full-capacity tier */
	void *view_of;          /* private to the library (NULL in a batch whose arrays are malloc'd): a bucket cut out of a shared pass by
	                         * ema_stream_* is a VIEW -- cand, cigar and status point into the pass's batch, which lives until its last view
	                         * is freed; a batch from ema_engine_fetch_ticket points into a pooled page-locked buffer.  Always release a
	                         * batch with ema_batch_free, never with free() on its members. */
} ema_batch_out;

/* Whole hot path for a batch: reads are ASCII, read r at bases[off[r] .. off[r+1]);
 * mate 1 of pair i is read 2i, mate 2 is read 2i+1 (off has 2*n_pairs+1 entries).
 * *out is allocated by the engine; free it with ema_batch_free().  n_pairs may exceed ema_engine_batch_capacity(): the
 * call then works through the input in capacity-sized pieces, alternating over two sets of batch buffers (the second one
 * is created on first need and lives as long as the engine; EMA_ALIGN_PIPELINE=0 in the environment keeps it to one). */
int ema_engine_align_pairs(ema_engine_t *e, const char *bases, const uint32_t *off, size_t n_pairs, ema_batch_out **out);
void ema_batch_free(ema_batch_out *out);

/* The same in three steps, so that a caller (and bench.py) can keep inputs resident in HBM and
 * time the kernels alone:  stage = nt4-convert + H2D;  run = queue all kernels (asynchronous; several runs may be
 * queued back to back, each overwrites the previous results);  sync = wait;  fetch = wait + D2H + assemble.
 * n_pairs must not exceed ema_engine_batch_capacity().
 * Per-read result slots come in two tiers: every pair first runs with lean capacities; the few pairs with a read over
 * one of them are redone on the device with the full capacities (at most ema_engine_full_tier_capacity() per batch). */
size_t ema_engine_batch_capacity(const ema_engine_t *e);
int ema_engine_max_read_len(void);      /* longest read the engine takes (255; the reference's MAX_READ_LEN is 200) */
size_t ema_engine_full_tier_capacity(const ema_engine_t *e);
int ema_engine_stage(ema_engine_t *e, const char *bases, const uint32_t *off, size_t n_pairs);
/* Several batches resident at once: stage_slot puts a batch into input slot 0 <= slot < EMA_MAX_SLOTS (its device buffers
 * are allocated on first use, ~0.7 GB per Mi pairs of capacity; ema_engine_stage is slot 0), run_slot queues one pass over
 * that slot's batch.  Runs on different slots may be queued back to back like runs on one; sync/fetch/timing refer to
 * the last run queued. */
#define EMA_MAX_SLOTS 16
#define EMA_MAX_INFLIGHT 3
int ema_engine_stage_slot(ema_engine_t *e, int slot, const char *bases, const uint32_t *off, size_t n_pairs);
int ema_engine_run_slot(ema_engine_t *e, int slot);
/* Asynchronous passes: run_async queues one pass over the batch in `slot`, with the result layout and the packing done on the
 * device in the pass's own streams, and returns a ticket; the next pass may be queued at once (EMA_MAX_INFLIGHT in flight), and
 * fetch_ticket waits for one pass and downloads its batch while the following pass runs (the batch's layout -- slices and
 * full-capacity tier merged in read order -- is made on the device behind the pass; the arrays of the ema_batch_out it returns
 * lie in a page-locked buffer the engine recycles when ema_batch_free is called on it).  stage_async stages a
 * slot for this path: it waits only for the last pass that read that slot and leaves the passes in flight alone.  This is the
 * form ema_stream_* (include/ema_stream.h) drives: staging of batch k+1, kernels of batch k and fetching of batch k-1 overlap
 * on one set of batch buffers. */
int ema_engine_stage_async(ema_engine_t *e, int slot, const char *bases, const uint32_t *off, size_t n_pairs);
/* stage_async for buckets read by ema_bucket_read_device (include/ema_ingest.h) on the engine's device: they are laid end to end in
 * the slot in device memory (bucket k's pairs follow bucket k-1's); no read crosses the host.  The caller checks the read lengths. */
struct ema_bucket;
int ema_engine_stage_async_dev(ema_engine_t *e, int slot, const struct ema_bucket *const *buckets, size_t n_buckets);
int ema_engine_run_async(ema_engine_t *e, int slot, int *ticket);
int ema_engine_fetch_ticket(ema_engine_t *e, int ticket, ema_batch_out **out);
int ema_engine_run(ema_engine_t *e);
int ema_engine_run_serial(ema_engine_t *e);   /* ema_engine_run with the slices one after another: kernel times in isolation */
int ema_engine_sync(ema_engine_t *e);
int ema_engine_fetch(ema_engine_t *e, ema_batch_out **out);

/* Stage-level access used by the parity tests and the profiler (the staged batch must fit the full-capacity tier,
 * on which these run): seed intervals of the staged batch (what bwa's mem_collect_intv leaves in aux->mem).  Runs K1 only.
 * intv: 4 x u64 per interval {k, k', size, start<<32|end}; n_intv per read; caller frees with free().  With the k-mer table
 * (ema_engine_index_info: kmer_k > 0) K1 does not produce k' (0), and it may hand a single-occurrence interval over BY POSITION:
 * k' == 1 << 63, size == 1 and k is the occurrence's place in the text -- what bwt_sa() returns for the row (k_seed.hip, "anchors"). */
int ema_engine_debug_seeds(ema_engine_t *e, uint64_t **intv, int32_t **n_intv, int32_t *cap_per_read);

/* Regions of every staged read after seeding, chaining, extension and dedup (bwa's mem_align1_core result,
 * i.e. before mate rescue).  Runs K1 + K2.  regs: n_reads x cap records of reg_bytes bytes laid out as
 * {i64 rb, re; i32 qb, qe, rid, score, truesc, sub, csub, w, seedcov, secondary, seedlen0, n_comp, is_alt; f32 frac_rep};
 * caller frees the three arrays with free(). */
int ema_engine_debug_regions(ema_engine_t *e, void **regs, int32_t **n_regs, int32_t **status, int32_t *cap_per_read,
                             int32_t *reg_bytes);

/* The three wave DPs in isolation, one task per wavefront, for parity tests against the oracle
 * (the pipeline kernels call the same device code).  Sequences are nt4 codes (0..3, 4 = N); task t uses
 * qbuf[qoff[t]..qoff[t+1]) and tbuf[toff[t]..toff[t+1]).  kind 0 = extension (ksw_extend2;
 * prm[4t..] = w, end_bonus, zdrop, h0; out[6t..] = score, qle, tle, gtle, gscore, max_off), 1 = global
 * (ksw_global2; prm[t] = w; out[2t..] = score, n_cigar; cigar[t*cigar_cap..]), 2 = one local pass
 * (ksw_u8/i16; prm[3t..] = p (16|8), minsc, endsc; out[5t..] = score, te, qe, score2, te2). */
int ema_engine_debug_dp(ema_engine_t *e, int kind, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                        const uint32_t *toff, const int32_t *prm, int n_tasks, int32_t *out, uint32_t *cigar,
                        int cigar_cap);

/* The per-call entry points the libbwa-shaped face (include/ema_bwaabi.h) is built from; also used by its parity tests.
 * set_opts: new scoring/seeding/chaining options on an open engine (what the reference does by editing its mem_opt_t,
 *   src/align.c:184-185); batch geometry and capacities stay as opened.
 * debug_matesw: one mem_matesw call (reference src/bwabridge.c:267,281): anchor region, the mate (ASCII), the mate's regions
 *   ma[0..*n_ma) with room for cap, FR window; *n_sw = alignments run.  Region records as in ema_engine_debug_regions.
 * debug_final: mem_reg2aln (src/bwabridge.c:304) for n_regs given regions of one read: out[i] (region fields + pos, is_rev,
 *   NM, n_cigar, cigar_off into `cigar`). */
int ema_engine_set_opts(ema_engine_t *e, const ema_engine_opts *o);
/* Development knobs (no reference counterpart: the reference has no such switches): one comma-separated "key=value" string, read
 * when an engine is opened -- A/B switches, the parity tests' forced routes, profiling levels (keys: ema_amd/csrc/engine.hip,
 * ema_tuning_get call sites).  NULL returns to the default, which is the environment variable EMA_TUNING if
 * set and nothing otherwise.  Process-wide; the library reads no other EMA_* variable when it opens an engine and never changes
 * its host's environment (GPU_MAX_HW_QUEUES is the embedding program's to set: INTEGRATION.md). */
int ema_engine_set_tuning(const char *kv);
int ema_engine_debug_matesw(ema_engine_t *e, const void *anchor, const char *mate, int l_mate, void *ma, int32_t *n_ma, int cap,
                            int pes_low, int pes_high, int32_t *n_sw);
int ema_engine_debug_final(ema_engine_t *e, const char *read, int l_read, const void *regs, int n_regs, ema_cand_t *out, uint32_t *cigar,
                           int cigar_cap, int32_t *n_cigar_total);

/* Profiling aid: with EMA_PHASE_PROFILE=2 in the environment when the engine is opened, K2b logs one record per read
 * {read, intervals or -1, seed occurrences, chains, seeds, regions before dedup, extension DPs, shader clocks / 16};
 * this returns and resets the log (n records of 8 ints; caller frees). */
int ema_engine_debug_readlog(ema_engine_t *e, int32_t **log, size_t *n);

/* Region de-duplication in isolation (bwa's mem_sort_dedup_patch as mem_matesw calls it, i.e. without patching),
 * one task per wavefront: task t owns regs[t*cap .. t*cap + n_in[t]) (records laid out as in ema_engine_debug_regions);
 * sorted/compacted in place, n_out[t] = regions kept. */
int ema_engine_debug_dedup(ema_engine_t *e, void *regs, const int32_t *n_in, int32_t *n_out, int cap, int n_tasks);
/* The device's contig look-ups (bwa's bns_intv2rid / bns_pos2rid, reached from mem_chain and bns_fetch_seq) on a contig layout of
 * the caller's: ctg_off[0..n_seqs] = the contigs' first forward positions and, last, l_pac.  For query i -- the half-open
 * interval [rb[i], re[i]) in bwa's forward-reverse coordinates -- out[2i] = bns_intv2rid, out[2i+1] = bns_pos2rid of rb[i]'s
 * forward position.  (The engine answers them from a coarse table instead of bwa's bisection; tests/test_gpu_contigs.py.) */
int ema_engine_debug_contigs(ema_engine_t *e, const int64_t *ctg_off, int n_seqs, const int64_t *rb, const int64_t *re, int n, int32_t *out);
/* Rows [first, first + n) of the suffix array as it sits in HBM (text positions, widened to 64 bits): what bwa's bwt_sa() returns
 * for those rows.  tests/test_gpu_bwa_index.py holds an engine opened on a stock bwa index (sampled .sa, expanded on the device)
 * against one opened on this repo's flat .fsa. */
int ema_engine_debug_sa(ema_engine_t *e, uint64_t first, uint64_t n, uint64_t *out);

/* The host stage right behind the engine: what the reference's append_alignments() (src/align.c:986-1061) derives from a
 * pair's candidates before the records go on to the cloud stage -- clip filter (:1017,:1042), search-depth filter with
 * the best_dist it shares between the two mates (:1021-1024,:1046-1049), mem_approx_mapq_se_insist (:959-984),
 * score_alignment (:846-911), the `unique` flag (:1032,:1057).  Records come in the order the reference appends its
 * SAMRecords: per pair, mate 1's surviving candidates then mate 2's.  Pure host arithmetic (double precision), no GPU.
 * off: the read offsets the batch was aligned with (read lengths); error_rate: the platform's (reference src/techs.c: 0.001
 * for 10x).  Free with ema_aln_free(). */
typedef struct {
	uint32_t pair;
	uint8_t mate, unique, pad_[2];
	uint64_t cand;                 /* index into ema_batch_out.cand */
	int32_t clip, clip_edit_dist;  /* unaligned read bases; NM + clip */
	int32_t mapq, score_mapq;
	double score;                  /* log-likelihood of the alignment */
} ema_aln_rec;
typedef struct {
	size_t n_pairs, n;
	uint64_t *pair_off;            /* n_pairs + 1: records of pair p are rec[pair_off[p] .. pair_off[p+1]) */
	ema_aln_rec *rec;
} ema_aln_out;
int ema_batch_append_alignments(const ema_batch_out *b, const uint32_t *off, const ema_engine_opts *opts, double error_rate,
                                ema_aln_out **out);
void ema_aln_free(ema_aln_out *o);

/* mean launch duration of each kernel in the last ema_engine_run, ms (HIP events on the stream the kernel was launched on;
 * one launch per slice, launches of different slices overlap) */
typedef struct {
	float seed_ms, chain_ms, extend_ms, rescue_ms, final_ms, total_ms;
	float full_tier_ms;     /* K1..K4 of the full-capacity tier (one launch each) */
	float full_ms[4];       /* ... and each of them */
} ema_engine_timing;
int ema_engine_last_timing(ema_engine_t *e, ema_engine_timing *t);
int ema_engine_n_streams(const ema_engine_t *e);
int ema_engine_device(const ema_engine_t *e);          /* the HIP device the engine was opened on */
int ema_engine_seed_launches(const ema_engine_t *e);   /* launches of the seeding kernel per slice and run (its re-packing series) */

#ifdef __cplusplus
}
#endif
#endif
