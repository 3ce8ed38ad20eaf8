/* include/ema_clouds.h -- C ABI of the cloud / EM / duplicate-marking stage behind the hot path.
 *
 * Replaces what find_clouds_and_align() does with every barcode group once append_alignments() has produced its records
 * (reference src/align.c:347-608) together with the dictionary it works through (src/samdict.c:11-243): the records of the
 * group are ordered by (chromosome, position, identifier) (record_cmp, src/samrecord.c:51-73), swept into "clouds" of
 * alignments no further than dist_thresh apart (:358-408; a cloud that holds two candidates of one read is marked bad and
 * re-entered by read name), every read's candidate posteriors (gammas) are initialised from the alignment likelihoods and
 * refined by EM_ITERS = 5 rounds against the cloud weights and the best consistent mate (:411-525; full EM from 30 read
 * pairs per barcode on), the best candidate of every read and of its mate is selected with an XA entry for a close
 * runner-up (find_best_record, src/samdict.c:177-243), duplicates are marked among the selected records
 * (dup_cmp, src/align.c:85-122, :575-585) and the pairs are printed in that order (:587-603).
 *
 * Here: one call per bucket.  Barcode groups are independent, so they are worked through by the host's threads; the result
 * is the reference's `-t 1` output order (groups in bucket order, pairs in the duplicate comparator's order), and the cloud
 * numbers printed as MI:i are those of a single-threaded run: group g's clouds are numbered from first_cloud_id + the number
 * of clouds of all earlier groups, which is known after the sweep (the reference's static counter, src/align.c:19-23, makes
 * them depend on thread timing under -t N).  All floating-point steps are the reference's expressions in
 * the reference's order, in double precision, without contraction.  Not included: -d (mark_optimal_alignments_in_cloud,
 * src/split.c), which seeds rand() from the wall clock in the reference.
 *
 * The output is the formatter's input (include/ema_sam.h): lines[] are (record, mate) pairs ready for ema_sam_write();
 * they point into the bucket, the batch and this object, which must all outlive their use.  Host code only.
 */
#ifndef EMA_CLOUDS_H
#define EMA_CLOUDS_H

#include <stddef.h>
#include <stdint.h>
#include "ema_engine.h"
#include "ema_ingest.h"
#include "ema_sam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	uint32_t dist_thresh;     /* tech->dist_thresh (reference src/techs.c:74-119): 50000 for 10x, haplotag, dbs, tellseq */
	int32_t many_clouds;      /* tech->many_clouds (tru, cpt): per-read cloud weights, no cloud sets, no duplicate marking */
	int32_t n_threads;        /* host threads over barcode groups (0 = default: min(32, hardware threads)) */
	int32_t first_cloud_id;   /* the reference's cloud counter when this bucket starts: 0 for `ema align -s bucket` */
	/* `ema align -d` (src/split.c:38-338, hook at src/align.c:396-397): in a cloud with a read-name collision, simulated annealing over
	 * the multi-mapped reads' alignments against the platform's read-density model decides which alignments stay `active`.  The
	 * reference draws its moves from libc's rand(), seeded ONCE per process from time(); this library draws from the same rand(), so the
	 * result equals the reference's for the same seed (ema_clouds_reseed) and the same order of work: the barcode groups that reach the
	 * optimiser (the ones with a bad cloud) run on ONE thread, in group order, after the others -- which draw nothing and run on the
	 * host's threads as without -d -- so the draws fall as in a `-t 1` run (the reference's own -d under -t N is not reproducible: SURVEY 0.5-1) */
	int32_t density_opt;      /* -d */
	int32_t n_density_probs;  /* tech->n_density_probs, tech->density_probs (src/techs.c:74-127): 4 x {0.6, 0.05, 0.2, 0.01} for 10x */
	double density_probs[16];
	int32_t emit;             /* what ema_clouds_out carries: 0 lines / recs / alts / idents (ema_sam_write's input); 1 the compact form only
	                           * (descs / xas / sel_at: ema_sam_dev_write's input -- no per-record structs, no copies of the names); 2 both */
	/* -d, where the draws come from.  0: the process's libc rand() stream (ema_clouds_reseed), shared by every call -- ONE reference process
	 * over everything this process prints, the `-x` run.  1: this call draws from a stream of its own, seeded with `seed` exactly as
	 * srand(seed) seeds rand() (glibc's random_r on a 128-byte state is the generator behind rand()) -- the bucket as its own
	 * `ema align -s bucket` process, which is how the reference runs buckets in parallel (README.md:127-130; every process seeds once, from
	 * time(), at its first bad cloud: src/split.c:54-59).  Calls with streams of their own may run concurrently. */
	int32_t seed_private;
	uint32_t seed;
	int32_t pad_;
} ema_cloud_opts;
void ema_cloud_opts_default(ema_cloud_opts *o);   /* 50000, 0, 0, 0; no -d, the 10x density model, the process's rand() stream */
/* srand(seed) for -d, as the reference's first bad cloud does with time(NULL) (src/split.c:54-59); without a call the library seeds
 * from the clock at its first use, like the reference */
void ema_clouds_reseed(unsigned seed);

/* per-bucket SAM statistics (the record the ranks gather) */
typedef struct {
	uint64_t groups, clouds, bad_clouds;      /* barcode groups with records; clouds found; clouds with a read-name collision */
	uint64_t lines, mapped, unmapped_mates;   /* SAM lines; lines of aligned records; lines standing in for an unaligned mate */
	uint64_t proper, duplicates, with_xa;     /* FLAG 0x2 lines; FLAG 0x400 lines; lines with an XA tag */
	uint64_t mapq_hist[7];                    /* printed MAPQ of the mapped lines: 0, 1-9, 10-19, 20-29, 30-39, 40-59, 60 */
	double select_s, write_s;                 /* wall seconds in ema_clouds_select; in ema_sam_write (set by ema_stream_sam) */
} ema_sam_stats;

typedef struct ema_clouds_out {
	size_t n_lines;
	ema_sam_line *lines;      /* print order: print_sam_record(rec, mate) then print_sam_record(mate, rec) per selected pair */
	size_t n_recs;
	ema_sam_rec *recs;        /* the selected records the lines point at */
	ema_sam_alt *alts;        /* their XA entries */
	char *idents;             /* NUL-terminated read names */
	int32_t next_cloud_id;    /* the cloud counter after this bucket (first_cloud_id of the next one in an -x run) */
	ema_sam_stats stats;
	/* emit 1 / 2: the selected records by index (include/ema_sam.h), in the order of recs[] */
	size_t n_descs, n_xas, n_sel;
	ema_sam_desc *descs;
	ema_sam_xa *xas;
	uint32_t *sel_at;         /* n_sel: first descriptor of selected pair i (lines 2i and 2i+1) */
	uint64_t cigar_lo, cigar_hi;      /* the CIGAR operations the descriptors and XA entries name lie in [cigar_lo, cigar_hi) of the batch's array */
} ema_clouds_out;

/* bk: the bucket as read (barcodes, groups, names, reads, qualities); b, a: its candidates and append_alignments records
 * (ema_engine_align_pairs / ema_batch_append_alignments, or ema_stream_buckets' sink arguments); contig_names[rid].
 * EMA_EARG on inconsistent inputs; EMA_EFORMAT where the reference would assert (an XA source with 64 or more CIGAR
 * operations, src/samdict.c:217).  *out is freed with ema_clouds_free(). */
int ema_clouds_select(const ema_bucket *bk, const ema_batch_out *b, const ema_aln_out *a, const char *const *contig_names,
                      int32_t n_contigs, const ema_cloud_opts *o, ema_clouds_out **out);
void ema_clouds_free(ema_clouds_out *out);

#ifdef __cplusplus
}
#endif
#endif
