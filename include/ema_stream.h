/* include/ema_stream.h -- C ABI of the bucket loop around the hot path: many barcode buckets through reader -> engine ->
 * append stage on one GPU, pipelined, results handed over in input order.
 *
 * Replaces the reference's outer loops: `ema align -s bucket` handles one bucket per process (reference src/main.c:380-394),
 * `-x` walks a list of them one after another (src/main.c:396-406); inside, find_clouds_and_align() reads the whole bucket
 * (read_special_fastq, src/align.c:258) and calls append_alignments() for every pair (src/align.c:307-349).  Here the same
 * three steps -- ema_bucket_read (include/ema_ingest.h), ema_engine_align_pairs, ema_batch_append_alignments
 * (include/ema_engine.h) -- run as a pipeline over the list: a reader thread parses buckets ahead; a stager converts and
 * uploads the next pass's input; the engine thread queues asynchronous passes on ONE set of batch buffers (up to
 * EMA_MAX_INFLIGHT in flight), fetches the oldest and runs its append stage while the younger ones compute; and the caller's
 * sink sees bucket 0, 1, 2, ... in order, each with its candidates (ema_batch_out) and its append_alignments records
 * (ema_aln_out).  Small buckets that are waiting behind one another share a pass (laid end to end up to the batch capacity,
 * the results cut apart again): the engine's kernels want batches of a million pairs, preproc's buckets are 100-200 K.
 * BASELINE configs[2] (500 buckets streamed on one GPU) is this call; with G GPUs each process calls it on its own buckets
 * (b mod G, no data-path collective).
 * Host code only; links into libema_engine.so.
 */
#ifndef EMA_STREAM_H
#define EMA_STREAM_H

#include <stddef.h>
#include <stdint.h>
#include "ema_engine.h"
#include "ema_ingest.h"
#include "ema_clouds.h"
#include "ema_sam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int bc_len, is_haplotag, max_read_len;   /* bucket reader: the platform's barcode (reference src/techs.c:74-119), longest read */
	double error_rate;                       /* append stage: the platform's error rate (reference src/techs.c; 0.001 for 10x) */
	int n_engines;                           /* 0 / 1 (default) = one set of batch buffers, asynchronous passes (ema_engine_run_async): staging of
	                                          * bucket k+1, kernels of bucket k, fetch + append of bucket k-1 overlap; 2 = the older schedule, alternate
	                                          * buckets on the engine and its ema_engine_peer(), one pass each */
	int read_ahead;                          /* buckets parsed ahead of the engine (0 = default 2) -- or, with small buckets, as many as make up
	                                          * that many full batches */
	int fastq_input;                         /* 0: paths are bucket files (-s / -x); 1: barcode-sorted FASTQ as `ema align -1 [-2]` takes it
	                                          * (ema_fastq_read, include/ema_ingest.h): paths[k] and, unless NULL, paths2[k] */
	int fastq_name_style;                    /* ema_fastq_read's name_style */
	const char *const *paths2;               /* fastq_input: mate-2 files, or NULL for interleaved input */
} ema_stream_opts;
void ema_stream_opts_default(ema_stream_opts *o);   /* 16, 0, 255, 0.001, 0, 0 */

/* per-bucket statistics (the record the ranks gather at the end of a multi-GPU run) */
typedef struct {
	uint64_t pairs, candidates, reads_with_candidates, records, unique_records, redone_pairs, barcode_groups;
	uint64_t mapq_hist[7];                   /* append-stage MAPQ of the records: [0], 1-9, 10-19, 20-29, 30-39, 40-59, >= 60 */
	int32_t capacity_flags, rc;              /* OR of the reads' status bits; the bucket's return code */
	double read_s, align_s, append_s;        /* wall seconds inside the three steps (they overlap across buckets) */
	float seed_ms, extend_ms, rescue_ms, final_ms, full_tier_ms;   /* this bucket's kernel launches (ema_engine_timing; 0 for a bucket beyond one batch) */
	float pad_;
} ema_bucket_stats;

/* Called once per bucket, in input order, on the calling thread.  `bk` is the parsed bucket (NULL for ema_stream_batches),
 * `b` its candidates, `a` its append_alignments records; all three are freed when the sink returns.  A non-zero return
 * stops the stream and becomes its return code. */
typedef int (*ema_stream_sink)(void *user, size_t index, const ema_bucket *bk, const ema_batch_out *b, const ema_aln_out *a);

/* paths[0..n): bucket files in `ema preproc`'s special-FASTQ form.  stats: n entries or NULL.  Returns 0, the first failing
 * bucket's code (EMA_EIO / EMA_EFORMAT / EMA_E*; ema_stream_last_error() has the text), or the sink's code.  EMA_ELIMIT
 * buckets (a read over an engine capacity) are still delivered, flags in stats[k].capacity_flags, and the call ends with EMA_ELIMIT. */
int ema_stream_buckets(ema_engine_t *e, const char *const *paths, size_t n, const ema_stream_opts *o, ema_stream_sink sink,
                       void *user, ema_bucket_stats *stats);

/* The same pipeline on batches already in host memory (no reader): batch k = bases[k], off[k] (2*n_pairs[k]+1 entries), as
 * ema_engine_align_pairs takes them. */
int ema_stream_batches(ema_engine_t *e, const char *const *bases, const uint32_t *const *off, const size_t *n_pairs, size_t n,
                       const ema_stream_opts *o, ema_stream_sink sink, void *user, ema_bucket_stats *stats);

/* The same pipeline on batches that are already in HBM (ema_engine_stage_slot): with s sets of batch buffers in use (1; 2 when
 * o->n_engines == 2 and the peer could be created), batch k must have been staged into slot (k / s) % slots_per_set of
 * set k % s (set 0 = e, set 1 = ema_engine_peer(e)); off[k] are its read offsets (the append stage needs the read lengths).
 * Nothing crosses PCIe towards the device: this is the rate with inputs resident, outputs delivered to the host. */
int ema_stream_resident(ema_engine_t *e, const uint32_t *const *off, const size_t *n_pairs, size_t n, int slots_per_set,
                        const ema_stream_opts *o, ema_stream_sink sink, void *user, ema_bucket_stats *stats);

/* Bucket files to SAM text: the whole `ema align -s` / `-x` body (reference src/main.c:380-406 -> find_clouds_and_align,
 * src/align.c:213-628) through this library's C-ABI stages only -- ema_stream_buckets (reader, engine, append stage) with a
 * sink that runs ema_clouds_select (include/ema_clouds.h) and ema_sam_write (include/ema_sam.h) on every bucket, in order, to
 * the file descriptor fd.  The header is the caller's (ema_sam_header).  continue_cloud_ids = 0: every bucket numbers its
 * clouds from clouds.first_cloud_id, as one `ema align -s bucket` process per bucket does; 1: the count runs on from bucket
 * to bucket, as `-x` does.  bstats / sstats: n entries each or NULL. */
typedef struct {
	ema_stream_opts stream;
	ema_cloud_opts clouds;
	ema_sam_opts sam;
	int32_t continue_cloud_ids;
} ema_sam_run_opts;
void ema_sam_run_opts_default(ema_sam_run_opts *o);
/* The defaults, then what `ema align -p <platform>` takes from the reference's platform table (get_platform_profile_by_name,
 * src/techs.c:74-135): barcode length, whether barcodes are haplotag codes, the error rate of the append stage, the cloud
 * distance threshold and the many-clouds mode.  name: "10x", "haplotag", "dbs", "tellseq" (bucketed text with 16 / 12 / 20 / 18
 * base barcodes), "tru", "cpt" (many-clouds platforms: 15 kb / 3.5 kb clouds, integer barcodes in the FASTQ names: `-1 / -2`
 * input only, bc_len 0; cpt also brings its own density model for -d).  stream.fastq_name_style is set to what
 * ema_fastq_read needs for the platform's names.  EMA_EARG for any other name. */
int ema_sam_run_opts_platform(const char *name, ema_sam_run_opts *o);
int ema_stream_sam(ema_engine_t *e, const char *const *paths, size_t n, const ema_sam_run_opts *o, int fd, ema_bucket_stats *bstats,
                   ema_sam_stats *sstats);

const char *ema_stream_last_error(void);   /* of the last failed call on this thread */

/* Diagnostics.  CPU seconds (user + system, summed over every thread that worked for the stage) the host stages of this library
 * have used in this process since it was loaded or since the last call with reset != 0:  out[0] bucket / FASTQ reader,
 * [1] staging (nt4 conversion, 2-bit packing, upload), [2] fetch (waiting for a pass, download, result assembly, cutting a shared
 * pass into buckets), [3] the append_alignments stage, [4] clouds / EM / duplicate marking, [5] formatter and write.
 * The stages of a stream overlap in time and spread over many short-lived threads: wall clocks cannot say where the cores went. */
void ema_host_cpu_seconds(double out[6], int reset);

#ifdef __cplusplus
}
#endif
#endif
