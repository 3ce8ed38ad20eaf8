/* include/ema_preproc.h -- C ABI of `ema preproc` (second half): barcode correction and bucketing of an
 * interleaved FASTQ stream.
 *
 * Replaces correct() (reference cpp/correct.cc:271-633; declared cpp/correct.h:25-35; called from src/main.c:201 with a
 * 10 MB buffer per bucket and the interleaved FASTQ on stdin), given the files `ema count` wrote (include/ema_count.h):
 *   1. the whitelist (or all 96^4 haplotag codes) with, per barcode, the prior (count + 1) / sum over the whitelist, from the
 *      .ema-ncnt files (:283-330; load_barcode_count :188-206);
 *   2. (10x) every barcode-with-qualities string of the .ema-fcnt files is corrected (correct_barcode :66-184): kept if it is
 *      on the whitelist (with do_h2, after weighing every two-base neighbour, :107-134), else replaced by the whitelisted
 *      barcode one base away -- or, with one N, by the candidates at the N -- of largest prior x error probability, accepted when
 *      its share of the candidates' total exceeds BC_CONF_THRESH = 0.975 (:157-164); accepted barcodes collect their reads'
 *      counts (:172-180);
 *   3. barcodes are dealt to n_buckets files, each to the currently smallest (ties: lowest number), in the iteration order of
 *      the reference's std::unordered_map (:389-395);
 *   4. the stream is read again: every pair whose mate 1 has >= 32 bases (and passes the quality test of cpp/count.cc) goes, with
 *      its corrected barcode, 16 + 7 bases trimmed off mate 1 (haplotag: untrimmed), to its barcode's bucket as one line
 *      `BC NAME R1 Q1 R2 Q2` (or, with do_bx_format, as interleaved FASTQ with BX:Z: tags), or to ema-nobc as interleaved FASTQ
 *      when it has no whitelisted barcode (:427-617).
 * Output: <output_dir>/ema-bin-000 .. and <output_dir>/ema-nobc, byte-identical to the reference's for well-formed input: the
 * containers whose iteration order decides floating-point sums and the deal are the reference's own kinds filled in the same
 * order, and every double-precision expression is the reference's, in its order (same libm).  Reproduced on purpose: in
 * haplotag mode the reference tests the BX tag's position against the length of the PREVIOUS pair's last line (cpp/correct.cc:446
 * uses `s`, not `n`), so the first pair of a haplotag stream is always dropped; a quality line LONGER than
 * its read is cut to the read's length (the reference copies one length and advances by the other, :558-565: its next write
 * overwrites the surplus); a stream that ends inside a pair without a final line end is written from the strings the reference's
 * failed getlines leave untouched (the previous pair's lines, :427-430,573,596,607-608; golden vectors cut_short_*).  Not
 * reproduced: a quality line SHORTER than its read leaves a gap of whatever the reference's buffer held; that is EMA_EFORMAT
 * here.  Host code; links into libema_engine.so.
 */
#ifndef EMA_PREPROC_H
#define EMA_PREPROC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef EMA_OK
#define EMA_OK 0
#endif
#ifndef EMA_EARG
#define EMA_EARG (-1)
#endif
#ifndef EMA_EIO
#define EMA_EIO (-6)
#endif
#ifndef EMA_EFORMAT
#define EMA_EFORMAT (-7)
#endif

typedef struct {
	int64_t no_change, no_barcode, h1_corrected, h2_corrected;   /* reads by what became of their barcode (":: Stats:", cpp/correct.cc:352-357) */
	int64_t corrected_strings;                                   /* barcode-with-qualities strings that map to another barcode */
	int64_t pairs_written, pairs_nobc, pairs_skipped;            /* to the buckets; to ema-nobc; dropped (short mate 1, low quality, no BX tag) */
	int64_t whitelist;
} ema_preproc_stats;

/* ncnt_paths[0..n_paths): the <prefix>.ema-ncnt files of `ema count` (the .ema-fcnt file beside each is read too, 10x only);
 * output_dir is created if missing; in_fd: the interleaved FASTQ (the reference reads stdin: pass 0); buffer_size: bytes buffered
 * per bucket before a write (the reference passes 10 MB); n_threads: threads of the correction step (results do not depend on
 * it); st may be NULL. */
int ema_preproc_fastq(const char *known_barcodes_path, const char *const *ncnt_paths, int n_paths, const char *output_dir, int do_h2,
                      size_t buffer_size, int do_bx_format, int n_threads, int n_buckets, int is_haplotag, int in_fd,
                      ema_preproc_stats *st);
const char *ema_preproc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
