/* include/ema_count.h -- C ABI of `ema count` (first half): barcode counting over an interleaved FASTQ
 * stream.
 *
 * Replaces count() (reference cpp/count.cc:38-182; declared cpp/count.h:19-23; called from src/main.c:239 with
 * max_map_size = 1 GB and the interleaved FASTQ on stdin).  For every read pair (8 lines: mate 1's name, bases, '+',
 * qualities, then mate 2's four lines, which are skipped) whose mate 1 has at least MIN_READ_SIZE = 32 bases:
 *   10x (is_haplotag = 0): the first BC_LEN = 16 bases are the barcode.  A quality character below '!' drops the pair
 *     (cpp/count.cc:113-117); qualities are capped at QUAL_BASE - 1 = 33 (:118-121).  The barcode's 2-bit code (N as A,
 *     cpp/common.h:76-89) is counted if it is on the whitelist and the barcode holds no N (:129-135); the barcode WITH its
 *     qualities -- 16 bytes, base code (N = 4) * 34 + capped quality (:123) -- is counted in an ordered map that is flushed to
 *     <prefix>.ema-fcnt as a block {int64 n; n x (16 bytes, int64 count)} whenever its estimated size (72 bytes per entry,
 *     cpp/common.h:110-115) reaches max_map_size, and once more at the end (:137-140, :171-175; dump_map :18-34).
 *   haplotag (is_haplotag = 1): the barcode is the BX:Z:AxxCxxBxxDxx tag of mate 1's name line (:91-103), packed as
 *     A << 24 | C << 16 | B << 8 | D (cpp/common.h:69-71); the whitelist is all 96^4 combinations (:56-59); no .ema-fcnt file.
 * <prefix>.ema-ncnt = {int64 n; n x (uint32 code, int64 count)} for the whitelisted barcodes seen at least once, in the
 * iteration order of the reference's std::unordered_map<uint32_t, int64_t> (:157-167) -- reproduced by keeping the counts in the
 * same container filled in the same order (same libstdc++: same order), so both files are byte-identical to the reference's.
 * Where the reference exits (unreadable whitelist, the all-A barcode on it, an output file that cannot be written) this
 * returns an error code; ema_count_last_error() has the text.  Host code (text in, two small files out); links into
 * libema_engine.so.
 */
#ifndef EMA_COUNT_H
#define EMA_COUNT_H

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
	int64_t total_reads;       /* pairs counted (":: Reads with OK barcode: nice out of total", cpp/count.cc:150) */
	int64_t nice_reads;        /* of those, pairs whose barcode is on the whitelist and holds no N */
	int64_t ignored_reads;     /* pairs dropped: mate 1 shorter than 32 bases, a quality below '!', no BX tag (haplotag) */
	int64_t bytes;             /* the reference's `sz`: line lengths + 1 over everything read */
	int64_t whitelist;         /* barcodes on the whitelist */
	int64_t nice_barcodes;     /* entries of <prefix>.ema-ncnt */
	int64_t full_blocks;       /* blocks written to <prefix>.ema-fcnt */
} ema_count_stats;

/* known_barcodes_path: the 10x whitelist, one barcode per line (ignored for haplotag); in_fd: the interleaved FASTQ (the
 * reference reads stdin: pass 0); output_prefix: <prefix>.ema-fcnt and <prefix>.ema-ncnt are written; max_map_size: bytes (the
 * reference passes 1 GB); st may be NULL. */
int ema_count_fastq(const char *known_barcodes_path, int in_fd, const char *output_prefix, size_t max_map_size, int is_haplotag,
                    ema_count_stats *st);
const char *ema_count_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
