/* include/ema_ingest.h -- C ABI of the bucket reader in front of the hot path.
 *
 * Replaces, on the reference's side, read_special_fastq() (reference src/align.c:759-806) together with the grouping
 * that seek_next_barcode_group() (src/align.c:808-843) does on its result: one barcode bucket in the "special FASTQ"
 * form written by `ema preproc` -- one pair per line,
 *
 *     BARCODE ID READ1 QUAL1 READ2 QUAL2
 *
 * fields separated by one whitespace character (copy_until_space, src/util.c:11-21) -- is read, ordered by its first
 * bc_len bytes exactly as the reference's qsort()/strncmp() comparison orders the lines (src/align.c:751-757, :773),
 * pairs of equal key staying in file order, and laid out as what ema_engine_align_pairs() takes: all reads in one
 * byte array, read 2i = mate 1 of pair i, read 2i+1 = its mate 2.  Barcodes are encoded as the reference does
 * (encode_bc, src/util.c:41-76).  Host code only; links into libema_engine.so beside the engine.
 *
 * Where the reference has undefined behaviour the reader fails with EMA_EFORMAT and names the line instead:
 * a line of 5000 bytes or more (fgets splits it, src/align.c:762,768), fewer than six fields, an identifier that is
 * empty (it would read as the end-of-array sentinel, include/samrecord.h:17) or longer than 149 bytes (id[150],
 * include/samrecord.h:12), a read longer than max_read_len or whose quality string has another length, a barcode field
 * that is not bc_len bytes of ACGT/acgt (assert, src/util.c:54) or, for haplotag, not of the form AddCddBddDdd.
 */
#ifndef EMA_INGEST_H
#define EMA_INGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef EMA_EARG
#define EMA_EARG (-1)       /* bad argument (same value as in ema_engine.h) */
#endif
#define EMA_EIO (-6)        /* file cannot be opened or read */
#define EMA_EFORMAT (-7)    /* malformed bucket; ema_bucket_last_error() names the line */

typedef struct ema_bucket {
	size_t n_pairs;
	size_t n_groups;        /* barcode groups: maximal runs of equal encoded barcode, as seek_next_barcode_group() finds them */
	uint64_t *group_off;    /* n_groups + 1: group g is pairs [group_off[g], group_off[g+1]) */
	uint64_t *bc;           /* n_pairs: encoded barcode (bc_t, reference include/util.h:21) */
	uint32_t *off;          /* 2*n_pairs + 1: read r is bases[off[r] .. off[r+1]), and quals at the same offsets */
	char *bases;            /* ASCII, as in the file */
	char *quals;
	uint32_t *id_off;       /* n_pairs + 1 */
	char *ids;              /* identifiers back to back (both mates share one, src/align.c:790) */
	void *dev;              /* private to the library.  NULL: every array above is in host memory.  Set by ema_bucket_read_device: bases and
	                         * quals are NULL on the host and live in device memory (with copies of the other arrays), where the engine's
	                         * staging and the SAM formatter read them; ema_bucket_dev_fetch copies them back */
} ema_bucket;

/* Reads a whole bucket file.  bc_len = the platform's barcode length (reference src/techs.c:74-119; 16 for 10x),
 * 1..32; is_haplotag selects encode_bc_haplotag (bc_len 12); max_read_len = longest read accepted (the reference's
 * MAX_READ_LEN is 200, include/align.h:61; the engine takes up to 255).  *out is freed with ema_bucket_free(). */
int ema_bucket_read(const char *path, int bc_len, int is_haplotag, int max_read_len, ema_bucket **out);

/* The same -- read_special_fastq, reference src/align.c:759-806 -- with the parsing on the device (csrc/ingest_dev.hip): the file is read
 * into page-locked memory and uploaded as it is;
 * line table, field scan with the reader's checks, stable radix sort by barcode, prefix sums and the gather of bases / qualities /
 * names into their sorted places are kernels.  The bucket that comes back is the one ema_bucket_read returns, except that bases and
 * quals stay on the device (bucket->dev; NULL on the host): ema_stream_sam hands them to the engine and to the SAM formatter there.
 * 10x-style and haplotag barcodes.  Whatever the kernels do not take -- a line the checks refuse, a NUL byte, 4 GB of text -- is
 * read by ema_bucket_read instead (then bases / quals are host arrays and dev is NULL): same result, same error texts. */
int ema_bucket_read_device(const char *path, int bc_len, int is_haplotag, int max_read_len, int device, ema_bucket **out);
/* bases / quals of a device-resident bucket copied to the caller's arrays (off[2 * n_pairs] bytes each); EMA_EARG if dev is NULL */
int ema_bucket_dev_fetch(const ema_bucket *b, char *bases, char *quals);
const char *ema_bucket_dev_last_error(void);

/* The same on a bucket already in memory (text[0 .. len)); text is not modified and need not end in a newline or NUL. */
int ema_bucket_parse(const char *text, size_t len, int bc_len, int is_haplotag, int max_read_len, ema_bucket **out);

/* `ema align -1 a.fq [-2 b.fq]` (reference src/align.c:637-744, src/techs.c:5-69): barcode-sorted FASTQ, the barcode in the read
 * name.  path2 == NULL: one file with the mates interleaved (read_fastq_rec_bc_group_interleaved), else mate 1 in path1 and mate 2
 * in path2 (read_fastq_rec_bc_group on each).  Per record, as extract_bc_10x / _haplotag do: the barcode is the bc_len characters
 * after the LAST ':' of the name line, the identifier is the name without its '@', cut at that ':' and at the first blank
 * (Long Ranger style names); name_style 1 = tellseq (src/techs.c:31-54: a " BX:Z:" comment carries the barcode); name_style 2 =
 * TruSeq SLR (extract_bc_truseq: the barcode is atoi() of the name behind its '@', the name stays whole) and 3 = CPT-seq
 * (extract_bc_cptseq: the name is cut at its last ':' and the barcode is atoi() of what follows that ':' and two more
 * characters) -- integer barcodes, bc_len 0 as in the reference's platform table.  Barcode groups are
 * the runs of equal barcode IN FILE ORDER -- the reference trusts the input to be sorted and so does this reader -- and both
 * mates of a pair must carry the same barcode and the same identifier (the reference asserts the former, src/align.c:708,733).
 * The result is laid out as ema_bucket_read's.  EMA_EFORMAT names the record where the reference would assert or read past a
 * buffer (truncated record, name of 150 bytes or more, read beyond max_read_len, quality string of another length, bad barcode). */
int ema_fastq_read(const char *path1, const char *path2, int name_style, int bc_len, int is_haplotag, int max_read_len, ema_bucket **out);

void ema_bucket_free(ema_bucket *b);

/* Message of the last failed call on this thread ("" if none). */
const char *ema_bucket_last_error(void);

/* encode_bc / decode_bc (reference src/util.c:41-95) for one barcode; encode returns EMA_EFORMAT on a bad barcode.
 * decode writes bc_len (or 12) bytes and no terminator. */
int ema_barcode_encode(const char *bc, int bc_len, int is_haplotag, uint64_t *out);
void ema_barcode_decode(uint64_t bc, int bc_len, int is_haplotag, char *out);

#ifdef __cplusplus
}
#endif
#endif
