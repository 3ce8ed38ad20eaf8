/* include/ema_sam.h -- C ABI of the SAM record formatter behind the hot path (the writer part).
 *
 * Replaces, on the reference's side, print_sam_record() (reference src/samrecord.c:104-284) as `ema align` calls it
 * for every selected alignment and its mate under the output lock (src/align.c:597-602): one fprintf/fputc at a time
 * there, here a whole batch of lines formatted on the host's cores into one buffer that the caller writes with a single
 * call.  Byte for byte the reference's text: flags, MAPQ = min(gamma-, score- and bwa-mapq) clamped to [0, 60]
 * (:139-146), CIGAR with hard clips shown as soft (:178-186, :270-276), mate fields and template length (:189-211),
 * reversed records reverse-complemented (:215-225), the NM / BX / XG / MI / XF tags in their 10x and haplotag forms
 * (:239-258, including the literal "-1" barcode suffix of a line that stands in for an unmapped read), RG up to the
 * first whitespace (:260-264) and XA (:266-279).  Host code only; clouds, EM and duplicate marking, which fill these
 * fields, stay with the caller.
 *
 * Where the reference asserts (a base outside ACGTN in a reversed read, :90-102; both records NULL, :110) the call
 * returns EMA_EFORMAT / EMA_EARG instead.
 */
#ifndef EMA_SAM_H
#define EMA_SAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef EMA_EARG
#define EMA_EARG (-1)
#endif
#ifndef EMA_EFORMAT
#define EMA_EFORMAT (-7)
#endif
#ifndef EMA_EIO
#define EMA_EIO (-6)
#endif

typedef struct ema_sam_alt {      /* struct xa, reference include/align.h:37-44 */
	const char *chrom;
	uint32_t pos;
	int32_t edit_dist, rev, n_cigar;
	const uint32_t *cigar;        /* BAM-packed: len << 4 | op */
} ema_sam_alt;

typedef struct ema_sam_rec {      /* what print_sam_record reads of one SAMRecord and the FASTQRecord, Cloud and
                                   * SingleReadAlignment it points at (reference include/samrecord.h:21-56) */
	const char *ident;            /* rec->ident */
	const char *chrom;            /* chrom_lookup(rec->chrom) */
	uint32_t chrom_id, pos;       /* rec->chrom, rec->pos */
	int32_t mapq, score_mapq;     /* rec->mapq (bwa's), rec->score_mapq */
	double gamma;                 /* rec->gamma */
	uint8_t mate, rev, duplicate, pad_;
	int32_t cloud_id, cloud_bad;  /* rec->cloud->id, ->bad */
	uint64_t bc;                  /* rec->bc */
	const char *read, *qual;      /* rec->fq->read, ->qual */
	int32_t read_len;             /* rec->fq->rlen */
	int32_t mate_read_len;        /* rec->fq_mate->rlen */
	const char *mate_read, *mate_qual;      /* rec->fq_mate: printed when this record's unmapped mate gets its line */
	int64_t aln_pos;              /* rec->aln.pos */
	int32_t aln_rev, edit_dist, n_cigar, pad2_;
	const uint32_t *cigar;        /* rec->aln.cigar */
	const ema_sam_alt *alts;      /* rec->alts, rec->n_alts */
	size_t n_alts;
} ema_sam_rec;

typedef struct ema_sam_line { const ema_sam_rec *rec, *mate; } ema_sam_line;      /* print_sam_record(rec, mate, ...): one may be NULL */

typedef struct ema_sam_opts {
	const char *rg_id;            /* NULL: no RG tag.  `ema align` without -R uses "@RG\tID:rg1\tSM:sample1" (reference src/main.c:25),
	                               * i.e. rg_id "rg1\tSM:sample1": the tag stops at the first whitespace */
	const char *bx_index;         /* the reference's global, "1" by default (src/main.c:26) */
	int32_t bc_len, is_haplotag;  /* BC_LEN; -p haplotag */
	int32_t insert_min, insert_max;      /* INSERT_MIN / INSERT_MAX of is_pair(), -35 / 750 (include/align.h:66-67) */
} ema_sam_opts;

void ema_sam_opts_default(ema_sam_opts *o);

/* Formats lines[0..n) in order.  *text (n_bytes long, not NUL-terminated) is freed with ema_sam_free(). */
int ema_sam_format(const ema_sam_line *lines, size_t n, const ema_sam_opts *o, char **text, size_t *n_bytes);
void ema_sam_free(char *text);

/* The same text written to an open file descriptor (the formatted pieces go out one after another, without being joined
 * first); *n_bytes, if not NULL, receives the number of bytes written.  EMA_EIO if a write fails. */
int ema_sam_write(int fd, const ema_sam_line *lines, size_t n, const ema_sam_opts *o, size_t *n_bytes);

/* ---- the same lines in compact form, for the formatter on the device (ema_sam_dev_*, below) ----
 * A selected record by index instead of by pointer: the device holds the bucket's text (names, bases, qualities, barcodes), the
 * batch's CIGAR operations and the contig names, so a line needs 52 bytes from the host.  The host keeps what needs libm or libc:
 * the printed MAPQ (min of the gamma-, score- and bwa-mapq clamped to [0, 60]: log10, reference src/samrecord.c:139-146) and the
 * "%.5g" text of gamma (:247). */
typedef struct ema_sam_desc {
	uint32_t pair;                /* pair index in the bucket: name, reads, barcode */
	int32_t rid;                  /* rec->chrom */
	uint32_t pos;                 /* rec->pos (1-based); rec->aln.pos is pos - 1 */
	uint32_t cigar_off;           /* first operation in the batch's CIGAR array */
	int32_t n_cigar, edit_dist;
	int32_t cloud_id;
	int32_t xa;                   /* index of the XA entry, -1: none */
	uint8_t mate, rev, duplicate, cloud_bad;
	uint8_t mapq;                 /* as printed */
	uint8_t gamma_len;            /* bytes of gamma[] */
	uint8_t has_mate;             /* on the first record of a selected pair: the next descriptor is its mate's */
	uint8_t pad_;
	char gamma[12];               /* "%.5g" of rec->gamma */
} ema_sam_desc;
typedef struct ema_sam_xa { int32_t rid; uint32_t pos; uint32_t cigar_off; int32_t n_cigar, edit_dist, rev; } ema_sam_xa;

/* The formatter as kernels (ema_amd/csrc/k_sam.hip): one lane renders one line -- a length pass, a prefix sum, a writing pass into
 * one text buffer -- from the bucket's arrays uploaded as they are, the compact records above and the batch's CIGAR operations
 * [cigar_lo, cigar_hi) (cigar points at operation cigar_lo).  sel_at[i] is the first descriptor of selected pair i, which prints two
 * lines: (rec, mate) and (mate, rec), the second one standing in for an unaligned mate when there is no mate record.  The text
 * comes back to a page-locked buffer and goes to fd with write(); byte for byte ema_sam_write()'s.  contig_names as for
 * ema_clouds_select.  EMA_EFORMAT as ema_sam_write; EMA_EARG on inconsistent input; a HIP failure is EMA_EIO with the message in
 * ema_sam_dev_last_error(). */
struct ema_bucket;
typedef struct ema_sam_dev ema_sam_dev_t;
int ema_sam_dev_open(int device, const char *const *contig_names, int32_t n_contigs, ema_sam_dev_t **out);
void ema_sam_dev_close(ema_sam_dev_t *d);
int ema_sam_dev_write(ema_sam_dev_t *d, int fd, const struct ema_bucket *bk, const uint32_t *cigar, uint64_t cigar_lo, uint64_t cigar_hi,
                      const ema_sam_desc *descs, size_t n_descs, const ema_sam_xa *xas, size_t n_xas, const uint32_t *sel_at, size_t n_sel,
                      const ema_sam_opts *o, size_t *n_bytes);
/* the same, the text returned instead of written (tests); free with ema_sam_free() */
int ema_sam_dev_format(ema_sam_dev_t *d, const struct ema_bucket *bk, const uint32_t *cigar, uint64_t cigar_lo, uint64_t cigar_hi,
                       const ema_sam_desc *descs, size_t n_descs, const ema_sam_xa *xas, size_t n_xas, const uint32_t *sel_at, size_t n_sel,
                       const ema_sam_opts *o, char **text, size_t *n_bytes);
const char *ema_sam_dev_last_error(void);

/* The header write_sam_header() prints (reference src/align.c:192-211): @HD VN:1.3 SO:unsorted, one @SQ per contig,
 * the read-group line as given (NULL: none), and @PG ID:ema PN:ema VN:<version> CL:<argv joined by spaces>.  Contig
 * names and lengths are what ema_engine_contig_name / _contig_len return.  *text is freed with ema_sam_free(). */
int ema_sam_header(const char *const *contig_names, const int32_t *contig_lens, int32_t n_contigs, const char *rg_line,
                   const char *version, int argc, const char *const *argv, char **text, size_t *n_bytes);

#ifdef __cplusplus
}
#endif
#endif
