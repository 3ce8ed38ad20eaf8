/*
 * oracle/ingest.c -- CPU oracle for the bucket reader in front of the hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Unlike the bwa half, the code restated here IS in the reference tree, so each function follows a cited original, one
 * line at a time on one thread as the reference does it:
 *   src/util.c:11-21     copy_until_space     -> orc_copy_until_space()
 *   src/util.c:41-76     encode_bc*           -> orc_encode_bc()
 *   src/util.c:78-95     decode_bc*           -> orc_decode_bc()
 *   src/align.c:751-806  read_special_fastq   -> orc_read_special_fastq()
 *   src/align.c:808-843  seek_next_barcode_group -> orc_next_group()
 * Pinned where the reference can run here: tests/test_ingest.py checks the three util.c restatements against
 * oracle/_ref/libref_util.so, which is the reference's own src/util.c compiled as it lies (oracle/Makefile).
 * read_special_fastq itself is a static function of src/align.c, which cannot be compiled here (it includes bwa's
 * headers); its order among lines of equal barcode is qsort()'s, which C leaves unspecified -- this restatement keeps
 * such lines in file order (what glibc's merge-sort qsort does).  Inputs are assumed well-formed (the reference has
 * undefined behaviour otherwise); malformed lines trip an assertion.
 */
#include <assert.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

void orc_copy_until_space(char *dest, char **src)
{
	size_t i = 0;
	while (**src && !isspace((unsigned char)**src)) { dest[i++] = **src; ++*src; }
	dest[i] = '\0';
	++*src;      /* the separator (or the terminator) is stepped over */
}

uint64_t orc_encode_bc(const char *bc, int bc_len, int is_haplotag)
{
	if (is_haplotag) {      /* src/util.c:63-70: A<a>C<c>B<b>D<d> -> a<<24 | c<<16 | b<<8 | d */
		const int a = 10 * (bc[1] - '0') + (bc[2] - '0'), c = 10 * (bc[4] - '0') + (bc[5] - '0');
		const int b = 10 * (bc[7] - '0') + (bc[8] - '0'), d = 10 * (bc[10] - '0') + (bc[11] - '0');
		return (uint64_t)(((uint32_t)a << 24) | ((uint32_t)c << 16) | ((uint32_t)b << 8) | (uint32_t)d);
	}
	uint64_t v = 0;
	for (int i = 0; i < bc_len; ++i) {      /* src/util.c:45-56: walks from the last base down */
		const char ch = bc[bc_len - 1 - i];
		v <<= 2;
		switch (ch) {
		case 'A': case 'a': v |= 0; break;
		case 'C': case 'c': v |= 1; break;
		case 'G': case 'g': v |= 2; break;
		case 'T': case 't': v |= 3; break;
		default: assert(0 && "barcode base outside ACGT");
		}
	}
	return v;
}

void orc_decode_bc(uint64_t bc, int bc_len, int is_haplotag, char *out)
{
	if (is_haplotag) {
		sprintf(out, "A%02uC%02uB%02uD%02u", (unsigned)((bc >> 24) & 127), (unsigned)((bc >> 16) & 127),
		        (unsigned)((bc >> 8) & 127), (unsigned)(bc & 127));
		return;
	}
	for (int i = 0; i < bc_len; ++i) { out[i] = "ACGT"[bc & 3]; bc >>= 2; }
}

/* stable merge sort of line pointers by strncmp(.., bc_len) (src/align.c:751-757, :773) */
static void sort_lines(char **a, char **tmp, size_t n, int bc_len)
{
	if (n < 2) return;
	const size_t h = n / 2;
	sort_lines(a, tmp, h, bc_len);
	sort_lines(a + h, tmp, n - h, bc_len);
	size_t i = 0, j = h, k = 0;
	while (i < h && j < n) tmp[k++] = strncmp(a[j], a[i], (size_t)bc_len) < 0 ? a[j++] : a[i++];
	while (i < h) tmp[k++] = a[i++];
	while (j < n) tmp[k++] = a[j++];
	memcpy(a, tmp, n * sizeof *a);
}

int orc_read_special_fastq(const char *path, int bc_len, int is_haplotag, orc_fastq_rec_t **r1, orc_fastq_rec_t **r2, size_t *n_out)
{
	FILE *fq = fopen(path, "r");
	if (!fq) return -1;
	char buf[5000];
	size_t n = 0, cap = 1024;
	char **lines = malloc(cap * sizeof *lines);
	while (fgets(buf, sizeof buf, fq)) {      /* src/align.c:768-772 */
		if (n == cap) lines = realloc(lines, (cap *= 2) * sizeof *lines);
		lines[n] = malloc(strlen(buf) + 1);
		strcpy(lines[n++], buf);
	}
	fclose(fq);
	char **tmp = malloc((n + 1) * sizeof *tmp);
	sort_lines(lines, tmp, n, bc_len);
	free(tmp);
	*r1 = calloc(n + 1, sizeof **r1);
	*r2 = calloc(n + 1, sizeof **r2);
	char bc_str[64];
	for (size_t i = 0; i < n; ++i) {      /* src/align.c:778-799 */
		char *p = lines[i];
		assert(strcspn(p, " \t\n\v\f\r") == (size_t)bc_len);
		orc_copy_until_space(bc_str, &p);
		const uint64_t bc = orc_encode_bc(bc_str, bc_len, is_haplotag);
		orc_fastq_rec_t *a = &(*r1)[i], *b = &(*r2)[i];
		a->bc = b->bc = bc;
		assert(strcspn(p, " \t\n\v\f\r") < sizeof a->id);
		orc_copy_until_space(a->id, &p);
		strcpy(b->id, a->id);
		orc_copy_until_space(a->read, &p);
		orc_copy_until_space(a->qual, &p);
		orc_copy_until_space(b->read, &p);
		orc_copy_until_space(b->qual, &p);
		a->rlen = (unsigned short)strlen(a->read);
		b->rlen = (unsigned short)strlen(b->read);
		free(lines[i]);
	}
	free(lines);      /* entry n stays zeroed: id[0] == '\0' is the end-of-array sentinel (include/samrecord.h:17-18) */
	*n_out = n;
	return 0;
}

/* One step of seek_next_barcode_group (src/align.c:808-843): from record `at`, the run of records with its barcode.
 * Returns the run's length, 0 at the sentinel. */
size_t orc_next_group(const orc_fastq_rec_t *recs, size_t at)
{
	if (recs[at].id[0] == '\0') return 0;
	size_t k = 0;
	while (recs[at + k].id[0] != '\0' && recs[at + k].bc == recs[at].bc) ++k;
	return k;
}
