/*
 * oracle/sam.c -- CPU oracle for the SAM record formatter.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * A line-for-line restatement of print_sam_record() (reference src/samrecord.c:104-284) with its helpers get_rlen
 * (:75-84), rc (:86-102) and is_pair (src/align.c:27-40), printing through stdio one call at a time as the reference
 * does.  The original is in the reference tree but cannot be compiled here (samrecord.c includes bwa's headers through
 * bwabridge.h), so this part is unpinned; decode_bc is the restatement of oracle/ingest.c, which IS pinned to the
 * reference's compiled util.c.
 */
#define _GNU_SOURCE
#include <assert.h>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static int get_rlen(int n_cigar, const uint32_t *cigar)
{
	int k, l;
	for (k = l = 0; k < n_cigar; k++) {
		int op = cigar[k] & 0xf;
		if (op == 0 || op == 2) l += (cigar[k] >> 4);
	}
	return l;
}

static char rc(const char c)
{
	switch (c) {
	case 'A': return 'T';
	case 'C': return 'G';
	case 'G': return 'C';
	case 'T': return 'A';
	case 'N': return 'N';
	}
	assert(0);
	return 0;
}

static int is_pair(const orc_sam_rec_t *r1, const orc_sam_rec_t *r2, const orc_sam_opts_t *o)
{
	if (r1->rev == r2->rev || r1->chrom_id != r2->chrom_id) return 0;
	if (r2->rev) { const orc_sam_rec_t *rt = r2; r2 = r1; r1 = rt; }
	const int64_t d = r1->pos - r2->pos;      /* two uint32_t, as in the reference: the difference wraps, never negative */
	return o->insert_min <= d && d <= o->insert_max;
}

static void print_cigar(FILE *out, const uint32_t *cigar, int cigar_len)
{
	for (int i = 0; i < cigar_len; i++) {
		const uint32_t op = cigar[i], type = op & 0xf, n = op >> 4;
		fprintf(out, "%u%c", n, "MIDSS"[type]);
	}
}

static void orc_print_sam_record(FILE *out, const orc_sam_rec_t *rec, const orc_sam_rec_t *mate, const orc_sam_opts_t *o)
{
	assert(rec != NULL || mate != NULL);
	int flag = 1;      /* SAM_READ_PAIRED */
	const char *ident, *chrom = "*", *read, *qual;
	uint32_t pos = 0;
	int mapq = 0, read_len;
	uint64_t bc;
	if (rec != NULL) {
		ident = rec->ident; chrom = rec->chrom; pos = rec->pos; read_len = rec->read_len; bc = rec->bc;
		read = rec->read; qual = rec->qual;
		const int gamma_mapq = ((rec->gamma <= 0.999999) ? (int)(-10 * log10(1 - rec->gamma)) : 60);
		mapq = gamma_mapq < rec->score_mapq ? gamma_mapq : rec->score_mapq;
		mapq = mapq < rec->mapq ? mapq : rec->mapq;
		mapq = mapq > 0 ? mapq : 0;
		mapq = mapq < 60 ? mapq : 60;
		if (rec->rev) flag |= 16;
		if (rec->duplicate) flag |= 1024;
		flag |= ((rec->mate == 0) ? 64 : 128);
	} else {
		ident = mate->ident; read_len = mate->mate_read_len; bc = mate->bc; read = mate->mate_read; qual = mate->mate_qual;
		flag |= 4;
		flag |= ((mate->mate == 0) ? 128 : 64);
	}
	if (mate != NULL) {
		if (rec != NULL && is_pair(rec, mate, o)) flag |= 2;
		if (mate->rev) flag |= 32;
	} else {
		flag |= 8;
	}
	fprintf(out, "%s\t%d\t%s\t%u\t%d\t", ident, flag, chrom, pos, mapq);
	if (rec != NULL) print_cigar(out, rec->cigar, rec->n_cigar);
	else fputc('*', out);
	if (mate != NULL) {
		const int same_chrom = (rec != NULL) && (mate->chrom_id == rec->chrom_id);
		fprintf(out, "\t%s\t%d", same_chrom ? "=" : mate->chrom, mate->pos);
		if (same_chrom) {
			int64_t p0 = rec->aln_pos + (rec->aln_rev ? get_rlen(rec->n_cigar, rec->cigar) - 1 : 0);
			int64_t p1 = mate->aln_pos + (mate->aln_rev ? get_rlen(mate->n_cigar, mate->cigar) - 1 : 0);
			if (mate->n_cigar == 0 || rec->n_cigar == 0) fprintf(out, "\t0");
			else fprintf(out, "\t%ld", (long)(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0))));
		} else {
			fprintf(out, "\t0");
		}
	} else {
		fprintf(out, "\t*\t0\t0");
	}
	fputc('\t', out);
	if (rec != NULL && rec->rev) {
		for (int i = read_len - 1; i >= 0; i--) fputc(rc(read[i]), out);
		fputc('\t', out);
		for (int i = read_len - 1; i >= 0; i--) fputc(qual[i], out);
	} else {
		for (int i = 0; i < read_len; i++) fputc(read[i], out);
		fputc('\t', out);
		for (int i = 0; i < read_len; i++) fputc(qual[i], out);
	}
	char bc_str[64];
	memset(bc_str, 0, sizeof bc_str);
	orc_decode_bc(bc, o->bc_len, o->is_haplotag, bc_str);
	if (o->is_haplotag) {
		if (rec != NULL)
			fprintf(out, "\tNM:i:%d\tBX:Z:%s\tXG:f:%.5g\tMI:i:%d\tXF:i:%d", rec->edit_dist, bc_str, rec->gamma, rec->cloud_id, rec->cloud_bad);
		else
			fprintf(out, "\tBX:Z:%s", bc_str);
	} else {
		if (rec != NULL)
			fprintf(out, "\tNM:i:%d\tBX:Z:%s-%s\tXG:f:%.5g\tMI:i:%d\tXF:i:%d", rec->edit_dist, bc_str, o->bx_index, rec->gamma, rec->cloud_id, rec->cloud_bad);
		else
			fprintf(out, "\tBX:Z:%s-1", bc_str);
	}
	if (o->rg_id != NULL) {
		fprintf(out, "\tRG:Z:");
		for (size_t i = 0; o->rg_id[i] != '\0' && !isspace((unsigned char)o->rg_id[i]); i++) fputc(o->rg_id[i], out);
	}
	if (rec != NULL && rec->n_alts > 0) {
		fprintf(out, "\tXA:Z:");
		for (size_t i = 0; i < rec->n_alts; i++) {
			const orc_sam_alt_t *alt = &rec->alts[i];
			fprintf(out, "%s,%s%d,", alt->chrom, alt->rev ? "-" : "+", alt->pos);
			print_cigar(out, alt->cigar, alt->n_cigar);
			fprintf(out, ",%d;", alt->edit_dist);
		}
	}
	fputc('\n', out);
}

/* lines[0..n) through orc_print_sam_record into one malloc'd buffer (freed by the caller with free()) */
int orc_sam_format(const orc_sam_line_t *lines, size_t n, const orc_sam_opts_t *o, char **text, size_t *n_bytes)
{
	FILE *f = open_memstream(text, n_bytes);
	if (!f) return -1;
	for (size_t i = 0; i < n; i++) orc_print_sam_record(f, lines[i].rec, lines[i].mate, o);
	fclose(f);
	return 0;
}

/* write_sam_header (reference src/align.c:192-211) into a malloc'd buffer */
int orc_sam_header(const char *const *names, const int32_t *lens, int32_t n, const char *rg, const char *version, int pg_argc,
                   const char *const *pg_argv, char **text, size_t *n_bytes)
{
	FILE *out_file = open_memstream(text, n_bytes);
	if (!out_file) return -1;
	fprintf(out_file, "@HD\tVN:1.3\tSO:unsorted\n");
	for (int32_t i = 0; i < n; i++) fprintf(out_file, "@SQ\tSN:%s\tLN:%d\n", names[i], lens[i]);
	if (rg != NULL) fprintf(out_file, "%s\n", rg);
	fprintf(out_file, "@PG\tID:ema\tPN:ema\tVN:%s\tCL:%s", version, pg_argv[0]);
	for (int i = 1; i < pg_argc; i++) fprintf(out_file, " %s", pg_argv[i]);
	fprintf(out_file, "\n");
	fclose(out_file);
	return 0;
}
