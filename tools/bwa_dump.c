/* tools/bwa_dump.c -- differential-test driver: the reference's bridge (bwa_mem_mate_sw + bwa_smith_waterman, reference
 * src/bwabridge.c:204-311) written against the nine libbwa symbols only, dumping for every read pair the region lists
 * after rescue and every hit's final alignment.  It compiles unchanged against
 *   - a real bwa checkout  (cc -I$BWADIR tools/bwa_dump.c $BWADIR/libbwa.a -lz -lm -lpthread  -DREAL_BWA), and
 *   - this repository's face (cc -Iinclude tools/bwa_dump.c -Lema_amd -lema_bwaabi),
 * so that the two outputs can be diffed line by line (tools/diff_vs_bwa.sh).  TEST TOOLING, not product code.
 * Input: one pair per line "READ1 READ2" (ASCII).  Output per pair: "P <i>" then per mate and hit
 *   "H <mate> rb re qb qe rid score truesc sub csub w seedcov secondary seedlen0 frac_rep pos is_rev NM cigar" */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef REAL_BWA
#include "bwamem.h"
#include "bntseq.h"
#include "bwa.h"
extern mem_alnreg_v mem_align1_core(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int l_seq, char *seq, void *buf);
extern int mem_matesw(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], const mem_alnreg_t *a, int l_ms, const uint8_t *ms, mem_alnreg_v *ma);
#else
#include "ema_bwaabi.h"
#endif

static void dump(const mem_opt_t *opt, bwaidx_t *idx, int mate, int len, char *seq, mem_alnreg_v *rv)
{
	size_t k;
	int j;
	for (k = 0; k < rv->n; ++k) {
		mem_alnreg_t *g = &rv->a[k];
		mem_aln_t a = mem_reg2aln(opt, idx->bns, idx->pac, len, seq, g);
		printf("H %d %lld %lld %d %d %d %d %d %d %d %d %d %d %d %.9g %lld %d %d ", mate, (long long)g->rb, (long long)g->re, g->qb, g->qe, g->rid,
		       g->score, g->truesc, g->sub, g->csub, g->w, g->seedcov, g->secondary, g->seedlen0, (double)g->frac_rep, (long long)a.pos,
		       (int)a.is_rev, (int)a.NM);
		for (j = 0; j < a.n_cigar; ++j) printf("%u%c", a.cigar[j] >> 4, "MIDSH"[a.cigar[j] & 0xf]);
		printf("\n");
		free(a.cigar); free(a.XA);
	}
}

int main(int argc, char **argv)
{
	char line[4096], r1[2048], r2[2048];
	long n_pair = 0;
	int i, score_delta = 25;
	mem_pestat_t pes[4];
	if (argc < 3) { fprintf(stderr, "usage: bwa_dump INDEX_PREFIX PAIRS.txt [max_occ]\n"); return 2; }
	bwaidx_t *idx = bwa_idx_load(argv[1], BWA_IDX_ALL);
	if (!idx) return 1;
	mem_opt_t *opt = mem_opt_init();
	opt->max_occ = argc > 3 ? atoi(argv[3]) : 3000;      /* reference src/align.c:185 */
	for (i = 0; i < 4; ++i) { pes[i].failed = i != 1; pes[i].low = -35; pes[i].high = 500; pes[i].avg = 200.0; pes[i].std = 100.0; }
	FILE *f = fopen(argv[2], "r");
	if (!f) { perror(argv[2]); return 1; }
	while (fgets(line, sizeof line, f)) {
		if (sscanf(line, "%2047s %2047s", r1, r2) != 2) continue;
		int l1 = (int)strlen(r1), l2 = (int)strlen(r2), num, best1 = 0, best2 = 0;
		size_t k;
		for (i = 0; i < l1; ++i) r1[i] = (char)nst_nt4_table[(unsigned char)r1[i]];
		for (i = 0; i < l2; ++i) r2[i] = (char)nst_nt4_table[(unsigned char)r2[i]];
		mem_alnreg_v v1 = mem_align1_core(opt, idx->bwt, idx->bns, idx->pac, l1, r1, 0);
		mem_alnreg_v v2 = mem_align1_core(opt, idx->bwt, idx->bns, idx->pac, l2, r2, 0);
		for (k = 0; k < v1.n; ++k) if (v1.a[k].score > best1) best1 = v1.a[k].score;
		for (k = 0; k < v2.n; ++k) if (v2.a[k].score > best2) best2 = v2.a[k].score;
		for (k = 0, num = 0; k < v2.n && num < 50; ++k)
			if (v2.a[k].score >= best2 - score_delta) { ++num; mem_matesw(opt, idx->bns, idx->pac, pes, &v2.a[k], l1, (uint8_t *)r1, &v1); }
		for (k = 0, num = 0; k < v1.n && num < 50; ++k)
			if (v1.a[k].score >= best1 - score_delta) { mem_alnreg_t anchor = v1.a[k]; ++num; mem_matesw(opt, idx->bns, idx->pac, pes, &anchor, l2, (uint8_t *)r2, &v2); }
		printf("P %ld\n", n_pair++);
		dump(opt, idx, 1, l1, r1, &v1);
		dump(opt, idx, 2, l2, r2, &v2);
		free(v1.a); free(v2.a);
	}
	fclose(f);
	free(opt);
	bwa_idx_destroy(idx);
	return 0;
}
