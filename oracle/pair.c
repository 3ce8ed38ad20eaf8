/*
 * oracle/pair.c -- mate rescue, final alignment and the reference's bridge.
 * TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle.h).
 *
 *   orc_matesw      bwa mem_matesw / mem_infer_dir   <- reference src/bwabridge.c:267,281
 *   orc_gen_cigar2  bwa bwa_gen_cigar2
 *   orc_reg2aln     bwa mem_reg2aln / infer_bw       <- reference src/bwabridge.c:304
 *   orc_mate_sw     reference src/bwabridge.c:204-299 (bwa_mem_mate_sw), restated
 *   orc_align_pair  candidate part of reference src/align.c:986-1061
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include <malloc.h>
#include "oracle.h"
#include <omp.h>

static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	int64_t p2;
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

int orc_matesw(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], const orc_reg_t *a,
               int l_ms, const uint8_t *ms, orc_reg_v *ma)
{
	int64_t l_pac = idx->l_pac;
	int i, r, skip[4], n = 0, rid = -1;
	for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (i = 0; (size_t)i < ma->n; ++i) {
		int64_t dist;
		r = infer_dir(l_pac, a->rb, ma->a[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		int is_rev, is_larger;
		uint8_t *seq, *rev = 0, *ref = 0;
		int64_t rb, re;
		if (skip[r]) continue;
		is_rev = (r >> 1 != (r & 1));
		is_larger = !(r >> 1);
		if (is_rev) {
			rev = malloc(l_ms);
			for (i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
			seq = rev;
		} else {
			rev = malloc(l_ms);
			memcpy(rev, ms, l_ms);
			seq = rev;
		}
		if (!is_rev) {
			rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
			re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
			re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) ref = orc_fetch_seq(idx, &rb, (rb + re) >> 1, &re, &rid);
		if (a->rid == rid && re - rb >= opt->min_seed_len) {
			orc_kswr_t aln;
			orc_reg_t b;
			int tmp, xtra = ORC_KSW_XSUBO | ORC_KSW_XSTART | (l_ms * opt->a < 250 ? ORC_KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
			aln = orc_ksw_align2(l_ms, seq, (int)(re - rb), ref, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra);
			memset(&b, 0, sizeof(b));
			if (aln.score >= opt->min_seed_len && aln.qb >= 0) {
				b.rid = a->rid;
				b.is_alt = a->is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				if (ma->n == ma->m) { ma->m = ma->m ? ma->m << 1 : 8; ma->a = realloc(ma->a, ma->m * sizeof(orc_reg_t)); }
				++ma->n;
				for (i = 0; (size_t)i < ma->n - 1; ++i)
					if (ma->a[i].score < b.score) break;
				tmp = i;
				for (i = (int)ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i - 1];
				ma->a[i] = b;
			}
			++n;
		}
		if (n) ma->n = orc_sort_dedup_patch(opt, 0, 0, (int)ma->n, ma->a);
		free(rev);
		free(ref);
	}
	return n;
}

/* ------------------------------------------------------------------ */

uint32_t *orc_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac,
                         const uint8_t *pac, int l_query, uint8_t *query, int64_t rb, int64_t re,
                         int *score, int *n_cigar, int *NM)
{
	uint32_t *cigar = 0;
	uint8_t tmp, *rseq;
	int i;
	int64_t rlen;

	if (n_cigar) *n_cigar = 0;
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	rseq = orc_get_seq(l_pac, pac, rb, re, &rlen);
	if (re - rb != rlen) goto done;
	if (rb >= l_pac) {   /* reverse both so that indels are left-aligned on the forward strand */
		for (i = 0; i < l_query >> 1; ++i)
			tmp = query[i], query[i] = query[l_query - 1 - i], query[l_query - 1 - i] = tmp;
		for (i = 0; i < rlen >> 1; ++i)
			tmp = rseq[i], rseq[i] = rseq[rlen - 1 - i], rseq[rlen - 1 - i] = tmp;
	}
	if (l_query == re - rb && w_ == 0) {   /* gap-free */
		if (n_cigar) {
			cigar = malloc(4);
			cigar[0] = (uint32_t)l_query << 4 | 0;
			*n_cigar = 1;
		}
		for (i = 0, *score = 0; i < l_query; ++i)
			*score += mat[rseq[i] * 5 + query[i]];
	} else {
		int w, max_gap, max_ins, max_del, min_w;
		max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
		max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
		max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		*score = orc_ksw_global2(l_query, query, (int)rlen, rseq, 5, mat, o_del, e_del, o_ins, e_ins, w, n_cigar, &cigar);
	}
	if (NM && n_cigar) {
		int k, x, y, n_mm = 0, n_gap = 0;
		for (k = 0, x = y = 0; k < *n_cigar; ++k) {
			int op = cigar[k] & 0xf, len = cigar[k] >> 4;
			if (op == 0) {
				for (i = 0; i < len; ++i)
					if (query[x + i] != rseq[y + i]) ++n_mm;
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < *n_cigar - 1) n_gap += len;   /* a terminal D is not counted */
				y += len;
			} else if (op == 1) x += len, n_gap += len;
		}
		*NM = n_mm + n_gap;
	}
	if (rb >= l_pac)
		for (i = 0; i < l_query >> 1; ++i)
			tmp = query[i], query[i] = query[l_query - 1 - i], query[l_query - 1 - i] = tmp;
done:
	free(rseq);
	return cigar;
}

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

orc_aln_t orc_reg2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query_, const orc_reg_t *ar)
{
	orc_aln_t a;
	int i, w2, tmp, qb, qe, NM = -1, score = 0, is_rev, last_sc = -(1 << 30);
	int64_t pos, rb, re;
	uint8_t *query;

	memset(&a, 0, sizeof(a));
	if (ar == 0 || ar->rb < 0 || ar->re < 0) {
		a.rid = -1; a.pos = -1; a.flag |= 0x4;
		return a;
	}
	qb = ar->qb; qe = ar->qe;
	rb = ar->rb; re = ar->re;
	query = malloc(l_query);
	for (i = 0; i < l_query; ++i) query[i] = query_[i] < 5 ? query_[i] : orc_nt4_table[(int)query_[i]];
	a.mapq = 0;   /* upstream: mem_approx_mapq_se; the reference overwrites it (src/align.c:1027,1052) */
	if (ar->secondary >= 0) a.flag |= 0x100;
	tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
	w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
	i = 0; a.cigar = 0;
	do {
		free(a.cigar);
		w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
		a.cigar = orc_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, idx->l_pac, idx->pac,
		                         qe - qb, &query[qb], rb, re, &score, &a.n_cigar, &NM);
		if (score == last_sc || w2 == opt->w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - opt->a);
	a.NM = NM;
	is_rev = (rb < idx->l_pac ? rb : re - 1) >= idx->l_pac;
	pos = is_rev ? (idx->l_pac << 1) - 1 - (re - 1) : rb;
	a.is_rev = is_rev;
	if (a.n_cigar > 0) {   /* squeeze out a leading or trailing deletion */
		if ((a.cigar[0] & 0xf) == 2) {
			pos += a.cigar[0] >> 4;
			--a.n_cigar;
			memmove(a.cigar, a.cigar + 1, a.n_cigar * 4);
		} else if ((a.cigar[a.n_cigar - 1] & 0xf) == 2) {
			--a.n_cigar;
		}
	}
	if (qb != 0 || qe != l_query) {
		int clip5, clip3;
		clip5 = is_rev ? l_query - qe : qb;
		clip3 = is_rev ? qb : l_query - qe;
		a.cigar = realloc(a.cigar, 4 * (a.n_cigar + 2));
		if (clip5) {
			memmove(a.cigar + 1, a.cigar, a.n_cigar * 4);
			a.cigar[0] = (uint32_t)clip5 << 4 | 3;
			++a.n_cigar;
		}
		if (clip3) a.cigar[a.n_cigar++] = (uint32_t)clip3 << 4 | 3;
	}
	a.rid = orc_pos2rid(idx, pos);
	assert(a.rid == ar->rid);
	a.pos = pos - idx->anns[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	free(query);
	return a;
}

/* ------------------------------------------------------------------ */
/* reference src/bwabridge.c:204-299 */

void orc_mate_sw(const orc_opt_t *opt, const orc_idx_t *idx, const char *read1, int len1, const char *read2, int len2,
                 int score_delta, orc_reg_v *r1, orc_reg_v *r2)
{
	orc_pestat_t pes[4];
	uint8_t *s1 = malloc(len1 > 0 ? len1 : 1), *s2 = malloc(len2 > 0 ? len2 : 1);
	int i, num, best1 = 0, best2 = 0;
	size_t k;

	/* src/bwabridge.c:216-227: only FR, insert in [-35, 500] */
	for (i = 0; i < 4; ++i) { pes[i].low = -35; pes[i].high = 500; pes[i].avg = 200.0; pes[i].std = 100.0; pes[i].failed = 1; }
	pes[1].failed = 0;
	for (i = 0; i < len1; ++i) s1[i] = orc_nt4_table[(unsigned char)read1[i]];   /* seq_convert, :151-157 */
	for (i = 0; i < len2; ++i) s2[i] = orc_nt4_table[(unsigned char)read2[i]];

	*r1 = orc_align1_core(opt, idx, len1, s1);   /* :236 */
	*r2 = orc_align1_core(opt, idx, len2, s2);   /* :237 */
	for (k = 0; k < r1->n; ++k) if (r1->a[k].score > best1) best1 = r1->a[k].score;   /* :245-252 */
	for (k = 0; k < r2->n; ++k) if (r2->a[k].score > best2) best2 = r2->a[k].score;   /* :254-261 */

	/* :263-269: rescue mate 1 around good hits of mate 2.  The loop bound is
	 * the pre-rescue count of mate 2; the scores compared are pre-rescue too. */
	{
		size_t n2 = r2->n;
		int *sc2 = malloc(sizeof(int) * (n2 + 1));
		orc_reg_t *hit2 = malloc(sizeof(orc_reg_t) * (n2 + 1));
		for (k = 0; k < n2; ++k) sc2[k] = r2->a[k].score, hit2[k] = r2->a[k];
		for (k = 0, num = 0; k < n2 && num < 50; ++k)
			if (sc2[k] >= best2 - score_delta) {
				++num;
				orc_matesw(opt, idx, pes, &hit2[k], len1, s1, r1);
			}
		free(sc2); free(hit2);
	}
	/* :271-283: rescue mate 2 around hits of the UPDATED mate-1 list, threshold from the pre-rescue best1 */
	{
		size_t n1 = r1->n;
		orc_reg_t *hit1 = malloc(sizeof(orc_reg_t) * (n1 + 1));
		for (k = 0; k < n1; ++k) hit1[k] = r1->a[k];
		for (k = 0, num = 0; k < n1 && num < 50; ++k)
			if (hit1[k].score >= best1 - score_delta) {
				++num;
				orc_matesw(opt, idx, pes, &hit1[k], len2, s2, r2);
			}
		free(hit1);
	}
	free(s1); free(s2);
}

void orc_align_pair(const orc_opt_t *opt, const orc_idx_t *idx, const char *read1, int len1,
                    const char *read2, int len2, orc_pair_out_t *out)
{
	orc_reg_v r[2];
	const char *rd[2] = { read1, read2 };
	int ln[2] = { len1, len2 };
	size_t k, n = 0, pool_m = 0;
	int m;

	orc_mate_sw(opt, idx, read1, len1, read2, len2, 25, &r[0], &r[1]);   /* src/align.c:1005 */
	out->n1 = r[0].n; out->n2 = r[1].n;
	out->c = calloc(out->n1 + out->n2 + 1, sizeof(orc_cand_t));
	out->pool = 0; out->n_pool = 0;
	for (m = 0; m < 2; ++m) {
		uint8_t *s = malloc(ln[m] > 0 ? ln[m] : 1);
		int i;
		for (i = 0; i < ln[m]; ++i) s[i] = orc_nt4_table[(unsigned char)rd[m][i]];
		for (k = 0; k < r[m].n; ++k, ++n) {   /* src/align.c:1010-1013 / 1035-1038 */
			orc_cand_t *c = &out->c[n];
			orc_aln_t a = orc_reg2aln(opt, idx, ln[m], s, &r[m].a[k]);
			c->reg = r[m].a[k];
			c->pos = a.pos; c->is_rev = a.is_rev; c->NM = a.NM; c->n_cigar = a.n_cigar;
			c->aln_score = a.score; c->aln_sub = a.sub;
			c->cigar_off = (uint32_t)out->n_pool;
			if (out->n_pool + a.n_cigar > pool_m) {
				pool_m = (out->n_pool + a.n_cigar) * 2 + 16;
				out->pool = realloc(out->pool, pool_m * 4);
			}
			if (a.n_cigar) memcpy(out->pool + out->n_pool, a.cigar, a.n_cigar * 4);
			out->n_pool += a.n_cigar;
			orc_stats.n_cigar += a.n_cigar;
			free(a.cigar);
		}
		orc_stats.n_regs += r[m].n;
		free(s);
		free(r[m].a);
	}
}

void orc_pair_out_free(orc_pair_out_t *out)
{
	free(out->c); free(out->pool);
	out->c = 0; out->pool = 0;
}

double orc_bench_pairs(const orc_opt_t *opt, const orc_idx_t *idx, const char *bases, const uint32_t *off, size_t n_pairs,
                       int n_threads, uint64_t *n_cand)
{
	uint64_t total = 0;
	long i;
	double t0;
	if (n_threads < 1) n_threads = 1;
	/* Like bwa, this code allocates its per-read working sets with malloc/calloc.  glibc gives every thread its own arena, but
	 * an arena hands memory back to the kernel whenever its top chunk passes the trim threshold and asks for it again for the
	 * next repeat-rich read; both are address-space updates under the process-wide memory-map lock, and with dozens of threads
	 * they serialise the run.  Keep freed memory in the arenas for the duration of the timing. */
	if (!getenv("ORC_BENCH_DEFAULT_MALLOC")) {
		mallopt(M_TRIM_THRESHOLD, 1 << 30);
		mallopt(M_MMAP_THRESHOLD, 1 << 30);
		mallopt(M_TOP_PAD, 64 << 20);
	}
	t0 = omp_get_wtime();
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 16) reduction(+ : total)
	for (i = 0; i < (long)n_pairs; ++i) {
		orc_pair_out_t o;
		orc_align_pair(opt, idx, bases + off[2 * i], (int)(off[2 * i + 1] - off[2 * i]), bases + off[2 * i + 1],
		               (int)(off[2 * i + 2] - off[2 * i + 1]), &o);
		total += o.n1 + o.n2;
		orc_pair_out_free(&o);
	}
	if (n_cand) *n_cand = total;
	return omp_get_wtime() - t0;
}

/* A 64-bit digest of each read's candidate list (regions, positions, NM, CIGARs, in order) for a whole batch on n_threads:
 * the large spot checks of bench.py and the chr20-scale parity test compare these with the same digest of the engine's candidates
 * (tests/oracle_lib.py, cand_digest) instead of walking 10^5 candidate lists in Python.  digest[2 * n_pairs]. */
#define ORC_DG(k) ((uint64_t)(k))
static uint64_t digest_cand(const orc_cand_t *c, const uint32_t *pool)
{
	uint64_t m = (uint64_t)c->reg.rb * ORC_DG(0x9E3779B97F4A7C15ULL) + (uint64_t)c->reg.re * ORC_DG(0xC2B2AE3D27D4EB4FULL)
	           + (uint64_t)(int64_t)c->reg.qb * ORC_DG(0x165667B19E3779F9ULL) + (uint64_t)(int64_t)c->reg.qe * ORC_DG(0x85EBCA77C2B2AE63ULL)
	           + (uint64_t)(int64_t)c->reg.score * ORC_DG(0x27D4EB2F165667C5ULL) + (uint64_t)c->pos * ORC_DG(0xD6E8FEB86659FD93ULL)
	           + (uint64_t)(int64_t)c->NM * ORC_DG(0xFF51AFD7ED558CCDULL) + (uint64_t)(int64_t)c->n_cigar * ORC_DG(0xC4CEB9FE1A85EC53ULL)
	           + (uint64_t)(int64_t)c->is_rev * ORC_DG(0x2545F4914F6CDD1DULL) + (uint64_t)(int64_t)c->reg.csub * ORC_DG(0x94D049BB133111EBULL)
	           + (uint64_t)(int64_t)c->reg.seedcov * ORC_DG(0xBF58476D1CE4E5B9ULL);
	int j;
	for (j = 0; j < c->n_cigar; ++j) m += (uint64_t)pool[c->cigar_off + j] * (ORC_DG(0xA0761D6478BD642FULL) + (uint64_t)j * ORC_DG(0xE7037ED1A0B428DBULL));
	m ^= m >> 29; m *= ORC_DG(0x8EBC6AF09C88C6E3ULL); m ^= m >> 32;
	return m;
}
double orc_digest_pairs(const orc_opt_t *opt, const orc_idx_t *idx, const char *bases, const uint32_t *off, size_t n_pairs,
                        int n_threads, uint64_t *digest)
{
	long i;
	double t0;
	if (n_threads < 1) n_threads = 1;
	if (!getenv("ORC_BENCH_DEFAULT_MALLOC")) {
		mallopt(M_TRIM_THRESHOLD, 1 << 30);
		mallopt(M_MMAP_THRESHOLD, 1 << 30);
		mallopt(M_TOP_PAD, 64 << 20);
	}
	t0 = omp_get_wtime();
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 4)
	for (i = 0; i < (long)n_pairs; ++i) {
		orc_pair_out_t o;
		size_t k;
		uint64_t d[2] = {0, 0};
		orc_align_pair(opt, idx, bases + off[2 * i], (int)(off[2 * i + 1] - off[2 * i]), bases + off[2 * i + 1],
		               (int)(off[2 * i + 2] - off[2 * i + 1]), &o);
		for (k = 0; k < o.n1 + o.n2; ++k) {
			const int m = k < o.n1 ? 0 : 1;
			const uint64_t r = m ? k - o.n1 : k;
			d[m] += digest_cand(&o.c[k], o.pool) * (2 * r + 1);
		}
		digest[2 * i] = d[0] + (uint64_t)o.n1 * ORC_DG(0x9FB21C651E98DF25ULL);
		digest[2 * i + 1] = d[1] + (uint64_t)o.n2 * ORC_DG(0x9FB21C651E98DF25ULL);
		orc_pair_out_free(&o);
	}
	return omp_get_wtime() - t0;
}

/* ------------------------------------------------------------------ */
/* Host stage behind the bridge calls, reference src/align.c:
 *   :959-984   mem_approx_mapq_se_insist   -> approx_mapq()
 *   :846-911   score_alignment             -> the likelihood block below
 *   :986-1061  append_alignments (filters, unique flag, record order) -> orc_append_alignments()
 * Constants: include/align.h:70-73 (INDEL_RATE, CLIP_RATE, EXTRA_SEARCH_DEPTH); MEM_MAPQ_COEF = 30 is bwa's.
 * TEST INFRASTRUCTURE (see oracle.h). */
static int approx_mapq(const orc_opt_t *opt, const orc_reg_t *a)
{
	int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
	if (a->score == 0) mapq = 0;
	else if (opt->mapQ_coef_len > 0) {
		double tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
	} else {
		mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 254) mapq = 254;
	if (mapq < 0) mapq = 0;
	return (int)(mapq * (1. - a->frac_rep) + .499);
}

/* For one pair's candidates (orc_align_pair): the records append_alignments would emit.  Arrays of capacity n1 + n2:
 * which[k] = index of the candidate in p->c, then clip, clip_edit_dist, mapq, score_mapq, unique, score.  Returns the
 * number of records. */
int orc_append_alignments(const orc_opt_t *opt, const orc_pair_out_t *p, int len1, int len2, double error_rate, int *which,
                          int *clip_, int *dist_, int *mapq_, int *score_mapq_, int *unique_, double *score_)
{
	const double lm = log(1 - error_rate), lx = log(error_rate), li = log(0.0001), lc = log(0.03);
	const double gx = log10(error_rate), gi = log10(0.0001), gc = log10(0.03);
	int n = 0, best_dist = -1;
	for (int mate = 0; mate < 2; ++mate) {
		const size_t first = mate ? p->n1 : 0, count = mate ? p->n2 : p->n1;
		const int len = mate ? len2 : len1;
		int added = 0;
		for (size_t i = 0; i < count; ++i) {
			const orc_cand_t *c = &p->c[first + i];
			const int clip = len - (c->reg.qe - c->reg.qb);
			int matches = 0, mismatches, indels = 0, events = 0, clipping = 0, dist;
			if (clip >= len / 2) continue;
			dist = c->NM + clip;
			if (i == 0) best_dist = dist;
			else if (dist - best_dist > 12) continue;
			for (int k = 0; k < c->n_cigar; ++k) {
				const uint32_t op = p->pool[c->cigar_off + k], type = op & 0xf, len_op = op >> 4;
				switch (type) {
				case 0: matches += len_op; break;
				case 1: case 2: indels += len_op; ++events; break;
				default: clipping += len_op; break;
				}
			}
			mismatches = c->NM - indels;
			matches -= mismatches;
			which[n] = (int)(first + i); clip_[n] = clip; dist_[n] = dist; unique_[n] = 0;
			mapq_[n] = approx_mapq(opt, &c->reg);
			score_[n] = matches * lm + mismatches * lx + events * li + clipping * lc;
			score_mapq_[n] = (int)(60.0 + mismatches * gx + events * gi + clipping * gc);
			++n; ++added;
		}
		if (added == 1) unique_[n - 1] = 1;
	}
	return n;
}
