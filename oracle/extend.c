/*
 * oracle/extend.c -- chain -> alignment regions (bwa's mem_chain2aln,
 * mem_sort_dedup_patch, mem_patch_reg, mem_align1_core), reached from the
 * reference at src/bwabridge.c:236-237.
 * TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle.h).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "oracle.h"
#include "introsort.h"

#define MAX_BAND_TRY 2
#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

static inline int cal_max_gap(const orc_opt_t *opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

static inline orc_reg_t *reg_pushp(orc_reg_v *v)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 8; v->a = realloc(v->a, v->m * sizeof(orc_reg_t)); }
	return &v->a[v->n++];
}

#define u64lt(a, b) ((a) < (b))
ORC_SORT_INIT(srt64, uint64_t, u64lt)

/* Dev aid (tools/cpu_extend_profile.py): would the engine's lane-per-read pass (k_align_lane.hip, lane_extend_diag) know this
 * extension's outcome without the dynamic program?  It does when the first qlen target bases differ from the query in at most one
 * position, nothing is ambiguous, tlen >= qlen and h0 > 0.  Per thread: [0] extensions, [1] decided, and why not: [2] target too short
 * or h0 <= 0, [3] ambiguous base, [4] two or more mismatches, [5] one mismatch but its conditions fail; [6] reads (align1_core calls),
 * [7] reads with at least one extension not decided, [8] of those: the first such extension was in the read's FIRST chain. */
__thread uint64_t orc_extprof[16];
static __thread int extprof_read_bad, extprof_chain_no;
void orc_extprof_get(uint64_t *out) { memcpy(out, orc_extprof, sizeof(orc_extprof)); }
void orc_extprof_reset(void) { memset(orc_extprof, 0, sizeof(orc_extprof)); }
/* Off unless tools/cpu_extend_profile.py switches it on (orc_extprof_enable): the timed oracle path -- bench.py's cpu_baseline and its
 * spot check run orc_chain2aln -- pays one predictable branch per extension and nothing else. */
static int extprof_on = 0;
void orc_extprof_enable(int on) { extprof_on = on; }
static void extprof_call(const orc_opt_t *opt, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int h0)
{
	int j, n_mm = 0, p_mm = -1, why = 1;
	++orc_extprof[0];
	if (!(tlen >= qlen && h0 > 0)) why = 2;
	else {
		for (j = 0; j < qlen; ++j) {
			if (q[j] > 3 || t[j] > 3) { why = 3; break; }
			if (q[j] != t[j]) { ++n_mm; p_mm = j; }
		}
		if (why == 1 && n_mm >= 2) why = 4;
		if (why == 1 && n_mm == 1) {
			const int oe_del = opt->o_del + opt->e_del, oe_ins = opt->o_ins + opt->e_ins;
			const int gap_min = oe_del < oe_ins + opt->a ? oe_del : oe_ins + opt->a;
			if (!(opt->a > 0 && gap_min > opt->a + opt->b && (opt->zdrop <= 0 || opt->a + opt->b <= opt->zdrop) && h0 + p_mm * opt->a - opt->b > 0)) why = 5;
		}
	}
	++orc_extprof[why];
	if (why != 1 && !extprof_read_bad) { extprof_read_bad = 1; ++orc_extprof[7]; if (extprof_chain_no == 0) ++orc_extprof[8]; }
}

void orc_chain2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query,
                   const orc_chain_t *c, orc_reg_v *av)
{
	int i, k, rid, max_off[2], aw[2];
	int64_t l_pac = idx->l_pac, rmax[2], tmp, max = 0;
	const orc_seed_t *s;
	uint8_t *rseq = 0;
	uint64_t *srt;

	if (c->n == 0) return;
	/* widest reference window any seed of the chain could extend into */
	rmax[0] = l_pac << 1; rmax[1] = 0;
	for (i = 0; i < c->n; ++i) {
		int64_t b, e;
		const orc_seed_t *t = &c->seeds[i];
		b = t->rbeg - (t->qbeg + cal_max_gap(opt, t->qbeg));
		e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(opt, l_query - t->qbeg - t->len));
		rmax[0] = rmax[0] < b ? rmax[0] : b;
		rmax[1] = rmax[1] > e ? rmax[1] : e;
		if (t->len > max) max = t->len;
	}
	rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) {
		if (c->seeds[0].rbeg < l_pac) rmax[1] = l_pac;
		else rmax[0] = l_pac;
	}
	rseq = orc_fetch_seq(idx, &rmax[0], c->seeds[0].rbeg, &rmax[1], &rid);
	assert(c->rid == rid);

	srt = malloc(c->n * 8);
	for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | (uint32_t)i;
	orc_introsort_srt64(c->n, srt);

	for (k = c->n - 1; k >= 0; --k) {
		orc_reg_t *a;
		s = &c->seeds[(uint32_t)srt[k]];

		for (i = 0; (size_t)i < av->n; ++i) {   /* already covered by an earlier extension? */
			orc_reg_t *p = &av->a[i];
			int64_t rd;
			int qd, w, max_gap;
			if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
			if (s->len - p->seedlen0 > .1 * l_query) continue;
			qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
			max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
			max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if ((size_t)i < av->n) {
			for (i = k + 1; i < c->n; ++i) {   /* unless an extended, overlapping seed sits on another diagonal */
				const orc_seed_t *t;
				if (srt[i] == 0) continue;
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue;
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) {
				srt[k] = 0;
				continue;
			}
		}

		a = reg_pushp(av);
		memset(a, 0, sizeof(orc_reg_t));
		a->w = aw[0] = aw[1] = opt->w;
		a->score = a->truesc = -1;
		a->rid = c->rid;

		if (s->qbeg) {   /* left extension on the reversed prefixes */
			uint8_t *rs, *qs;
			int qle, tle, gtle, gscore;
			qs = malloc(s->qbeg);
			for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
			tmp = s->rbeg - rmax[0];
			rs = malloc(tmp > 0 ? tmp : 1);
			for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
			for (i = 0; i < MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[0] = opt->w << i;
				if (i == 0 && extprof_on) extprof_call(opt, s->qbeg, qs, (int)tmp, rs, s->len * opt->a);
				a->score = orc_ksw_extend2(s->qbeg, qs, (int)tmp, rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins,
				                           aw[0], opt->pen_clip5, opt->zdrop, s->len * opt->a, &qle, &tle, &gtle, &gscore, &max_off[0]);
				if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip5) {
				a->qb = s->qbeg - qle; a->rb = s->rbeg - tle;
				a->truesc = a->score;
			} else {
				a->qb = 0; a->rb = s->rbeg - gtle;
				a->truesc = gscore;
			}
			free(qs); free(rs);
		} else a->score = a->truesc = s->len * opt->a, a->qb = 0, a->rb = s->rbeg;

		if (s->qbeg + s->len != l_query) {   /* right extension */
			int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
			qe = s->qbeg + s->len;
			re = (int)(s->rbeg + s->len - rmax[0]);
			assert(re >= 0);
			for (i = 0; i < MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[1] = opt->w << i;
				if (i == 0 && extprof_on) extprof_call(opt, l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, sc0);
				a->score = orc_ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, 5, opt->mat,
				                           opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw[1], opt->pen_clip3, opt->zdrop, sc0,
				                           &qle, &tle, &gtle, &gscore, &max_off[1]);
				if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip3) {
				a->qe = qe + qle; a->re = rmax[0] + re + tle;
				a->truesc += a->score - sc0;
			} else {
				a->qe = l_query; a->re = rmax[0] + re + gtle;
				a->truesc += gscore - sc0;
			}
		} else a->qe = l_query, a->re = s->rbeg + s->len;

		for (i = 0, a->seedcov = 0; i < c->n; ++i) {
			const orc_seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
				a->seedcov += t->len;
		}
		a->w = aw[0] > aw[1] ? aw[0] : aw[1];
		a->seedlen0 = s->len;
		a->frac_rep = c->frac_rep;
	}
	free(srt); free(rseq);
}

/* ------------------------------------------------------------------ */

static int patch_reg(const orc_opt_t *opt, const orc_idx_t *idx, uint8_t *query, const orc_reg_t *a, const orc_reg_t *b, int *_w)
{
	int w, score, q_s, r_s;
	double r;
	if (idx == 0 || query == 0) return 0;
	assert(a->rid == b->rid && a->rb <= b->rb);
	if (a->rb < idx->l_pac && b->rb >= idx->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0 ? w : -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0. ? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
	} else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
	w += a->w + b->w;
	w = w < opt->w << 2 ? w : opt->w << 2;
	score = 0;
	orc_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, idx->l_pac, idx->pac,
	               b->qe - a->qb, query + a->qb, a->rb, b->re, &score, 0, 0);
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
	*_w = w;
	return score;
}

#define ars2_lt(a, b) ((a).re < (b).re)
ORC_SORT_INIT(ars2, orc_reg_t, ars2_lt)
#define ars_lt(a, b) ((a).score > (b).score || ((a).score == (b).score && ((a).rb < (b).rb || ((a).rb == (b).rb && (a).qb < (b).qb))))
ORC_SORT_INIT(ars, orc_reg_t, ars_lt)

int orc_sort_dedup_patch(const orc_opt_t *opt, const orc_idx_t *idx, uint8_t *query, int n, orc_reg_t *a)
{
	int m, i, j;
	if (n <= 1) return n;
	orc_introsort_ars2(n, a);   /* by END position */
	for (i = 0; i < n; ++i) a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		orc_reg_t *p = &a[i];
		if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
			orc_reg_t *q = &a[j];
			int64_t or_, oq, mr, mq;
			int score, w;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
				if (p->score < q->score) {
					p->qe = p->qb;
					break;
				} else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(opt, idx, query, q, p, &w)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) {
			if (m != i) a[m++] = a[i];
			else ++m;
		}
	n = m;
	orc_introsort_ars(n, a);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb)
			a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) {
			if (m != i) a[m++] = a[i];
			else ++m;
		}
	return m;
}

orc_reg_v orc_align1_core(const orc_opt_t *opt, const orc_idx_t *idx, int l_seq, uint8_t *seq)
{
	int i;
	orc_chain_v chn;
	orc_reg_v regs = {0, 0, 0};

	for (i = 0; i < l_seq; ++i) seq[i] = seq[i] < 4 ? seq[i] : orc_nt4_table[(int)seq[i]];
	orc_stats.l_read += l_seq;
	chn = orc_chain(opt, idx, l_seq, seq);
	chn.n = orc_chain_flt(opt, (int)chn.n, chn.a);
	/* mem_flt_chained_seeds: returns at once while MEM_MINSC_COEF*ln(l) > MEM_SEEDSW_COEF*l, i.e. l < ~700 */
	assert(5.5 * log(l_seq > 1 ? l_seq : 2) > 0.05 * l_seq);
	if (extprof_on) { ++orc_extprof[6]; extprof_read_bad = 0; }
	for (i = 0; (size_t)i < chn.n; ++i) {
		if (extprof_on) extprof_chain_no = i;
		orc_chain2aln(opt, idx, l_seq, seq, &chn.a[i], &regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs.n = orc_sort_dedup_patch(opt, idx, seq, (int)regs.n, regs.a);
	for (i = 0; (size_t)i < regs.n; ++i) {
		orc_reg_t *p = &regs.a[i];
		if (p->rid >= 0 && idx->anns[p->rid].is_alt) p->is_alt = 1;
	}
	return regs;
}
