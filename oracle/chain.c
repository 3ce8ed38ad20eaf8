/*
 * oracle/chain.c -- seed location, chaining and chain filtering
 * (bwa's mem_chain, test_and_merge, mem_chain_weight, mem_chain_flt),
 * reached from reference src/bwabridge.c:236-237 through mem_align1_core.
 * TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle.h).
 *
 * Upstream keeps open chains in a kbtree keyed by the first seed's reference
 * position.  Its observable behaviour is restated on a sorted array:
 *   lookup(key) = first element with pos == key if one exists, otherwise the
 *                 greatest element with pos < key            (kb_intervalp)
 *   insert(key) = directly after the element lookup() returns (kb_putp)
 *   traversal   = ascending array order                      (__kb_traverse)
 * This is what a kbtree leaf does; it can differ from a multi-level tree only
 * when two chains share the same pos AND the tree has split (> 9 chains).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "oracle.h"
#include "introsort.h"

static int test_and_merge(const orc_opt_t *opt, int64_t l_pac, orc_chain_t *c, const orc_seed_t *p, int seed_rid)
{
	int64_t qend, rend, x, y;
	const orc_seed_t *last = &c->seeds[c->n - 1];
	qend = last->qbeg + last->len;
	rend = last->rbeg + last->len;
	if (seed_rid != c->rid) return 0;
	if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
		return 1;   /* contained: absorbed without being stored */
	if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
	x = p->qbeg - last->qbeg;
	y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
		if (c->n == c->m) {
			c->m <<= 1;
			c->seeds = realloc(c->seeds, c->m * sizeof(orc_seed_t));
		}
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0;
}

orc_chain_v orc_chain(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq)
{
	orc_chain_v chain = {0, 0, 0};
	orc_intv_v mem = {0, 0, 0};
	size_t i;
	int b, e, l_rep;
	int64_t l_pac = idx->l_pac;

	if (len < opt->min_seed_len) return chain;
	orc_collect_intv(opt, idx, len, seq, &mem);
	for (i = 0, b = e = l_rep = 0; i < mem.n; ++i) {
		orc_intv_t *p = &mem.a[i];
		int sb = (int)(p->info >> 32), se = (int)(uint32_t)p->info;
		if (p->x[2] <= (uint64_t)opt->max_occ) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	for (i = 0; i < mem.n; ++i) {
		orc_intv_t *p = &mem.a[i];
		int step, count, slen = (int)((uint32_t)p->info - (p->info >> 32));
		int64_t k;
		step = p->x[2] > (uint64_t)opt->max_occ ? (int)(p->x[2] / opt->max_occ) : 1;
		for (k = count = 0; (uint64_t)k < p->x[2] && count < opt->max_occ; k += step, ++count) {
			orc_seed_t s;
			int rid, to_add = 0;
			size_t lo, hi, at = 0;
			s.rbeg = (int64_t)orc_sa(idx, p->x[0] + k);
			s.qbeg = (int32_t)(p->info >> 32);
			s.score = s.len = slen;
			rid = orc_intv2rid(idx, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;
			if (chain.n) {
				orc_chain_t *lower = 0;
				lo = 0; hi = chain.n;           /* first element with pos >= key */
				while (lo < hi) {
					size_t mid = (lo + hi) >> 1;
					if (chain.a[mid].pos < s.rbeg) lo = mid + 1; else hi = mid;
				}
				if (lo < chain.n && chain.a[lo].pos == s.rbeg) { lower = &chain.a[lo]; at = lo + 1; }
				else if (lo > 0) { lower = &chain.a[lo - 1]; at = lo; }
				else { lower = 0; at = 0; }
				if (!lower || !test_and_merge(opt, l_pac, lower, &s, rid)) to_add = 1;
			} else to_add = 1, at = 0;
			if (to_add) {
				orc_chain_t tmp;
				memset(&tmp, 0, sizeof(tmp));
				tmp.n = 1; tmp.m = 4;
				tmp.seeds = calloc(tmp.m, sizeof(orc_seed_t));
				tmp.seeds[0] = s;
				tmp.rid = rid;
				tmp.pos = s.rbeg;
				tmp.is_alt = !!idx->anns[rid].is_alt;
				if (chain.n == chain.m) { chain.m = chain.m ? chain.m << 1 : 16; chain.a = realloc(chain.a, chain.m * sizeof(orc_chain_t)); }
				memmove(&chain.a[at + 1], &chain.a[at], (chain.n - at) * sizeof(orc_chain_t));
				chain.a[at] = tmp;
				++chain.n;
			}
		}
	}
	for (i = 0; i < chain.n; ++i) chain.a[i].frac_rep = (float)l_rep / len;
	free(mem.a);
	return chain;
}

static int chain_weight(const orc_chain_t *c)
{
	int64_t end;
	int j, w = 0, tmp;
	for (j = 0, end = 0; j < c->n; ++j) {
		const orc_seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0; j < c->n; ++j) {
		const orc_seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
		end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

#define flt_lt(a, b) ((a).w > (b).w)
ORC_SORT_INIT(flt, orc_chain_t, flt_lt)

#define chn_beg(ch) ((ch).seeds->qbeg)
#define chn_end(ch) ((ch).seeds[(ch).n - 1].qbeg + (ch).seeds[(ch).n - 1].len)

int orc_chain_flt(const orc_opt_t *opt, int n_chn, orc_chain_t *a)
{
	int i, k, n_kept = 0, *kept_idx;
	if (n_chn == 0) return 0;
	for (i = k = 0; i < n_chn; ++i) {
		orc_chain_t *c = &a[i];
		c->first = -1; c->kept = 0;
		c->w = (uint32_t)chain_weight(c);
		if ((int)c->w < opt->min_chain_weight) free(c->seeds);
		else a[k++] = *c;
	}
	n_chn = k;
	orc_introsort_flt(n_chn, a);
	kept_idx = malloc(sizeof(int) * (n_chn + 1));
	a[0].kept = 3;
	kept_idx[n_kept++] = 0;
	for (i = 1; i < n_chn; ++i) {
		int large_ovlp = 0;
		for (k = 0; k < n_kept; ++k) {
			int j = kept_idx[k];
			int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
			int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				int li = chn_end(a[i]) - chn_beg(a[i]);
				int lj = chn_end(a[j]) - chn_beg(a[j]);
				int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
					large_ovlp = 1;
					if (a[j].first < 0) a[j].first = i;
					if ((int)a[i].w < (int)a[j].w * opt->drop_ratio && (int)a[j].w - (int)a[i].w >= opt->min_seed_len << 1)
						break;
				}
			}
		}
		if (k == n_kept) {
			kept_idx[n_kept++] = i;
			a[i].kept = large_ovlp ? 2 : 3;
		}
	}
	for (i = 0; i < n_kept; ++i) {
		orc_chain_t *c = &a[kept_idx[i]];
		if (c->first >= 0) a[c->first].kept = 1;
	}
	free(kept_idx);
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= opt->max_chain_extend) break;
	}
	for (; i < n_chn; ++i)
		if (a[i].kept < 3) a[i].kept = 0;
	for (i = k = 0; i < n_chn; ++i) {
		orc_chain_t *c = &a[i];
		if (c->kept == 0) free(c->seeds);
		else a[k++] = a[i];
	}
	return k;
}
