/*
 * oracle/fmindex.c -- index loading, FM-index primitives and SMEM seeding.
 * TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle.h).
 *
 * Restates bwa's bwt.c / bntseq.c / the seeding part of bwamem.c, which the
 * reference reaches through bwa_idx_load (src/bwabridge.c:79) and
 * mem_align1_core (src/bwabridge.c:236-237).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "oracle.h"
#include "introsort.h"

__thread orc_stats_t orc_stats;
void orc_stats_get(orc_stats_t *out) { *out = orc_stats; }
void orc_stats_reset(void) { memset(&orc_stats, 0, sizeof(orc_stats)); }

const unsigned char orc_nt4_table[256] = {
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 5 /*'-'*/, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 0, 4, 1,  4, 4, 4, 2,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  3, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 0, 4, 1,  4, 4, 4, 2,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  3, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4
};

/* ------------------------------------------------------------------ */
/* options: mem_opt_init() of bwamem.c, then reference src/align.c:185 */

void orc_opt_init(orc_opt_t *o)
{
	int i, j, k;
	memset(o, 0, sizeof(*o));
	o->a = 1; o->b = 4;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100;
	o->T = 30;
	o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20;
	o->min_seed_len = 19;
	o->split_width = 10;
	o->max_occ = 500;
	o->max_chain_gap = 10000;
	o->max_ins = 10000;
	o->mask_level = 0.50f;
	o->drop_ratio = 0.50f;
	o->XA_drop_ratio = 0.80f;
	o->split_factor = 1.5f;
	o->chunk_size = 10000000;
	o->n_threads = 1;
	o->max_XA_hits = 5;
	o->max_XA_hits_alt = 200;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95f;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50;
	o->mapQ_coef_fac = 3;     /* (int)log(50) */
	for (i = k = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) o->mat[k++] = i == j ? o->a : -o->b;
		o->mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) o->mat[k++] = -1;
	o->max_occ = 3000;        /* reference src/align.c:185 */
}

/* ------------------------------------------------------------------ */
/* index files (bwa_idx_load(prefix, BWA_IDX_ALL); layouts: SURVEY D.2) */

static void *slurp(const char *prefix, const char *ext, size_t *size)
{
	char path[4096];
	FILE *f;
	long n;
	void *buf;
	snprintf(path, sizeof(path), "%s%s", prefix, ext);
	f = fopen(path, "rb");
	if (!f) return NULL;
	fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
	buf = malloc(n > 0 ? n : 1);
	if (fread(buf, 1, n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
	fclose(f);
	*size = n;
	return buf;
}

orc_idx_t *orc_idx_load(const char *prefix)
{
	orc_idx_t *idx = calloc(1, sizeof(*idx));
	size_t sz, i;
	uint64_t *raw;
	char path[4096], line[8192];
	FILE *f;

	/* .bwt: u64 primary, u64 L2[1..4], then the interleaved occ/bwt words */
	raw = slurp(prefix, ".bwt", &sz);
	if (!raw) goto fail;
	idx->primary = raw[0];
	idx->L2[0] = 0;
	for (i = 0; i < 4; ++i) idx->L2[i + 1] = raw[1 + i];
	idx->seq_len = idx->L2[4];
	idx->bwt_size = (sz - 40) / 4;
	idx->bwt = malloc(idx->bwt_size * 4);
	memcpy(idx->bwt, raw + 5, idx->bwt_size * 4);
	free(raw);

	/* .sa: u64 primary, 4 x u64 (skipped), u64 sa_intv, u64 seq_len, n_sa-1 values */
	raw = slurp(prefix, ".sa", &sz);
	if (!raw) goto fail;
	if (raw[0] != idx->primary || raw[6] != idx->seq_len) { free(raw); goto fail; }
	idx->sa_intv = (int)raw[5];
	idx->n_sa = (idx->seq_len + idx->sa_intv) / idx->sa_intv;
	idx->sa = malloc(idx->n_sa * 8);
	idx->sa[0] = (uint64_t)-1;
	memcpy(idx->sa + 1, raw + 7, (idx->n_sa - 1) * 8);
	free(raw);

	/* .ann */
	snprintf(path, sizeof(path), "%s.ann", prefix);
	f = fopen(path, "r");
	if (!f) goto fail;
	{
		long long l_pac; int n_seqs; unsigned seed;
		if (!fgets(line, sizeof(line), f) || sscanf(line, "%lld %d %u", &l_pac, &n_seqs, &seed) != 3) { fclose(f); goto fail; }
		idx->l_pac = l_pac; idx->n_seqs = n_seqs;
		idx->anns = calloc(n_seqs, sizeof(orc_ann_t));
		for (i = 0; i < (size_t)n_seqs; ++i) {
			orc_ann_t *p = &idx->anns[i];
			unsigned gi; char name[4096]; long long off; int len, nambs;
			char *sp;
			if (!fgets(line, sizeof(line), f)) { fclose(f); goto fail; }
			if (sscanf(line, "%u %4095s", &gi, name) != 2) { fclose(f); goto fail; }
			p->gi = gi; p->name = strdup(name);
			sp = strstr(line, name) + strlen(name);
			while (*sp == ' ') ++sp;
			sp[strcspn(sp, "\n")] = 0;
			p->anno = strdup(sp);
			if (!fgets(line, sizeof(line), f) || sscanf(line, "%lld %d %d", &off, &len, &nambs) != 3) { fclose(f); goto fail; }
			p->offset = off; p->len = len; p->n_ambs = nambs;
		}
	}
	fclose(f);

	/* .alt (optional; bwa_idx_load_from_disk): every line that does not start with '@' names an ALT contig by its first
	 * tab- or space-delimited field (the file is SAM-formatted in bwa.kit; a plain list of names works the same way) */
	snprintf(path, sizeof(path), "%s.alt", prefix);
	f = fopen(path, "r");
	if (f) {
		while (fgets(line, sizeof(line), f)) {
			char *e = line;
			if (line[0] == '@') continue;
			while (*e && *e != '\t' && *e != ' ' && *e != '\n' && *e != '\r') ++e;
			*e = 0;
			if (!line[0]) continue;
			for (i = 0; i < (size_t)idx->n_seqs; ++i)
				if (strcmp(idx->anns[i].name, line) == 0) { idx->anns[i].is_alt = 1; break; }
		}
		fclose(f);
	}

	/* .pac: forward strand, 4 bases/byte, first base in the high bits */
	idx->pac = slurp(prefix, ".pac", &sz);
	if (!idx->pac) goto fail;
	if (2 * (uint64_t)idx->l_pac != idx->seq_len) goto fail;
	return idx;
fail:
	orc_idx_destroy(idx);
	return NULL;
}

void orc_idx_destroy(orc_idx_t *idx)
{
	int i;
	if (!idx) return;
	free(idx->bwt); free(idx->sa); free(idx->pac);
	if (idx->anns) {
		for (i = 0; i < idx->n_seqs; ++i) { free(idx->anns[i].name); free(idx->anns[i].anno); }
		free(idx->anns);
	}
	free(idx);
}

/* ------------------------------------------------------------------ */
/* occ / extend / sa (bwt.c) */

/* the 16-word block holding BWT position k (k already primary-adjusted) */
static inline const uint32_t *occ_block(const orc_idx_t *b, uint64_t k) { return b->bwt + ((k >> 7) << 4); }
static inline int bwt_base(const orc_idx_t *b, uint64_t k)
{
	const uint32_t *blk = occ_block(b, k);
	return blk[8 + ((k & 127) >> 4)] >> ((~k & 15) << 1) & 3;
}

/* number of symbol c in B[0..k] (inclusive), k in the with-sentinel row space */
void orc_occ4(const orc_idx_t *b, uint64_t k, uint64_t cnt[4])
{
	const uint32_t *blk;
	uint64_t c64[4];
	int r, i, nb;
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= b->primary);     /* '$' is not stored */
	blk = occ_block(b, k);
	memcpy(c64, blk, 32);
	r = (int)(k & 127);         /* count bases 0..r of the block */
	nb = r + 1;
	for (i = 0; i < nb; ++i) {
		int base = blk[8 + (i >> 4)] >> ((~i & 15) << 1) & 3;
		++c64[base];
	}
	memcpy(cnt, c64, 32);
}

uint64_t orc_occ(const orc_idx_t *b, uint64_t k, int c)
{
	uint64_t cnt[4];
	if (k == b->seq_len) return b->L2[c + 1] - b->L2[c];
	if (k == (uint64_t)-1) return 0;
	orc_occ4(b, k, cnt);
	return cnt[c];
}

/* Seeding profile (diagnostic, per thread): bwt_extend calls by class (0/1 pass-1 forward/backward, 2/3 pass-2 forward/backward,
 * 4 pass 3) and by the length of the string the extend produces (>= 63 in the last bin) -- what share of the rank queries a
 * table of all k-mer intervals up to a given k could answer. */
__thread uint64_t orc_seedprof[5][64];
static __thread int sp_class, sp_len;
void orc_seedprof_get(uint64_t *out) { memcpy(out, orc_seedprof, sizeof(orc_seedprof)); }
void orc_seedprof_reset(void) { memset(orc_seedprof, 0, sizeof(orc_seedprof)); }
static __thread int sp_pass = 1;

void orc_extend(const orc_idx_t *b, const orc_intv_t *ik, orc_intv_t ok[4], int is_back)
{
	++orc_seedprof[sp_class][sp_len < 63 ? (sp_len < 0 ? 0 : sp_len) : 63];
	uint64_t tk[4], tl[4];
	int i, nb = !is_back;
	++orc_stats.n_ext;
	orc_occ4(b, ik->x[nb] - 1, tk);
	orc_occ4(b, ik->x[nb] - 1 + ik->x[2], tl);
	for (i = 0; i < 4; ++i) {
		ok[i].x[nb] = b->L2[i] + 1 + tk[i];
		ok[i].x[2] = tl[i] - tk[i];
	}
	ok[3].x[is_back] = ik->x[is_back] + (ik->x[nb] <= b->primary && ik->x[nb] + ik->x[2] - 1 >= b->primary);
	ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
	ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
	ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

static inline uint64_t inv_psi(const orc_idx_t *b, uint64_t k)
{
	uint64_t x;
	int c;
	if (k == b->primary) return 0;
	x = k - (k > b->primary);
	c = bwt_base(b, x);
	return b->L2[c] + orc_occ(b, k, c);
}

uint64_t orc_sa(const orc_idx_t *b, uint64_t k)
{
	uint64_t steps = 0, mask = b->sa_intv - 1;
	++orc_stats.n_occ;
	while (k & mask) {
		++steps; ++orc_stats.n_lf;
		k = inv_psi(b, k);
	}
	return steps + b->sa[k / b->sa_intv];
}

/* ------------------------------------------------------------------ */
/* SMEM search (bwt_smem1a, bwt_seed_strategy1) */

static inline void iv_push(orc_intv_v *v, const orc_intv_t *p)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 16; v->a = realloc(v->a, v->m * sizeof(orc_intv_t)); }
	v->a[v->n++] = *p;
}
static void iv_reverse(orc_intv_v *v)
{
	size_t i;
	for (i = 0; i < v->n >> 1; ++i) { orc_intv_t t = v->a[i]; v->a[i] = v->a[v->n - 1 - i]; v->a[v->n - 1 - i] = t; }
}
static inline void set_intv(const orc_idx_t *b, int c, orc_intv_t *ik)
{
	ik->x[0] = b->L2[c] + 1; ik->x[2] = b->L2[c + 1] - b->L2[c]; ik->x[1] = b->L2[3 - c] + 1; ik->info = 0;
}

int orc_smem1a(const orc_idx_t *b, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv,
               orc_intv_v *mem, orc_intv_v *tmp0, orc_intv_v *tmp1)
{
	int i, c, ret;
	size_t j;
	orc_intv_t ik, ok[4];
	orc_intv_v *prev = tmp0, *curr = tmp1, *swap;

	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	set_intv(b, q[x], &ik);
	ik.info = x + 1;

	/* forward: remember the interval each time its size is about to change */
	for (i = x + 1, curr->n = 0; i < len; ++i) {
		if (ik.x[2] < max_intv) {
			iv_push(curr, &ik);
			break;
		} else if (q[i] < 4) {
			c = 3 - q[i];
			sp_class = sp_pass == 1 ? 0 : 2; sp_len = i - x + 1;
			orc_extend(b, &ik, ok, 0);
			if (ok[c].x[2] != ik.x[2]) {
				iv_push(curr, &ik);
				if (ok[c].x[2] < (uint64_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else {
			iv_push(curr, &ik);
			break;
		}
	}
	if (i == len) iv_push(curr, &ik);
	iv_reverse(curr);               /* longest match first */
	ret = (int)curr->a[0].info;
	swap = curr; curr = prev; prev = swap;

	/* backward: extend every surviving interval; emit those that die first */
	for (i = x - 1; i >= -1; --i) {
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < prev->n; ++j) {
			orc_intv_t *p = &prev->a[j];
			if (c >= 0 && ik.x[2] >= max_intv) { sp_class = sp_pass == 1 ? 1 : 3; sp_len = (int)(uint32_t)p->info - i; orc_extend(b, p, ok, 1); }
			if (c < 0 || ik.x[2] < max_intv || ok[c].x[2] < (uint64_t)min_intv) {
				if (curr->n == 0) {
					if (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n - 1].info >> 32) {
						ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
						iv_push(mem, &ik);
					}
				}
			} else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2]) {
				ok[c].info = p->info;
				iv_push(curr, &ok[c]);
			}
		}
		if (curr->n == 0) break;
		swap = curr; curr = prev; prev = swap;
	}
	iv_reverse(mem);                /* by start coordinate */
	return ret;
}

int orc_seed_strategy1(const orc_idx_t *b, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem)
{
	int i, c;
	orc_intv_t ik, ok[4];
	memset(mem, 0, sizeof(*mem));
	if (q[x] > 3) return x + 1;
	set_intv(b, q[x], &ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			sp_class = 4; sp_len = i - x + 1;
			orc_extend(b, &ik, ok, 0);
			if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) {
				*mem = ok[c];
				mem->info = (uint64_t)x << 32 | (uint32_t)(i + 1);
				return i + 1;
			}
			ik = ok[c];
		} else return i + 1;
	}
	return len;
}

#define intv_lt(a, b) ((a).info < (b).info)
ORC_SORT_INIT(intv, orc_intv_t, intv_lt)
#define u64_lt(a, b) ((a) < (b))
ORC_SORT_INIT(k64, uint64_t, u64_lt)
void orc_introsort_u64(size_t n, uint64_t *a) { orc_introsort_k64(n, a); }

void orc_collect_intv(const orc_opt_t *opt, const orc_idx_t *b, int len, const uint8_t *seq, orc_intv_v *mem)
{
	int x = 0, split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	size_t i, k, old_n;
	orc_intv_v mem1 = {0, 0, 0}, t0 = {0, 0, 0}, t1 = {0, 0, 0};
	mem->n = 0;
	/* pass 1: all SMEMs */
	sp_pass = 1;
	while (x < len) {
		if (seq[x] < 4) {
			x = orc_smem1a(b, len, seq, x, 1, 0, &mem1, &t0, &t1);
			for (i = 0; i < mem1.n; ++i) {
				orc_intv_t *p = &mem1.a[i];
				int slen = (int)((uint32_t)p->info - (p->info >> 32));
				if (slen >= opt->min_seed_len) iv_push(mem, p);
			}
		} else ++x;
	}
	/* pass 2: re-seed inside long, rare SMEMs */
	old_n = mem->n;
	sp_pass = 2;
	for (k = 0; k < old_n; ++k) {
		orc_intv_t *p = &mem->a[k];
		int start = (int)(p->info >> 32), end = (int32_t)p->info;
		if (end - start < split_len || p->x[2] > (uint64_t)opt->split_width) continue;
		orc_smem1a(b, len, seq, (start + end) >> 1, (int)p->x[2] + 1, 0, &mem1, &t0, &t1);
		for (i = 0; i < mem1.n; ++i)
			if ((int)((uint32_t)mem1.a[i].info - (mem1.a[i].info >> 32)) >= opt->min_seed_len)
				iv_push(mem, &mem1.a[i]);
	}
	/* pass 3: LAST-like seeds */
	if (opt->max_mem_intv > 0) {
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				orc_intv_t m;
				x = orc_seed_strategy1(b, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, &m);
				if (m.x[2] > 0) iv_push(mem, &m);
			} else ++x;
		}
	}
	orc_introsort_intv(mem->n, mem->a);
	free(mem1.a); free(t0.a); free(t1.a);
}

/* ------------------------------------------------------------------ */
/* bntseq.c */

int orc_pos2rid(const orc_idx_t *idx, int64_t pos_f)
{
	int left, mid, right;
	if (pos_f >= idx->l_pac) return -1;
	left = 0; mid = 0; right = idx->n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= idx->anns[mid].offset) {
			if (mid == idx->n_seqs - 1) break;
			if (pos_f < idx->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

static inline int64_t depos(const orc_idx_t *idx, int64_t pos, int *is_rev)
{
	return (*is_rev = (pos >= idx->l_pac)) ? (idx->l_pac << 1) - 1 - pos : pos;
}

int orc_intv2rid(const orc_idx_t *idx, int64_t rb, int64_t re)
{
	int is_rev, rid_b, rid_e;
	if (rb < idx->l_pac && re > idx->l_pac) return -2;
	assert(rb <= re);
	rid_b = orc_pos2rid(idx, depos(idx, rb, &is_rev));
	rid_e = rb < re ? orc_pos2rid(idx, depos(idx, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

#define PAC_GET(pac, l) ((pac)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

uint8_t *orc_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len)
{
	uint8_t *seq = 0;
	if (end < beg) { int64_t t = beg; beg = end; end = t; }
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	if (beg >= l_pac || end <= l_pac) {
		int64_t k, l = 0;
		*len = end - beg;
		seq = malloc(end - beg > 0 ? end - beg : 1);
		if (beg >= l_pac) {
			int64_t beg_f = (l_pac << 1) - 1 - end, end_f = (l_pac << 1) - 1 - beg;
			for (k = end_f; k > beg_f; --k) seq[l++] = 3 - PAC_GET(pac, k);
		} else {
			for (k = beg; k < end; ++k) seq[l++] = PAC_GET(pac, k);
		}
		orc_stats.w_ref += (uint64_t)(end - beg);
	} else *len = 0;
	return seq;
}

uint8_t *orc_fetch_seq(const orc_idx_t *idx, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	int64_t far_beg, far_end, len;
	int is_rev;
	uint8_t *seq;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	assert(*beg <= mid && mid < *end);
	*rid = orc_pos2rid(idx, depos(idx, mid, &is_rev));
	far_beg = idx->anns[*rid].offset;
	far_end = far_beg + idx->anns[*rid].len;
	if (is_rev) {
		int64_t t = far_beg;
		far_beg = (idx->l_pac << 1) - far_end;
		far_end = (idx->l_pac << 1) - t;
	}
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	seq = orc_get_seq(idx->l_pac, idx->pac, *beg, *end, &len);
	assert(seq && *end - *beg == len);
	return seq;
}
