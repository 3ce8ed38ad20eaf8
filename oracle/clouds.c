/*
 * oracle/clouds.c -- CPU oracle for the cloud / EM / duplicate-marking stage.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * A single-thread, line-for-line restatement of what find_clouds_and_align() does with one barcode group once
 * append_alignments() has filled `records` (reference src/align.c:347-608, i.e. `-t 1` semantics), together with the
 * dictionary it works through (src/samdict.c:11-243), the comparators (src/samrecord.c:51-73 record_cmp, src/align.c:71-122
 * name_cmp / dup_cmp), the hashes (src/samrecord.c:12-49, src/util.c:121-128), init_cloud (src/align.c:17-25),
 * normalize_cloud_probabilities (src/align.c:124-143) and normalize_log_probs (src/util.c:130-163).  libc's qsort is called
 * where the reference calls it, on the same element sizes' worth of ordering (the records array is sorted through an index
 * array with the same comparator and a stable fallback: glibc's qsort is a merge sort for these sizes).
 * [r5] The -d density optimisation (mark_optimal_alignments_in_cloud, src/split.c:15-339 with include/split.h's constants) is
 * restated below as well, statement for statement -- the draws from libc's rand() and the order of the double-precision sums ARE
 * its output; off unless orc_clouds_set_density switches it on (apply_opt, src/align.c:396).
 * These files ARE in the reference tree but include bwa's headers (through bwabridge.h), so they cannot be compiled here:
 * unpinned, like oracle/sam.c.
 */
#define _GNU_SOURCE
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define EM_ITERS 5
#define INSERT_MIN (-35)
#define INSERT_MAX 750
#define UNPAIRED_PENALTY (-15.0)
#define SECONDARY_ALIGN_THRESH 0.9
#define MAX_CANDIDATES 5000
#define DEFAULT_CANDS 5
#define SAM_DICT_CAP_SMALL (1 << 16)
#define SAM_DICT_CAP_LARGE (1 << 25)

typedef struct cloud {
	double exp_cov, weight;
	struct cloud *parent, *child;
	int id;
	unsigned bad : 1;
} Cloud;

typedef orc_crec_t SAMRecord;

typedef struct sam_dict_ent {
	SAMRecord *key;
	struct sam_dict_ent *clash_next, *link_next, *mate;
	size_t num_cands, cap_cands;
	SAMRecord **cand_records;
	Cloud **cand_clouds;
	double *gammas;
	unsigned visited : 1;
} SAMDictEnt;

typedef struct { SAMDictEnt *head; SAMDictEnt **entries; uint32_t count, cap; } SAMDict;

static uint32_t hash_ident(const char *ident)
{
	uint32_t h = 0;
	for (size_t i = 0; ident[i] != '\0'; i++) h = 31 * h + ident[i];
	return h;
}
static uint32_t record_hash(SAMRecord *r)
{
	if (r->hashed) return r->hash;
	r->hash = hash_ident(r->ident) * (r->mate + 1); r->hashed = 1;
	return r->hash;
}
static uint32_t record_hash_mate(SAMRecord *r)
{
	if (r->mate_hashed) return r->mate_hash;
	r->mate_hash = hash_ident(r->ident) * (2 - r->mate); r->mate_hashed = 1;
	return r->mate_hash;
}
static uint32_t record_eq(SAMRecord *r1, SAMRecord *r2) { return record_hash(r1) == record_hash(r2) && r1->mate == r2->mate && strcmp(r1->ident, r2->ident) == 0; }
static uint32_t record_eq_mate(SAMRecord *r1, SAMRecord *r2) { return record_hash_mate(r1) == record_hash(r2) && r1->mate != r2->mate && strcmp(r1->ident, r2->ident) == 0; }

static int record_cmp(const void *v1, const void *v2)
{
	const SAMRecord *r1 = *(SAMRecord *const *)v1, *r2 = *(SAMRecord *const *)v2;
	int c = (r1->bc > r2->bc) - (r1->bc < r2->bc);
	if (c != 0) return c;
	const uint8_t chrom1 = (uint8_t)r1->chrom, chrom2 = (uint8_t)r2->chrom;      /* the reference narrows to 8 bits here */
	c = (chrom1 > chrom2) - (chrom1 < chrom2);
	if (c != 0) return c;
	c = (r1->pos > r2->pos) - (r1->pos < r2->pos);
	if (c != 0) return c;
	c = strcmp(r1->ident, r2->ident);
	if (c != 0) return c;
	return (r1->orig > r2->orig) - (r1->orig < r2->orig);      /* equal keys keep their order: what glibc's merge sort gives */
}

static int name_cmp(const void *v1, const void *v2)
{
	SAMRecord **r1 = (SAMRecord **)v1, **r2 = (SAMRecord **)v2;
	const uint32_t m1 = (*r1)->mate, m2 = (*r2)->mate;
	const int cmp1 = strcmp((*r1)->ident, (*r2)->ident), cmp2 = (m1 > m2) - (m1 < m2);
	return (cmp1 != 0) ? cmp1 : cmp2;
}

static int dup_cmp(const void *v1, const void *v2)
{
	int cmp[6];
	SAMRecord *r1 = *((SAMRecord **)v1), *r2 = *((SAMRecord **)v2);
	SAMRecord *mate1 = r1->sel_mate, *mate2 = r2->sel_mate;
	const uint32_t m1 = r1->mate, m2 = r2->mate, rev1 = r1->rev, rev2 = r2->rev, c1 = r1->chrom, c2 = r2->chrom, p1 = r1->pos, p2 = r2->pos;
	const uint32_t mc1 = mate1 ? mate1->chrom : (uint32_t)-1, mc2 = mate2 ? mate2->chrom : (uint32_t)-1;
	const uint32_t mp1 = mate1 ? mate1->pos : (uint32_t)-1, mp2 = mate2 ? mate2->pos : (uint32_t)-1;
	cmp[0] = (m1 > m2) - (m1 < m2); cmp[1] = (rev1 > rev2) - (rev1 < rev2); cmp[2] = (c1 > c2) - (c1 < c2);
	cmp[3] = (p1 > p2) - (p1 < p2); cmp[4] = (mc1 > mc2) - (mc1 < mc2); cmp[5] = (mp1 > mp2) - (mp1 < mp2);
	for (size_t i = 0; i < 6; i++) if (cmp[i] != 0) return cmp[i];
	return 0;
}

static void normalize_log_probs(double *p, const size_t n)
{
	if (n == 1) { p[0] = 1.0; return; }
	const double thresh = log(1e-50) - log(n);
	double p_max = p[0];
	for (size_t i = 1; i < n; i++) if (p[i] > p_max) p_max = p[i];
	double total = 0;
	for (size_t i = 0; i < n; i++) {
		p[i] -= p_max;
		if (p[i] < thresh) p[i] = 0; else p[i] = exp(p[i]);
		total += p[i];
	}
	for (size_t i = 0; i < n; i++) p[i] /= total;
}

static void normalize_cloud_probabilities(Cloud *clouds, const size_t nc)
{
	for (size_t i = 0; i < nc; i++) {
		Cloud *c = &clouds[i];
		if (c->parent != NULL) continue;
		double total = 0.0;
		for (Cloud *child = c; child != NULL; child = child->child) total += child->weight;
		for (Cloud *child = c; child != NULL; child = child->child) child->weight /= total;
	}
}

static SAMDictEnt *sde_new(SAMRecord *key, Cloud *v)
{
	SAMDictEnt *sde = malloc(sizeof(*sde));
	sde->key = key; sde->mate = NULL;
	sde->cand_records = malloc(DEFAULT_CANDS * sizeof(*(sde->cand_records)));
	sde->cand_clouds = malloc(DEFAULT_CANDS * sizeof(*(sde->cand_clouds)));
	sde->gammas = malloc(DEFAULT_CANDS * sizeof(*(sde->gammas)));
	sde->cand_records[0] = key; sde->cand_clouds[0] = v; sde->gammas[0] = key->score;
	sde->cap_cands = DEFAULT_CANDS; sde->num_cands = 1; sde->visited = 0;
	return sde;
}
static void sde_free(SAMDictEnt *sde) { free(sde->cand_records); free(sde->cand_clouds); free(sde->gammas); free(sde); }

static SAMDictEnt *find_for_key(SAMDict *sd, SAMRecord *k)
{
	const uint32_t idx = (record_hash(k) & (sd->cap - 1));
	for (SAMDictEnt *e = sd->entries[idx]; e != NULL; e = e->clash_next) if (record_eq(k, e->key)) return e;
	return NULL;
}
static SAMDictEnt *find_mate_for_key(SAMDict *sd, SAMRecord *k)
{
	const uint32_t idx = (record_hash_mate(k) & (sd->cap - 1));
	for (SAMDictEnt *e = sd->entries[idx]; e != NULL; e = e->clash_next) if (record_eq_mate(k, e->key)) return e;
	return NULL;
}

static int sam_dict_add(SAMDict *sd, SAMRecord *k, Cloud *v, const int force, int many_clouds)
{
	SAMDictEnt *e = find_for_key(sd, k);
	if (e != NULL) {
		const size_t num_cands = e->num_cands;
		if (num_cands < MAX_CANDIDATES) {
			if (num_cands > 0) {
				Cloud *parent = e->cand_clouds[num_cands - 1];
				if (parent == v && !force) return 1;
				if (!many_clouds) {      /* link in the disjoint-set structure */
					Cloud *root1 = parent;
					while (root1->parent != NULL) root1 = root1->parent;
					Cloud *root2 = v;
					while (root2->parent != NULL) root2 = root2->parent;
					if (root1 != root2) {
						Cloud *leaf = parent;
						while (leaf->child != NULL) leaf = leaf->child;
						root2->parent = leaf; leaf->child = root2;
					}
				}
			}
			if (num_cands == e->cap_cands) {
				const size_t new_cap = (e->cap_cands * 3) / 2 + 1;
				e->cand_records = realloc(e->cand_records, new_cap * sizeof(*(e->cand_records)));
				e->cand_clouds = realloc(e->cand_clouds, new_cap * sizeof(*(e->cand_clouds)));
				e->gammas = realloc(e->gammas, new_cap * sizeof(*(e->gammas)));
				e->cap_cands = new_cap;
			}
			e->cand_clouds[num_cands] = v; e->cand_records[num_cands] = k; e->gammas[num_cands] = k->score;
			++(e->num_cands);
		}
	} else {
		const uint32_t idx = (record_hash(k) & (sd->cap - 1));
		e = sde_new(k, v);
		e->link_next = sd->head; sd->head = e;
		e->clash_next = sd->entries[idx]; sd->entries[idx] = e;
		SAMDictEnt *mate = find_mate_for_key(sd, k);
		if (mate != NULL) { e->mate = mate; mate->mate = e; }
		++(sd->count);
	}
	return 0;
}
static void sam_dict_del(SAMDict *sd, SAMRecord *k)
{
	SAMDictEnt *e = find_for_key(sd, k);
	if (e != NULL) --(e->num_cands);
}

static SAMRecord *find_best_record(SAMDictEnt *e)
{
	SAMRecord **cand_records = e->cand_records;
	double *gammas = e->gammas;
	size_t best = 0;
	double best_gamma = -1.0;
	const size_t num_cands = e->num_cands;
	for (size_t i = 0; i < num_cands; i++) {
		if (!cand_records[i]->active) continue;
		if (gammas[i] > best_gamma) { best = i; best_gamma = gammas[i]; }
	}
	SAMRecord *chosen = cand_records[best];
	chosen->alt = -1;
	chosen->gamma = best_gamma;
	chosen->cloud_id = e->cand_clouds[best]->id;
	chosen->cloud_bad = (int)e->cand_clouds[best]->bad;
	if (best_gamma <= SECONDARY_ALIGN_THRESH) {
		size_t second_best = 0;
		double second_best_gamma = -1.0;
		for (size_t i = 0; i < num_cands; i++) {
			if (!cand_records[i]->active) continue;
			if (i != best && gammas[i] > second_best_gamma) { second_best = i; second_best_gamma = gammas[i]; }
		}
		if (second_best_gamma > 0) chosen->alt = (int)cand_records[second_best]->orig;      /* the record the XA entry is copied from */
	}
	return chosen;
}

/* ---- -d: src/split.c ---------------------------------------------------------------------------------------------------- */
#define SPLIT_EXTRA_SEARCH_DEPTH 5       /* include/align.h:74 */
#define TMAX_LOG 0.0                     /* include/split.h:8-17 */
#define TMIN_LOG (-12.0)
#define SIM_ANNEAL_ITERS 50000
#define BIN_SIZE 1000
#define MAX_BINS (1000000 / BIN_SIZE)    /* MAX_FRAG / BIN_SIZE */
#define SCORE_SCALE 20
#define MAX_NO_MOVE 500
#define BUF_SIZE 50000
static int g_apply_opt = 0, g_n_density = 4;
static double g_density_probs[16] = {0.6, 0.05, 0.2, 0.01};      /* src/techs.c:76-79: every platform but cpt */
void orc_clouds_set_density(int apply_opt, int n_probs, const double *probs)
{
	g_apply_opt = apply_opt;
	if (probs && n_probs > 0 && n_probs <= 16) { g_n_density = n_probs; for (int i = 0; i < n_probs; i++) g_density_probs[i] = probs[i]; }
}
void orc_clouds_reseed(unsigned seed) { srand(seed); }

static double log_density_prob(unsigned int density)      /* src/split.c:15-35 */
{
	const size_t size = (size_t)g_n_density;
	if (density < size) return log(g_density_probs[density]);
	return log(g_density_probs[size - 1]) - (density - size + 1) * log(2.0);
}

static int split_is_pair(SAMRecord *r1, SAMRecord *r2)      /* is_pair, src/align.c:27-40 (pos is uint32_t there: the difference wraps) */
{
	if (r1->rev == r2->rev || r1->chrom != r2->chrom) return 0;
	if (r2->rev) { SAMRecord *rt = r2; r2 = r1; r1 = rt; }
	const int64_t d = (int64_t)(uint32_t)(r1->pos - r2->pos);
	return INSERT_MIN <= d && d <= INSERT_MAX;
}

#define BIN_IDX_FOR_POS(pos, lo) (((pos) - (lo)) / BIN_SIZE)
/* caution: `records` should be name-sorted */
static void mark_optimal_alignments_in_cloud(SAMRecord **records, size_t n_records)      /* src/split.c:38-339 */
{
	double log_config_prob = 0;
	static __thread unsigned short bins[MAX_BINS];
	struct uniquemapped_read { size_t idx; };
	struct multimapped_read { size_t idx; int n, mate_umap, mate_mmap, active; };
	size_t n_umaps = 0, n_mmaps = 0;
	uint32_t cloud_lo = 0xffffffff, cloud_hi = 0x00000000;
	if (n_records >= BUF_SIZE || n_records <= 5) return;
	memset(bins, 0, sizeof(bins));
	struct uniquemapped_read *umaps = malloc((n_records + 1) * sizeof(*umaps));
	struct multimapped_read *mmaps = malloc((n_records + 1) * sizeof(*mmaps));
	/* remove records that are too far from the lowest edit distance */
	SAMRecord **records_clean = malloc(n_records * sizeof(*records_clean));
	size_t n_records_clean = 0;
	for (size_t i = 0; i < n_records;) {
		size_t j = i + 1;
		while (j < n_records && record_eq(records[j], records[i])) ++j;
		const size_t n = j - i;
		if (n > 1) {
			size_t min_edit_dist = 0;
			for (size_t k = 0; k < n; k++) if (records[i + k]->clip_edit_dist < records[i + min_edit_dist]->clip_edit_dist) min_edit_dist = k;
			SAMRecord *best = records[i + min_edit_dist];
			const int edit_dist_cutoff = best->clip_edit_dist + SPLIT_EXTRA_SEARCH_DEPTH;
			for (size_t k = 0; k < n; k++) {
				if (records[i + k]->clip_edit_dist <= edit_dist_cutoff) records_clean[n_records_clean++] = records[i + k];
				else records[i + k]->active = 0;
			}
		} else records_clean[n_records_clean++] = records[i];
		i = j;
	}
	records = records_clean;
	n_records = n_records_clean;
	/* find the multi-mapped reads, record highest */
	for (size_t i = 0; i < n_records;) {
		if (records[i]->pos < cloud_lo) cloud_lo = records[i]->pos;
		if (records[i]->pos > cloud_hi) cloud_hi = records[i]->pos;
		size_t j = i + 1;
		while (j < n_records && record_eq(records[j], records[i])) {
			if (records[j]->pos < cloud_lo) cloud_lo = records[j]->pos;
			if (records[j]->pos > cloud_hi) cloud_hi = records[j]->pos;
			++j;
		}
		const size_t n = j - i;
		if (n > 1) {
			size_t max_score = 0;
			for (size_t k = 0; k < n; k++) if (records[i + k]->score > records[i + max_score]->score) max_score = k;
			int mate_umap = -1, mate_mmap = -1;
			for (size_t k = 0; k < n_umaps; k++) if (record_eq_mate(records[i], records[umaps[k].idx])) { mate_umap = (int)k; break; }
			if (mate_umap < 0) {
				for (size_t k = 0; k < n_mmaps; k++) {
					if (record_eq_mate(records[i], records[mmaps[k].idx])) { mate_mmap = (int)k; mmaps[k].mate_mmap = (int)n_mmaps; break; }
				}
			}
			mmaps[n_mmaps++] = (struct multimapped_read){.idx = i, .n = (int)n, .mate_umap = mate_umap, .mate_mmap = mate_mmap, .active = (int)max_score};
			log_config_prob += records[i + max_score]->score / SCORE_SCALE;
		} else {
			for (size_t k = 0; k < n_mmaps; k++) if (record_eq_mate(records[i], records[mmaps[k].idx])) { mmaps[k].mate_umap = (int)n_umaps; break; }
			umaps[n_umaps++] = (struct uniquemapped_read){.idx = i};
			log_config_prob += records[i]->score / SCORE_SCALE;
		}
		i = j;
	}
	/* get initial configuration probability */
	const size_t n_bins = (cloud_hi - cloud_lo) / BIN_SIZE + 1;
	if (n_bins >= MAX_BINS || n_records <= 5 || n_mmaps == 0) { free(records); free(umaps); free(mmaps); return; }
	for (size_t i = 0; i < n_records; i++) records[i]->active = 0;      /* we will re-set the active ones later */
	for (size_t i = 0; i < n_umaps; i++) ++bins[BIN_IDX_FOR_POS(records[umaps[i].idx]->pos, cloud_lo)];
	for (size_t i = 0; i < n_mmaps; i++) ++bins[BIN_IDX_FOR_POS(records[mmaps[i].idx + mmaps[i].active]->pos, cloud_lo)];
	for (size_t i = 0; i < n_bins; i++) log_config_prob += log_density_prob(bins[i]);
	/* simulated annealing to minimize distance variance */
	int no_move_count = 0;
	for (size_t k = 0; k < SIM_ANNEAL_ITERS; k++) {
		const double t = pow(10.0, TMAX_LOG - ((TMAX_LOG - TMIN_LOG) * k) / SIM_ANNEAL_ITERS);
		size_t r = rand() % n_mmaps;
		size_t r_old = mmaps[r].active;
		size_t r_new = rand() % (mmaps[r].n - 1);
		if (r_new >= r_old) ++r_new;
		SAMRecord *active_mate = NULL;
		size_t mate_r = 0;
		int mate_is_mmap = 0;
		if (mmaps[r].mate_umap >= 0) { mate_r = mmaps[r].mate_umap; active_mate = records[umaps[mate_r].idx]; }
		else if (mmaps[r].mate_mmap >= 0) { mate_r = mmaps[r].mate_mmap; active_mate = records[mmaps[mate_r].idx + mmaps[mate_r].active]; mate_is_mmap = 1; }
		SAMRecord *rec_old = records[mmaps[r].idx + r_old];
		SAMRecord *rec_new = records[mmaps[r].idx + r_new];
		double density_prob_change = 0.0, score_prob_change = 0.0;
		int force_move = 0, mate_new_active = -1;
		size_t mate_old_bin = 0, mate_new_bin = 0;
		const int old_paired = active_mate != NULL && split_is_pair(rec_old, active_mate);
		const int new_paired = active_mate != NULL && split_is_pair(rec_new, active_mate);
		if (!old_paired && new_paired) force_move = 1;
		else if (old_paired && !new_paired && mate_is_mmap) {
			for (int i = 0; i < mmaps[mate_r].n; i++) {
				SAMRecord *mate_rec_new = records[mmaps[mate_r].idx + i];
				if (split_is_pair(rec_new, mate_rec_new)) {
					SAMRecord *mate_rec_old = active_mate;
					mate_new_active = i;
					mate_old_bin = BIN_IDX_FOR_POS(mate_rec_old->pos, cloud_lo);
					mate_new_bin = BIN_IDX_FOR_POS(mate_rec_new->pos, cloud_lo);
					score_prob_change += (mate_rec_new->score - mate_rec_old->score) / SCORE_SCALE;
					break;
				}
			}
		}
		const size_t old_bin = BIN_IDX_FOR_POS(rec_old->pos, cloud_lo), new_bin = BIN_IDX_FOR_POS(rec_new->pos, cloud_lo);
		const int p1 = (mate_new_active >= 0 && old_bin == mate_old_bin) ? 2 : 1;
		const int p2 = (mate_new_active >= 0 && new_bin == mate_new_bin) ? 2 : 1;
		{
			const double old_bin_prob_old = log_density_prob(bins[old_bin]);
			const double old_bin_prob_new = log_density_prob(bins[old_bin] - p1);
			const double new_bin_prob_old = log_density_prob(bins[new_bin]);
			const double new_bin_prob_new = log_density_prob(bins[new_bin] + p2);
			density_prob_change += (old_bin_prob_new - old_bin_prob_old) + (new_bin_prob_new - new_bin_prob_old);
		}
		if (p1 == 1 && mate_new_active >= 0) {
			const double a = log_density_prob(bins[mate_old_bin]), b = log_density_prob(bins[mate_old_bin] - 1);
			density_prob_change += (b - a);
		}
		if (p2 == 1 && mate_new_active >= 0) {
			const double a = log_density_prob(bins[mate_new_bin]), b = log_density_prob(bins[mate_new_bin] + 1);
			density_prob_change += (b - a);
		}
		score_prob_change += (rec_new->score - rec_old->score) / SCORE_SCALE;
		const double prob_change = density_prob_change + score_prob_change;
		if (force_move || prob_change > 0 || exp(prob_change / t) >= ((double)rand()) / RAND_MAX) {
			log_config_prob += prob_change;
			mmaps[r].active = (int)r_new;
			bins[old_bin] -= 1;
			bins[new_bin] += 1;
			if (mate_new_active >= 0) {
				mmaps[mate_r].active = mate_new_active;
				bins[mate_old_bin] -= 1;
				bins[mate_new_bin] += 1;
			}
		} else ++no_move_count;
		if (no_move_count >= MAX_NO_MOVE) break;
	}
	(void)log_config_prob;
	for (size_t i = 0; i < n_umaps; i++) records[umaps[i].idx]->active = 1;
	for (size_t i = 0; i < n_mmaps; i++) records[mmaps[i].idx + mmaps[i].active]->active = 1;
	free(records); free(umaps); free(mmaps);
}

static double mate_dist_penalty(const int64_t mate1_pos, const int64_t mate2_pos)
{
	const int64_t d = mate1_pos - mate2_pos;
	return (INSERT_MIN <= d && d <= INSERT_MAX) ? 0.0 : UNPAIRED_PENALTY;
}

/* One barcode group.  recs[0..n): the group's records in append_alignments' order (recs[i].orig == i on entry; every record
 * `active`, none duplicate); n_pairs: read pairs of the group (full EM from 30 on).  *cloud_id: the process-wide counter of
 * init_cloud.  order[]: room for 2 * n ints; on return the print order as (record, its selected mate or -1) index pairs --
 * the reference prints print_sam_record(rec, mate) then print_sam_record(mate, rec) for each.  Returns the number of pairs
 * written to order[].  Outputs per record: gamma, cloud_id, cloud_bad, duplicate, alt. */
size_t orc_clouds_group(orc_crec_t *recs, size_t n, size_t n_pairs, uint32_t dist_thresh, int many_clouds, int *cloud_id, int *order)
{
	if (n == 0) return 0;
	const uint64_t bc = recs[0].bc;
	SAMRecord **sorted = malloc((n + 1) * sizeof(*sorted));
	for (size_t i = 0; i < n; i++) sorted[i] = &recs[i];
	qsort(sorted, n, sizeof(*sorted), record_cmp);
	/* the reference sorts the structs themselves; here `records` is that array seen through the index */
	SAMRecord sentinel; memset(&sentinel, 0, sizeof(sentinel)); sentinel.bc = 0;
	sorted[n] = &sentinel;
	Cloud *clouds = malloc((n + 1) * sizeof(*clouds));
	size_t nc = 0;
	SAMDict sdv, *sd = &sdv;
	sd->cap = many_clouds ? SAM_DICT_CAP_LARGE : SAM_DICT_CAP_SMALL;
	sd->entries = calloc(sd->cap, sizeof(*sd->entries));
	sd->head = NULL; sd->count = 0;
	const int worth_doing_full_em = (n_pairs >= 30);
	size_t n_records_final = 0;
	SAMRecord **records_final = malloc(2 * n_pairs * sizeof(*records_final) + 16);

	size_t at = 0;      /* `record` */
	while (sorted[at]->bc == bc) {
		size_t r = at;
		Cloud *c = &clouds[nc];
		c->exp_cov = 0.0; c->parent = NULL; c->child = NULL; c->id = (*cloud_id)++; c->bad = 0;      /* init_cloud */
		sam_dict_add(sd, sorted[r], c, 0, many_clouds);
		size_t cov = 1;
		int collision_detected = 0;
		while (sorted[r + 1]->bc == sorted[r]->bc && sorted[r + 1]->chrom == sorted[r]->chrom &&
		       (uint32_t)(sorted[r + 1]->pos - sorted[r]->pos) <= dist_thresh) {
			++r;
			if (!collision_detected && sam_dict_add(sd, sorted[r], c, 0, many_clouds)) {
				collision_detected = 1;
				for (size_t i = 0; i < cov; i++) sam_dict_del(sd, sorted[at + i]);
			}
			++cov;
		}
		if (collision_detected) {
			c->bad = 1;
			SAMRecord **cloud_to_split = malloc(cov * sizeof(*cloud_to_split));
			for (size_t i = 0; i < cov; i++) cloud_to_split[i] = sorted[at + i];
			qsort(cloud_to_split, cov, sizeof(*cloud_to_split), name_cmp);
			if (g_apply_opt) mark_optimal_alignments_in_cloud(cloud_to_split, cov);      /* src/align.c:396-397 */
			for (size_t i = 0; i < cov; i++) sam_dict_add(sd, cloud_to_split[i], &clouds[nc], 1, many_clouds);
			free(cloud_to_split);
		}
		++nc;
		at = r + 1;
	}

	for (SAMDictEnt *e = sd->head; e != NULL; e = e->link_next) {
		normalize_log_probs(e->gammas, e->num_cands);
		for (size_t i = 0; i < e->num_cands; i++) e->cand_clouds[i]->exp_cov += e->gammas[i];
	}
	for (size_t i = 0; i < nc; i++) clouds[i].weight = clouds[i].exp_cov;
	if (!many_clouds) normalize_cloud_probabilities(clouds, nc);

	for (int q = 0; q < EM_ITERS; q++) {
		if (!worth_doing_full_em) break;
		for (size_t i = 0; i < nc; i++) clouds[i].exp_cov = 0.0;
		for (SAMDictEnt *e = sd->head; e != NULL; e = e->link_next) {
			SAMDictEnt *mate = e->mate;
			SAMRecord **records = e->cand_records;
			Cloud **cl = e->cand_clouds;
			double *gammas = e->gammas;
			const size_t num_cands = e->num_cands;
			double *cloud_weights = malloc((num_cands + 1) * sizeof(double));
			double cw_tot = 0;
			if (many_clouds) {
				for (size_t i = 0; i < num_cands; i++) { cloud_weights[i] = cl[i]->weight; cw_tot += cloud_weights[i]; }
				for (size_t i = 0; i < num_cands; i++) cloud_weights[i] /= cw_tot;
			}
			for (size_t i = 0; i < num_cands; i++) {
				double best_mate_score = UNPAIRED_PENALTY;
				if (mate != NULL) {
					for (size_t j = 0; j < mate->num_cands; j++) {
						if (mate->cand_records[j]->chrom == records[i]->chrom && mate->cand_records[j]->rev != records[i]->rev &&
						    mate->cand_clouds[j] == cl[i] && mate->gammas[j] != 0.0) {
							const double penalty = records[i]->rev ? mate_dist_penalty(records[i]->pos, mate->cand_records[j]->pos)
							                                       : mate_dist_penalty(mate->cand_records[j]->pos, records[i]->pos);
							const double mate_score = penalty + log(mate->gammas[j]);
							if (mate_score > best_mate_score) best_mate_score = mate_score;
						}
					}
				}
				gammas[i] = records[i]->score + (many_clouds ? log(cloud_weights[i]) : log(cl[i]->weight)) + best_mate_score;
			}
			normalize_log_probs(gammas, num_cands);
			free(cloud_weights);
		}
		for (SAMDictEnt *e = sd->head; e != NULL; e = e->link_next)
			for (size_t i = 0; i < e->num_cands; i++)
				if (e->cand_records[i]->active && !e->cand_records[i]->duplicate) e->cand_clouds[i]->exp_cov += e->gammas[i];
		for (size_t i = 0; i < nc; i++) clouds[i].weight = clouds[i].exp_cov;
		if (!many_clouds) normalize_cloud_probabilities(clouds, nc);
	}

	SAMDictEnt *e = sd->head;
	while (e != NULL) {
		if (!e->visited) {
			SAMDictEnt *m = e->mate;
			SAMRecord *best = find_best_record(e);
			SAMRecord *best_mate = (m != NULL) ? find_best_record(m) : NULL;
			records_final[n_records_final++] = best;
			best->sel_mate = best_mate;
			if (best_mate != NULL) { records_final[n_records_final++] = best_mate; best_mate->sel_mate = best; }
			e->visited = 1;
			if (m != NULL) { m->visited = 1; m->mate = NULL; }
		}
		SAMDictEnt *t = e;
		e = e->link_next;
		sde_free(t);
	}
	if (!many_clouds) {
		qsort(records_final, n_records_final, sizeof(*records_final), dup_cmp);
		for (size_t i = 0; i < n_records_final;) {
			size_t j = i + 1;
			while (j < n_records_final && dup_cmp(&records_final[i], &records_final[j]) == 0) { records_final[j]->duplicate = 1; j++; }
			i = j;
		}
	}
	size_t n_out = 0;
	for (size_t i = 0; i < n_records_final; i++) {
		SAMRecord *best = records_final[i], *best_mate = best->sel_mate;
		if (best->visited) continue;
		if (best_mate != NULL) best_mate->visited = 1;
		order[2 * n_out] = (int)best->orig;
		order[2 * n_out + 1] = best_mate ? (int)best_mate->orig : -1;
		++n_out;
	}
	free(records_final); free(sd->entries); free(clouds); free(sorted);
	return n_out;
}
