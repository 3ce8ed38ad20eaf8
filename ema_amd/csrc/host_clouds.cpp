// ema_amd/csrc/host_clouds.cpp -- the cloud / EM / duplicate-marking stage behind the hot path (include/ema_clouds.h):
// what the reference's find_clouds_and_align() does with every barcode group after append_alignments()
// (reference src/align.c:347-608) and its SAMDict (src/samdict.c:11-243), for a whole bucket, barcode groups in parallel.
//
// Structure.  The reference keeps pointers between SAMRecords, Clouds and dictionary entries and a hash table keyed by read
// name; here a group's records, clouds and entries are index-addressed arrays in per-thread scratch that is reused from
// group to group, and the dictionary is a direct table over (rank of the read name within the group, mate) -- the same
// equivalence classes the reference's hash + strcmp gives.  Everything that decides an output is kept as the reference has
// it: the three sort orders (its qsort calls are merge sorts in glibc, i.e. stable: std::stable_sort with the same
// comparators), the order in which dictionary entries are visited (newest first: the reference pushes at the list head),
// the order of every floating-point accumulation, the expressions themselves (double precision, no contraction: the file is
// built with -ffp-contract=off), and the cloud numbering of a single-threaded run.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <thread>
#include <vector>
#include "ema_clouds.h"
#include "host_cpuacct.h"
#include "host_fmt.h"
#include "host_pool.h"

namespace {

// reference include/align.h:49-74, include/samdict.h:9
const int kEmIters = 5;
const int kInsertMin = -35, kInsertMax = 750;
const double kUnpairedPenalty = -15.0;
const double kSecondaryAlignThresh = 0.9;
const size_t kMaxCandidates = 5000;

struct Rec {
	uint32_t chrom, pos;          // chromosome index; 1-based position (alignment_to_sam_rec, src/align.c:922-923)
	int32_t rank;                 // rank of the read name among the group's names (strcmp order)
	double score, gamma;
	uint8_t mate, rev, duplicate, visited;
	uint8_t active;               // cleared by the density optimiser (-d) only
	int32_t clip_edit_dist;       // edit distance including clipping (src/align.c:936), for -d
	int32_t cloud, alt, sel_mate; // local cloud; record the XA entry is copied from; selected mate (record index) or -1
	uint64_t gi;                  // index in ema_aln_out.rec
};
struct Cloud { double exp_cov, weight; int32_t parent, child; uint8_t bad; };
struct Cand { int32_t rec, cloud; double gamma; };
struct Entry { int32_t key, mate; bool visited; std::vector<Cand> c; };

struct Name { const char *p; uint32_t len; uint32_t pair; };

inline int name_cmp(const Name &a, const Name &b)      // strcmp on the (NUL-free) names
{
	const uint32_t n = a.len < b.len ? a.len : b.len;
	const int c = memcmp(a.p, b.p, n);
	if (c) return c;
	return (a.len > b.len) - (a.len < b.len);
}

void normalize_log_probs(std::vector<Cand> &c)      // src/util.c:130-163
{
	const size_t n = c.size();
	if (n == 1) { c[0].gamma = 1.0; return; }
	if (n == 0) return;
	const double thresh = std::log(1e-50) - std::log((double)n);
	double p_max = c[0].gamma;
	for (size_t i = 1; i < n; i++) if (c[i].gamma > p_max) p_max = c[i].gamma;
	double total = 0;
	for (size_t i = 0; i < n; i++) {
		c[i].gamma -= p_max;
		if (c[i].gamma < thresh) c[i].gamma = 0; else c[i].gamma = std::exp(c[i].gamma);
		total += c[i].gamma;
	}
	for (size_t i = 0; i < n; i++) c[i].gamma /= total;
}

double mate_dist_penalty(int64_t mate1_pos, int64_t mate2_pos)      // src/align.c:57-67
{
	const int64_t d = mate1_pos - mate2_pos;
	return (kInsertMin <= d && d <= kInsertMax) ? 0.0 : kUnpairedPenalty;
}

struct Sel { uint64_t rec, mate; };      // indices into ema_aln_out.rec; mate == ~0: none

// per-thread scratch, reused from group to group
struct Work {
	std::vector<Rec> recs;
	std::vector<int32_t> ord, final_, split;
	std::vector<Cloud> clouds;
	std::vector<Entry> ents;
	size_t n_ents = 0;
	std::vector<int32_t> ent_of;      // (rank, mate) -> entry
	std::vector<Name> names;
	std::vector<int32_t> rank_of_pair;
	std::vector<double> log_weight;      // EM: log of a cloud's weight, taken once per round
	std::vector<uint8_t> have_log_weight;
	// [r6] the two sorts of a group on keys made once per element instead of through two indirections per comparison
	// (record order 22 % and duplicate order 18 % of this stage's CPU, r06: a quarter of both now)
	struct OrdKey { uint64_t key; int32_t idx; };
	struct DupKey { uint32_t k[6]; int32_t idx; };
	std::vector<OrdKey> ord_keys;
	std::vector<DupKey> dup_keys;
};

struct Shared {
	const ema_bucket *bk;
	const ema_batch_out *b;
	const ema_aln_out *a;
	ema_cloud_opts o;
	// per record of ema_aln_out (only the selected ones are read back)
	std::vector<double> gamma;
	std::vector<int32_t> cloud, alt_of;      // local cloud id; aln record the XA entry comes from (-1: none)
	std::vector<uint8_t> flags;              // bit 0 duplicate, bit 1 cloud bad
	std::vector<Sel> sel;                    // per group, from the group's first record index on
	std::vector<uint32_t> n_sel, n_clouds, n_bad;      // per group
	// -d with ema_cloud_opts.seed_private: this call's own stream of draws (glibc's rand() is random() on a 128-byte state; random_r
	// on a state of that size seeded by initstate_r is the same generator, minus the lock).  Drawn from on one thread, in group order.
	mutable struct random_data rng;
	mutable char rng_state[128];
	bool rng_private = false;
	int draw() const
	{
		if (!rng_private) return rand();
		int32_t v = 0;
		(void)random_r(&rng, &v);
		return (int)v;
	}
};

int sam_dict_add(Work &w, const Shared &S, int32_t k, int32_t v, bool force)      // src/samdict.c:79-148
{
	const Rec &kr = w.recs[(size_t)k];
	const size_t slot = 2 * (size_t)kr.rank + kr.mate;
	int32_t ei = w.ent_of[slot];
	if (ei >= 0) {
		Entry &e = w.ents[(size_t)ei];
		const size_t num = e.c.size();
		if (num < kMaxCandidates) {
			if (num > 0) {
				const int32_t parent = e.c[num - 1].cloud;
				if (parent == v && !force) return 1;
				if (!S.o.many_clouds) {      // link the two clouds' sets (chains through parent/child)
					int32_t root1 = parent;
					while (w.clouds[(size_t)root1].parent >= 0) root1 = w.clouds[(size_t)root1].parent;
					int32_t root2 = v;
					while (w.clouds[(size_t)root2].parent >= 0) root2 = w.clouds[(size_t)root2].parent;
					if (root1 != root2) {
						int32_t leaf = parent;
						while (w.clouds[(size_t)leaf].child >= 0) leaf = w.clouds[(size_t)leaf].child;
						w.clouds[(size_t)root2].parent = leaf;
						w.clouds[(size_t)leaf].child = root2;
					}
				}
			}
			e.c.push_back(Cand{k, v, kr.score});
		}
	} else {
		if (w.n_ents == w.ents.size()) w.ents.emplace_back();
		ei = (int32_t)w.n_ents++;
		Entry &e = w.ents[(size_t)ei];
		e.key = k; e.mate = -1; e.visited = false;
		e.c.clear();
		e.c.push_back(Cand{k, v, kr.score});
		w.ent_of[slot] = ei;
		const int32_t mi = w.ent_of[2 * (size_t)kr.rank + (1 - kr.mate)];      // the entry of the same read name, other mate
		if (mi >= 0) { e.mate = mi; w.ents[(size_t)mi].mate = ei; }
	}
	return 0;
}

void sam_dict_del(Work &w, int32_t k)      // src/samdict.c:150-157
{
	const Rec &kr = w.recs[(size_t)k];
	const int32_t ei = w.ent_of[2 * (size_t)kr.rank + kr.mate];
	if (ei >= 0 && !w.ents[(size_t)ei].c.empty()) w.ents[(size_t)ei].c.pop_back();
}

int32_t find_best_record(Work &w, Entry &e)      // src/samdict.c:177-243 (records the density optimiser switched off are passed over)
{
	size_t best = 0;
	double best_gamma = -1.0;
	const size_t n = e.c.size();
	for (size_t i = 0; i < n; i++)
		if (w.recs[(size_t)e.c[i].rec].active && e.c[i].gamma > best_gamma) { best = i; best_gamma = e.c[i].gamma; }
	Rec &chosen = w.recs[(size_t)e.c[best].rec];
	chosen.alt = -1;
	chosen.gamma = best_gamma;
	chosen.cloud = e.c[best].cloud;
	if (best_gamma <= kSecondaryAlignThresh) {
		size_t second = 0;
		double second_gamma = -1.0;
		for (size_t i = 0; i < n; i++)
			if (w.recs[(size_t)e.c[i].rec].active && i != best && e.c[i].gamma > second_gamma) { second = i; second_gamma = e.c[i].gamma; }
		if (second_gamma > 0) chosen.alt = e.c[second].rec;
	}
	return e.c[best].rec;
}

void normalize_cloud_probabilities(std::vector<Cloud> &cl, size_t nc)      // src/align.c:124-143
{
	for (size_t i = 0; i < nc; i++) {
		if (cl[i].parent >= 0) continue;
		double total = 0.0;
		for (int32_t c = (int32_t)i; c >= 0; c = cl[(size_t)c].child) total += cl[(size_t)c].weight;
		for (int32_t c = (int32_t)i; c >= 0; c = cl[(size_t)c].child) cl[(size_t)c].weight /= total;
	}
}

// ---- `ema align -d`: mark_optimal_alignments_in_cloud (src/split.c:38-338) on the name-sorted records of a bad cloud.
// Same decisions, same order, same draws from libc's rand() as the reference, same floating-point expressions in the same order
// (-ffp-contract=off); the arrays are sized by need instead of 50,000-entry stack buffers.
std::once_flag g_rand_once;
std::atomic<bool> g_rand_seeded{false};

double log_density_prob(const Shared &S, unsigned int density)      // src/split.c:15-35
{
	const size_t size = (size_t)S.o.n_density_probs;
	if (density < size) return std::log(S.o.density_probs[density]);
	return std::log(S.o.density_probs[size - 1]) - (density - size + 1) * std::log(2.0);      // (unsigned arithmetic, as there)
}

bool split_is_pair(const Rec *r1, const Rec *r2)      // is_pair, src/align.c:27-41: two uint32_t positions, the difference wraps
{
	if (r1->rev == r2->rev || r1->chrom != r2->chrom) return false;
	if (r2->rev) { const Rec *t = r2; r2 = r1; r1 = t; }
	const int64_t d = (int64_t)(uint32_t)(r1->pos - r2->pos);
	return kInsertMin <= d && d <= kInsertMax;
}

void mark_optimal_alignments_in_cloud(Work &w, const Shared &S, const std::vector<int32_t> &sorted)
{
	const int kMaxNoMove = 500, kBinSize = 1000, kMaxBins = 1000000 / 1000, kScoreScale = 20, kSplitExtraDepth = 5, kIters = 50000;
	const size_t kBuf = 50000;
	if (!S.rng_private && !g_rand_seeded.load()) std::call_once(g_rand_once, [] { if (!g_rand_seeded.load()) { srand((unsigned)time(nullptr)); g_rand_seeded.store(true); } });
	size_t n_records = sorted.size();
	if (n_records >= kBuf || n_records <= 5) return;
	auto R = [&](int32_t i) -> Rec & { return w.recs[(size_t)i]; };
	auto rec_eq = [&](int32_t a, int32_t b) { return R(a).rank == R(b).rank && R(a).mate == R(b).mate; };
	auto rec_eq_mate = [&](int32_t a, int32_t b) { return R(a).rank == R(b).rank && R(a).mate != R(b).mate; };
	// records too far from their read's lowest edit distance go
	std::vector<int32_t> records;
	records.reserve(n_records);
	for (size_t i = 0; i < n_records;) {
		size_t j = i + 1;
		while (j < n_records && rec_eq(sorted[j], sorted[i])) ++j;
		const size_t n = j - i;
		if (n > 1) {
			size_t min_edit = 0;
			for (size_t k = 0; k < n; k++) if (R(sorted[i + k]).clip_edit_dist < R(sorted[i + min_edit]).clip_edit_dist) min_edit = k;
			const int cutoff = R(sorted[i + min_edit]).clip_edit_dist + kSplitExtraDepth;
			for (size_t k = 0; k < n; k++) {
				if (R(sorted[i + k]).clip_edit_dist <= cutoff) records.push_back(sorted[i + k]);
				else R(sorted[i + k]).active = 0;
			}
		} else records.push_back(sorted[i]);
		i = j;
	}
	n_records = records.size();
	struct MMap { size_t idx; int n, mate_umap, mate_mmap, active; };
	std::vector<size_t> umaps;
	std::vector<MMap> mmaps;
	double log_config_prob = 0;
	uint32_t cloud_lo = 0xffffffffu, cloud_hi = 0;
	auto bounds = [&](const Rec &r) { if (r.pos < cloud_lo) cloud_lo = r.pos; if (r.pos > cloud_hi) cloud_hi = r.pos; };
	for (size_t i = 0; i < n_records;) {
		bounds(R(records[i]));
		size_t j = i + 1;
		while (j < n_records && rec_eq(records[j], records[i])) { bounds(R(records[j])); ++j; }
		const size_t n = j - i;
		if (n > 1) {
			size_t max_score = 0;
			for (size_t k = 0; k < n; k++) if (R(records[i + k]).score > R(records[i + max_score]).score) max_score = k;
			int mate_umap = -1, mate_mmap = -1;
			for (size_t k = 0; k < umaps.size(); k++) if (rec_eq_mate(records[i], records[umaps[k]])) { mate_umap = (int)k; break; }
			if (mate_umap < 0)
				for (size_t k = 0; k < mmaps.size(); k++)
					if (rec_eq_mate(records[i], records[mmaps[k].idx])) { mate_mmap = (int)k; mmaps[k].mate_mmap = (int)mmaps.size(); break; }
			mmaps.push_back(MMap{i, (int)n, mate_umap, mate_mmap, (int)max_score});
			log_config_prob += R(records[i + max_score]).score / kScoreScale;
		} else {
			for (size_t k = 0; k < mmaps.size(); k++)
				if (rec_eq_mate(records[i], records[mmaps[k].idx])) { mmaps[k].mate_umap = (int)umaps.size(); break; }
			umaps.push_back(i);
			log_config_prob += R(records[i]).score / kScoreScale;
		}
		i = j;
	}
	const size_t n_umaps = umaps.size(), n_mmaps = mmaps.size();
	const size_t n_bins = (size_t)(cloud_hi - cloud_lo) / (size_t)kBinSize + 1;
	if (n_bins >= (size_t)kMaxBins || n_records <= 5 || n_mmaps == 0) return;
	std::vector<unsigned short> bins((size_t)kMaxBins, 0);
	auto bin_of = [&](uint32_t pos) { return (size_t)((pos - cloud_lo) / (uint32_t)kBinSize); };
	for (size_t i = 0; i < n_records; i++) R(records[i]).active = 0;      // the active ones are set again below
	for (size_t i = 0; i < n_umaps; i++) ++bins[bin_of(R(records[umaps[i]]).pos)];
	for (size_t i = 0; i < n_mmaps; i++) ++bins[bin_of(R(records[mmaps[i].idx + (size_t)mmaps[i].active]).pos)];
	for (size_t i = 0; i < n_bins; i++) log_config_prob += log_density_prob(S, bins[i]);
	int no_move_count = 0;
	for (size_t k = 0; k < (size_t)kIters; k++) {
		const double t = std::pow(10.0, 0.0 - ((0.0 - (-12.0)) * k) / kIters);
		const size_t r = (size_t)S.draw() % n_mmaps;
		const size_t r_old = (size_t)mmaps[r].active;
		size_t r_new = (size_t)(S.draw() % (mmaps[r].n - 1));
		if (r_new >= r_old) ++r_new;
		const Rec *active_mate = nullptr;
		size_t mate_r = 0;
		int mate_is_mmap = 0;
		if (mmaps[r].mate_umap >= 0) { mate_r = (size_t)mmaps[r].mate_umap; active_mate = &R(records[umaps[mate_r]]); }
		else if (mmaps[r].mate_mmap >= 0) { mate_r = (size_t)mmaps[r].mate_mmap; active_mate = &R(records[mmaps[mate_r].idx + (size_t)mmaps[mate_r].active]); mate_is_mmap = 1; }
		const Rec *rec_old = &R(records[mmaps[r].idx + r_old]), *rec_new = &R(records[mmaps[r].idx + r_new]);
		double density_prob_change = 0.0, score_prob_change = 0.0;
		int force_move = 0, mate_new_active = -1;
		size_t mate_old_bin = 0, mate_new_bin = 0;
		const int old_paired = active_mate != nullptr && split_is_pair(rec_old, active_mate);
		const int new_paired = active_mate != nullptr && split_is_pair(rec_new, active_mate);
		if (!old_paired && new_paired) force_move = 1;
		else if (old_paired && !new_paired && mate_is_mmap) {      // try to move the mate along
			for (int i = 0; i < mmaps[mate_r].n; i++) {
				const Rec *mate_rec_new = &R(records[mmaps[mate_r].idx + (size_t)i]);
				if (split_is_pair(rec_new, mate_rec_new)) {
					const Rec *mate_rec_old = active_mate;
					mate_new_active = i;
					mate_old_bin = bin_of(mate_rec_old->pos);
					mate_new_bin = bin_of(mate_rec_new->pos);
					score_prob_change += (mate_rec_new->score - mate_rec_old->score) / kScoreScale;
					break;
				}
			}
		}
		const size_t old_bin = bin_of(rec_old->pos), new_bin = bin_of(rec_new->pos);
		const int p1 = (mate_new_active >= 0 && old_bin == mate_old_bin) ? 2 : 1;
		const int p2 = (mate_new_active >= 0 && new_bin == mate_new_bin) ? 2 : 1;
		{
			const double old_bin_prob_old = log_density_prob(S, bins[old_bin]);
			const double old_bin_prob_new = log_density_prob(S, (unsigned int)(bins[old_bin] - p1));
			const double new_bin_prob_old = log_density_prob(S, bins[new_bin]);
			const double new_bin_prob_new = log_density_prob(S, (unsigned int)(bins[new_bin] + p2));
			density_prob_change += (old_bin_prob_new - old_bin_prob_old) + (new_bin_prob_new - new_bin_prob_old);
		}
		if (p1 == 1 && mate_new_active >= 0) {
			const double a = log_density_prob(S, bins[mate_old_bin]), b = log_density_prob(S, (unsigned int)(bins[mate_old_bin] - 1));
			density_prob_change += (b - a);
		}
		if (p2 == 1 && mate_new_active >= 0) {
			const double a = log_density_prob(S, bins[mate_new_bin]), b = log_density_prob(S, (unsigned int)(bins[mate_new_bin] + 1));
			density_prob_change += (b - a);
		}
		score_prob_change += (rec_new->score - rec_old->score) / kScoreScale;
		const double prob_change = density_prob_change + score_prob_change;
		if (force_move || prob_change > 0 || std::exp(prob_change / t) >= ((double)S.draw()) / RAND_MAX) {
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
		if (no_move_count >= kMaxNoMove) break;
	}
	(void)log_config_prob;
	for (size_t i = 0; i < n_umaps; i++) R(records[umaps[i]]).active = 1;
	for (size_t i = 0; i < n_mmaps; i++) R(records[mmaps[i].idx + (size_t)mmaps[i].active]).active = 1;
}

// one barcode group: pairs [p0, p1), records [r0, r1) of ema_aln_out.  may_draw (-d only): this call may reach the density optimiser,
// i.e. libc's rand(); a call that may not returns false at the group's first bad cloud, having written nothing (the caller runs
// such groups afterwards, in group order, on one thread: groups without a bad cloud draw nothing, so the draws fall exactly as in
// a one-thread run over all groups -- the reference's `-t 1`)
bool do_group(Work &w, Shared &S, size_t g, size_t p0, size_t p1, uint64_t r0, uint64_t r1, bool may_draw)
{
	const ema_bucket *bk = S.bk;
	const size_t n = (size_t)(r1 - r0), n_pairs = p1 - p0;
	S.n_sel[g] = 0; S.n_clouds[g] = 0; S.n_bad[g] = 0;
	if (n == 0) return true;
	// read names of the group (the identifier field without its first character, src/align.c:925-929), ranked in strcmp order
	w.names.resize(n_pairs);
	for (size_t p = p0; p < p1; ++p) {
		const uint32_t b = bk->id_off[p], e = bk->id_off[p + 1];
		w.names[p - p0] = Name{bk->ids + b + (e > b ? 1 : 0), e > b ? e - b - 1 : 0, (uint32_t)(p - p0)};
	}
	std::sort(w.names.begin(), w.names.end(), [](const Name &x, const Name &y) { const int c = name_cmp(x, y); return c ? c < 0 : x.pair < y.pair; });
	w.rank_of_pair.resize(n_pairs);
	int32_t n_rank = 0;
	for (size_t i = 0; i < n_pairs; ++i) {
		if (i > 0 && name_cmp(w.names[i - 1], w.names[i]) != 0) ++n_rank;
		w.rank_of_pair[w.names[i].pair] = n_rank;
	}
	++n_rank;
	w.ent_of.assign(2 * (size_t)n_rank, -1);
	w.n_ents = 0;
	// the group's records in append_alignments' order
	w.recs.resize(n);
	for (size_t i = 0; i < n; ++i) {
		const ema_aln_rec &ar = S.a->rec[r0 + i];
		const ema_cand_t &c = S.b->cand[ar.cand];
		Rec &r = w.recs[i];
		r.chrom = (uint32_t)c.rid; r.pos = (uint32_t)(c.pos + 1);
		r.rank = w.rank_of_pair[ar.pair - p0];
		r.score = ar.score; r.gamma = 0;
		r.mate = ar.mate; r.rev = (uint8_t)(c.is_rev != 0); r.duplicate = 0; r.visited = 0;
		r.active = 1; r.clip_edit_dist = ar.clip_edit_dist;
		r.cloud = -1; r.alt = -1; r.sel_mate = -1;
		r.gi = r0 + i;
	}
	// qsort(records, record_cmp): (barcode,) chromosome narrowed to 8 bits, position, read name (src/samrecord.c:51-73)
	w.ord.resize(n + 1);
	if (n_rank < (1 << 24)) {      // the three keys in one word: chromosome (8 bits), position (32), read-name rank (24); stable on ties, as above
		w.ord_keys.resize(n);
		for (size_t i = 0; i < n; ++i) {
			const Rec &r = w.recs[i];
			w.ord_keys[i] = Work::OrdKey{(uint64_t)(uint8_t)r.chrom << 56 | (uint64_t)r.pos << 24 | (uint64_t)(uint32_t)r.rank, (int32_t)i};
		}
		std::stable_sort(w.ord_keys.begin(), w.ord_keys.end(), [](const Work::OrdKey &a, const Work::OrdKey &b) { return a.key < b.key; });
		for (size_t i = 0; i < n; ++i) w.ord[i] = w.ord_keys[i].idx;
	} else {
		for (size_t i = 0; i < n; ++i) w.ord[i] = (int32_t)i;
		std::stable_sort(w.ord.begin(), w.ord.begin() + (long)n, [&](int32_t x, int32_t y) {
			const Rec &a = w.recs[(size_t)x], &b = w.recs[(size_t)y];
			const uint8_t ca = (uint8_t)a.chrom, cb = (uint8_t)b.chrom;
			if (ca != cb) return ca < cb;
			if (a.pos != b.pos) return a.pos < b.pos;
			return a.rank < b.rank;
		});
	}
	// clouds (src/align.c:358-408)
	w.clouds.clear();
	size_t at = 0;
	while (at < n) {
		size_t r = at;
		const int32_t ci = (int32_t)w.clouds.size();
		w.clouds.push_back(Cloud{0.0, 0.0, -1, -1, 0});
		sam_dict_add(w, S, w.ord[r], ci, false);
		size_t cov = 1;
		bool collision = false;
		while (r + 1 < n && w.recs[(size_t)w.ord[r + 1]].chrom == w.recs[(size_t)w.ord[r]].chrom &&
		       (uint32_t)(w.recs[(size_t)w.ord[r + 1]].pos - w.recs[(size_t)w.ord[r]].pos) <= S.o.dist_thresh) {
			++r;
			if (!collision && sam_dict_add(w, S, w.ord[r], ci, false)) {
				collision = true;
				for (size_t i = 0; i < cov; i++) sam_dict_del(w, w.ord[at + i]);
			}
			++cov;
		}
		if (collision) {      // two candidates of one read in this cloud: re-enter it by read name, every record forced in
			if (S.o.density_opt && !may_draw) { S.n_bad[g] = 0; return false; }
			w.clouds[(size_t)ci].bad = 1;
			++S.n_bad[g];
			w.split.assign(w.ord.begin() + (long)at, w.ord.begin() + (long)(at + cov));
			std::stable_sort(w.split.begin(), w.split.end(), [&](int32_t x, int32_t y) {      // name_cmp, src/align.c:70-82
				const Rec &a = w.recs[(size_t)x], &b = w.recs[(size_t)y];
				if (a.rank != b.rank) return a.rank < b.rank;
				return a.mate < b.mate;
			});
			if (S.o.density_opt) mark_optimal_alignments_in_cloud(w, S, w.split);      // -d, src/align.c:396-397
			for (size_t i = 0; i < cov; i++) sam_dict_add(w, S, w.split[i], ci, true);
		}
		at = r + 1;
	}
	const size_t nc = w.clouds.size();
	S.n_clouds[g] = (uint32_t)nc;
	std::vector<Cloud> &cl = w.clouds;
	// initialisation (src/align.c:411-430); entries are visited newest first, as the reference's list is
	for (size_t k = w.n_ents; k-- > 0;) {
		Entry &e = w.ents[k];
		normalize_log_probs(e.c);
		for (const Cand &c : e.c) cl[(size_t)c.cloud].exp_cov += c.gamma;
	}
	for (size_t i = 0; i < nc; i++) cl[i].weight = cl[i].exp_cov;
	if (!S.o.many_clouds) normalize_cloud_probabilities(cl, nc);
	// EM (src/align.c:432-525)
	const bool full_em = n_pairs >= 30;
	std::vector<double> cw;
	// (two savings that change no bit: a read with ONE candidate has gamma 1 whatever its score, cloud weight and mate say -- normalize_log_probs
	// of one value -- so the round's arithmetic is skipped for it; and a cloud's log(weight), constant within a round, is taken once)
	std::vector<double> &lw = w.log_weight;
	std::vector<uint8_t> &have_lw = w.have_log_weight;
	lw.resize(nc); have_lw.resize(nc);
	for (int q = 0; q < kEmIters && full_em; q++) {
		for (size_t i = 0; i < nc; i++) cl[i].exp_cov = 0.0;
		std::fill(have_lw.begin(), have_lw.end(), (uint8_t)0);
		for (size_t k = w.n_ents; k-- > 0;) {
			Entry &e = w.ents[k];
			const Entry *m = e.mate >= 0 ? &w.ents[(size_t)e.mate] : nullptr;
			const size_t num = e.c.size();
			if (num == 1) { e.c[0].gamma = 1.0; continue; }
			if (S.o.many_clouds) {      // with many clouds the weights are normalised per read
				cw.resize(num);
				double tot = 0;
				for (size_t i = 0; i < num; i++) { cw[i] = cl[(size_t)e.c[i].cloud].weight; tot += cw[i]; }
				for (size_t i = 0; i < num; i++) cw[i] /= tot;
			}
			for (size_t i = 0; i < num; i++) {
				const Rec &ri = w.recs[(size_t)e.c[i].rec];
				double best_mate_score = kUnpairedPenalty;
				if (m) {
					for (const Cand &mc : m->c) {
						const Rec &rj = w.recs[(size_t)mc.rec];
						if (rj.chrom == ri.chrom && rj.rev != ri.rev && mc.cloud == e.c[i].cloud && mc.gamma != 0.0) {
							const double penalty = ri.rev ? mate_dist_penalty(ri.pos, rj.pos) : mate_dist_penalty(rj.pos, ri.pos);
							const double mate_score = penalty + (mc.gamma == 1.0 ? 0.0 : std::log(mc.gamma));      // (log(1.0) is +0.0: the mate with one candidate)
							if (mate_score > best_mate_score) best_mate_score = mate_score;
						}
					}
				}
				double log_w;
				if (S.o.many_clouds) log_w = std::log(cw[i]);
				else {
					const size_t ci = (size_t)e.c[i].cloud;
					if (!have_lw[ci]) { lw[ci] = std::log(cl[ci].weight); have_lw[ci] = 1; }
					log_w = lw[ci];
				}
				e.c[i].gamma = ri.score + log_w + best_mate_score;
			}
			normalize_log_probs(e.c);
		}
		for (size_t k = w.n_ents; k-- > 0;)
			for (const Cand &c : w.ents[k].c) if (w.recs[(size_t)c.rec].active) cl[(size_t)c.cloud].exp_cov += c.gamma;      // src/align.c:530 (none is a duplicate yet)
		for (size_t i = 0; i < nc; i++) cl[i].weight = cl[i].exp_cov;
		if (!S.o.many_clouds) normalize_cloud_probabilities(cl, nc);
	}
	// best alignments (src/align.c:527-558)
	w.final_.clear();
	for (size_t k = w.n_ents; k-- > 0;) {
		Entry &e = w.ents[k];
		if (e.visited) continue;
		const int32_t best = find_best_record(w, e);
		int32_t best_mate = -1;
		if (e.mate >= 0) best_mate = find_best_record(w, w.ents[(size_t)e.mate]);
		w.final_.push_back(best);
		w.recs[(size_t)best].sel_mate = best_mate;
		if (best_mate >= 0) { w.final_.push_back(best_mate); w.recs[(size_t)best_mate].sel_mate = best; }
		e.visited = true;
		if (e.mate >= 0) { w.ents[(size_t)e.mate].visited = true; w.ents[(size_t)e.mate].mate = -1; }
	}
	// duplicates (src/align.c:560-573): Lariat's definition, dup_cmp (src/align.c:85-122)
	if (!S.o.many_clouds) {
		auto key = [&](int32_t x, uint32_t k[6]) {
			const Rec &r = w.recs[(size_t)x];
			k[0] = r.mate; k[1] = r.rev; k[2] = r.chrom; k[3] = r.pos;
			k[4] = r.sel_mate >= 0 ? w.recs[(size_t)r.sel_mate].chrom : 0xffffffffu;
			k[5] = r.sel_mate >= 0 ? w.recs[(size_t)r.sel_mate].pos : 0xffffffffu;
		};
		auto cmp = [](const Work::DupKey &a, const Work::DupKey &b) {
			for (int i = 0; i < 6; ++i) if (a.k[i] != b.k[i]) return a.k[i] < b.k[i] ? -1 : 1;
			return 0;
		};
		const size_t nf = w.final_.size();
		w.dup_keys.resize(nf);
		for (size_t i = 0; i < nf; ++i) { key(w.final_[i], w.dup_keys[i].k); w.dup_keys[i].idx = w.final_[i]; }
		std::stable_sort(w.dup_keys.begin(), w.dup_keys.end(), [&](const Work::DupKey &a, const Work::DupKey &b) { return cmp(a, b) < 0; });
		for (size_t i = 0; i < nf; ++i) w.final_[i] = w.dup_keys[i].idx;
		for (size_t i = 0; i < nf;) {
			size_t j = i + 1;
			while (j < nf && cmp(w.dup_keys[i], w.dup_keys[j]) == 0) { w.recs[(size_t)w.final_[j]].duplicate = 1; j++; }
			i = j;
		}
	}
	// print order (src/align.c:587-603)
	uint32_t n_out = 0;
	for (int32_t x : w.final_) {
		Rec &best = w.recs[(size_t)x];
		if (best.visited) continue;
		if (best.sel_mate >= 0) w.recs[(size_t)best.sel_mate].visited = 1;
		S.sel[r0 + n_out] = Sel{best.gi, best.sel_mate >= 0 ? w.recs[(size_t)best.sel_mate].gi : ~(uint64_t)0};
		++n_out;
		for (int k = 0; k < 2; ++k) {
			const int32_t y = k == 0 ? x : best.sel_mate;
			if (y < 0) continue;
			const Rec &r = w.recs[(size_t)y];
			S.gamma[r.gi] = r.gamma;
			S.cloud[r.gi] = r.cloud;
			S.alt_of[r.gi] = r.alt >= 0 ? (int32_t)(w.recs[(size_t)r.alt].gi - r0) : -1;
			S.flags[r.gi] = (uint8_t)(r.duplicate | (cl[(size_t)r.cloud].bad << 1));
		}
	}
	S.n_sel[g] = n_out;
	return true;
}

int n_host_threads(int want)
{
	return want > 0 ? want : EmaPool::get().size();      // (pieces of a pass: the pool runs as many at a time as it has threads)
}

}  // namespace

extern "C" {

void ema_cloud_opts_default(ema_cloud_opts *o)
{
	if (!o) return;
	o->dist_thresh = 50000; o->many_clouds = 0; o->n_threads = 0; o->first_cloud_id = 0;
	o->density_opt = 0; o->n_density_probs = 4; o->emit = 0; o->seed_private = 0; o->seed = 0; o->pad_ = 0;
	for (double &p : o->density_probs) p = 0;
	o->density_probs[0] = 0.6; o->density_probs[1] = 0.05; o->density_probs[2] = 0.2; o->density_probs[3] = 0.01;      // src/techs.c: every platform but cpt
}

void ema_clouds_reseed(unsigned seed)
{
	srand(seed);
	g_rand_seeded.store(true);
}

void ema_clouds_free(ema_clouds_out *out)
{
	if (!out) return;
	free(out->lines); free(out->recs); free(out->alts); free(out->idents);
	free(out->descs); free(out->xas); free(out->sel_at);
	free(out);
}

int ema_clouds_select(const ema_bucket *bk, const ema_batch_out *b, const ema_aln_out *a, const char *const *contig_names,
                      int32_t n_contigs, const ema_cloud_opts *opts, ema_clouds_out **out)
{
	if (!out) return EMA_EARG;
	EMA_CPU(EMA_CPU_CLOUDS);
	*out = nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	if (!bk || !b || !a || (n_contigs > 0 && !contig_names)) return EMA_EARG;
	if (bk->n_pairs != b->n_pairs || bk->n_pairs != a->n_pairs) return EMA_EARG;
	Shared S;
	S.bk = bk; S.b = b; S.a = a;
	if (opts) S.o = *opts; else ema_cloud_opts_default(&S.o);
	if (S.o.density_opt && S.o.seed_private) {
		memset(&S.rng, 0, sizeof(S.rng));
		if (initstate_r(S.o.seed, S.rng_state, sizeof(S.rng_state), &S.rng) != 0) return EMA_EARG;
		S.rng_private = true;
	}
	const size_t n_rec = a->n, n_groups = bk->n_groups;
	for (size_t i = 0; i < n_rec; ++i) {
		const ema_cand_t &c = b->cand[a->rec[i].cand];
		if (c.rid < 0 || c.rid >= n_contigs) return EMA_EARG;
	}
	S.gamma.assign(n_rec, 0.0); S.cloud.assign(n_rec, -1); S.alt_of.assign(n_rec, -1); S.flags.assign(n_rec, 0);
	S.sel.resize(n_rec + 1);
	S.n_sel.assign(n_groups + 1, 0); S.n_clouds.assign(n_groups + 1, 0); S.n_bad.assign(n_groups + 1, 0);
	{   // barcode groups over the host's threads, a few at a time from a shared counter
		std::atomic<size_t> next{0};
		std::vector<uint8_t> deferred(S.o.density_opt ? n_groups : 0, 0);
		auto run = [&] {
			EMA_CPU(EMA_CPU_CLOUDS);
			Work w;
			for (;;) {
				const size_t g0 = next.fetch_add(8);
				if (g0 >= n_groups) break;
				for (size_t g = g0; g < g0 + 8 && g < n_groups; ++g) {
					const size_t p0 = bk->group_off[g], p1 = bk->group_off[g + 1];
					if (!do_group(w, S, g, p0, p1, a->pair_off[p0], a->pair_off[p1], !S.o.density_opt)) deferred[g] = 1;
				}
			}
		};
		const int nt = (int)std::min<size_t>((size_t)n_host_threads(S.o.n_threads), n_groups / 16 + 1);
		EmaPool::get().run((size_t)nt, [&](size_t) { run(); });
		// -d draws from libc's rand(): the groups that reach the optimiser (the ones with a bad cloud: few) run here, on one thread, in
		// order, so that the draws fall as in a `-t 1` run of the reference.  (Until r05 every group of a -d run went this way.)
		if (S.o.density_opt) {
			Work w;
			for (size_t g = 0; g < n_groups; ++g) if (deferred[g]) {
				const size_t p0 = bk->group_off[g], p1 = bk->group_off[g + 1];
				(void)do_group(w, S, g, p0, p1, a->pair_off[p0], a->pair_off[p1], true);
			}
		}
	}
	// assembly: cloud numbers of a single-threaded run, the selected records as formatter input, statistics -- on the host's threads
	// as well (round 2 did this part on one: more than half of the stage's wall time, r03): the offsets of every group's records
	// and lines follow from the per-group counts, so groups are laid out independently
	ema_clouds_out *o = (ema_clouds_out *)calloc(1, sizeof(ema_clouds_out));
	if (!o) return EMA_EARG;
	const bool want_lines = S.o.emit != 1, want_descs = S.o.emit >= 1;
	std::vector<int32_t> cloud_base(n_groups + 1);
	std::vector<size_t> rec_base(n_groups + 1), line_base(n_groups + 1), xa_base(n_groups + 1);
	int32_t next_id = S.o.first_cloud_id;
	size_t n_sel = 0, n_recs = 0, n_xas = 0;
	for (size_t g = 0; g < n_groups; ++g) {
		cloud_base[g] = next_id;
		next_id += (int32_t)S.n_clouds[g];
		rec_base[g] = n_recs; line_base[g] = 2 * n_sel; xa_base[g] = n_xas;
		const uint64_t r0 = a->pair_off[bk->group_off[g]];
		for (uint32_t k = 0; k < S.n_sel[g]; ++k) {
			const Sel &sl = S.sel[r0 + k];
			n_recs += sl.mate != ~(uint64_t)0 ? 2 : 1;
			if (want_descs) n_xas += (size_t)(S.alt_of[sl.rec] >= 0) + (size_t)(sl.mate != ~(uint64_t)0 && S.alt_of[sl.mate] >= 0);
		}
		n_sel += S.n_sel[g];
		if (S.n_clouds[g]) ++o->stats.groups;
		o->stats.clouds += S.n_clouds[g]; o->stats.bad_clouds += S.n_bad[g];
	}
	rec_base[n_groups] = n_recs; line_base[n_groups] = 2 * n_sel; xa_base[n_groups] = n_xas;
	o->next_cloud_id = next_id;
	std::vector<size_t> ident_at;
	if (want_lines) {
		o->n_lines = 2 * n_sel;
		o->n_recs = n_recs;
		o->lines = (ema_sam_line *)malloc((2 * n_sel + 1) * sizeof(ema_sam_line));
		o->recs = (ema_sam_rec *)malloc((2 * n_sel + 1) * sizeof(ema_sam_rec));
		o->alts = (ema_sam_alt *)malloc((2 * n_sel + 1) * sizeof(ema_sam_alt));
		ident_at.resize(bk->n_pairs + 1);
		size_t id_bytes = 0;
		for (size_t p = 0; p < bk->n_pairs; ++p) {      // names without the first character ('@'), NUL-terminated
			const uint32_t ib = bk->id_off[p], ie = bk->id_off[p + 1];
			ident_at[p] = id_bytes;
			id_bytes += (size_t)(ie > ib ? ie - ib - 1 : 0) + 1;
		}
		o->idents = (char *)malloc(id_bytes + 1);
		if (!o->lines || !o->recs || !o->alts || !o->idents) { ema_clouds_free(o); return EMA_EARG; }
	}
	if (want_descs) {      // the compact form (include/ema_sam.h): indices into the bucket and the batch instead of pointers
		o->n_descs = n_recs; o->n_xas = n_xas; o->n_sel = n_sel;
		o->descs = (ema_sam_desc *)malloc((n_recs + 1) * sizeof(ema_sam_desc));
		o->xas = (ema_sam_xa *)malloc((n_xas + 1) * sizeof(ema_sam_xa));
		o->sel_at = (uint32_t *)malloc((n_sel + 1) * sizeof(uint32_t));
		if (!o->descs || !o->xas || !o->sel_at || n_recs >= 0xffffffffu) { ema_clouds_free(o); return EMA_EARG; }
	}
	std::atomic<int> rc_all{EMA_OK};
	const int nt_asm = (int)std::min<size_t>((size_t)n_host_threads(S.o.n_threads), n_groups / 64 + 1);
	std::vector<ema_sam_stats> part((size_t)nt_asm);
	std::vector<uint64_t> cig_lo((size_t)nt_asm, ~(uint64_t)0), cig_hi((size_t)nt_asm, 0);
	for (auto &ps : part) memset(&ps, 0, sizeof(ps));
	// one selected record as the statistics see it (what print_sam_record flags and prints of it, src/samrecord.c:104-175)
	struct Brief { uint32_t rid, pos; int mapq; bool rev, dup, xa; };
	auto assemble = [&](int tid) {
		EMA_CPU(EMA_CPU_CLOUDS);
		// this thread's stretch of pairs (names) and of groups (records, lines, statistics)
		if (want_lines) {
			const size_t pa = bk->n_pairs * (size_t)tid / (size_t)nt_asm, pb = bk->n_pairs * (size_t)(tid + 1) / (size_t)nt_asm;
			for (size_t p = pa; p < pb; ++p) {
				const uint32_t ib = bk->id_off[p], ie = bk->id_off[p + 1];
				const uint32_t len = ie > ib ? ie - ib - 1 : 0;
				memcpy(o->idents + ident_at[p], bk->ids + ib + (ie > ib ? 1 : 0), len);
				o->idents[ident_at[p] + len] = '\0';
			}
		}
		ema_sam_stats st;      // (this thread's own: the threads' slots of part[] / cig_lo[] share cache lines)
		memset(&st, 0, sizeof(st));
		uint64_t clo = ~(uint64_t)0, chi = 0;
		const size_t ga = n_groups * (size_t)tid / (size_t)nt_asm, gb = n_groups * (size_t)(tid + 1) / (size_t)nt_asm;
		for (size_t g = ga; g < gb; ++g) {
			const uint64_t r0 = a->pair_off[bk->group_off[g]];
			size_t at = rec_base[g], ln = line_base[g], xat = xa_base[g];
			auto fill = [&](uint64_t gi, bool has_mate, Brief &bf) -> ema_sam_rec * {
				const ema_aln_rec &ar = a->rec[gi];
				const ema_cand_t &c = b->cand[ar.cand];
				const size_t p = ar.pair;
				const double gamma = S.gamma[gi];
				const int gamma_mapq = gamma <= 0.999999 ? (int)(-10 * std::log10(1 - gamma)) : 60;
				int q = gamma_mapq < ar.score_mapq ? gamma_mapq : ar.score_mapq;
				q = q < ar.mapq ? q : ar.mapq;
				q = q > 0 ? q : 0; q = q < 60 ? q : 60;
				bf.rid = (uint32_t)c.rid; bf.pos = (uint32_t)(c.pos + 1); bf.mapq = q; bf.rev = c.is_rev != 0; bf.dup = S.flags[gi] & 1; bf.xa = S.alt_of[gi] >= 0;
				const ema_cand_t *x = nullptr;
				if (S.alt_of[gi] >= 0) {
					x = &b->cand[a->rec[r0 + (uint64_t)S.alt_of[gi]].cand];
					if (x->n_cigar >= 64) rc_all.store(EMA_EFORMAT);      // the reference asserts (struct xa holds 64 operations)
				}
				if (want_descs) {
					ema_sam_desc &d = o->descs[at];
					memset(&d, 0, sizeof(d));
					if (c.pos < 0 || c.pos >= 0xffffffffll || c.n_cigar < 0) rc_all.store(EMA_EARG);
					d.pair = (uint32_t)p; d.rid = c.rid; d.pos = (uint32_t)(c.pos + 1); d.cigar_off = c.cigar_off; d.n_cigar = c.n_cigar; d.edit_dist = c.NM;
					d.cloud_id = cloud_base[g] + S.cloud[gi]; d.xa = -1;
					d.mate = ar.mate; d.rev = (uint8_t)(c.is_rev != 0); d.duplicate = S.flags[gi] & 1; d.cloud_bad = (S.flags[gi] >> 1) & 1;
					d.mapq = (uint8_t)q; d.has_mate = has_mate;
					if (gamma == 1.0) { d.gamma[0] = '1'; d.gamma_len = 1; }      // the two values most records carry, without the library call
					else if (gamma == 0.0) { d.gamma[0] = '0'; d.gamma_len = 1; }
					else if (const int k5 = ema_fmt_g5(gamma, d.gamma)) d.gamma_len = (uint8_t)k5;      // host_fmt.h: the same text, exactly
					else {
						char t[48];
						const int k = snprintf(t, sizeof t, "%.5g", gamma);
						if (k < 0 || k > (int)sizeof d.gamma) rc_all.store(EMA_EARG);      // (a gamma is in [0, 1]: at most 11 characters)
						else { memcpy(d.gamma, t, (size_t)k); d.gamma_len = (uint8_t)k; }
					}
					if (c.n_cigar > 0) { clo = std::min<uint64_t>(clo, c.cigar_off); chi = std::max<uint64_t>(chi, (uint64_t)c.cigar_off + (uint64_t)c.n_cigar); }
					if (x) {
						ema_sam_xa &xa = o->xas[xat];
						if (x->pos < 0 || x->pos >= 0xffffffffll || x->n_cigar < 0) rc_all.store(EMA_EARG);
						xa.rid = x->rid; xa.pos = (uint32_t)(x->pos + 1); xa.cigar_off = x->cigar_off; xa.n_cigar = x->n_cigar; xa.edit_dist = x->NM; xa.rev = x->is_rev != 0;
						if (x->n_cigar > 0) { clo = std::min<uint64_t>(clo, x->cigar_off); chi = std::max<uint64_t>(chi, (uint64_t)x->cigar_off + (uint64_t)x->n_cigar); }
						d.xa = (int32_t)xat++;
					}
				}
				ema_sam_rec *out_rec = nullptr;
				if (want_lines) {
					ema_sam_rec &r = o->recs[at];
					memset(&r, 0, sizeof(r));
					r.ident = o->idents + ident_at[p];
					r.chrom = contig_names[c.rid]; r.chrom_id = (uint32_t)c.rid; r.pos = (uint32_t)(c.pos + 1);
					r.mapq = ar.mapq; r.score_mapq = ar.score_mapq; r.gamma = gamma;
					r.mate = ar.mate; r.rev = (uint8_t)(c.is_rev != 0); r.duplicate = S.flags[gi] & 1;
					r.cloud_id = cloud_base[g] + S.cloud[gi]; r.cloud_bad = (S.flags[gi] >> 1) & 1;
					r.bc = bk->bc[p];
					const size_t rd = 2 * p + ar.mate, md = 2 * p + (1 - ar.mate);
					r.read = bk->bases + bk->off[rd]; r.qual = bk->quals + bk->off[rd]; r.read_len = (int32_t)(bk->off[rd + 1] - bk->off[rd]);
					r.mate_read = bk->bases + bk->off[md]; r.mate_qual = bk->quals + bk->off[md]; r.mate_read_len = (int32_t)(bk->off[md + 1] - bk->off[md]);
					r.aln_pos = c.pos; r.aln_rev = c.is_rev; r.edit_dist = c.NM; r.n_cigar = c.n_cigar; r.cigar = b->cigar + c.cigar_off;
					r.alts = nullptr; r.n_alts = 0;
					if (x) {
						ema_sam_alt &al = o->alts[at];
						al.chrom = contig_names[x->rid]; al.pos = (uint32_t)(x->pos + 1); al.edit_dist = x->NM; al.rev = x->is_rev != 0;
						al.n_cigar = x->n_cigar; al.cigar = b->cigar + x->cigar_off;
						r.alts = &al; r.n_alts = 1;
					}
					out_rec = &r;
				}
				++at;
				return out_rec;
			};
			// statistics of the lines as print_sam_record will flag them (src/samrecord.c:104-175)
			auto count_line = [&](const Brief *rec, const Brief *mate) {
				++st.lines;
				if (!rec) { ++st.unmapped_mates; return; }
				++st.mapped;
				if (rec->dup) ++st.duplicates;
				if (rec->xa) ++st.with_xa;
				if (mate && rec->rev != mate->rev && rec->rid == mate->rid) {      // is_pair, src/align.c:27-40
					const Brief *r1 = rec, *r2 = mate;
					if (r2->rev) { r1 = mate; r2 = rec; }
					const int64_t d = (int64_t)(uint32_t)(r1->pos - r2->pos);      // two uint32_t: the difference wraps
					if (kInsertMin <= d && d <= kInsertMax) ++st.proper;
				}
				const int q = rec->mapq;
				++st.mapq_hist[q == 0 ? 0 : q < 10 ? 1 : q < 20 ? 2 : q < 30 ? 3 : q < 40 ? 4 : q < 60 ? 5 : 6];
			};
			for (uint32_t k = 0; k < S.n_sel[g]; ++k) {
				const Sel sl = S.sel[r0 + k];
				const bool has_mate = sl.mate != ~(uint64_t)0;
				Brief b1, b2;
				if (want_descs) o->sel_at[ln / 2] = (uint32_t)at;
				const ema_sam_rec *rec = fill(sl.rec, has_mate, b1);
				const ema_sam_rec *mate = has_mate ? fill(sl.mate, false, b2) : nullptr;
				if (want_lines) { o->lines[ln] = ema_sam_line{rec, mate}; o->lines[ln + 1] = ema_sam_line{mate, rec}; }
				ln += 2;
				count_line(&b1, has_mate ? &b2 : nullptr);
				count_line(has_mate ? &b2 : nullptr, &b1);
			}
		}
		part[(size_t)tid] = st; cig_lo[(size_t)tid] = clo; cig_hi[(size_t)tid] = chi;
	};
	EmaPool::get().run((size_t)nt_asm, [&](size_t t) { assemble((int)t); });
	o->cigar_lo = ~(uint64_t)0; o->cigar_hi = 0;
	for (int t = 0; t < nt_asm; ++t) { o->cigar_lo = std::min(o->cigar_lo, cig_lo[(size_t)t]); o->cigar_hi = std::max(o->cigar_hi, cig_hi[(size_t)t]); }
	if (o->cigar_lo > o->cigar_hi) o->cigar_lo = o->cigar_hi = 0;
	for (const auto &ps : part) {
		o->stats.lines += ps.lines; o->stats.mapped += ps.mapped; o->stats.unmapped_mates += ps.unmapped_mates; o->stats.proper += ps.proper;
		o->stats.duplicates += ps.duplicates; o->stats.with_xa += ps.with_xa;
		for (int i = 0; i < 7; ++i) o->stats.mapq_hist[i] += ps.mapq_hist[i];
	}
	const int rc = rc_all.load();
	o->stats.select_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
	*out = o;
	return rc;
}

}  // extern "C"
