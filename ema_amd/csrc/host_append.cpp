// ema_amd/csrc/host_append.cpp -- host stage right behind the engine: what the reference's append_alignments()
// (reference src/align.c:986-1061) does with a pair's candidates once it has them -- the clip filter (:1017, :1042),
// the search-depth filter on edit distance + clipping with its best_dist shared by both mates (:1021-1024,
// :1046-1049), the approximate mapping quality (mem_approx_mapq_se_insist, :959-984), the alignment likelihood and
// its Phred-like companion from the CIGAR (score_alignment, :846-911), and the `unique` flag (:1032-1033, :1057-1058).
// Pure host arithmetic in double precision, as in the reference (a2, a7, a8 stay on the host for
// floating-point identity); no GPU work.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "ema_engine.h"
#include "host_cpuacct.h"
#include "host_pool.h"

extern "C" void ema_aln_free(ema_aln_out *o);
namespace {

// reference include/align.h:70-73
const double kIndelRate = 0.0001, kClipRate = 0.03;
const int kExtraSearchDepth = 12;
const double kMemMapqCoef = 30.0;      // bwa's MEM_MAPQ_COEF (reference src/align.c:975)

int approx_mapq(const ema_engine_opts &o, const ema_cand_t &c)
{
	int sub = c.sub ? c.sub : o.min_seed_len * o.a;
	if (c.csub > sub) sub = c.csub;
	if (sub >= c.score) return 0;
	const int lq = c.qe - c.qb;
	const int64_t lr = c.re - c.rb;
	const int l = lq > lr ? lq : (int)lr;
	const double identity = 1. - (double)(l * o.a - c.score) / (o.a + o.b) / l;
	int mapq;
	if (c.score == 0) mapq = 0;
	else if (o.mapq_coef_len > 0) {
		double t = l < o.mapq_coef_len ? 1. : o.mapq_coef_fac / std::log((double)l);
		t *= identity * identity;
		mapq = (int)(6.02 * (c.score - sub) / o.a * t * t + .499);
	} else {
		mapq = (int)(kMemMapqCoef * (1. - (double)sub / c.score) * std::log((double)c.seedcov) + .499);
		if (identity < 0.95) mapq = (int)(mapq * identity * identity + .499);
	}
	if (c.sub_n > 0) mapq -= (int)(4.343 * std::log((double)(c.sub_n + 1)) + .499);
	if (mapq > 254) mapq = 254;
	if (mapq < 0) mapq = 0;
	return (int)(mapq * (1. - c.frac_rep) + .499);
}

}  // namespace

extern "C" int ema_batch_append_alignments(const ema_batch_out *b, const uint32_t *off, const ema_engine_opts *opts,
                                           double error_rate, ema_aln_out **out)
{
	if (!b || !off || !opts || !out || !(error_rate > 0. && error_rate < 1.)) return EMA_EARG;
	EMA_CPU(EMA_CPU_APPEND);
	*out = nullptr;
	const double ln_match = std::log(1. - error_rate), ln_mis = std::log(error_rate), ln_indel = std::log(kIndelRate),
	             ln_clip = std::log(kClipRate);
	const double lg_mis = std::log10(error_rate), lg_indel = std::log10(kIndelRate), lg_clip = std::log10(kClipRate);
	ema_aln_out *o = (ema_aln_out *)calloc(1, sizeof(ema_aln_out));
	if (!o) return EMA_EDEVICE;      // (out of host memory)
	o->n_pairs = b->n_pairs;
	o->pair_off = (uint64_t *)malloc((b->n_pairs + 1) * sizeof(uint64_t));
	if (!o->pair_off) { ema_aln_free(o); return EMA_EDEVICE; }
	// pairs are independent: chunks of pairs on the host's cores (EMA_HOST_THREADS, default min(32, hardware threads)),
	// each into its own list; the lists are then laid end to end
	int n_thr = 1;
	{
		n_thr = EmaPool::get().size();      // host_pool.h
		if (b->n_pairs < 4096) n_thr = 1;
	}
	std::vector<std::vector<ema_aln_rec>> part(n_thr);
	std::vector<int> bad(n_thr, 0);
	const size_t per = (b->n_pairs + n_thr - 1) / n_thr;
	auto work = [&](int t) {
		EMA_CPU(EMA_CPU_APPEND);
		std::vector<ema_aln_rec> &recs = part[t];
		const size_t p0 = (size_t)t * per, p1 = p0 + per < b->n_pairs ? p0 + per : b->n_pairs;
		recs.reserve((p1 > p0 ? p1 - p0 : 0) * 2 + 16);
		for (size_t p = p0; p < p1; ++p) {
			o->pair_off[p] = recs.size();      // relative to the chunk; rebased below
			int best_dist = -1;      // shared by the two mates, reset only by a mate's first candidate when it passes the clip filter
			for (int m = 0; m < 2; ++m) {
				const size_t r = 2 * p + m;
				const int len = (int)(off[r + 1] - off[r]);
				const uint64_t lo = b->cand_off[r], hi = b->cand_off[r + 1];
				size_t added = 0;
				for (uint64_t i = lo; i < hi; ++i) {
					const ema_cand_t &c = b->cand[i];
					const int clip = len - (c.qe - c.qb);
					if (clip >= len / 2) continue;
					const int dist = c.NM + clip;
					if (i == lo) best_dist = dist;
					else if (dist - best_dist > kExtraSearchDepth) continue;
					ema_aln_rec rec;
					memset(&rec, 0, sizeof(rec));
					rec.pair = (uint32_t)p; rec.mate = (uint8_t)m; rec.cand = i;
					rec.clip = clip; rec.clip_edit_dist = dist;
					rec.mapq = approx_mapq(*opts, c);
					// score_alignment: matches / mismatches / indel events / clipped bases from the CIGAR and NM
					int matches = 0, indels = 0, indel_events = 0, clipping = 0;
					const uint32_t *cig = b->cigar + c.cigar_off;
					for (int k = 0; k < c.n_cigar; ++k) {
						const uint32_t type = cig[k] & 0xf, n = cig[k] >> 4;
						if (type == 0) matches += (int)n;
						else if (type == 1 || type == 2) { indels += (int)n; ++indel_events; }
						else if (type == 3 || type == 4) clipping += (int)n;
						else bad[t] = 1;
					}
					const int mismatches = c.NM - indels;
					matches -= mismatches;
					rec.score = matches * ln_match + mismatches * ln_mis + indel_events * ln_indel + clipping * ln_clip;
					rec.score_mapq = (int)(60.0 + mismatches * lg_mis + indel_events * lg_indel + clipping * lg_clip);
					recs.push_back(rec);
					++added;
				}
				if (added == 1) recs.back().unique = 1;
			}
		}
	};
	EmaPool::get().run((size_t)n_thr, [&](size_t t) { work((int)t); });
	for (int t = 0; t < n_thr; ++t) if (bad[t]) { free(o->pair_off); free(o); return EMA_EARG; }
	size_t total = 0;
	for (int t = 0; t < n_thr; ++t) total += part[t].size();
	o->n = total;
	o->rec = (ema_aln_rec *)malloc((total + 1) * sizeof(ema_aln_rec));
	if (!o->rec) { ema_aln_free(o); return EMA_EDEVICE; }
	size_t at = 0;
	for (int t = 0; t < n_thr; ++t) {
		const size_t p0 = (size_t)t * per, p1 = p0 + per < b->n_pairs ? p0 + per : b->n_pairs;
		for (size_t p = p0; p < p1; ++p) o->pair_off[p] += at;
		if (!part[t].empty()) memcpy(o->rec + at, part[t].data(), part[t].size() * sizeof(ema_aln_rec));
		at += part[t].size();
	}
	o->pair_off[b->n_pairs] = total;
	*out = o;
	return EMA_OK;
}

extern "C" void ema_aln_free(ema_aln_out *o)
{
	if (!o) return;
	free(o->rec); free(o->pair_off); free(o);
}
