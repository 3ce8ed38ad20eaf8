// ema_amd/csrc/bwaabi.cpp -- libema_bwaabi.so: the nine symbols EMA links from -lbwa (include/ema_bwaabi.h), each a thin
// host wrapper around the engine's C ABI (include/ema_engine.h).  The reference's bridge calls them once per read / region
// (reference src/bwabridge.c:173,236-237,267,281,304); here every such call stages a batch of one and runs the same HIP
// kernels the batched path runs.  No alignment arithmetic in this file except bwa's mapq formula for mem_aln_t.mapq (a
// field the reference never reads: it recomputes its own at src/align.c:959-984) and the host-side reference fetch of
// bns_fetch_seq.  Nothing here touches oracle/.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "ema_bwaabi.h"
#include "ema_engine.h"

extern "C" {
unsigned char nst_nt4_table[256] = {
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5 /*'-'*/, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
}

namespace {

// region record of the engine's debug entry points (include/ema_engine.h, ema_engine_debug_regions)
struct Reg {
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, csub, w, seedcov, secondary, seedlen0, n_comp, is_alt;
	float frac_rep;
};

struct Handle {
	bwaidx_t idx;
	bwt_t bwt;
	bntseq_t bns;
	ema_engine_t *eng = nullptr;
	ema_engine_opts eopts;
	std::vector<bntann1_t> anns;
	std::vector<std::string> names;
	std::vector<uint8_t> pac;
	std::mutex mu;
};

std::mutex g_mu;
std::vector<Handle *> g_handles;

Handle *find_by(const void *bwt, const void *bns)
{
	std::lock_guard<std::mutex> lk(g_mu);
	for (Handle *h : g_handles)
		if ((bwt && bwt == &h->bwt) || (bns && bns == &h->bns)) return h;
	return g_handles.size() == 1 ? g_handles[0] : nullptr;      // a copied struct: with one index open there is no doubt
}

// mem_opt_t -> the engine's option record (geometry fields are ignored by ema_engine_set_opts)
void to_engine_opts(const mem_opt_t *o, ema_engine_opts *e)
{
	ema_engine_opts_default(e);
	e->a = o->a; e->b = o->b; e->o_del = o->o_del; e->e_del = o->e_del; e->o_ins = o->o_ins; e->e_ins = o->e_ins;
	e->pen_clip5 = o->pen_clip5; e->pen_clip3 = o->pen_clip3; e->w = o->w; e->zdrop = o->zdrop;
	e->min_seed_len = o->min_seed_len; e->split_width = o->split_width; e->max_mem_intv = (int)o->max_mem_intv; e->max_occ = o->max_occ;
	e->max_chain_gap = o->max_chain_gap; e->min_chain_weight = o->min_chain_weight; e->max_chain_extend = o->max_chain_extend;
	e->split_factor = o->split_factor; e->mask_level = o->mask_level; e->drop_ratio = o->drop_ratio; e->mask_level_redun = o->mask_level_redun;
	e->mapq_coef_len = (int)o->mapQ_coef_len; e->mapq_coef_fac = o->mapQ_coef_fac;
}

bool same_algo_opts(const ema_engine_opts &x, const ema_engine_opts &y)
{
	return x.a == y.a && x.b == y.b && x.o_del == y.o_del && x.e_del == y.e_del && x.o_ins == y.o_ins && x.e_ins == y.e_ins &&
	       x.pen_clip5 == y.pen_clip5 && x.pen_clip3 == y.pen_clip3 && x.w == y.w && x.zdrop == y.zdrop && x.min_seed_len == y.min_seed_len &&
	       x.split_width == y.split_width && x.max_mem_intv == y.max_mem_intv && x.max_occ == y.max_occ && x.max_chain_gap == y.max_chain_gap &&
	       x.min_chain_weight == y.min_chain_weight && x.max_chain_extend == y.max_chain_extend && x.split_factor == y.split_factor &&
	       x.mask_level == y.mask_level && x.drop_ratio == y.drop_ratio && x.mask_level_redun == y.mask_level_redun;
}

// the caller's options become the engine's (the reference edits max_occ after mem_opt_init, src/align.c:185); h->mu held
bool use_opts(Handle *h, const mem_opt_t *opt)
{
	ema_engine_opts want;
	to_engine_opts(opt, &want);
	if (same_algo_opts(want, h->eopts)) return true;
	if (ema_engine_set_opts(h->eng, &want) != EMA_OK) { fprintf(stderr, "[ema_bwaabi] %s\n", ema_engine_strerror(h->eng)); return false; }
	ema_engine_get_opts(h->eng, &h->eopts);
	return true;
}

void to_ascii(const char *seq, int n, std::string &out)      // nt4 codes or ASCII -> ASCII
{
	out.resize((size_t)n);
	for (int i = 0; i < n; ++i) {
		const unsigned char c = (unsigned char)seq[i];
		out[(size_t)i] = "ACGTN"[c < 4 ? c : (nst_nt4_table[c] < 4 ? nst_nt4_table[c] : 4)];
	}
}

Reg to_reg(const mem_alnreg_t &a)
{
	Reg r;
	r.rb = a.rb; r.re = a.re; r.qb = a.qb; r.qe = a.qe; r.rid = a.rid; r.score = a.score; r.truesc = a.truesc; r.sub = a.sub; r.csub = a.csub;
	r.w = a.w; r.seedcov = a.seedcov; r.secondary = a.secondary; r.seedlen0 = a.seedlen0; r.n_comp = a.n_comp; r.is_alt = a.is_alt;
	r.frac_rep = a.frac_rep;
	return r;
}

mem_alnreg_t from_reg(const Reg &r)
{
	mem_alnreg_t a;
	memset(&a, 0, sizeof(a));
	a.rb = r.rb; a.re = r.re; a.qb = r.qb; a.qe = r.qe; a.rid = r.rid; a.score = r.score; a.truesc = r.truesc; a.sub = r.sub; a.csub = r.csub;
	a.w = r.w; a.seedcov = r.seedcov; a.secondary = r.secondary; a.seedlen0 = r.seedlen0; a.n_comp = r.n_comp; a.is_alt = r.is_alt;
	a.frac_rep = r.frac_rep;
	a.secondary_all = 0; a.alt_sc = 0; a.sub_n = 0; a.hash = 0;
	return a;
}

int pos2rid(const bntseq_t *bns, int64_t pos_f)
{
	if (pos_f >= bns->l_pac) return -1;
	int left = 0, right = bns->n_seqs, mid = 0;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= bns->anns[mid].offset) {
			if (mid == bns->n_seqs - 1 || pos_f < bns->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

inline int pac_get(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

// bwa's mem_approx_mapq_se (cap 60)
int approx_mapq_se(const mem_opt_t *opt, const mem_alnreg_t *a)
{
	int sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	const int l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	const double identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
	int mapq;
	if (a->score == 0) mapq = 0;
	else if (opt->mapQ_coef_len > 0) {
		double tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / std::log((double)l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
	} else {
		mapq = (int)(MEM_MAPQ_COEF * (1. - (double)sub / a->score) * std::log((double)a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * std::log((double)(a->sub_n + 1)) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

}  // namespace

extern "C" {

size_t ema_bwaabi_sizeof(int which)
{
	switch (which) {
	case 0: return sizeof(mem_opt_t);
	case 1: return sizeof(mem_alnreg_t);
	case 2: return sizeof(mem_aln_t);
	case 3: return sizeof(mem_pestat_t);
	case 4: return sizeof(bntann1_t);
	case 5: return sizeof(bntseq_t);
	case 6: return sizeof(bwaidx_t);
	case 7: return sizeof(bwt_t);
	case 8: return sizeof(mem_chain_t);
	default: return 0;
	}
}

bwaidx_t *bwa_idx_load(const char *hint, int which)
{
	(void)which;      // the engine always needs BWT + BNS + PAC
	if (!hint) return nullptr;
	Handle *h = new Handle();
	const char *dev = getenv("EMA_DEVICE");
	ema_engine_opts o;
	ema_engine_opts_default(&o);
	// per-call use: small batches (the batched C ABI of include/ema_engine.h is the fast path)
	o.batch_pairs = 256; o.n_streams = 1; o.full_tier_pairs = 256;
	const int rc = ema_engine_open(hint, dev ? atoi(dev) : 0, &o, &h->eng);
	if (rc != EMA_OK) {
		fprintf(stderr, "[E::bwa_idx_load] %s\n", h->eng ? ema_engine_strerror(h->eng) : "cannot open the engine");
		if (h->eng) ema_engine_close(h->eng);
		delete h;
		return nullptr;
	}
	ema_engine_get_opts(h->eng, &h->eopts);
	const int n = ema_engine_n_contigs(h->eng);
	h->names.resize((size_t)n);
	h->anns.resize((size_t)n);
	for (int i = 0; i < n; ++i) {
		h->names[(size_t)i] = ema_engine_contig_name(h->eng, i);
		bntann1_t &a = h->anns[(size_t)i];
		memset(&a, 0, sizeof(a));
		a.offset = ema_engine_contig_offset(h->eng, i);
		a.len = (int32_t)ema_engine_contig_len(h->eng, i);
		a.is_alt = ema_engine_contig_is_alt(h->eng, i);
		a.name = const_cast<char *>(h->names[(size_t)i].c_str());
		a.anno = const_cast<char *>("");
	}
	memset(&h->bns, 0, sizeof(h->bns));
	h->bns.l_pac = ema_engine_l_pac(h->eng);
	h->bns.n_seqs = n;
	h->bns.seed = 11;
	h->bns.anns = h->anns.data();
	memset(&h->bwt, 0, sizeof(h->bwt));
	h->bwt.seq_len = (bwtint_t)(2 * h->bns.l_pac);
	{   // host copy of the packed forward strand, for bns_fetch_seq
		FILE *f = fopen((std::string(hint) + ".pac").c_str(), "rb");
		if (f) {
			h->pac.resize((size_t)(h->bns.l_pac / 4 + 2), 0);
			const size_t got = fread(h->pac.data(), 1, h->pac.size(), f);
			(void)got;
			fclose(f);
		}
	}
	memset(&h->idx, 0, sizeof(h->idx));
	h->idx.bwt = &h->bwt; h->idx.bns = &h->bns; h->idx.pac = h->pac.data();
	std::lock_guard<std::mutex> lk(g_mu);
	g_handles.push_back(h);
	return &h->idx;
}

void bwa_idx_destroy(bwaidx_t *idx)
{
	if (!idx) return;
	Handle *h = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		for (size_t i = 0; i < g_handles.size(); ++i)
			if (&g_handles[i]->idx == idx) { h = g_handles[i]; g_handles.erase(g_handles.begin() + (long)i); break; }
	}
	if (!h) return;
	ema_engine_close(h->eng);
	delete h;
}

mem_opt_t *mem_opt_init(void)
{
	mem_opt_t *o = (mem_opt_t *)calloc(1, sizeof(mem_opt_t));
	if (!o) return nullptr;
	o->flag = 0;
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
	o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)std::log(o->mapQ_coef_len);
	int k = 0;
	for (int i = 0; i < 4; ++i) {
		for (int j = 0; j < 4; ++j) o->mat[k++] = (int8_t)(i == j ? o->a : -o->b);
		o->mat[k++] = -1;
	}
	for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
	return o;
}

mem_alnreg_v mem_align1_core(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int l_seq, char *seq, void *buf)
{
	(void)pac; (void)buf;
	mem_alnreg_v out = {0, 0, nullptr};
	Handle *h = find_by(bwt, bns);
	if (!h || !opt || !seq || l_seq < 0) return out;
	for (int i = 0; i < l_seq; ++i) seq[i] = (char)((unsigned char)seq[i] < 4 ? seq[i] : (char)nst_nt4_table[(unsigned char)seq[i]]);      // as bwa does
	std::string ascii;
	to_ascii(seq, l_seq, ascii);
	std::lock_guard<std::mutex> lk(h->mu);
	if (!use_opts(h, opt)) return out;
	const uint32_t off[3] = {0, (uint32_t)l_seq, (uint32_t)l_seq};      // the read and an empty mate
	if (ema_engine_stage(h->eng, ascii.data(), off, 1) != EMA_OK) { fprintf(stderr, "[ema_bwaabi] %s\n", ema_engine_strerror(h->eng)); return out; }
	void *regs = nullptr; int32_t *n_regs = nullptr, *status = nullptr, cap = 0, bytes = 0;
	if (ema_engine_debug_regions(h->eng, &regs, &n_regs, &status, &cap, &bytes) != EMA_OK || bytes != (int32_t)sizeof(Reg)) {
		fprintf(stderr, "[ema_bwaabi] %s\n", ema_engine_strerror(h->eng));
		free(regs); free(n_regs); free(status);
		return out;
	}
	if (status[0]) fprintf(stderr, "[ema_bwaabi] mem_align1_core: the read exceeded an engine capacity (status %d)\n", status[0]);
	const int n = status[0] ? 0 : n_regs[0];
	out.n = (size_t)n; out.m = (size_t)(n > 0 ? n : 1);
	out.a = (mem_alnreg_t *)malloc(out.m * sizeof(mem_alnreg_t));
	const Reg *r = (const Reg *)regs;
	for (int i = 0; i < n; ++i) out.a[i] = from_reg(r[i]);
	free(regs); free(n_regs); free(status);
	return out;
}

mem_chain_v mem_chain(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, int len, const uint8_t *seq, void *buf)
{
	(void)opt; (void)bwt; (void)bns; (void)len; (void)seq; (void)buf;
	mem_chain_v v = {0, 0, nullptr};
	return v;
}

int mem_matesw(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], const mem_alnreg_t *a, int l_ms,
               const uint8_t *ms, mem_alnreg_v *ma)
{
	(void)pac;
	Handle *h = find_by(nullptr, bns);
	if (!h || !opt || !pes || !a || !ms || !ma) return 0;
	if (!(pes[0].failed && !pes[1].failed && pes[2].failed && pes[3].failed)) {
		static bool said = false;
		if (!said) { said = true; fprintf(stderr, "[ema_bwaabi] mem_matesw: only the FR-only insert model of the reference (src/bwabridge.c:216-227) is supported\n"); }
		return 0;
	}
	std::string ascii;
	to_ascii((const char *)ms, l_ms, ascii);
	std::lock_guard<std::mutex> lk(h->mu);
	if (!use_opts(h, opt)) return 0;
	const int cap = (int)ma->n + 1;
	std::vector<Reg> regs((size_t)cap + 1);
	for (size_t i = 0; i < ma->n; ++i) regs[i] = to_reg(ma->a[i]);
	const Reg anchor = to_reg(*a);
	int32_t n = (int32_t)ma->n, n_sw = 0;
	const int rc = ema_engine_debug_matesw(h->eng, &anchor, ascii.data(), l_ms, regs.data(), &n, cap, pes[1].low, pes[1].high, &n_sw);
	if (rc != EMA_OK) { fprintf(stderr, "[ema_bwaabi] %s\n", ema_engine_strerror(h->eng)); return 0; }
	if ((size_t)n > ma->m || !ma->a) {
		ma->m = (size_t)(n > 0 ? n : 1);
		ma->a = (mem_alnreg_t *)realloc(ma->a, ma->m * sizeof(mem_alnreg_t));
	}
	// fields the engine does not carry (hash, alt_sc, sub_n, secondary_all) are 0 on this path in bwa as well: mem_matesw's new
	// region is memset to 0 and mem_align1_core never sets them before mem_mark_primary_se, which the bridge does not call
	for (int i = 0; i < n; ++i) ma->a[i] = from_reg(regs[(size_t)i]);
	ma->n = (size_t)n;
	return n_sw;
}

mem_aln_t mem_reg2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_seq, const char *seq, const mem_alnreg_t *ar)
{
	(void)pac;
	mem_aln_t a;
	memset(&a, 0, sizeof(a));
	if (!ar || ar->rb < 0 || ar->re < 0) {      // an unmapped record
		a.rid = -1; a.pos = -1; a.flag |= 0x4;
		return a;
	}
	Handle *h = find_by(nullptr, bns);
	if (!h || !opt || !seq) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	std::string ascii;
	to_ascii(seq, l_seq, ascii);
	std::lock_guard<std::mutex> lk(h->mu);
	if (!use_opts(h, opt)) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	const Reg r = to_reg(*ar);
	ema_cand_t c;
	std::vector<uint32_t> cig(4096);
	int32_t n_cig = 0;
	const int rc = ema_engine_debug_final(h->eng, ascii.data(), l_seq, &r, 1, &c, cig.data(), (int)cig.size(), &n_cig);
	if (rc != EMA_OK) { fprintf(stderr, "[ema_bwaabi] %s\n", ema_engine_strerror(h->eng)); a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	a.mapq = (uint32_t)(ar->secondary < 0 ? approx_mapq_se(opt, ar) : 0);
	if (ar->secondary >= 0) a.flag |= 0x100;
	a.pos = c.pos; a.rid = ar->rid; a.is_rev = (uint32_t)(c.is_rev != 0); a.is_alt = (uint32_t)(ar->is_alt != 0);
	a.NM = (uint32_t)c.NM & 0x3fffff;
	a.n_cigar = c.n_cigar;
	a.cigar = (uint32_t *)malloc((size_t)(c.n_cigar > 0 ? c.n_cigar : 1) * 4);
	memcpy(a.cigar, cig.data() + c.cigar_off, (size_t)c.n_cigar * 4);
	a.XA = nullptr;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub; a.alt_sc = ar->alt_sc;
	return a;
}

uint8_t *bns_fetch_seq(const bntseq_t *bns, const uint8_t *pac, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	if (!bns || !pac || !beg || !end || !rid) return nullptr;
	if (*end < *beg) { const int64_t t = *beg; *beg = *end; *end = t; }
	const int is_rev = mid >= bns->l_pac;
	const int64_t mid_f = is_rev ? (bns->l_pac << 1) - 1 - mid : mid;
	*rid = pos2rid(bns, mid_f);
	if (*rid < 0) return nullptr;
	int64_t far_beg = bns->anns[*rid].offset, far_end = far_beg + bns->anns[*rid].len;
	if (is_rev) { const int64_t t = far_beg; far_beg = (bns->l_pac << 1) - far_end; far_end = (bns->l_pac << 1) - t; }
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	const int64_t len = *end - *beg;
	uint8_t *seq = (uint8_t *)malloc((size_t)(len > 0 ? len : 1));
	int64_t l = 0;
	if (*beg >= bns->l_pac) {      // reverse strand: complement, read from the far end
		const int64_t beg_f = (bns->l_pac << 1) - 1 - *end, end_f = (bns->l_pac << 1) - 1 - *beg;
		for (int64_t k = end_f; k > beg_f; --k) seq[l++] = (uint8_t)(3 - pac_get(pac, k));
	} else for (int64_t k = *beg; k < *end; ++k) seq[l++] = (uint8_t)pac_get(pac, k);
	return seq;
}

}  // extern "C"
