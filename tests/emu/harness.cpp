// tests/emu/harness.cpp -- runs the HIP kernels of ema_amd/csrc through the host SIMT
// interpreter (simt_emu.h).  TEST INFRASTRUCTURE for machines without a GPU; never shipped.
#include <hip/hip_runtime.h>   // resolves to tests/emu/hip/hip_runtime.h
#include <string>
#include <vector>
#include "host_index.h"
// the library's development knobs (engine.hip: ema_engine_set_tuning / EMA_TUNING) as the interpreter sees them: EMU_<KNOB> in the environment
const char *ema_tuning_get(const char *key)
{
	std::string name = "EMU_";
	for (const char *p = key; *p; ++p) name += (char)toupper((unsigned char)*p);
	return getenv(name.c_str());
}
#include "opts.h"

// kernels + launchers, compiled as plain C++
#include "k_kmer.hip"
#include "k_seed.hip"
#include "k_seed_p3.hip"
#include "k_seed_wave.hip"
#include "k_dp_test.hip"
#include "k_align.hip"
#include "k_align_lane.hip"
#include "k_pair.hip"
#include "k_final.hip"
#include "k_sam.hip"
#include "ingest_kernels.hpp"

// per-read capacities of the emulated launches (the strides of the arrays tests/emu_lib.py allocates)
#define EMU_INTV_CAP 512
#define EMU_REG_CAP 256
#define EMU_CIG_CAP 1024
static DevOpts emu_dev_opts(const ema_engine_opts &o_in)
{
	ema_engine_opts o = o_in;
	if (const char *v = getenv("EMU_MIN_SEED_LEN")) o.min_seed_len = atoi(v);      // (bwa's -k, for the seeding shortcuts that depend on it)
	DevOpts d = ema_make_dev_opts(o);
	d.intv_cap = EMU_INTV_CAP; d.reg_cap = EMU_REG_CAP; d.cig_cap = EMU_CIG_CAP;
	// pass 3 as its own kernel behind K1 (k_seed_p3.hip), as the product runs it; EMU_SEED_SPLIT3=0: inside K1's machine
	static std::vector<int32_t> ext;
	const char *sp = getenv("EMU_SEED_SPLIT3");
	if (!sp || atoi(sp) != 0) { ext.assign(1 << 20, 0); d.seed_flags |= 8; d.seed_ext = ext.data(); }
	return d;
}
// K1c behind a K1 launch (series) of the harness
static void emu_seed_p3(const DevIndex &di, const DevOpts &d, const uint32_t *qp, const uint32_t *off, int n_reads, const int *n_dev, const int *map,
                        Intv *intv, int *n_intv, int *status, int n_blocks)
{
	if (!ema_seed_splits_pass3(&d, nullptr)) return;
	int ctr = 0;
	ema_launch_seed_p3(&di, &d, qp, off, n_reads, n_dev, map, intv, n_intv, status, d.seed_ext, &ctr, nullptr, nullptr, 0, n_blocks, nullptr);
}

static std::vector<uint32_t> pack_reads(const uint8_t *bases, const uint32_t *off, int n_reads)
{
	std::vector<uint32_t> q((size_t)n_reads * 24 + 8, 0);
	for (int r = 0; r < n_reads; ++r)
		for (uint32_t i = 0; i < off[r + 1] - off[r]; ++i) {
			const uint8_t b = bases[off[r] + i];
			q[(size_t)r * 24 + (i >> 4)] |= (uint32_t)(b & 3) << ((i & 15) << 1);
			if (b > 3) q[(size_t)r * 24 + 16 + (i >> 5)] |= 1u << (i & 31);
		}
	return q;
}

// K2 as the engine runs it: K2a (one lane per read) for the small reads, then K2b (one wavefront per read) over K2a's
// todo list.  EMU_LANE_ALIGN=0 sends every read through K2b.
static void emu_run_align(const DevIndex &di, const DevOpts &d, const uint8_t *bases, const uint32_t *qp, const uint32_t *off, int n_reads,
                          const int *n_dev, const int *map, Intv *intv, int *n_intv, DevReg *regs, int *n_regs, int *status,
                          uint8_t *slabs, int n_blocks)
{
	const char *v = getenv("EMU_LANE_ALIGN");
	const bool lane = !v || atoi(v) != 0;
	std::vector<int> todo(n_reads + 1);
	std::vector<uint8_t> hand((size_t)n_reads * EMA_HAND_BYTES);
	int n_todo = 0, n_hand = 0, c0 = 0, c1 = 0;
	if (lane) {
		std::vector<uint8_t> scratch((size_t)n_blocks * 4 * ema_align_lane_wave_bytes());
		ema_launch_align_simple(&di, &d, qp, off, n_reads, n_dev, map, intv, n_intv, regs, n_regs, status, scratch.data(), &c0, todo.data(),
		                        &n_todo, hand.data(), &n_hand, n_blocks, nullptr, nullptr);
		fprintf(stderr, "emu K2a: %d of %d reads handed over with their chains, %d left for K2b's full path\n", n_hand, n_reads, n_todo);
	}
	// EMU_HEAVY_CHAINS=n: K2b sets reads with at least n chains to extend aside for K2c / K2d (EMU_HEAVY_ARENA: arena bytes, to run it full)
	const char *vh = getenv("EMU_HEAVY_CHAINS");
	const int heavy_chains = vh ? atoi(vh) : 0;
	HeavyCtl hv;
	std::vector<uint8_t> arena;
	std::vector<unsigned long long> hreads, htasks, used(1, 0);
	int hn[4] = {0, 0, 0, 0};
	if (heavy_chains > 0) {
		const char *va = getenv("EMU_HEAVY_ARENA");
		arena.resize(va ? (size_t)atol(va) : (size_t)256 << 20);
		hreads.resize(getenv("EMU_HEAVY_READS") ? atoi(getenv("EMU_HEAVY_READS")) : n_reads + 1);
		htasks.resize(getenv("EMU_HEAVY_TASKS") ? atoi(getenv("EMU_HEAVY_TASKS")) : 1 << 20);
		hv.arena = arena.data(); hv.arena_bytes = arena.size(); hv.arena_used = used.data(); hv.reads = hreads.data(); hv.tasks = htasks.data();
		hv.n_reads = &hn[0]; hv.n_tasks = &hn[1]; hv.reads_cap = (int)hreads.size(); hv.tasks_cap = (int)htasks.size(); hv.min_chains = heavy_chains;
	}
	if (heavy_chains <= 0) { hv.arena = nullptr; hv.arena_bytes = 0; hv.arena_used = nullptr; hv.reads = hv.tasks = nullptr; hv.n_reads = hv.n_tasks = nullptr; hv.reads_cap = hv.tasks_cap = 0; hv.min_chains = 1 << 30; }
	int c3 = 0;
	if (lane)      // K2a's hand-overs on their own build
		ema_launch_align(&di, &d, bases, off, n_reads, n_dev, map, intv, n_intv, regs, n_regs, status, nullptr, &n_hand, hand.data(), slabs,
		                 &c3, n_blocks, nullptr, nullptr, nullptr, &hv, 3);
	ema_launch_align(&di, &d, bases, off, n_reads, n_dev, map, intv, n_intv, regs, n_regs, status, lane ? todo.data() : nullptr, &n_todo, hand.data(), slabs,
	                 &c1, n_blocks, nullptr, nullptr, nullptr, &hv, 0);
	if (heavy_chains > 0) {
		fprintf(stderr, "emu K2b: %d reads set aside, %d chain tasks, %llu arena bytes\n", hn[0], hn[1], used[0]);
		for (int mode = 1; mode <= 2; ++mode)
			ema_launch_align(&di, &d, bases, off, n_reads, n_dev, map, intv, n_intv, regs, n_regs, status, nullptr, nullptr, hand.data(), slabs,
			                 &hn[1 + mode], n_blocks, nullptr, nullptr, nullptr, &hv, mode);
	}
}

extern "C" {

void *emu_index_load(const char *prefix, char *err, int errlen)
{
	HostIndex *ix = new HostIndex();
	std::string e = host_index_load(prefix, *ix);
	if (!e.empty()) { snprintf(err, errlen, "%s", e.c_str()); delete ix; return nullptr; }
	if (const char *v = getenv("EMU_KMER_K")) {      // the k-mer interval table, built by the device kernel under the interpreter
		const int k = atoi(v);
		if (k > 0 && k <= EMA_KMER_MAX) {
			const int w = k < EMA_KMER_WIDE ? k : EMA_KMER_WIDE;
			ix->kmer_wide.assign(2 * ((((size_t)1 << (2 * (w + 1))) - 4) / 3) + 2, 0);
			if (k > EMA_KMER_WIDE) ix->kmer_narrow.assign((((size_t)1 << (2 * (k + 1))) - ((size_t)1 << (2 * (EMA_KMER_WIDE + 1)))) / 3 + 1, 0);
			int overflow = 0;
			for (int L = 1; L <= k; ++L) {
				DevIndex di = ix->view();
				ema_launch_kmer_level(&di, L, ix->kmer_wide.data(), ix->kmer_narrow.empty() ? nullptr : ix->kmer_narrow.data(), &overflow, nullptr);
			}
			if (overflow) { snprintf(err, errlen, "k-mer table overflow"); delete ix; return nullptr; }
			ix->kmer_k = k;
			const char *t = getenv("EMU_SEED_TAIL");
			if (!t || atoi(t) != 0) {      // the 2-bit text K1's tails read, by the device kernel under the interpreter
				ix->text2.assign(ema_text2_words(ix->l_pac), 0);
				ema_launch_text2(ix->pac.data(), ix->l_pac, ix->text2.data(), nullptr);
			}
		}
	}
	return ix;
}
void emu_index_free(void *h) { delete (HostIndex *)h; }
// rows of the loaded suffix array, widened to 64 bits (tests/test_index_build.py: the flat array expanded from bwa's sampled .sa)
void emu_index_sa(void *h, uint64_t *out, uint64_t n)
{
	const HostIndex *ix = (const HostIndex *)h;
	for (uint64_t i = 0; i < n; ++i)
		out[i] = ix->sa_width == 4 ? (uint64_t)((const uint32_t *)ix->sa_bytes.data())[i] : ((const uint64_t *)ix->sa_bytes.data())[i];
}

// bases: nt4 codes.  intv: n_reads*EMU_INTV_CAP*4 u64.  Returns the interval capacity per read.
int emu_seed(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, uint64_t *intv, int *n_intv, int *status,
             int n_blocks)
{
	HostIndex *ix = (HostIndex *)h;
	ema_engine_opts o; ema_fill_default_opts(&o);
	DevOpts d = emu_dev_opts(o);
	DevIndex di = ix->view();
	std::vector<Intv> lists((size_t)n_blocks * 256 * 2 * EMA_LIST_CAP);
	int seed_counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	std::vector<int> emu_order;      // the engine's heavy-first order of the reads (k_seed.hip, ema_k_seed_order), in table mode
	if (di.kmer_k > 0 && !(getenv("EMU_SEED_ORDER") && atoi(getenv("EMU_SEED_ORDER")) == 0)) {
		emu_order.assign((size_t)n_reads, -1);
		int cnt[2] = {0, 0};
		ema_launch_seed_order(&di, qp.data(), off, n_reads, emu_order.data(), cnt, 6, 16, nullptr);
		fprintf(stderr, "emu_seed order: %d reads expected long first, %d after them\n", cnt[0], cnt[1]);
	}
	{   // the engine's series of launches: fresh reads, then the machines parked by retiring waves (n_blocks < 0: no parking)
		const int nb = n_blocks < 0 ? -n_blocks : n_blocks, park_max = n_blocks < 0 ? 0 : 40, rounds = n_blocks < 0 ? 1 : 4;
		lists.assign((size_t)nb * 256 * 2 * EMA_LIST_CAP, Intv());
		std::vector<uint8_t> park[2];
		for (auto &pk : park) pk.resize((size_t)nb * 4 * 64 * ema_seed_park_bytes());
		int n_park[2] = {0, 0}, ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int r = 0; r < rounds; ++r) {
			const bool last = r == rounds - 1;
			const int in = (r + 1) & 1, out = r & 1;
			if (r >= 2) n_park[out] = 0;
			ema_launch_seed(&di, &d, qp.data(), off, n_reads, nullptr, nullptr, (Intv *)intv, n_intv, status, lists.data(), &ctr[r],
			                r == 0 ? nullptr : park[in].data(), &n_park[in], last ? nullptr : park[out].data(), &n_park[out],
			                last ? 0 : park_max, nullptr, nullptr, 0, emu_order.empty() ? nullptr : emu_order.data(), nb, nullptr, nullptr);
			fprintf(stderr, "emu_seed round %d: parked %d\n", r, last ? 0 : n_park[out]);
		}
		emu_seed_p3(di, d, qp.data(), off, n_reads, nullptr, nullptr, (Intv *)intv, n_intv, status, nb);
	}
	for (int r = 0; r < n_reads; ++r) {
		Intv *a = (Intv *)intv + (size_t)r * EMU_INTV_CAP;
		std::stable_sort(a, a + n_intv[r], [](const Intv &x, const Intv &y) { return x.info < y.info; });
	}
	return EMU_INTV_CAP;
}

// K1w (one wavefront per read) on host memory; same outputs as emu_seed
extern "C" int emu_seed_wave(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, uint64_t *intv, int *n_intv, int *status)
{
	HostIndex *ix = (HostIndex *)h;
	ema_engine_opts o; ema_fill_default_opts(&o);
	DevOpts d = emu_dev_opts(o);
	DevIndex di = ix->view();
	int counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	ema_launch_seed_wave(&di, &d, qp.data(), off, n_reads, nullptr, nullptr, nullptr, (Intv *)intv, n_intv, status, &counter, 1, nullptr);
	for (int r = 0; r < n_reads; ++r) {
		Intv *a = (Intv *)intv + (size_t)r * EMU_INTV_CAP;
		std::stable_sort(a, a + n_intv[r], [](const Intv &x, const Intv &y) { return x.info < y.info; });
	}
	return EMU_INTV_CAP;
}

static DevOpts default_dev_opts() { ema_engine_opts o; ema_fill_default_opts(&o); return emu_dev_opts(o); }

void emu_dp_extend(const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf, const uint32_t *toff, const int *prm,
                   int n, int *out)
{
	DevOpts d = default_dev_opts();
	ema_launch_test_extend(&d, qbuf, qoff, tbuf, toff, prm, n, out, nullptr);
}
void emu_dp_global(const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf, const uint32_t *toff, const int *prm,
                   int n, int *out, uint32_t *cig, int cap)
{
	DevOpts d = default_dev_opts();
	size_t zs = 256 * 1024;
	std::vector<uint8_t> z((size_t)n * zs);
	ema_launch_test_global(&d, qbuf, qoff, tbuf, toff, prm, n, out, cig, cap, z.data(), zs, nullptr);
}
void emu_dp_local(const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf, const uint32_t *toff, const int *prm,
                  int n, int *out)
{
	DevOpts d = default_dev_opts();
	size_t bs = 2048;
	std::vector<uint64_t> b((size_t)n * bs);
	ema_launch_test_local(&d, qbuf, qoff, tbuf, toff, prm, n, out, b.data(), bs, nullptr);
}

// K1 + K2 on host memory.  regs: n_reads * EMU_REG_CAP DevReg (80 bytes each); returns EMU_REG_CAP
int emu_align(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, void *regs, int *n_regs, int *status,
              int n_blocks)
{
	HostIndex *ix = (HostIndex *)h;
	DevOpts d = default_dev_opts();
	DevIndex di = ix->view();
	std::vector<Intv> intv((size_t)n_reads * EMU_INTV_CAP);
	std::vector<int> n_intv(n_reads);
	std::vector<Intv> lists((size_t)1 * 256 * 2 * EMA_LIST_CAP);
	for (int i = 0; i < n_reads; ++i) status[i] = 0;
	int seed_counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	ema_launch_seed(&di, &d, qp.data(), off, n_reads, nullptr, nullptr, intv.data(), n_intv.data(), status, lists.data(), &seed_counter, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 1, nullptr, nullptr);
	emu_seed_p3(di, d, qp.data(), off, n_reads, nullptr, nullptr, intv.data(), n_intv.data(), status, 1);
	std::vector<uint8_t> slabs((size_t)n_blocks * 4 * ema_align_slab_bytes());
	int counter = 0;
	emu_run_align(di, d, bases, qp.data(), off, n_reads, nullptr, nullptr, intv.data(), n_intv.data(), (DevReg *)regs, n_regs, status, slabs.data(), n_blocks);
	return EMU_REG_CAP;
}
int emu_sizeof_reg() { return (int)sizeof(DevReg); }
// dev_sort.hpp: the wavefront's introsort and the single-lane one on copies of the same keys (tests/test_emu_dp.py)
void emu_sort(uint64_t *wave_io, uint64_t *seq_io, const int *n, int cap, int n_tasks, int by_weight) { ema_launch_test_sort(wave_io, seq_io, n, cap, n_tasks, by_weight, nullptr); }

// k_sam.hip: the SAM formatter's three kernels on host memory (tests/test_sam_format.py, tests/test_golden_sam.py); the job's pointers
// are the caller's arrays.  Returns the text's length (text may be NULL to ask for it), -7 where the kernel flags a bad base.
long long emu_sam_format(const SamJob *job, char *text, long long cap)
{
	SamJob J = *job;
	const uint32_t n_chunks = (J.n_lines + 63u) / 64u;
	std::vector<uint32_t> local((size_t)J.n_lines + 64), ctot((size_t)n_chunks + 1);
	std::vector<uint64_t> cbase((size_t)n_chunks + 1);
	uint64_t total = 0;
	int bad = 0;
	ema_launch_sam_len(J, local.data(), ctot.data(), nullptr);
	ema_launch_sam_tops(n_chunks, ctot.data(), cbase.data(), &total, nullptr);
	if (!text || (long long)total > cap) return (long long)total;
	ema_launch_sam_write(J, local.data(), cbase.data(), text, &bad, nullptr);
	return bad ? -7 : (long long)total;
}
int emu_sizeof_sam_job() { return (int)sizeof(SamJob); }

// ingest_kernels.hpp: the bucket reader's kernels on host memory, with the driver's library passes (newline select, radix sort,
// prefix sums: rocPRIM on the device) done by plain loops here.  Returns the kernels' irregularity flags (0: the bucket is in the
// caller's arrays, *n_out pairs); bases / quals / ids must hold len bytes each, bc n_cap, off 2 n_cap + 1, id_off n_cap + 1 entries.
int emu_ingest(const char *text, uint32_t len, int bc_len, int is_haplotag, uint32_t max_read_len, uint32_t n_cap, uint64_t *bc, uint32_t *off, uint32_t *id_off,
               char *bases, char *quals, char *ids, uint32_t *n_out)
{
	unsigned long long n_nl64 = 0;
	int irr = 0;
	*n_out = 0;
	if (!len) return 0;
	std::vector<char> t(text, text + len);
	t.resize((size_t)len + 64, 0);
	hipLaunchKernelGGL(ema_k_ing_count, dim3((len + 4095u) / 4096u), dim3(256), 0, nullptr, t.data(), len, &n_nl64, &irr);
	if (irr) return irr;
	std::vector<uint32_t> nl;
	for (uint32_t i = 0; i < len; ++i) if (t[i] == '\n') nl.push_back(i);
	if (nl.size() != n_nl64) return -1;
	const uint32_t n_nl = (uint32_t)nl.size(), n = n_nl + (t[len - 1] != '\n' ? 1u : 0u);
	if (n > n_cap) return -2;
	nl.push_back(0);
	std::vector<Fields> fields(n);
	std::vector<uint64_t> codes(n), codes_s(n);
	std::vector<uint32_t> idx(n), order(n), lo(n), rlen(2 * (size_t)n + 1), ilen((size_t)n + 1);
	hipLaunchKernelGGL(ema_k_ing_parse, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, t.data(), len, nl.data(), n_nl, n, bc_len, is_haplotag, max_read_len,
	                   fields.data(), codes.data(), lo.data(), idx.data(), &irr);
	if (irr) return irr;
	order = idx;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return codes[a] != codes[b] ? codes[a] < codes[b] : (is_haplotag && lo[a] < lo[b]); });
	for (uint32_t i = 0; i < n; ++i) codes_s[i] = codes[order[i]];
	hipLaunchKernelGGL(ema_k_ing_lens, dim3(n / 256u + 1u), dim3(256), 0, nullptr, order.data(), fields.data(), n, rlen.data(), ilen.data());
	uint32_t run = 0;
	for (size_t r = 0; r <= 2 * (size_t)n; ++r) { off[r] = run; run += rlen[r]; }
	run = 0;
	for (size_t p = 0; p <= n; ++p) { id_off[p] = run; run += ilen[p]; }
	hipLaunchKernelGGL(ema_k_ing_gather, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, t.data(), len, nl.data(), n_nl, order.data(), fields.data(), codes_s.data(), n,
	                   bc_len, is_haplotag, off, id_off, bases, quals, ids, bc);
	*n_out = n;
	return 0;
}

// the whole pipeline K1..K4 on host memory (n_reads even: pairs)
// K4's set-aside path (k_final.hip: K4t / K4r) on host memory: lists, arena and counters; EMU_K4_HEAVY = regions a read must have left
// for K4b to be set aside (default 2 here, so that the path runs on ordinary test reads; 0: never)
struct EmuFinalHeavy {
	std::vector<uint8_t> arena;
	std::vector<unsigned long long> reads, tasks;
	int counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long used = 0;
	HeavyCtl hv;
	int min_regions, min_attempts;
	EmuFinalHeavy() : arena((size_t)8 << 20), reads(4096), tasks(65536)
	{
		memset(&hv, 0, sizeof(hv));
		hv.arena = arena.data(); hv.arena_bytes = arena.size(); hv.reads = reads.data(); hv.tasks = tasks.data();
		hv.reads_cap = (int)reads.size(); hv.tasks_cap = (int)tasks.size();
		const char *v = getenv("EMU_K4_HEAVY");
		min_regions = v ? atoi(v) : 2;
		const char *u = getenv("EMU_K3_HEAVY");      // candidate rescue anchors for K3b to set a pair aside (default 2 here; 0: never)
		min_attempts = u ? atoi(u) : 2;
	}
};

int emu_pipeline(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, void *regs, int *n_regs, void *alns,
                 uint32_t *cigars, int *cig_n, int *status, int upto)
{
	HostIndex *ix = (HostIndex *)h;
	ema_engine_opts eo; ema_fill_default_opts(&eo);
	DevOpts d = emu_dev_opts(eo);
	DevIndex di = ix->view();
	std::vector<Intv> intv((size_t)n_reads * EMU_INTV_CAP);
	std::vector<int> n_intv(n_reads);
	std::vector<Intv> lists((size_t)1 * 256 * 2 * EMA_LIST_CAP);
	for (int i = 0; i < n_reads; ++i) status[i] = 0;
	int seed_counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	ema_launch_seed(&di, &d, qp.data(), off, n_reads, nullptr, nullptr, intv.data(), n_intv.data(), status, lists.data(), &seed_counter, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 1, nullptr, nullptr);
	emu_seed_p3(di, d, qp.data(), off, n_reads, nullptr, nullptr, intv.data(), n_intv.data(), status, 1);
	size_t slab = ema_align_slab_bytes();
	if (ema_pair_slab_bytes() > slab) slab = ema_pair_slab_bytes();
	if (ema_final_slab_bytes() > slab) slab = ema_final_slab_bytes();
	std::vector<uint8_t> slabs((size_t)4 * slab);
	int counter[3] = {0, 0, 0};
	emu_run_align(di, d, bases, qp.data(), off, n_reads, nullptr, nullptr, intv.data(), n_intv.data(), (DevReg *)regs, n_regs, status, slabs.data(), 1);
	if (upto >= 3)
	{
		std::vector<int> todo(n_reads); int n_todo = 0;
		EmuFinalHeavy phv;
		ema_launch_pair(&di, &d, eo.score_delta, eo.max_rescue, eo.pes_low, eo.pes_high, bases, off, n_reads / 2, nullptr, nullptr, (DevReg *)regs,
		                n_regs, status, getenv("EMU_NO_K3A") ? nullptr : todo.data(), &n_todo, slabs.data(), &counter[1], 1, nullptr, nullptr,
		                &phv.hv, phv.counters, &phv.used, phv.min_attempts);
		fprintf(stderr, "emu K3: %d pairs set aside, %d + %d attempt tasks\n", phv.counters[0], phv.counters[1], phv.counters[2]);
		fprintf(stderr, "emu K3a: %d of %d pairs need a rescue alignment\n", n_todo, n_reads / 2);
	}
	if (upto >= 4)
	{
		std::vector<int> kdone(n_reads), todo(n_reads); int n_todo = 0;
		EmuFinalHeavy fhv;
		ema_launch_final(&di, &d, bases, qp.data(), off, n_reads, nullptr, nullptr, (DevReg *)regs, n_regs, (DevAln *)alns, cigars, cig_n, EMU_CIG_CAP, status,
		                 kdone.data(), todo.data(), &n_todo, slabs.data(), &counter[2], 1, nullptr, nullptr, &fhv.hv, fhv.counters, &fhv.used, fhv.min_regions);
		fprintf(stderr, "emu K4: %d reads set aside, %d region tasks\n", fhv.counters[0], fhv.counters[1]);
	}
	return EMU_CIG_CAP;
}

// K1..K4 of one tier on host memory
struct TierBuf {
	std::vector<Intv> intv; std::vector<int> n_intv, n_regs, cig_n, status;
	std::vector<DevReg> regs; std::vector<DevAln> alns; std::vector<uint32_t> cigars;
	void size(int n_reads, const DevOpts &d)
	{
		intv.assign((size_t)n_reads * d.intv_cap, Intv()); n_intv.assign(n_reads, 0); n_regs.assign(n_reads, 0); cig_n.assign(n_reads, 0);
		status.assign(n_reads, 0); regs.assign((size_t)n_reads * d.reg_cap, DevReg()); alns.assign((size_t)n_reads * d.reg_cap, DevAln());
		cigars.assign((size_t)n_reads * d.cig_cap, 0);
	}
};

static void run_tier(const DevIndex &di, const DevOpts &d, const ema_engine_opts &eo, const uint8_t *bases, const uint32_t *off,
                     const uint32_t *qp, int n_pairs, const int *n_dev, const int *map, TierBuf &t)
{
	std::vector<Intv> lists((size_t)1 * 256 * 2 * EMA_LIST_CAP);
	size_t slab = ema_align_slab_bytes();
	if (ema_pair_slab_bytes() > slab) slab = ema_pair_slab_bytes();
	if (ema_final_slab_bytes() > slab) slab = ema_final_slab_bytes();
	std::vector<uint8_t> slabs((size_t)4 * slab);
	int counter[4] = {0, 0, 0, 0};
	ema_launch_seed(&di, &d, qp, off, 2 * n_pairs, n_dev, map, t.intv.data(), t.n_intv.data(), t.status.data(), lists.data(), &counter[3], nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 1, nullptr, nullptr);
	emu_seed_p3(di, d, qp, off, 2 * n_pairs, n_dev, map, t.intv.data(), t.n_intv.data(), t.status.data(), 1);
	emu_run_align(di, d, bases, qp, off, 2 * n_pairs, n_dev, map, t.intv.data(), t.n_intv.data(), t.regs.data(), t.n_regs.data(), t.status.data(),
	              slabs.data(), 1);
	std::vector<int> ptodo(n_pairs + 1); int n_ptodo = 0;
	EmuFinalHeavy phv;
	ema_launch_pair(&di, &d, eo.score_delta, eo.max_rescue, eo.pes_low, eo.pes_high, bases, off, n_pairs, n_dev, map, t.regs.data(),
	                t.n_regs.data(), t.status.data(), ptodo.data(), &n_ptodo, slabs.data(), &counter[1], 1, nullptr, nullptr,
	                &phv.hv, phv.counters, &phv.used, phv.min_attempts);
	std::vector<int> kdone(2 * n_pairs), todo(2 * n_pairs); int n_todo = 0;
	EmuFinalHeavy fhv;
	ema_launch_final(&di, &d, bases, qp, off, 2 * n_pairs, n_dev, map, t.regs.data(), t.n_regs.data(), t.alns.data(), t.cigars.data(),
	                 t.cig_n.data(), d.cig_cap, t.status.data(), kdone.data(), todo.data(), &n_todo, slabs.data(), &counter[2], 1, nullptr, nullptr,
	                 &fhv.hv, fhv.counters, &fhv.used, fhv.min_regions);
}

// The engine's two capacity tiers on host memory: lean tier with the given capacities, ema_k_collect, full tier over the
// listed pairs; results merged per read into the caller's arrays (strides EMU_REG_CAP / EMU_CIG_CAP).  tier[r] = 0 lean,
// 1 full.  Returns the number of listed pairs.
int emu_pipeline_tiers(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, int lean_intv, int lean_reg, int lean_cig,
                       int full_pairs, void *regs, int *n_regs, void *alns, uint32_t *cigars, int *cig_n, int *status, int *tier)
{
	HostIndex *ix = (HostIndex *)h;
	ema_engine_opts eo; ema_fill_default_opts(&eo);
	DevIndex di = ix->view();
	const int n_pairs = n_reads / 2;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	DevOpts dl = emu_dev_opts(eo), df = emu_dev_opts(eo);
	dl.intv_cap = lean_intv; dl.reg_cap = lean_reg; dl.cig_cap = lean_cig;
	TierBuf lean, full;
	lean.size(n_reads, dl);
	run_tier(di, dl, eo, bases, off, qp.data(), n_pairs, nullptr, nullptr, lean);
	std::vector<int> redo(full_pairs + 1, 0);
	ema_launch_collect(n_pairs, 0, lean.status.data(), redo.data(), redo.data() + 1, full_pairs, nullptr);
	full.size(2 * full_pairs, df);
	run_tier(di, df, eo, bases, off, qp.data(), full_pairs, redo.data(), redo.data() + 1, full);
	auto put = [&](int r, const TierBuf &t, const DevOpts &d, int i, int which) {
		tier[r] = which; n_regs[r] = t.n_regs[i]; cig_n[r] = t.cig_n[i]; status[r] = t.status[i];
		for (int k = 0; k < t.n_regs[i]; ++k) {
			((DevReg *)regs)[(size_t)r * EMU_REG_CAP + k] = t.regs[(size_t)i * d.reg_cap + k];
			((DevAln *)alns)[(size_t)r * EMU_REG_CAP + k] = t.alns[(size_t)i * d.reg_cap + k];
		}
		for (int k = 0; k < t.cig_n[i]; ++k) cigars[(size_t)r * EMU_CIG_CAP + k] = t.cigars[(size_t)i * d.cig_cap + k];
	};
	for (int r = 0; r < n_reads; ++r)
		if (!lean.status[r]) put(r, lean, dl, r, 0);
		else { tier[r] = -1; n_regs[r] = cig_n[r] = 0; status[r] = lean.status[r]; }
	const int n_redo = redo[0] < full_pairs ? redo[0] : full_pairs;
	for (int i = 0; i < 2 * n_redo; ++i) put(2 * redo[1 + (i >> 1)] + (i & 1), full, df, i, 1);
	return redo[0];
}
}
