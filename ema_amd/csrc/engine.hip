// ema_amd/csrc/engine.hip -- C ABI of the engine (include/ema_engine.h): index upload, batch
// staging, kernel pipeline, result assembly.  Host side of the drop-in boundary that replaces
// the reference's per-pair bridge calls (reference src/bwabridge.c:204-311).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <unistd.h>
#include <string>
#include <vector>
#include "ema_engine.h"
#include "dev_types.h"
#include "host_index.h"
#include "opts.h"

extern "C" void ema_launch_seed(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off,
                                int n_reads, Intv *intv, int *n_intv, int *status, Intv *lists, int *counter, int n_blocks,
                                hipStream_t stream);

extern "C" size_t ema_align_slab_bytes();
extern "C" void ema_launch_align(const DevIndex *ix, const DevOpts *opt, const uint8_t *bases, const uint32_t *off,
                                 int n_reads, const Intv *intv, const int *n_intv, DevReg *regs, int *n_regs, int *status,
                                 uint8_t *slabs, int *counter, int n_blocks, hipStream_t stream, int *dbg,
                                 unsigned long long *prof);
struct DevAln { int64_t pos; int32_t is_rev, NM, n_cigar; uint32_t cigar_off; };
extern "C" int ema_align_blocks_per_cu();
extern "C" int ema_pair_blocks_per_cu();
extern "C" int ema_final_blocks_per_cu();
extern "C" int ema_seed_blocks_per_cu();
extern "C" size_t ema_pair_slab_bytes();
extern "C" size_t ema_final_slab_bytes();
extern "C" size_t ema_sizeof_aln();
extern "C" void ema_launch_pair(const DevIndex *ix, const DevOpts *opt, int score_delta, int max_rescue, int pes_low,
                                int pes_high, const uint8_t *bases, const uint32_t *off, int n_pairs, DevReg *regs, int *n_regs,
                                int *status, uint8_t *slabs, int *counter, int n_blocks, hipStream_t stream, int *dbg);
extern "C" void ema_launch_final(const DevIndex *ix, const DevOpts *opt, const uint8_t *bases, const uint32_t *off,
                                 int n_reads, const DevReg *regs, const int *n_regs, DevAln *alns, uint32_t *cigars,
                                 int *cig_n, int cig_cap, int *status, uint8_t *slabs, int *counter, int n_blocks,
                                 hipStream_t stream, int *dbg);
extern "C" void ema_launch_pack(int n_reads, const DevReg *regs, const int *n_regs, const DevAln *alns, const uint32_t *cigars,
                                const int *cig_n, int cig_cap, const uint64_t *cand_off, const uint64_t *cig_off,
                                ema_cand_t *cand, uint32_t *cigar_out, int n_blocks, hipStream_t stream);
extern "C" void ema_launch_test_extend(const DevOpts *opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                                       const uint32_t *toff, const int *prm, int n_tasks, int *out, hipStream_t s);
extern "C" void ema_launch_test_global(const DevOpts *opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                                       const uint32_t *toff, const int *prm, int n_tasks, int *out, uint32_t *cig, int cap,
                                       uint8_t *zbuf, size_t z_stride, hipStream_t s);
extern "C" void ema_launch_test_local(const DevOpts *opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                                      const uint32_t *toff, const int *prm, int n_tasks, int *out, uint64_t *bsc,
                                      size_t b_stride, hipStream_t s);

extern "C" void ema_launch_test_dedup(const DevIndex *ix, const DevOpts *opt, DevReg *regs, const int *n_in, int *n_out, int cap,
                                      int n_tasks, DevReg *tmp, uint64_t *keys, hipStream_t s);

namespace {

const unsigned char kNt4[256] = {
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

template <typename T> struct DevBuf {
	T *p = nullptr;
	size_t n = 0;
	hipError_t alloc(size_t count)
	{
		release();
		n = count;
		return hipMalloc((void **)&p, count * sizeof(T) + 256);
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct ema_engine {
	ema_engine_opts opts;
	DevOpts dopts;
	DevIndex dix;
	std::vector<HostContig> contigs;
	int64_t l_pac = 0;
	int device = 0;
	int n_cu = 256;
	hipStream_t stream = nullptr;
	std::string err;
	// index in HBM
	DevBuf<OccSlot> d_occ;
	DevBuf<uint8_t> d_sa, d_pac;
	DevBuf<int64_t> d_ctg;
	// batch
	size_t cap_pairs = 0, n_pairs = 0;
	bool staged = false, ran = false;
	std::vector<uint8_t> h_nt4;
	std::vector<uint32_t> h_off, h_qpack;
	DevBuf<uint8_t> d_bases;
	DevBuf<uint32_t> d_off, d_qpack;   // d_qpack: 24 words per read (2-bit codes + N mask) for K1
	// K1
	int seed_blocks = 0;
	DevBuf<Intv> d_intv, d_lists;
	DevBuf<int> d_n_intv, d_status;
	// K2
	int align_blocks = 0;
	DevBuf<DevReg> d_regs;
	DevBuf<int> d_n_regs, d_counters;
	DevBuf<unsigned long long> d_prof;   // EMA_PHASE_PROFILE=1: per-phase shader-clock totals of K2
	DevBuf<uint8_t> d_slabs;
	// K3 / K4 / pack
	int pair_blocks = 0, final_blocks = 0;
	DevBuf<DevAln> d_alns;
	DevBuf<uint32_t> d_cigars, d_cigar_out;
	DevBuf<int> d_cig_n;
	DevBuf<uint64_t> d_cand_off, d_cig_off;
	DevBuf<ema_cand_t> d_cand;
	size_t cand_cap = 0, cigar_out_cap = 0;
	// development aid (EMA_WATCHDOG_S=<seconds>): host-visible per-wave progress words + a poll after every launch
	int *dbg = nullptr;
	int dbg_slots = 0;
	double watchdog_s = 0;
	hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	ema_engine_timing timing;
};

#define HIPCHK(e, call)                                                                              \
	do {                                                                                             \
		hipError_t rc_ = (call);                                                                     \
		if (rc_ != hipSuccess) {                                                                     \
			(e)->err = std::string(#call) + ": " + hipGetErrorString(rc_);                           \
			return EMA_EDEVICE;                                                                      \
		}                                                                                            \
	} while (0)

extern "C" {

void ema_engine_opts_default(ema_engine_opts *o) { ema_fill_default_opts(o); }

static int engine_alloc_batch(ema_engine *e)
{
	const size_t n_reads = 2 * e->cap_pairs;
	HIPCHK(e, e->d_bases.alloc(n_reads * (size_t)(EMA_MAX_READ + 1)));
	HIPCHK(e, e->d_off.alloc(n_reads + 1));
	HIPCHK(e, e->d_qpack.alloc(n_reads * 24 + 8));
	HIPCHK(e, e->d_intv.alloc(n_reads * (size_t)EMA_INTV_CAP));
	HIPCHK(e, e->d_n_intv.alloc(n_reads));
	HIPCHK(e, e->d_status.alloc(n_reads));
	e->seed_blocks = e->n_cu * ema_seed_blocks_per_cu();      // every resident lane carries one read
	HIPCHK(e, e->d_lists.alloc((size_t)e->seed_blocks * 256 * 2 * EMA_LIST_CAP));
	HIPCHK(e, e->d_regs.alloc(n_reads * (size_t)EMA_REG_CAP));
	HIPCHK(e, e->d_n_regs.alloc(n_reads));
	HIPCHK(e, e->d_counters.alloc(16));
	e->align_blocks = e->n_cu * ema_align_blocks_per_cu();    // one scratch slab per resident wave
	e->pair_blocks = e->n_cu * ema_pair_blocks_per_cu();
	e->final_blocks = e->n_cu * ema_final_blocks_per_cu();
	size_t slab = (size_t)e->align_blocks * 4 * ema_align_slab_bytes();      // the three stages run one after another
	if ((size_t)e->pair_blocks * 4 * ema_pair_slab_bytes() > slab) slab = (size_t)e->pair_blocks * 4 * ema_pair_slab_bytes();
	if ((size_t)e->final_blocks * 4 * ema_final_slab_bytes() > slab) slab = (size_t)e->final_blocks * 4 * ema_final_slab_bytes();
	HIPCHK(e, e->d_slabs.alloc(slab));
	if (ema_sizeof_aln() != sizeof(DevAln)) { e->err = "DevAln layout mismatch"; return EMA_EDEVICE; }
	HIPCHK(e, e->d_alns.alloc(n_reads * (size_t)EMA_REG_CAP));
	HIPCHK(e, e->d_cigars.alloc(n_reads * (size_t)EMA_CIG_CAP));
	HIPCHK(e, e->d_cig_n.alloc(n_reads));
	HIPCHK(e, e->d_cand_off.alloc(n_reads + 1));
	HIPCHK(e, e->d_cig_off.alloc(n_reads + 1));
	return EMA_OK;
}

int ema_engine_open(const char *index_prefix, int device, const ema_engine_opts *opts, ema_engine_t **out)
{
	if (!index_prefix || !out) return EMA_EARG;
	*out = nullptr;
	ema_engine *e = new ema_engine();
	*out = e;   // returned even on failure so that the caller can read the error text
	if (opts) e->opts = *opts; else ema_fill_default_opts(&e->opts);
	e->dopts = ema_make_dev_opts(e->opts);
	e->device = device;
	memset(&e->timing, 0, sizeof(e->timing));
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { e->err = "no HIP device available (the engine has no CPU fallback)"; return EMA_EDEVICE; }
	if (device < 0 || device >= n_dev) { e->err = "device index out of range"; return EMA_EARG; }
	HIPCHK(e, hipSetDevice(device));
	hipDeviceProp_t prop;
	HIPCHK(e, hipGetDeviceProperties(&prop, device));
	e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	HIPCHK(e, hipStreamCreate(&e->stream));
	for (auto &ev : e->ev) HIPCHK(e, hipEventCreate(&ev));

	HostIndex hix;
	std::string msg = host_index_load(index_prefix, hix);
	if (!msg.empty()) { e->err = msg; return EMA_EINDEX; }
	e->contigs = hix.contigs;
	e->l_pac = hix.l_pac;
	HIPCHK(e, e->d_occ.alloc(hix.occ.size()));
	HIPCHK(e, hipMemcpy(e->d_occ.p, hix.occ.data(), hix.occ.size() * sizeof(OccSlot), hipMemcpyHostToDevice));
	HIPCHK(e, e->d_sa.alloc(hix.sa_bytes.size()));
	HIPCHK(e, hipMemcpy(e->d_sa.p, hix.sa_bytes.data(), hix.sa_bytes.size(), hipMemcpyHostToDevice));
	HIPCHK(e, e->d_pac.alloc(hix.pac.size()));
	HIPCHK(e, hipMemcpy(e->d_pac.p, hix.pac.data(), hix.pac.size(), hipMemcpyHostToDevice));
	HIPCHK(e, e->d_ctg.alloc(hix.ctg_off.size()));
	HIPCHK(e, hipMemcpy(e->d_ctg.p, hix.ctg_off.data(), hix.ctg_off.size() * 8, hipMemcpyHostToDevice));
	e->dix = hix.view();
	e->dix.occ = e->d_occ.p; e->dix.sa = e->d_sa.p; e->dix.pac = e->d_pac.p; e->dix.ctg_off = e->d_ctg.p;

	if (getenv("EMA_PHASE_PROFILE")) { HIPCHK(e, e->d_prof.alloc(8)); HIPCHK(e, hipMemset(e->d_prof.p, 0, 64)); }
	if (const char *wd = getenv("EMA_WATCHDOG_S")) {
		e->watchdog_s = atof(wd);
		e->dbg_slots = e->n_cu * 8 * 4 + 64;
		if (!getenv("EMA_WATCHDOG_NOMARK")) {
			HIPCHK(e, hipHostMalloc((void **)&e->dbg, (size_t)e->dbg_slots * 4 * sizeof(int), hipHostMallocDefault));
			memset(e->dbg, 0xff, (size_t)e->dbg_slots * 4 * sizeof(int));
		}
	}
	e->cap_pairs = e->opts.batch_pairs > 0 ? (size_t)e->opts.batch_pairs : (size_t)131072;
	int rc = engine_alloc_batch(e);
	if (rc != EMA_OK) return rc;
	return EMA_OK;
}

void ema_engine_close(ema_engine_t *e)
{
	if (!e) return;
	(void)hipSetDevice(e->device);
	e->d_occ.release(); e->d_sa.release(); e->d_pac.release(); e->d_ctg.release();
	e->d_bases.release(); e->d_off.release(); e->d_qpack.release(); e->d_intv.release(); e->d_lists.release();
	e->d_n_intv.release(); e->d_status.release();
	e->d_regs.release(); e->d_n_regs.release(); e->d_counters.release(); e->d_slabs.release();
	e->d_alns.release(); e->d_cigars.release(); e->d_cigar_out.release(); e->d_cig_n.release();
	e->d_cand_off.release(); e->d_cig_off.release(); e->d_cand.release();
	for (auto &ev : e->ev) if (ev) (void)hipEventDestroy(ev);
	if (e->stream) (void)hipStreamDestroy(e->stream);
	delete e;
}

const char *ema_engine_strerror(const ema_engine_t *e) { return e ? e->err.c_str() : "null engine"; }
int ema_engine_n_contigs(const ema_engine_t *e) { return e ? (int)e->contigs.size() : 0; }
const char *ema_engine_contig_name(const ema_engine_t *e, int rid)
{
	return (e && rid >= 0 && rid < (int)e->contigs.size()) ? e->contigs[rid].name.c_str() : nullptr;
}
int64_t ema_engine_contig_len(const ema_engine_t *e, int rid)
{
	return (e && rid >= 0 && rid < (int)e->contigs.size()) ? e->contigs[rid].len : -1;
}
int64_t ema_engine_contig_offset(const ema_engine_t *e, int rid)
{
	return (e && rid >= 0 && rid < (int)e->contigs.size()) ? e->contigs[rid].offset : -1;
}
int64_t ema_engine_l_pac(const ema_engine_t *e) { return e ? e->l_pac : -1; }
size_t ema_engine_batch_capacity(const ema_engine_t *e) { return e ? e->cap_pairs : 0; }

int ema_engine_stage(ema_engine_t *e, const char *bases, const uint32_t *off, size_t n_pairs)
{
	if (!e || !bases || !off) return EMA_EARG;
	if (n_pairs > e->cap_pairs) { e->err = "batch larger than ema_engine_batch_capacity()"; return EMA_EARG; }
	HIPCHK(e, hipSetDevice(e->device));
	const size_t n_reads = 2 * n_pairs;
	e->h_off.resize(n_reads + 1);
	const uint32_t base0 = off[0];
	for (size_t r = 0; r <= n_reads; ++r) e->h_off[r] = off[r] - base0;
	for (size_t r = 0; r < n_reads; ++r)
		if (off[r + 1] < off[r] || off[r + 1] - off[r] > EMA_MAX_READ) { e->err = "read longer than EMA_MAX_READ"; return EMA_ELIMIT; }
	const size_t total = e->h_off[n_reads];
	e->h_nt4.resize(total + 1);
	const unsigned char *src = (const unsigned char *)bases + base0;
	for (size_t i = 0; i < total; ++i) e->h_nt4[i] = kNt4[src[i]];   // seq_convert, reference src/bwabridge.c:151-157
	e->h_qpack.assign(n_reads * 24 + 8, 0);
	for (size_t r = 0; r < n_reads; ++r) {
		uint32_t *w = e->h_qpack.data() + r * 24;
		const uint8_t *b = e->h_nt4.data() + e->h_off[r];
		const uint32_t len = e->h_off[r + 1] - e->h_off[r];
		for (uint32_t i = 0; i < len; ++i) {
			w[i >> 4] |= (uint32_t)(b[i] & 3) << ((i & 15) << 1);
			if (b[i] > 3) w[16 + (i >> 5)] |= 1u << (i & 31);
		}
	}
	HIPCHK(e, hipMemcpyAsync(e->d_qpack.p, e->h_qpack.data(), (n_reads * 24 + 8) * 4, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipMemcpyAsync(e->d_bases.p, e->h_nt4.data(), total, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipMemcpyAsync(e->d_off.p, e->h_off.data(), (n_reads + 1) * 4, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	e->n_pairs = n_pairs;
	e->staged = true; e->ran = false;
	return EMA_OK;
}

// EMA_WATCHDOG_S: wait for the stream with a deadline; on expiry print where every unfinished wave is and exit
static void watchdog(ema_engine *e, const char *what)
{
	if (e->watchdog_s <= 0) return;
	const int n_poll = (int)(e->watchdog_s * 100);
	for (int t = 0; t < n_poll && hipStreamQuery(e->stream) == hipErrorNotReady; ++t) usleep(10000);
	if (hipStreamQuery(e->stream) == hipErrorNotReady) {
		fprintf(stderr, "%s still running after %.1f s; unfinished waves (slot: unit stage value):\n", what, e->watchdog_s);
		int shown = 0;
		for (int sl = 0; sl < e->dbg_slots && shown < 64; ++sl)
			if (e->dbg && e->dbg[sl * 4] >= 0 && e->dbg[sl * 4 + 1] != 9) { fprintf(stderr, "  %d: %d %d %d 0x%x\n", sl, e->dbg[sl * 4], e->dbg[sl * 4 + 1], e->dbg[sl * 4 + 2], e->dbg[sl * 4 + 3]); ++shown; }
		fflush(stderr);
		_exit(3);
	}
	if (e->dbg) memset(e->dbg, 0xff, (size_t)e->dbg_slots * 4 * sizeof(int));
}

static int run_seed(ema_engine *e)
{
	const int n_reads = (int)(2 * e->n_pairs);
	HIPCHK(e, hipMemsetAsync(e->d_status.p, 0, (size_t)n_reads * 4, e->stream));
	HIPCHK(e, hipMemsetAsync(e->d_counters.p, 0, 16 * 4, e->stream));
	ema_launch_seed(&e->dix, &e->dopts, e->d_qpack.p, e->d_off.p, n_reads, e->d_intv.p, e->d_n_intv.p, e->d_status.p,
	                e->d_lists.p, e->d_counters.p + 3, e->seed_blocks, e->stream);
	HIPCHK(e, hipGetLastError());
	watchdog(e, "ema_k_seed");
	return EMA_OK;
}

static int run_align(ema_engine *e)
{
	const int n_reads = (int)(2 * e->n_pairs);
	ema_launch_align(&e->dix, &e->dopts, e->d_bases.p, e->d_off.p, n_reads, e->d_intv.p, e->d_n_intv.p, e->d_regs.p,
	                 e->d_n_regs.p, e->d_status.p, e->d_slabs.p, e->d_counters.p + 0, e->align_blocks, e->stream, e->dbg, e->d_prof.p);
	HIPCHK(e, hipGetLastError());
	watchdog(e, "ema_k_align");
	return EMA_OK;
}

static int run_pair(ema_engine *e)
{
	ema_launch_pair(&e->dix, &e->dopts, e->opts.score_delta, e->opts.max_rescue, e->opts.pes_low, e->opts.pes_high,
	                e->d_bases.p, e->d_off.p, (int)e->n_pairs, e->d_regs.p, e->d_n_regs.p, e->d_status.p, e->d_slabs.p,
	                e->d_counters.p + 1, e->pair_blocks, e->stream, e->dbg);
	HIPCHK(e, hipGetLastError());
	watchdog(e, "ema_k_pair");
	return EMA_OK;
}

static int run_final(ema_engine *e)
{
	const int n_reads = (int)(2 * e->n_pairs);
	ema_launch_final(&e->dix, &e->dopts, e->d_bases.p, e->d_off.p, n_reads, e->d_regs.p, e->d_n_regs.p, e->d_alns.p,
	                 e->d_cigars.p, e->d_cig_n.p, EMA_CIG_CAP, e->d_status.p, e->d_slabs.p, e->d_counters.p + 2,
	                 e->final_blocks, e->stream, e->dbg);
	HIPCHK(e, hipGetLastError());
	watchdog(e, "ema_k_final");
	return EMA_OK;
}

int ema_engine_run(ema_engine_t *e)
{
	if (!e) return EMA_EARG;
	if (!e->staged) { e->err = "ema_engine_run before ema_engine_stage"; return EMA_ESTATE; }
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipEventRecord(e->ev[0], e->stream));
	int rc = run_seed(e);
	if (rc) return rc;
	HIPCHK(e, hipEventRecord(e->ev[1], e->stream));
	if ((rc = run_align(e))) return rc;
	HIPCHK(e, hipEventRecord(e->ev[2], e->stream));
	if ((rc = run_pair(e))) return rc;
	HIPCHK(e, hipEventRecord(e->ev[3], e->stream));
	if ((rc = run_final(e))) return rc;
	HIPCHK(e, hipEventRecord(e->ev[4], e->stream));
	e->ran = true;
	return EMA_OK;
}

int ema_engine_sync(ema_engine_t *e)
{
	if (!e) return EMA_EARG;
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (e->ran) {
		HIPCHK(e, hipEventElapsedTime(&e->timing.seed_ms, e->ev[0], e->ev[1]));
		HIPCHK(e, hipEventElapsedTime(&e->timing.extend_ms, e->ev[1], e->ev[2]));
		HIPCHK(e, hipEventElapsedTime(&e->timing.rescue_ms, e->ev[2], e->ev[3]));
		HIPCHK(e, hipEventElapsedTime(&e->timing.final_ms, e->ev[3], e->ev[4]));
		HIPCHK(e, hipEventElapsedTime(&e->timing.total_ms, e->ev[0], e->ev[4]));
	}
	return EMA_OK;
}

int ema_engine_last_timing(ema_engine_t *e, ema_engine_timing *t)
{
	if (!e || !t) return EMA_EARG;
	if (e->d_prof.p) {
		unsigned long long h[8];
		if (hipMemcpy(h, e->d_prof.p, 64, hipMemcpyDeviceToHost) == hipSuccess) {
			fprintf(stderr, "K2 phase ticks (idle/fetch, chain, filter, chain2aln-ctl, extend-dp, dedup):");
			for (int i = 0; i < 6; ++i) fprintf(stderr, " %llu", h[i]);
			fprintf(stderr, "\n");
			(void)hipMemset(e->d_prof.p, 0, 64);
		}
	}
	*t = e->timing;
	return EMA_OK;
}

int ema_engine_debug_seeds(ema_engine_t *e, uint64_t **intv, int32_t **n_intv, int32_t *cap_per_read)
{
	if (!e || !intv || !n_intv || !cap_per_read) return EMA_EARG;
	if (!e->staged) { e->err = "ema_engine_debug_seeds before ema_engine_stage"; return EMA_ESTATE; }
	HIPCHK(e, hipSetDevice(e->device));
	int rc = run_seed(e);
	if (rc) return rc;
	HIPCHK(e, hipStreamSynchronize(e->stream));
	const size_t n_reads = 2 * e->n_pairs;
	*intv = (uint64_t *)malloc(n_reads * (size_t)EMA_INTV_CAP * sizeof(Intv) + 8);
	*n_intv = (int32_t *)malloc(n_reads * 4 + 8);
	HIPCHK(e, hipMemcpy(*intv, e->d_intv.p, n_reads * (size_t)EMA_INTV_CAP * sizeof(Intv), hipMemcpyDeviceToHost));
	HIPCHK(e, hipMemcpy(*n_intv, e->d_n_intv.p, n_reads * 4, hipMemcpyDeviceToHost));
	// K1 emits in discovery order; present the lists as mem_collect_intv leaves them: ordered by (start, end)
	for (size_t r = 0; r < n_reads; ++r) {
		Intv *a = (Intv *)(*intv) + r * EMA_INTV_CAP;
		std::stable_sort(a, a + (*n_intv)[r], [](const Intv &x, const Intv &y) { return x.info < y.info; });
	}
	*cap_per_read = EMA_INTV_CAP;
	return EMA_OK;
}

int ema_engine_debug_regions(ema_engine_t *e, void **regs, int32_t **n_regs, int32_t **status, int32_t *cap_per_read,
                             int32_t *reg_bytes)
{
	if (!e || !regs || !n_regs || !status || !cap_per_read || !reg_bytes) return EMA_EARG;
	if (!e->staged) { e->err = "ema_engine_debug_regions before ema_engine_stage"; return EMA_ESTATE; }
	HIPCHK(e, hipSetDevice(e->device));
	int rc = run_seed(e);
	if (rc) return rc;
	if ((rc = run_align(e))) return rc;
	HIPCHK(e, hipStreamSynchronize(e->stream));
	const size_t n_reads = 2 * e->n_pairs;
	*regs = malloc(n_reads * (size_t)EMA_REG_CAP * sizeof(DevReg) + 8);
	*n_regs = (int32_t *)malloc(n_reads * 4 + 8);
	*status = (int32_t *)malloc(n_reads * 4 + 8);
	HIPCHK(e, hipMemcpy(*regs, e->d_regs.p, n_reads * (size_t)EMA_REG_CAP * sizeof(DevReg), hipMemcpyDeviceToHost));
	HIPCHK(e, hipMemcpy(*n_regs, e->d_n_regs.p, n_reads * 4, hipMemcpyDeviceToHost));
	HIPCHK(e, hipMemcpy(*status, e->d_status.p, n_reads * 4, hipMemcpyDeviceToHost));
	*cap_per_read = EMA_REG_CAP;
	*reg_bytes = (int32_t)sizeof(DevReg);
	return EMA_OK;
}

int ema_engine_debug_dedup(ema_engine_t *e, void *regs, const int32_t *n_in, int32_t *n_out, int cap, int n_tasks)
{
	if (!e || !regs || !n_in || !n_out || cap <= 0 || n_tasks <= 0) return EMA_EARG;
	HIPCHK(e, hipSetDevice(e->device));
	DevBuf<DevReg> dr, dt;
	DevBuf<uint64_t> dk;
	DevBuf<int> dn, dm;
	const size_t tot = (size_t)n_tasks * cap;
	HIPCHK(e, dr.alloc(tot)); HIPCHK(e, dt.alloc(tot)); HIPCHK(e, dk.alloc(tot)); HIPCHK(e, dn.alloc(n_tasks)); HIPCHK(e, dm.alloc(n_tasks));
	HIPCHK(e, hipMemcpy(dr.p, regs, tot * sizeof(DevReg), hipMemcpyHostToDevice));
	HIPCHK(e, hipMemcpy(dn.p, n_in, (size_t)n_tasks * 4, hipMemcpyHostToDevice));
	ema_launch_test_dedup(&e->dix, &e->dopts, dr.p, dn.p, dm.p, cap, n_tasks, dt.p, dk.p, e->stream);
	HIPCHK(e, hipGetLastError());
	HIPCHK(e, hipStreamSynchronize(e->stream));
	HIPCHK(e, hipMemcpy(regs, dr.p, tot * sizeof(DevReg), hipMemcpyDeviceToHost));
	HIPCHK(e, hipMemcpy(n_out, dm.p, (size_t)n_tasks * 4, hipMemcpyDeviceToHost));
	dr.release(); dt.release(); dk.release(); dn.release(); dm.release();
	return EMA_OK;
}

int ema_engine_debug_dp(ema_engine_t *e, int kind, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                        const uint32_t *toff, const int32_t *prm, int n_tasks, int32_t *out, uint32_t *cigar, int cigar_cap)
{
	if (!e || !qbuf || !qoff || !tbuf || !toff || !prm || !out || n_tasks <= 0 || kind < 0 || kind > 2) return EMA_EARG;
	HIPCHK(e, hipSetDevice(e->device));
	const int n_prm = kind == 0 ? 4 : kind == 1 ? 1 : 3, n_out = kind == 0 ? 6 : kind == 1 ? 2 : 5;
	DevBuf<uint8_t> dq, dt, dz;
	DevBuf<uint32_t> dqo, dto, dc;
	DevBuf<int> dp, dout;
	DevBuf<uint64_t> db;
	const size_t z_stride = 256 * 1024, b_stride = 2048;
	HIPCHK(e, dq.alloc(qoff[n_tasks] + 1)); HIPCHK(e, dt.alloc(toff[n_tasks] + 1));
	HIPCHK(e, dqo.alloc(n_tasks + 1)); HIPCHK(e, dto.alloc(n_tasks + 1));
	HIPCHK(e, dp.alloc((size_t)n_tasks * n_prm)); HIPCHK(e, dout.alloc((size_t)n_tasks * n_out));
	HIPCHK(e, hipMemcpy(dq.p, qbuf, qoff[n_tasks], hipMemcpyHostToDevice));
	HIPCHK(e, hipMemcpy(dt.p, tbuf, toff[n_tasks], hipMemcpyHostToDevice));
	HIPCHK(e, hipMemcpy(dqo.p, qoff, (n_tasks + 1) * 4, hipMemcpyHostToDevice));
	HIPCHK(e, hipMemcpy(dto.p, toff, (n_tasks + 1) * 4, hipMemcpyHostToDevice));
	HIPCHK(e, hipMemcpy(dp.p, prm, (size_t)n_tasks * n_prm * 4, hipMemcpyHostToDevice));
	HIPCHK(e, hipEventRecord(e->ev[5], e->stream));
	if (kind == 0) ema_launch_test_extend(&e->dopts, dq.p, dqo.p, dt.p, dto.p, dp.p, n_tasks, dout.p, e->stream);
	else if (kind == 1) {
		if (!cigar || cigar_cap <= 0) return EMA_EARG;
		HIPCHK(e, dz.alloc((size_t)n_tasks * z_stride));
		HIPCHK(e, dc.alloc((size_t)n_tasks * cigar_cap));
		ema_launch_test_global(&e->dopts, dq.p, dqo.p, dt.p, dto.p, dp.p, n_tasks, dout.p, dc.p, cigar_cap, dz.p, z_stride, e->stream);
	} else {
		HIPCHK(e, db.alloc((size_t)n_tasks * b_stride));
		ema_launch_test_local(&e->dopts, dq.p, dqo.p, dt.p, dto.p, dp.p, n_tasks, dout.p, db.p, b_stride, e->stream);
	}
	HIPCHK(e, hipGetLastError());
	HIPCHK(e, hipEventRecord(e->ev[6], e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (getenv("EMA_DP_TIMING")) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, e->ev[5], e->ev[6]);
		fprintf(stderr, "debug_dp kind %d: %d tasks in %.3f ms\n", kind, n_tasks, ms);
	}
	HIPCHK(e, hipMemcpy(out, dout.p, (size_t)n_tasks * n_out * 4, hipMemcpyDeviceToHost));
	if (kind == 1) HIPCHK(e, hipMemcpy(cigar, dc.p, (size_t)n_tasks * cigar_cap * 4, hipMemcpyDeviceToHost));
	dq.release(); dt.release(); dz.release(); dqo.release(); dto.release(); dc.release(); dp.release(); dout.release(); db.release();
	return EMA_OK;
}

int ema_engine_fetch(ema_engine_t *e, ema_batch_out **out)
{
	if (!e || !out) return EMA_EARG;
	*out = nullptr;
	if (!e->ran) { e->err = "ema_engine_fetch before ema_engine_run"; return EMA_ESTATE; }
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	const size_t n_reads = 2 * e->n_pairs;
	std::vector<int> n_regs(n_reads), cig_n(n_reads);
	HIPCHK(e, hipMemcpy(n_regs.data(), e->d_n_regs.p, n_reads * 4, hipMemcpyDeviceToHost));
	HIPCHK(e, hipMemcpy(cig_n.data(), e->d_cig_n.p, n_reads * 4, hipMemcpyDeviceToHost));
	ema_batch_out *o = (ema_batch_out *)calloc(1, sizeof(ema_batch_out));
	o->n_pairs = e->n_pairs;
	o->cand_off = (uint64_t *)malloc((n_reads + 1) * 8);
	std::vector<uint64_t> cig_off(n_reads + 1);
	o->cand_off[0] = 0; cig_off[0] = 0;
	for (size_t r = 0; r < n_reads; ++r) {
		o->cand_off[r + 1] = o->cand_off[r] + (uint64_t)n_regs[r];
		cig_off[r + 1] = cig_off[r] + (uint64_t)cig_n[r];
	}
	const size_t n_cand = o->cand_off[n_reads], n_cig = cig_off[n_reads];
	o->n_cigar = n_cig;
	o->cand = (ema_cand_t *)malloc((n_cand + 1) * sizeof(ema_cand_t));
	o->cigar = (uint32_t *)malloc((n_cig + 1) * 4);
	o->status = (int32_t *)malloc((n_reads + 1) * 4);
	*out = o;
	if (n_cand + 1 > e->cand_cap) { e->cand_cap = (n_cand + 1) * 5 / 4 + 1024; HIPCHK(e, e->d_cand.alloc(e->cand_cap)); }
	if (n_cig + 1 > e->cigar_out_cap) { e->cigar_out_cap = (n_cig + 1) * 5 / 4 + 1024; HIPCHK(e, e->d_cigar_out.alloc(e->cigar_out_cap)); }
	HIPCHK(e, hipMemcpyAsync(e->d_cand_off.p, o->cand_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipMemcpyAsync(e->d_cig_off.p, cig_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, e->stream));
	ema_launch_pack((int)n_reads, e->d_regs.p, e->d_n_regs.p, e->d_alns.p, e->d_cigars.p, e->d_cig_n.p, EMA_CIG_CAP,
	                e->d_cand_off.p, e->d_cig_off.p, e->d_cand.p, e->d_cigar_out.p, e->n_cu * 4, e->stream);
	HIPCHK(e, hipGetLastError());
	HIPCHK(e, hipMemcpyAsync(o->cand, e->d_cand.p, n_cand * sizeof(ema_cand_t), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipMemcpyAsync(o->cigar, e->d_cigar_out.p, n_cig * 4, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipMemcpyAsync(o->status, e->d_status.p, n_reads * 4, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	for (size_t r = 0; r < n_reads; ++r)
		if (o->status[r]) { e->err = "a read exceeded an engine capacity; see ema_batch_out.status"; return EMA_ELIMIT; }
	return EMA_OK;
}

int ema_engine_align_pairs(ema_engine_t *e, const char *bases, const uint32_t *off, size_t n_pairs, ema_batch_out **out)
{
	int rc = ema_engine_stage(e, bases, off, n_pairs);
	if (rc) return rc;
	if ((rc = ema_engine_run(e))) return rc;
	if ((rc = ema_engine_sync(e))) return rc;
	return ema_engine_fetch(e, out);
}

void ema_batch_free(ema_batch_out *out)
{
	if (!out) return;
	free(out->cand_off); free(out->cand); free(out->cigar); free(out->status);
	free(out);
}

}  // extern "C"
