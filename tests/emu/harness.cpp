// tests/emu/harness.cpp -- runs the HIP kernels of ema_amd/csrc through the host SIMT
// interpreter (simt_emu.h).  TEST INFRASTRUCTURE for machines without a GPU; never shipped.
#include <hip/hip_runtime.h>   // resolves to tests/emu/hip/hip_runtime.h
#include <string>
#include <vector>
#include "host_index.h"
#include "opts.h"

// kernels + launchers, compiled as plain C++
#include "k_seed.hip"
#include "k_dp_test.hip"
#include "k_align.hip"
#include "k_pair.hip"
#include "k_final.hip"

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

extern "C" {

void *emu_index_load(const char *prefix, char *err, int errlen)
{
	HostIndex *ix = new HostIndex();
	std::string e = host_index_load(prefix, *ix);
	if (!e.empty()) { snprintf(err, errlen, "%s", e.c_str()); delete ix; return nullptr; }
	return ix;
}
void emu_index_free(void *h) { delete (HostIndex *)h; }

// bases: nt4 codes.  intv: n_reads*EMA_INTV_CAP*4 u64.  Returns the interval capacity per read.
int emu_seed(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, uint64_t *intv, int *n_intv, int *status,
             int n_blocks)
{
	HostIndex *ix = (HostIndex *)h;
	ema_engine_opts o; ema_fill_default_opts(&o);
	DevOpts d = ema_make_dev_opts(o);
	DevIndex di = ix->view();
	std::vector<Intv> lists((size_t)n_blocks * 256 * 2 * EMA_LIST_CAP);
	int seed_counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	ema_launch_seed(&di, &d, qp.data(), off, n_reads, (Intv *)intv, n_intv, status, lists.data(), &seed_counter, n_blocks, nullptr, nullptr);
	for (int r = 0; r < n_reads; ++r) {
		Intv *a = (Intv *)intv + (size_t)r * EMA_INTV_CAP;
		std::stable_sort(a, a + n_intv[r], [](const Intv &x, const Intv &y) { return x.info < y.info; });
	}
	return EMA_INTV_CAP;
}

static DevOpts default_dev_opts() { ema_engine_opts o; ema_fill_default_opts(&o); return ema_make_dev_opts(o); }

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

// K1 + K2 on host memory.  regs: n_reads * EMA_REG_CAP DevReg (80 bytes each); returns EMA_REG_CAP
int emu_align(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, void *regs, int *n_regs, int *status,
              int n_blocks)
{
	HostIndex *ix = (HostIndex *)h;
	DevOpts d = default_dev_opts();
	DevIndex di = ix->view();
	std::vector<Intv> intv((size_t)n_reads * EMA_INTV_CAP);
	std::vector<int> n_intv(n_reads);
	std::vector<Intv> lists((size_t)1 * 256 * 2 * EMA_LIST_CAP);
	for (int i = 0; i < n_reads; ++i) status[i] = 0;
	int seed_counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	ema_launch_seed(&di, &d, qp.data(), off, n_reads, intv.data(), n_intv.data(), status, lists.data(), &seed_counter, 1, nullptr, nullptr);
	std::vector<uint8_t> slabs((size_t)n_blocks * 4 * ema_align_slab_bytes());
	int counter = 0;
	ema_launch_align(&di, &d, bases, off, n_reads, intv.data(), n_intv.data(), (DevReg *)regs, n_regs, status, slabs.data(),
	                 &counter, n_blocks, nullptr, nullptr, nullptr);
	return EMA_REG_CAP;
}
int emu_sizeof_reg() { return (int)sizeof(DevReg); }

// the whole pipeline K1..K4 on host memory (n_reads even: pairs)
int emu_pipeline(void *h, const uint8_t *bases, const uint32_t *off, int n_reads, void *regs, int *n_regs, void *alns,
                 uint32_t *cigars, int *cig_n, int *status, int upto)
{
	HostIndex *ix = (HostIndex *)h;
	ema_engine_opts eo; ema_fill_default_opts(&eo);
	DevOpts d = ema_make_dev_opts(eo);
	DevIndex di = ix->view();
	std::vector<Intv> intv((size_t)n_reads * EMA_INTV_CAP);
	std::vector<int> n_intv(n_reads);
	std::vector<Intv> lists((size_t)1 * 256 * 2 * EMA_LIST_CAP);
	for (int i = 0; i < n_reads; ++i) status[i] = 0;
	int seed_counter = 0;
	std::vector<uint32_t> qp = pack_reads(bases, off, n_reads);
	ema_launch_seed(&di, &d, qp.data(), off, n_reads, intv.data(), n_intv.data(), status, lists.data(), &seed_counter, 1, nullptr, nullptr);
	size_t slab = ema_align_slab_bytes();
	if (ema_pair_slab_bytes() > slab) slab = ema_pair_slab_bytes();
	if (ema_final_slab_bytes() > slab) slab = ema_final_slab_bytes();
	std::vector<uint8_t> slabs((size_t)4 * slab);
	int counter[3] = {0, 0, 0};
	ema_launch_align(&di, &d, bases, off, n_reads, intv.data(), n_intv.data(), (DevReg *)regs, n_regs, status, slabs.data(),
	                 &counter[0], 1, nullptr, nullptr, nullptr);
	if (upto >= 3)
		ema_launch_pair(&di, &d, eo.score_delta, eo.max_rescue, eo.pes_low, eo.pes_high, bases, off, n_reads / 2, (DevReg *)regs,
		                n_regs, status, slabs.data(), &counter[1], 1, nullptr, nullptr);
	if (upto >= 4)
		ema_launch_final(&di, &d, bases, off, n_reads, (DevReg *)regs, n_regs, (DevAln *)alns, cigars, cig_n, EMA_CIG_CAP, status,
		                 slabs.data(), &counter[2], 1, nullptr, nullptr);
	return EMA_CIG_CAP;
}
}
