// ema_amd/csrc/k_dp_test.hip -- stand-alone launches of the three wave DPs, one task per wavefront.
// Exposed through the C ABI's debug entry points so that each DP can be parity-checked against the
// oracle in isolation (tests/test_gpu_dp.py); the pipeline kernels call the same device functions.
#include <hip/hip_runtime.h>
#include "dev_dp.hpp"

// task t: query = qbuf[qoff[t]..qoff[t+1]), target = tbuf[toff[t]..toff[t+1]); prm[t*4..] = {w, end_bonus, zdrop, h0}
__global__ void __launch_bounds__(64)
ema_k_test_extend(DevOpts opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf, const uint32_t *toff,
                  const int *prm, int n_tasks, int *out)
{
	const int t = blockIdx.x;
	if (t >= n_tasks) return;
	EmaSeq q{qbuf + qoff[t], 1}, tg{tbuf + toff[t], 1};
	const EmaExtRes r = ema_wave_extend(opt, (int)(qoff[t + 1] - qoff[t]), q, (int)(toff[t + 1] - toff[t]), tg, prm[t * 4],
	                                    prm[t * 4 + 1], prm[t * 4 + 2], prm[t * 4 + 3]);
	if (ema_lane() == 0) {
		int *o = out + t * 6;
		o[0] = r.score; o[1] = r.qle; o[2] = r.tle; o[3] = r.gtle; o[4] = r.gscore; o[5] = r.max_off;
	}
}

// prm[t] = band w.  out[t*2] = score, out[t*2+1] = n_cigar; cigar ops at cig[t*cap .. ) (forward order, left-packed)
__global__ void __launch_bounds__(64)
ema_k_test_global(DevOpts opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf, const uint32_t *toff,
                  const int *prm, int n_tasks, int *out, uint32_t *cig, int cap, uint8_t *zbuf, size_t z_stride)
{
	const int t = blockIdx.x;
	if (t >= n_tasks) return;
	const int qlen = (int)(qoff[t + 1] - qoff[t]), tlen = (int)(toff[t + 1] - toff[t]);
	EmaSeq q{qbuf + qoff[t], 1}, tg{tbuf + toff[t], 1};
	uint8_t *z = zbuf + (size_t)t * z_stride;
	const int score = ema_wave_global(opt, qlen, q, tlen, tg, prm[t], z);
	uint32_t *c = cig + (size_t)t * cap;
	ema_wave_sync();
	const int first = ema_traceback((const uint8_t *)z, qlen, tlen, prm[t], c, cap);
	ema_wave_sync();
	const int n = first < 0 ? -1 : cap - first;
	for (int k = 0; k < n; ++k) { const uint32_t v = c[first + k]; c[k] = v; }   // wave-uniform left-pack (first >= k)
	if (ema_lane() == 0) { out[t * 2] = score; out[t * 2 + 1] = n; }
}

// prm[t*3..] = {p (16|8), minsc, endsc}.  out[t*5..] = score, te, qe, score2, te2
__global__ void __launch_bounds__(64)
ema_k_test_local(DevOpts opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf, const uint32_t *toff,
                 const int *prm, int n_tasks, int *out, uint64_t *bsc, size_t b_stride)
{
	const int t = blockIdx.x;
	if (t >= n_tasks) return;
	EmaSeq q{qbuf + qoff[t], 1}, tg{tbuf + toff[t], 1};
	const EmaLocalRes r = ema_wave_local(opt, (int)(qoff[t + 1] - qoff[t]), prm[t * 3], q, (int)(toff[t + 1] - toff[t]), tg,
	                                     prm[t * 3 + 1], prm[t * 3 + 2], bsc + (size_t)t * b_stride);
	if (ema_lane() == 0) {
		int *o = out + t * 5;
		o[0] = r.score; o[1] = r.te; o[2] = r.qe; o[3] = r.score2; o[4] = r.te2;
	}
}

extern "C" void ema_launch_test_extend(const DevOpts *opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                                       const uint32_t *toff, const int *prm, int n_tasks, int *out, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_extend, dim3(n_tasks), dim3(64), 0, s, *opt, qbuf, qoff, tbuf, toff, prm, n_tasks, out);
}
extern "C" void ema_launch_test_global(const DevOpts *opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                                       const uint32_t *toff, const int *prm, int n_tasks, int *out, uint32_t *cig, int cap,
                                       uint8_t *zbuf, size_t z_stride, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_global, dim3(n_tasks), dim3(64), 0, s, *opt, qbuf, qoff, tbuf, toff, prm, n_tasks, out, cig,
	                   cap, zbuf, z_stride);
}
extern "C" void ema_launch_test_local(const DevOpts *opt, const uint8_t *qbuf, const uint32_t *qoff, const uint8_t *tbuf,
                                      const uint32_t *toff, const int *prm, int n_tasks, int *out, uint64_t *bsc,
                                      size_t b_stride, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_local, dim3(n_tasks), dim3(64), 0, s, *opt, qbuf, qoff, tbuf, toff, prm, n_tasks, out, bsc,
	                   b_stride);
}

// bns_intv2rid and bns_pos2rid on the coarse contig table (dev_ref.hpp), one lane per query
#include "dev_ref.hpp"
__global__ void ema_k_test_contigs(DevIndex ix, const int64_t *rb, const int64_t *re, int n, int *out)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= n) return;
	int is_rev;
	out[2 * i] = ema_intv2rid(ix, rb[i], re[i]);
	out[2 * i + 1] = ema_pos2rid(ix, ema_depos(ix, rb[i], is_rev));
}
extern "C" void ema_launch_test_contigs(const DevIndex *ix, const int64_t *rb, const int64_t *re, int n, int *out, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_contigs, dim3((n + 255) / 256), dim3(256), 0, s, *ix, rb, re, n, out);
}

// mem_sort_dedup_patch without patching (the form mem_matesw uses), one task per wavefront:
// task t owns regs[t*cap .. t*cap + n_in[t]); result in place, n_out[t] = surviving regions.
#include "dev_regions.hpp"
__global__ void __launch_bounds__(64)
ema_k_test_dedup(DevIndex ix, DevOpts opt, DevReg *regs, const int *n_in, int *n_out, int cap, int n_tasks, DevReg *tmp,
                 uint64_t *keys)
{
	__shared__ int stack[3 * 70];
	__shared__ uint8_t rseq[EMA_RSEQ_CAP];
	const int t = blockIdx.x;
	if (t >= n_tasks) return;
	EmaRegWork wk;
	wk.a = regs + (size_t)t * cap; wk.tmp = tmp + (size_t)t * cap; wk.keys = keys + (size_t)t * cap; wk.stack = stack; wk.rseq = rseq;
	int status = 0;
	const int n = ema_sort_dedup_patch(ix, opt, nullptr, n_in[t], wk, status);
	if (ema_lane() == 0) n_out[t] = n;
}
extern "C" void ema_launch_test_dedup(const DevIndex *ix, const DevOpts *opt, DevReg *regs, const int *n_in, int *n_out, int cap,
                                      int n_tasks, DevReg *tmp, uint64_t *keys, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_dedup, dim3(n_tasks), dim3(64), 0, s, *ix, *opt, regs, n_in, n_out, cap, n_tasks, tmp, keys);
}

// ema_introsort_wave against ema_introsort (dev_sort.hpp): task t sorts wave_io[t*cap .. +n[t]) with the wavefront's form and
// seq_io[t*cap .. +n[t]) (the same keys) with the single-lane form; the comparison is the chain filter's (high words, descending)
// when by_weight, else plain ascending.  The caller compares the two arrays.
__global__ void __launch_bounds__(64)
ema_k_test_sort(uint64_t *wave_io, uint64_t *seq_io, const int *n, int cap, int n_tasks, int by_weight)
{
	__shared__ int stack[3 * 70];
	__shared__ uint16_t scratch[512];
	__shared__ uint64_t keys[256];
	const int t = blockIdx.x;
	if (t >= n_tasks) return;
	const int lane = (int)ema_lane(), m = n[t];
	for (int i = lane; i < m; i += EMA_WAVE) keys[i] = wave_io[(size_t)t * cap + i];
	ema_wave_sync();
	if (by_weight) ema_introsort_wave(keys, m, [](uint64_t x, uint64_t y) { return (x >> 32) > (y >> 32); }, stack, scratch);
	else ema_introsort_wave(keys, m, [](uint64_t x, uint64_t y) { return x < y; }, stack, scratch);
	ema_wave_sync();
	for (int i = lane; i < m; i += EMA_WAVE) wave_io[(size_t)t * cap + i] = keys[i];
	ema_wave_sync();
	if (lane == 0) {
		if (by_weight) ema_introsort(seq_io + (size_t)t * cap, m, [](uint64_t x, uint64_t y) { return (x >> 32) > (y >> 32); }, stack);
		else ema_introsort(seq_io + (size_t)t * cap, m, [](uint64_t x, uint64_t y) { return x < y; }, stack);
	}
}
extern "C" void ema_launch_test_sort(uint64_t *wave_io, uint64_t *seq_io, const int *n, int cap, int n_tasks, int by_weight, hipStream_t s)
{
	hipLaunchKernelGGL(ema_k_test_sort, dim3(n_tasks), dim3(64), 0, s, wave_io, seq_io, n, cap, n_tasks, by_weight);
}
