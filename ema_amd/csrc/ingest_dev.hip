// ema_amd/csrc/ingest_dev.hip -- the bucket reader on the device (include/ema_ingest.h: ema_bucket_read_device; SURVEY 8f rank 2:
// "parallel parse + (GPU) radix sort by barcode").
//
// read_special_fastq (reference src/align.c:759-806: count_lines, fgets + strcpy per line, qsort of the lines by
// strncmp(.., BC_LEN), six copy_until_space calls) for a bucket of 10x-style barcodes (ACGT / acgt, at most 21 bases) or haplotag ones
// (AxxCxxBxxDxx: the twelve bytes are the sort key as they are, 96 bits in two stable sorts):
// the file goes into a page-locked buffer with pread() and up as it is; then, all on the device,
//   1. newline positions (one count, one rocPRIM select) -> the line table;
//   2. ema_k_ing_parse: one lane per line -- the six fields as copy_until_space finds them (the same scan as host_ingest.cpp's
//      next_field: bytes up to the next whitespace or NUL, then one byte skipped), the reader's checks in the reader's order,
//      the barcode as a sort code of 3 bits per base in the order strncmp gives the bytes (A C G T a c g t);
//   3. a stable radix sort of (code, line): equal barcodes keep file order, the order the host reader's sort gives (and the
//      reference's for distinct lines -- SURVEY 0.5: qsort itself is not stable);
//   4. prefix sums of the read and name lengths in sorted order -> off[], id_off[];
//   5. ema_k_ing_gather: bases, qualities, names to their sorted places, the encoded barcodes (encode_bc, src/util.c:41-76).
// The bucket that comes back has bc / off / id_off / ids / group_off on the host (the cloud stage's inputs) and bases / quals
// ON THE DEVICE ONLY (bucket->dev): the engine stages them device-to-device (ema_engine_stage_async_dev) and the SAM formatter
// (k_sam.hip) reads them where they are -- a read's 300 bytes never cross the host's caches again.
// Anything irregular -- a line the checks refuse, a NUL or a line of 5000 bytes, barcodes beyond 21 bases, a file of 4 GB -- goes to
// the host reader (ema_bucket_read), which owns the error messages and the odd cases: the result is the host reader's either way.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include "ema_ingest.h"
#include "dev_bucket.h"
#include "host_cpuacct.h"
#include "host_pool.h"
#include "ingest_kernels.hpp"
#include <chrono>
#include <time.h>

const char *ema_tuning_get(const char *key);      // engine.hip

namespace {

// ---- memory kept from bucket to bucket: hipFree waits for the whole device (the engine's streams included), so blocks are reused -------------
struct Block { void *p = nullptr; size_t cap = 0; int device = -1; };      // (device: where a device block lives; page-locked ones serve any)
struct BlockPool {
	std::mutex mu;
	std::vector<Block> idle;
	bool pinned;
	explicit BlockPool(bool pin) : pinned(pin) {}
	hipError_t take(size_t bytes, Block &out, int device)
	{
		Block evicted;      // (freed AFTER the lock is dropped: hipFree waits for the whole device, and the SAM writer's ema_bucket_dev_release waits for this mutex -- ADVICE r05)
		{
			std::lock_guard<std::mutex> lk(mu);
			int best = -1;
			for (size_t k = 0; k < idle.size(); ++k)
				if (idle[k].cap >= bytes && (pinned || idle[k].device == device) && (best < 0 || idle[k].cap < idle[(size_t)best].cap)) best = (int)k;
			if (best >= 0) { out = idle[(size_t)best]; idle.erase(idle.begin() + best); return hipSuccess; }
			if (idle.size() >= 6) {      // none fits and the shelf is full: the smallest one makes room
				size_t small = 0;
				for (size_t k = 1; k < idle.size(); ++k) if (idle[k].cap < idle[small].cap) small = k;
				evicted = idle[small];
				idle.erase(idle.begin() + (long)small);
			}
		}
		if (evicted.p) { if (pinned) (void)hipHostFree(evicted.p); else (void)hipFree(evicted.p); }
		out.cap = bytes + bytes / 8 + 65536;
		out.device = device;
		const hipError_t rc = pinned ? hipHostMalloc(&out.p, out.cap, hipHostMallocDefault) : hipMalloc(&out.p, out.cap);
		if (rc != hipSuccess) { out.p = nullptr; out.cap = 0; }
		return rc;
	}
	void give(Block b)
	{
		if (!b.p) return;
		std::lock_guard<std::mutex> lk(mu);
		idle.push_back(b);
	}
};
BlockPool &dev_pool() { static BlockPool p(false); return p; }
BlockPool &pin_pool() { static BlockPool p(true); return p; }

thread_local std::string g_dev_err;

struct Carve {      // consecutive 256-byte aligned pieces of one block
	char *base; size_t at = 0;
	explicit Carve(void *p) : base((char *)p) {}
	template <typename T> T *take(size_t n) { T *r = (T *)(base + at); at += (n * sizeof(T) + 255) & ~(size_t)255; return r; }
	static size_t need(std::initializer_list<size_t> bytes) { size_t t = 0; for (size_t b : bytes) t += (b + 255) & ~(size_t)255; return t; }
};

}  // namespace

struct ema_bucket_dev_impl {
	ema_bucket_dev pub;
	Block block;
};

extern "C" {

const char *ema_bucket_dev_last_error(void) { return g_dev_err.c_str(); }

void ema_bucket_dev_release(void *dev)
{
	if (!dev) return;
	ema_bucket_dev_impl *d = (ema_bucket_dev_impl *)dev;
	dev_pool().give(d->block);
	delete d;
}

const ema_bucket_dev *ema_bucket_dev_view(const ema_bucket *bk) { return bk && bk->dev ? &((const ema_bucket_dev_impl *)bk->dev)->pub : nullptr; }

int ema_bucket_dev_fetch(const ema_bucket *bk, char *bases, char *quals)
{
	const ema_bucket_dev *d = ema_bucket_dev_view(bk);
	if (!d || !bases || !quals) return EMA_EARG;
	if (hipSetDevice(d->device) != hipSuccess) return EMA_EIO;
	if (d->n_bases && (hipMemcpy(bases, d->bases, d->n_bases, hipMemcpyDeviceToHost) != hipSuccess ||
	                   hipMemcpy(quals, d->quals, d->n_bases, hipMemcpyDeviceToHost) != hipSuccess)) return EMA_EIO;
	return 0;
}

int ema_bucket_read_device(const char *path, int bc_len, int is_haplotag, int max_read_len, int device, ema_bucket **out)
{
	if (!out) return EMA_EARG;
	*out = nullptr;
	g_dev_err.clear();
	if (!path || bc_len < 1 || bc_len > 32 || max_read_len < 1 || max_read_len > 4096 || (is_haplotag && bc_len != 12))
		return ema_bucket_read(path, bc_len, is_haplotag, max_read_len, out);      // (the host reader words the refusal)
	if (!is_haplotag && bc_len > 21) return ema_bucket_read(path, bc_len, is_haplotag, max_read_len, out);      // (no platform has such barcodes)
	EMA_CPU(EMA_CPU_READER);
	const int fd = open(path, O_RDONLY);
	struct stat sb;
	if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size == 0 || (uint64_t)sb.st_size >= 0xfff00000ull) {
		if (fd >= 0) close(fd);
		return ema_bucket_read(path, bc_len, is_haplotag, max_read_len, out);
	}
	const size_t len = (size_t)sb.st_size;
	const bool prof = ema_tuning_get("ingest_prof") != nullptr;      // phase times on stderr (tools/ingest_rate.py --device)
	auto t_last = std::chrono::steady_clock::now();
	auto cpu_now = [] { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6; };
	double c_last = prof ? cpu_now() : 0.0;
	auto lap = [&](const char *what) {      // wall milliseconds of the phase, and the CPU milliseconds THIS thread spent in it (the pool's threads not counted)
		if (!prof) return;
		const auto t = std::chrono::steady_clock::now();
		const double c = cpu_now();
		fprintf(stderr, "[ingest_dev] %-22s %7.2f ms wall %7.2f ms cpu\n", what, std::chrono::duration<double, std::milli>(t - t_last).count(), c - c_last);
		t_last = t; c_last = c;
	};
	Block pin, txt, work, keep;
	ema_bucket *o = nullptr;
	hipStream_t st = nullptr;
	int rc = -1000;      // -1000: not decided; -1001: irregular, the host reader takes the bucket
	auto hip_fail = [&](hipError_t e, const char *what) { g_dev_err = std::string(what) + ": " + hipGetErrorString(e); rc = EMA_EIO; };
#define ING(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { hip_fail(e_, #call); goto done; } } while (0)
	{
		ING(hipSetDevice(device));
		ING(pin_pool().take(len + 64, pin, device));
		{   // the file, on the host's threads, straight into page-locked memory
			std::atomic<int> bad{0};
			char *buf = (char *)pin.p;
			const size_t t = std::max<size_t>(1, std::min<size_t>((size_t)EmaPool::get().size(), len >> 22));
			const size_t per = (len + t - 1) / t;
			EmaPool::get().run(t, [&](size_t k) {
				EMA_CPU(EMA_CPU_READER);
				size_t at = std::min(len, k * per);
				const size_t hi = std::min(len, at + per);
				while (at < hi) {
					const ssize_t got = pread(fd, buf + at, hi - at, (off_t)at);
					if (got < 0 && errno == EINTR) continue;
					if (got <= 0) { bad.store(1); return; }
					at += (size_t)got;
				}
			});
			if (bad.load()) { rc = -1001; goto done; }      // (the host reader reports the I/O error)
			memset(buf + len, 0, 64);
		}
		lap("pread (page-locked)");
		{   // [r6] the reader's kernels are small and the host waits for four of them per bucket (counts, sizes): on the queue of highest
			// priority they are dispatched ahead of the engine's long launches instead of behind them.  No effect with one reader; what
			// lets TWO readers overlap (profiles/r06_sam_leg_ab.txt).  Tuning knob ingest_priority=0: a queue like any other
			const char *v = ema_tuning_get("ingest_priority");
			int least = 0, greatest = 0;
			if (!(v && atoi(v) == 0) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
			    hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) != hipSuccess) { st = nullptr; (void)hipGetLastError(); }      // (no such queue: an ordinary one)
			if (!st) ING(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
		}
		ING(dev_pool().take(len + 64 + 512, txt, device));
		char *d_text = (char *)txt.p;
		unsigned long long *d_cnt = (unsigned long long *)(d_text + ((len + 64 + 255) & ~(size_t)255));      // [n_nl][select's count][irregular]
		ING(hipMemcpyAsync(d_text, pin.p, len + 64, hipMemcpyHostToDevice, st));
		ING(hipMemsetAsync(d_cnt, 0, 64, st));
		int *d_irr = (int *)(d_cnt + 2);
		hipLaunchKernelGGL(ema_k_ing_count, dim3((unsigned)((len + 4095) / 4096)), dim3(256), 0, st, d_text, (uint32_t)len, d_cnt, d_irr);
		unsigned long long h_cnt[4] = {0, 0, 0, 0};
		ING(hipMemcpyAsync(h_cnt, d_cnt, 32, hipMemcpyDeviceToHost, st));
		ING(hipStreamSynchronize(st));
		lap("upload + count");
		if (((int *)&h_cnt[2])[0]) { rc = -1001; goto done; }
		const uint32_t n_nl = (uint32_t)h_cnt[0];
		const uint32_t n = n_nl + (((const char *)pin.p)[len - 1] != '\n' ? 1u : 0u);
		if (n == 0 || n >= 0x7fffffffu) { rc = -1001; goto done; }
		// scratch for the lines
		size_t sel_tmp = 0, sort_tmp = 0, scan_tmp = 0;
		(void)rocprim::select(nullptr, sel_tmp, rocprim::make_counting_iterator<uint32_t>(0), (uint32_t *)nullptr, (size_t *)nullptr, len, IsNewline{d_text}, st);
		(void)rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, is_haplotag ? 64u : (unsigned)(3 * bc_len), st);
		if (is_haplotag) {
			size_t b32 = 0;
			(void)rocprim::radix_sort_pairs(nullptr, b32, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, 32u, st);
			sort_tmp = std::max(sort_tmp, b32);
		}
		(void)rocprim::exclusive_scan(nullptr, scan_tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)2 * n + 1, rocprim::plus<uint32_t>(), st);
		const size_t tmp_bytes = std::max(sel_tmp, std::max(sort_tmp, scan_tmp)) + 256;
		ING(dev_pool().take(Carve::need({(size_t)(n_nl + 1) * 4, (size_t)n * sizeof(Fields), (size_t)n * 8, (size_t)n * 8, (size_t)n * 4, (size_t)n * 4, (size_t)n * 8, (size_t)n * 4, (size_t)n * 4,
		                                 ((size_t)2 * n + 1) * 4, ((size_t)n + 1) * 4, ((size_t)2 * n + 1) * 4, ((size_t)n + 1) * 4, tmp_bytes, 64}), work, device));
		Carve cw(work.p);
		uint32_t *d_nl = cw.take<uint32_t>((size_t)n_nl + 1);
		Fields *d_fields = cw.take<Fields>(n);
		uint64_t *d_codes = cw.take<uint64_t>(n), *d_codes_s = cw.take<uint64_t>(n);
		uint32_t *d_idx = cw.take<uint32_t>(n), *d_order = cw.take<uint32_t>(n);
		uint64_t *d_pick = cw.take<uint64_t>(n);      // haplotag: the high key words after the first sort
		uint32_t *d_lo = cw.take<uint32_t>(n), *d_lo_s = cw.take<uint32_t>(n);
		uint32_t *d_rlen = cw.take<uint32_t>((size_t)2 * n + 1), *d_ilen = cw.take<uint32_t>((size_t)n + 1);
		uint32_t *d_off = cw.take<uint32_t>((size_t)2 * n + 1), *d_id_off = cw.take<uint32_t>((size_t)n + 1);
		void *d_tmp = cw.take<char>(tmp_bytes);
		size_t *d_sel_n = (size_t *)cw.take<char>(64);
		if (n_nl) { size_t b = tmp_bytes; ING(rocprim::select(d_tmp, b, rocprim::make_counting_iterator<uint32_t>(0), d_nl, d_sel_n, len, IsNewline{d_text}, st)); }
		hipLaunchKernelGGL(ema_k_ing_parse, dim3((n + 255) / 256), dim3(256), 0, st, d_text, (uint32_t)len, d_nl, n_nl, n, bc_len, is_haplotag, (uint32_t)max_read_len,
		                   d_fields, d_codes, d_lo, d_idx, d_irr);
		if (is_haplotag) {      // a 96-bit key: two stable sorts, the low word first
			{ size_t b = tmp_bytes; ING(rocprim::radix_sort_pairs(d_tmp, b, d_lo, d_lo_s, d_idx, d_order, (size_t)n, 0u, 32u, st)); }
			hipLaunchKernelGGL(ema_k_ing_pick, dim3((n + 255) / 256), dim3(256), 0, st, d_codes, d_order, n, d_pick);
			{ size_t b = tmp_bytes; ING(rocprim::radix_sort_pairs(d_tmp, b, d_pick, d_codes_s, d_order, d_idx, (size_t)n, 0u, 64u, st)); }
			std::swap(d_order, d_idx);      // (the final order is in what was d_idx)
		} else { size_t b = tmp_bytes; ING(rocprim::radix_sort_pairs(d_tmp, b, d_codes, d_codes_s, d_idx, d_order, (size_t)n, 0u, (unsigned)(3 * bc_len), st)); }
		hipLaunchKernelGGL(ema_k_ing_lens, dim3(n / 256 + 1), dim3(256), 0, st, d_order, d_fields, n, d_rlen, d_ilen);
		{ size_t b = tmp_bytes; ING(rocprim::exclusive_scan(d_tmp, b, d_rlen, d_off, 0u, (size_t)2 * n + 1, rocprim::plus<uint32_t>(), st)); }
		{ size_t b = tmp_bytes; ING(rocprim::exclusive_scan(d_tmp, b, d_ilen, d_id_off, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st)); }
		uint32_t totals[2] = {0, 0};
		int irr = 0;
		ING(hipMemcpyAsync(&totals[0], d_off + 2 * (size_t)n, 4, hipMemcpyDeviceToHost, st));
		ING(hipMemcpyAsync(&totals[1], d_id_off + n, 4, hipMemcpyDeviceToHost, st));
		ING(hipMemcpyAsync(&irr, d_irr, 4, hipMemcpyDeviceToHost, st));
		ING(hipStreamSynchronize(st));
		lap("lines, parse, sort, sums");
		if (irr) { rc = -1001; goto done; }
		const size_t nb = totals[0], ni = totals[1];      // (each below the file's size: no 32-bit overflow)
		// the bucket's arrays on the device, and their host copies (all but bases and qualities)
		ING(dev_pool().take(Carve::need({nb + 8, nb + 8, ni + 8, ((size_t)2 * n + 1) * 4, ((size_t)n + 1) * 4, (size_t)n * 8}), keep, device));
		Carve ck(keep.p);
		char *k_bases = ck.take<char>(nb + 8), *k_quals = ck.take<char>(nb + 8), *k_ids = ck.take<char>(ni + 8);
		uint32_t *k_off = ck.take<uint32_t>((size_t)2 * n + 1), *k_id_off = ck.take<uint32_t>((size_t)n + 1);
		uint64_t *k_bc = ck.take<uint64_t>(n);
		ING(hipMemcpyAsync(k_off, d_off, ((size_t)2 * n + 1) * 4, hipMemcpyDeviceToDevice, st));
		ING(hipMemcpyAsync(k_id_off, d_id_off, ((size_t)n + 1) * 4, hipMemcpyDeviceToDevice, st));
		hipLaunchKernelGGL(ema_k_ing_gather, dim3((n + 255) / 256), dim3(256), 0, st, d_text, (uint32_t)len, d_nl, n_nl, d_order, d_fields, d_codes_s, n, bc_len, is_haplotag,
		                   k_off, k_id_off, k_bases, k_quals, k_ids, k_bc);
		ING(hipGetLastError());
		o = (ema_bucket *)calloc(1, sizeof(ema_bucket));
		if (o) {
			o->n_pairs = n;
			o->bc = (uint64_t *)malloc(((size_t)n + 1) * 8); o->off = (uint32_t *)malloc(((size_t)2 * n + 1) * 4);
			o->id_off = (uint32_t *)malloc(((size_t)n + 1) * 4); o->ids = (char *)malloc(ni + 1);
		}
		if (!o || !o->bc || !o->off || !o->id_off || !o->ids) { g_dev_err = "out of memory"; rc = EMA_EIO; goto done; }
		ING(hipMemcpyAsync(o->bc, k_bc, (size_t)n * 8, hipMemcpyDeviceToHost, st));
		ING(hipMemcpyAsync(o->off, k_off, ((size_t)2 * n + 1) * 4, hipMemcpyDeviceToHost, st));
		ING(hipMemcpyAsync(o->id_off, k_id_off, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
		if (ni) ING(hipMemcpyAsync(o->ids, k_ids, ni, hipMemcpyDeviceToHost, st));
		ING(hipStreamSynchronize(st));
		lap("gather + download");
		size_t n_groups = 0;
		for (size_t i = 0; i < n; ++i) n_groups += (i == 0 || o->bc[i] != o->bc[i - 1]);
		o->n_groups = n_groups;
		o->group_off = (uint64_t *)malloc((n_groups + 1) * sizeof(uint64_t));
		if (!o->group_off) { g_dev_err = "out of memory"; rc = EMA_EIO; goto done; }
		size_t g = 0;
		for (size_t i = 0; i < n; ++i) if (i == 0 || o->bc[i] != o->bc[i - 1]) o->group_off[g++] = i;
		o->group_off[n_groups] = n;
		ema_bucket_dev_impl *d = new ema_bucket_dev_impl();
		d->pub.device = device; d->pub.bases = k_bases; d->pub.quals = k_quals; d->pub.ids = k_ids; d->pub.off = k_off; d->pub.id_off = k_id_off; d->pub.bc = k_bc;
		d->pub.n_pairs = n; d->pub.n_bases = nb; d->pub.n_ids = ni;
		d->block = keep; keep = Block();
		o->dev = d;
		rc = 0;
		lap("groups");
	}
done:
#undef ING
	close(fd);
	// (on a failure path copies and kernels may still be queued on st: the blocks go back to pools other readers take from, and the
	// targets of the device-to-host copies are about to leave scope -- wait for the stream first)
	if (st) { if (rc != 0) (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
	pin_pool().give(pin); dev_pool().give(txt); dev_pool().give(work); dev_pool().give(keep);
	if (rc == 0) { *out = o; return 0; }
	if (o) { free(o->bc); free(o->off); free(o->id_off); free(o->ids); free(o->group_off); free(o); }
	if (rc == -1001) return ema_bucket_read(path, bc_len, is_haplotag, max_read_len, out);
	return rc;
}

}  // extern "C"
