// ema_amd/csrc/sam_dev.hip -- host side of the SAM formatter on the device (include/ema_sam.h: ema_sam_dev_*; kernels in k_sam.hip).
//
// Per bucket: the reader's arrays go up as they are (names, bases, qualities, offsets, barcodes: ~700 bytes per pair), with them the
// compact records of the cloud stage (52 bytes per selected record) and the stretch of the batch's CIGAR array they name; the text
// comes back into one of two page-locked buffers and is written to the descriptor while the next stretch of lines is rendered
// (a stretch is at most 2^20 lines, ~450 MB of text).  Device and page-locked buffers grow to the largest bucket seen and stay.
// The host's part of a line is what ema_clouds_select already did (emit = 1); here it copies nothing per line.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <poll.h>
#include <unistd.h>
#include "ema_ingest.h"
#include "ema_sam.h"
#include "dev_sam.h"
#include "dev_bucket.h"
#include "host_cpuacct.h"

const char *ema_tuning_get(const char *key);      // engine.hip

namespace {

thread_local std::string g_dev_err;

struct DBuf {
	void *p = nullptr;
	size_t cap = 0;
	hipError_t need(size_t n)
	{
		if (n <= cap) return hipSuccess;
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		const size_t want = n + n / 8 + 4096;
		const hipError_t rc = hipMalloc(&p, want);
		if (rc == hipSuccess) cap = want;
		return rc;
	}
	void drop() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PBuf {      // page-locked host memory
	char *p = nullptr;
	size_t cap = 0;
	hipError_t need(size_t n)
	{
		if (n <= cap) return hipSuccess;
		if (p) (void)hipHostFree(p);
		p = nullptr; cap = 0;
		const size_t want = n + n / 8 + 4096;
		const hipError_t rc = hipHostMalloc((void **)&p, want, hipHostMallocDefault);
		if (rc == hipSuccess) cap = want;
		return rc;
	}
	void drop() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// lines rendered per round (a multiple of 64: the prefix sum's chunks); "sam_stretch_lines" in the tuning string makes it small for tests
uint32_t stretch_lines()
{
	const char *v = ema_tuning_get("sam_stretch_lines");
	const long n = v ? atol(v) : 0;
	return n >= 64 && n <= (1l << 24) ? (uint32_t)(n / 64 * 64) : 1u << 20;
}

}  // namespace

struct ema_sam_dev {
	int device = 0;
	hipStream_t st = nullptr;
	int32_t n_contigs = 0;
	DBuf names, name_off, bases, quals, off, ids, id_off, bc, cigar, desc, xa, sel, local, ctot, cbase, text[2], small;
	PBuf htext[2], hsmall;
	~ema_sam_dev()
	{
		for (DBuf *b : {&names, &name_off, &bases, &quals, &off, &ids, &id_off, &bc, &cigar, &desc, &xa, &sel, &local, &ctot, &cbase, &text[0], &text[1], &small}) b->drop();
		htext[0].drop(); htext[1].drop(); hsmall.drop();
		if (st) (void)hipStreamDestroy(st);
	}
};

namespace {

#define SAMCHK(call)                                                                                              \
	do {                                                                                                          \
		const hipError_t rc_ = (call);                                                                            \
		if (rc_ != hipSuccess) { g_dev_err = std::string(#call) + ": " + hipGetErrorString(rc_); return EMA_EIO; } \
	} while (0)

// an interrupted or momentarily refused write is retried (as ema_sam_write does); *done counts what is on the descriptor
int write_all(int fd, const char *p, size_t n, size_t *done)
{
	size_t at = 0;
	int refused = 0;
	while (at < n) {
		const ssize_t w = write(fd, p + at, n - at);
		if (w < 0 && errno == EINTR) continue;
		if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK) && ++refused <= 120) {
			struct pollfd pf; pf.fd = fd; pf.events = POLLOUT; pf.revents = 0;
			(void)poll(&pf, 1, 1000);
			continue;
		}
		if (w > 0) refused = 0;
		if (w <= 0) { *done += at; return EMA_EIO; }
		at += (size_t)w;
	}
	*done += n;
	return 0;
}

struct Sink {      // where the text goes: a descriptor, or one growing malloc'd block (ema_sam_dev_format)
	int fd = -1;
	char *block = nullptr;
	size_t n = 0, cap = 0;
	int put(const char *p, size_t k)
	{
		if (fd >= 0) return write_all(fd, p, k, &n);
		if (n + k + 1 > cap) {
			const size_t want = (n + k + 1) * 2;
			char *q = (char *)realloc(block, want);
			if (!q) return EMA_EARG;
			block = q; cap = want;
		}
		memcpy(block + n, p, k);
		n += k;
		return 0;
	}
};

int run(ema_sam_dev *d, Sink &out, const ema_bucket *bk, const uint32_t *cigar, uint64_t cigar_lo, uint64_t cigar_hi, const ema_sam_desc *descs,
        size_t n_descs, const ema_sam_xa *xas, size_t n_xas, const uint32_t *sel_at, size_t n_sel, const ema_sam_opts *opt)
{
	EMA_CPU(EMA_CPU_FORMAT);
	g_dev_err.clear();
	if (!d || !bk || !opt || !opt->bx_index || opt->bc_len < 0 || opt->bc_len > 32 || (opt->is_haplotag && opt->bc_len != 12)) return EMA_EARG;
	if ((n_descs && !descs) || (n_xas && !xas) || (n_sel && !sel_at) || cigar_hi < cigar_lo || (cigar_hi > cigar_lo && !cigar)) return EMA_EARG;
	if (n_sel >= (1ull << 31) || n_descs >= (1ull << 32) || cigar_lo >= (1ull << 32)) return EMA_EARG;
	if (n_sel == 0) return 0;
	SAMCHK(hipSetDevice(d->device));
	const size_t n_pairs = bk->n_pairs, n_reads = 2 * n_pairs;
	const size_t n_bases = bk->off[n_reads], n_ids = bk->id_off[n_pairs], n_cig = (size_t)(cigar_hi - cigar_lo);
	// what the lines share: the RG identifier up to its first whitespace (src/samrecord.c:260-264), bx_index; then the kernels' two results
	size_t rg_len = 0;
	if (opt->rg_id) for (size_t i = 0; opt->rg_id[i] && !(opt->rg_id[i] == ' ' || (opt->rg_id[i] >= '\t' && opt->rg_id[i] <= '\r')); ++i) rg_len = i + 1;
	const size_t bx_len = strlen(opt->bx_index);
	const size_t small_bytes = 16 + ((rg_len + 7) & ~(size_t)7) + ((bx_len + 7) & ~(size_t)7);      // [total u64][bad i32, pad][rg][bx]
	SAMCHK(d->hsmall.need(small_bytes));
	SAMCHK(d->small.need(small_bytes));
	memset(d->hsmall.p, 0, small_bytes);
	const size_t rg_at = 16, bx_at = rg_at + ((rg_len + 7) & ~(size_t)7);
	if (rg_len) memcpy(d->hsmall.p + rg_at, opt->rg_id, rg_len);
	memcpy(d->hsmall.p + bx_at, opt->bx_index, bx_len);
	SAMCHK(hipMemcpyAsync(d->small.p, d->hsmall.p, small_bytes, hipMemcpyHostToDevice, d->st));
	// a bucket read by ema_bucket_read_device has its arrays on the device already: nothing of the bucket goes up
	const ema_bucket_dev *twin = ema_bucket_dev_view(bk);
	if (twin && twin->device != d->device) twin = nullptr;
	if (!twin && (!bk->bases || !bk->quals)) return EMA_EARG;
	struct Up { DBuf *b; const void *src; size_t bytes; };
	const Up ups[] = {{&d->bases, bk->bases, n_bases}, {&d->quals, bk->quals, n_bases}, {&d->off, bk->off, (n_reads + 1) * 4}, {&d->ids, bk->ids, n_ids},
	                  {&d->id_off, bk->id_off, (n_pairs + 1) * 4}, {&d->bc, bk->bc, n_pairs * 8}, {&d->cigar, cigar, n_cig * 4},
	                  {&d->desc, descs, n_descs * sizeof(ema_sam_desc)}, {&d->xa, xas, n_xas * sizeof(ema_sam_xa)}, {&d->sel, sel_at, n_sel * 4}};
	for (size_t u_i = 0; u_i < sizeof ups / sizeof ups[0]; ++u_i) {
		const Up &u = ups[u_i];
		if (twin && u_i < 6) continue;
		SAMCHK(u.b->need(u.bytes + 8));
		if (u.bytes) SAMCHK(hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, d->st));
	}
	SamJob J;
	memset(&J, 0, sizeof J);
	if (twin) { J.bases = twin->bases; J.quals = twin->quals; J.off = twin->off; J.ids = twin->ids; J.id_off = twin->id_off; J.bc = twin->bc; }
	else {
		J.bases = (const char *)d->bases.p; J.quals = (const char *)d->quals.p; J.off = (const uint32_t *)d->off.p;
		J.ids = (const char *)d->ids.p; J.id_off = (const uint32_t *)d->id_off.p; J.bc = (const uint64_t *)d->bc.p;
	}
	J.cigar = (const uint32_t *)d->cigar.p; J.desc = (const ema_sam_desc *)d->desc.p; J.xa = (const ema_sam_xa *)d->xa.p;
	J.names = (const char *)d->names.p; J.name_off = (const uint32_t *)d->name_off.p;
	J.rg = (const char *)d->small.p + rg_at; J.bx = (const char *)d->small.p + bx_at;
	J.cigar_lo = (uint32_t)cigar_lo;
	J.has_rg = opt->rg_id != nullptr; J.rg_len = (int32_t)rg_len; J.bx_len = (int32_t)bx_len; J.bc_len = opt->bc_len; J.is_haplotag = opt->is_haplotag;
	J.insert_min = opt->insert_min; J.insert_max = opt->insert_max;
	uint64_t *d_total = (uint64_t *)d->small.p;
	int *d_bad = (int *)((char *)d->small.p + 8);
	const uint64_t n_lines = 2 * (uint64_t)n_sel;
	const uint32_t kStretch = stretch_lines();
	const uint32_t max_stretch = (uint32_t)std::min<uint64_t>(n_lines, kStretch);
	SAMCHK(d->local.need((size_t)max_stretch * 4));
	SAMCHK(d->ctot.need(((size_t)max_stretch / 64 + 1) * 4));
	SAMCHK(d->cbase.need(((size_t)max_stretch / 64 + 1) * 8));
	// stretch k's text is copied back and written while stretch k + 1 is rendered
	size_t pending = 0;      // bytes of the previous stretch, in htext[prev]
	int prev = -1, rc = 0, k = 0;
	for (uint64_t l0 = 0; l0 < n_lines; l0 += kStretch, ++k) {
		const int cur = k & 1;
		J.sel_at = (const uint32_t *)d->sel.p + l0 / 2;
		J.n_lines = (uint32_t)std::min<uint64_t>(kStretch, n_lines - l0);
		ema_launch_sam_len(J, (uint32_t *)d->local.p, (uint32_t *)d->ctot.p, d->st);
		ema_launch_sam_tops((J.n_lines + 63u) / 64u, (const uint32_t *)d->ctot.p, (uint64_t *)d->cbase.p, d_total, d->st);
		uint64_t total = 0;
		SAMCHK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, d->st));
		SAMCHK(hipStreamSynchronize(d->st));      // (the previous stretch's text is on the host now, too)
		SAMCHK(d->text[cur].need((size_t)total + 8));
		SAMCHK(d->htext[cur].need((size_t)total + 8));
		ema_launch_sam_write(J, (const uint32_t *)d->local.p, (const uint64_t *)d->cbase.p, (char *)d->text[cur].p, d_bad, d->st);
		SAMCHK(hipGetLastError());
		SAMCHK(hipMemcpyAsync(d->htext[cur].p, d->text[cur].p, (size_t)total, hipMemcpyDeviceToHost, d->st));
		if (prev >= 0 && rc == 0) rc = out.put(d->htext[prev].p, pending);
		prev = cur; pending = (size_t)total;
	}
	int bad = 0;
	SAMCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, d->st));
	SAMCHK(hipStreamSynchronize(d->st));
	if (bad) return EMA_EFORMAT;      // a base outside ACGTN in a reversed read (the reference asserts, src/samrecord.c:90-102)
	if (prev >= 0 && rc == 0) rc = out.put(d->htext[prev].p, pending);
	return rc;
}

}  // namespace

extern "C" {

const char *ema_sam_dev_last_error(void) { return g_dev_err.c_str(); }

int ema_sam_dev_open(int device, const char *const *contig_names, int32_t n_contigs, ema_sam_dev_t **out)
{
	if (!out || n_contigs < 0 || (n_contigs && !contig_names)) return EMA_EARG;
	*out = nullptr;
	g_dev_err.clear();
	ema_sam_dev *d = new ema_sam_dev();
	d->device = device; d->n_contigs = n_contigs;
	std::vector<char> names;
	std::vector<uint32_t> name_off((size_t)n_contigs + 1, 0);
	for (int32_t i = 0; i < n_contigs; ++i) {
		const size_t l = strlen(contig_names[i]);
		names.insert(names.end(), contig_names[i], contig_names[i] + l);
		name_off[(size_t)i + 1] = (uint32_t)names.size();
	}
	names.resize(names.size() + 8, 0);
	auto fail = [&](hipError_t rc, const char *what) { g_dev_err = std::string(what) + ": " + hipGetErrorString(rc); delete d; return EMA_EIO; };
	hipError_t rc;
	if ((rc = hipSetDevice(device)) != hipSuccess) return fail(rc, "hipSetDevice");
	if ((rc = hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking)) != hipSuccess) return fail(rc, "hipStreamCreateWithFlags");
	if ((rc = d->names.need(names.size())) != hipSuccess) return fail(rc, "hipMalloc");
	if ((rc = d->name_off.need(name_off.size() * 4)) != hipSuccess) return fail(rc, "hipMalloc");
	if ((rc = hipMemcpy(d->names.p, names.data(), names.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(rc, "hipMemcpy");
	if ((rc = hipMemcpy(d->name_off.p, name_off.data(), name_off.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(rc, "hipMemcpy");
	*out = d;
	return 0;
}

void ema_sam_dev_close(ema_sam_dev_t *d)
{
	if (!d) return;
	(void)hipSetDevice(d->device);
	delete d;
}

int ema_sam_dev_write(ema_sam_dev_t *d, int fd, const ema_bucket *bk, const uint32_t *cigar, uint64_t cigar_lo, uint64_t cigar_hi,
                      const ema_sam_desc *descs, size_t n_descs, const ema_sam_xa *xas, size_t n_xas, const uint32_t *sel_at, size_t n_sel,
                      const ema_sam_opts *o, size_t *n_bytes)
{
	if (n_bytes) *n_bytes = 0;
	if (fd < 0) return EMA_EARG;
	Sink s;
	s.fd = fd;
	const int rc = run(d, s, bk, cigar, cigar_lo, cigar_hi, descs, n_descs, xas, n_xas, sel_at, n_sel, o);
	if (n_bytes) *n_bytes = s.n;
	return rc;
}

int ema_sam_dev_format(ema_sam_dev_t *d, const ema_bucket *bk, const uint32_t *cigar, uint64_t cigar_lo, uint64_t cigar_hi,
                       const ema_sam_desc *descs, size_t n_descs, const ema_sam_xa *xas, size_t n_xas, const uint32_t *sel_at, size_t n_sel,
                       const ema_sam_opts *o, char **text, size_t *n_bytes)
{
	if (!text || !n_bytes) return EMA_EARG;
	*text = nullptr; *n_bytes = 0;
	Sink s;
	const int rc = run(d, s, bk, cigar, cigar_lo, cigar_hi, descs, n_descs, xas, n_xas, sel_at, n_sel, o);
	if (rc != 0) { free(s.block); return rc; }
	if (!s.block) s.block = (char *)malloc(1);
	*text = s.block; *n_bytes = s.n;
	return 0;
}

}  // extern "C"
