// ema_amd/csrc/host_ingest.cpp -- bucket reader in front of the hot path (include/ema_ingest.h).
//
// What the reference does per bucket, one line at a time on one thread -- count_lines, fgets + malloc + strcpy per line,
// qsort of the line pointers by strncmp(.., BC_LEN), six copy_until_space calls into 568-byte FASTQRecords (reference
// src/align.c:759-806, src/util.c:11-21,97-106) -- is here five data-parallel passes over the mapped file on the host's
// cores, producing the engine's input layout directly:
//   1. newline scan: line table (start, length), field extents, validation, and -- the common case: barcodes of ACGT / acgt, at
//      most 21 bases -- the barcode as a sort code of 3 bits per base in the order strncmp gives the bytes (A C G T a c g t);
//   2. order: a stable radix sort of (code, line) -- or, for haplotag barcodes and longer ones, the first bc_len bytes of each
//      line, cut at the line's end and zero-padded (what strncmp sees: the buffer holds the line, its '\n', then NUL), as
//      big-endian words + the line number, chunk-sorted and merged;
//   3. barcodes in sorted order (encode_bc, src/util.c:41-76): from the code, or from the text;
//   4. prefix sums -> off[], id_off[];
//   5. copies of bases / qualities / identifiers: lines taken in FILE order (the source streams through the cache) and written
//      to their sorted place -- a gather in sorted order waits for five cache misses per pair; barcode groups
//      (seek_next_barcode_group's runs of equal bc, src/align.c:829-837).
// No GPU work: parsing is byte-serial per line and the payload crosses PCIe once, as 2-bit packed reads, in
// ema_engine_stage.
#include <algorithm>
#include <chrono>
#include <array>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <atomic>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "ema_ingest.h"
#include "host_cpuacct.h"
#include "host_pool.h"

extern "C" void ema_bucket_dev_release(void *dev);

namespace {

thread_local std::string g_err;

const size_t kMaxLine = 4999;      // fgets(buf, 5000): at most 4999 bytes per call, '\n' included (src/align.c:762,768)
const size_t kMaxId = 149;         // id[150] (include/samrecord.h:12)

int n_threads() { return EmaPool::get().size(); }      // host_pool.h

// fn(k, b, e) on contiguous ranges of [0, n), one per thread of the pool
template <typename F> void parallel_ranges(size_t n, size_t min_per_thread, F fn)
{
	size_t t = std::min<size_t>((size_t)n_threads(), n / std::max<size_t>(min_per_thread, 1));
	if (t <= 1) { fn((size_t)0, (size_t)0, n); return; }
	const size_t per = (n + t - 1) / t;
	EmaPool::get().run(t, [&](size_t k) {
		EMA_CPU(EMA_CPU_READER);
		const size_t b = k * per, e = std::min(n, b + per);
		if (b < e) fn(k, b, e);
	});
}

// isspace() in the "C" locale, which is what the reference runs in (it never calls setlocale)
inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

struct Line { uint64_t start; uint32_t len; };      // len counts the '\n' when the line has one

template <int W> struct Key {
	std::array<uint64_t, W> w;      // key bytes, big-endian, so that word order is byte order
	uint32_t line;
	bool operator<(const Key &o) const
	{
		for (int i = 0; i < W; ++i) if (w[i] != o.w[i]) return w[i] < o.w[i];
		return line < o.line;
	}
};

template <int W> void sorted_lines(const char *text, const std::vector<Line> &lines, int bc_len, std::vector<uint32_t> &order)
{
	const size_t n = lines.size();
	std::vector<Key<W>> a(n), b;
	parallel_ranges(n, 1 << 14, [&](size_t, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			unsigned char k[8 * W];
			memset(k, 0, sizeof k);
			const unsigned char *p = (const unsigned char *)text + lines[i].start;
			const size_t m = std::min<size_t>((size_t)bc_len, lines[i].len);
			for (size_t j = 0; j < m && p[j]; ++j) k[j] = p[j];      // strncmp stops at a NUL
			for (int x = 0; x < W; ++x) {
				uint64_t v = 0;
				for (int j = 0; j < 8; ++j) v = v << 8 | k[8 * x + j];
				a[i].w[x] = v;
			}
			a[i].line = (uint32_t)i;
		}
	});
	// chunk sorts, then a merge tree between the two buffers
	size_t t = 1;
	while (t * 2 <= (size_t)n_threads() && n / (t * 2) >= (1 << 14)) t *= 2;
	const size_t per = (n + t - 1) / t;
	auto lo_of = [&](size_t k) { return std::min(n, k * per); };
	EmaPool::get().run(t, [&](size_t k) { EMA_CPU(EMA_CPU_READER); std::sort(a.begin() + lo_of(k), a.begin() + lo_of(k + 1)); });
	if (t > 1) b.resize(n);
	std::vector<Key<W>> *src = &a, *dst = &b;
	for (size_t width = 1; width < t; width *= 2) {
		EmaPool::get().run((t + 2 * width - 1) / (2 * width), [&](size_t j) {
			EMA_CPU(EMA_CPU_READER);
			const size_t k = j * 2 * width;
			const size_t l = lo_of(k), m = lo_of(std::min(t, k + width)), r = lo_of(std::min(t, k + 2 * width));
			std::merge(src->begin() + l, src->begin() + m, src->begin() + m, src->begin() + r, dst->begin() + l);
		});
		std::swap(src, dst);
	}
	order.resize(n);
	const std::vector<Key<W>> &s = *src;
	parallel_ranges(n, 1 << 16, [&](size_t, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) order[i] = s[i].line; });
}

struct Fields { uint16_t id_b, id_l, r1_b, r1_l, q1_b, r2_b, r2_l, q2_b; };      // byte offsets within the line

// Sort code of a barcode of ACGT / acgt: 3 bits per base, first base in the highest bits, in the order of the bytes' values
// ('A' < 'C' < 'G' < 'T' < 'a' < 'c' < 'g' < 't'); the low two bits of a base's code are its encode_bc value.  -1: another byte.
const struct BaseCode {
	int8_t t[256];
	BaseCode() { memset(t, -1, sizeof t); const char *b = "ACGTacgt"; for (int i = 0; i < 8; ++i) t[(unsigned char)b[i]] = (int8_t)i; }
} kBaseCode;
inline int base_code(unsigned char c) { return kBaseCode.t[c]; }

// order[] = the lines by (code, line number): least-significant-digit radix sort, 8 bits a pass (stable, so equal codes keep
// file order -- the order Key<W> + std::sort gives)
void radix_order(const std::vector<uint64_t> &code, int bits, std::vector<uint32_t> &order)
{
	struct KV { uint64_t k; uint32_t i; };
	const size_t n = code.size();
	order.clear();
	if (!n) return;
	std::vector<KV> a(n), b(n);
	for (size_t i = 0; i < n; ++i) { a[i].k = code[i]; a[i].i = (uint32_t)i; }
	KV *src = a.data(), *dst = b.data();
	for (int sh = 0; sh < bits; sh += 8) {
		size_t cnt[257] = {0};
		for (size_t i = 0; i < n; ++i) ++cnt[((src[i].k >> sh) & 255) + 1];
		if (cnt[((src[0].k >> sh) & 255) + 1] == n) continue;      // every key has the same digit here
		for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
		for (size_t i = 0; i < n; ++i) dst[cnt[(src[i].k >> sh) & 255]++] = src[i];
		std::swap(src, dst);
	}
	order.resize(n);
	for (size_t i = 0; i < n; ++i) order[i] = src[i].i;
}

enum Bad : uint8_t { kOk = 0, kLong, kFew, kBcLen, kBcBase, kIdEmpty, kIdLong, kReadLong, kQualLen };
const char *const kBadText[] = {"", "line of 5000 bytes or more", "fewer than six fields", "barcode field is not bc_len bytes",
                                "barcode has a byte outside ACGT/acgt", "empty identifier", "identifier longer than 149 bytes",
                                "read longer than max_read_len", "quality string and read differ in length"};

// End of the field that starts at s[at]: the first whitespace or NUL at or after it (or len).  Eight bytes at a time:
// bases, qualities and identifiers are all above ' ', so a word with no byte below 0x21 cannot hold a separator.
inline size_t field_end(const char *s, size_t at, size_t len)
{
	while (at + 8 <= len) {
		uint64_t x;
		memcpy(&x, s + at, 8);
		const uint64_t low = (x - 0x2121212121212121ULL) & ~x & 0x8080808080808080ULL;      // lowest flag = first byte < 0x21
		if (!low) { at += 8; continue; }
		at += (size_t)__builtin_ctzll(low) >> 3;
		if (!s[at] || is_space((unsigned char)s[at])) return at;
		++at;      // some other control byte: part of the field
	}
	while (at < len && s[at] && !is_space((unsigned char)s[at])) ++at;
	return at;
}

// One field as copy_until_space reads it (src/util.c:11-21): bytes up to the next whitespace or NUL, then one byte skipped.
inline bool next_field(const char *s, size_t len, size_t &at, size_t &b, size_t &l)
{
	if (at > len) return false;      // the previous field ran into the end of the line: nothing left to read
	b = at;
	at = field_end(s, at, len);
	l = at - b;
	at = (at < len && s[at]) ? at + 1 : len + 1;
	return true;
}

inline bool acgt_only(const char *s, size_t n)
{
	for (size_t i = 0; i < n; ++i) {
		const char c = (char)(s[i] & ~0x20);      // upper case
		if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
	}
	return true;
}

int encode_default(const char *bc, int bc_len, uint64_t *out)
{
	uint64_t v = 0;
	for (int i = bc_len - 1; i >= 0; --i) {      // first base in the lowest two bits
		uint64_t c;
		switch (bc[i]) {
		case 'A': case 'a': c = 0; break;
		case 'C': case 'c': c = 1; break;
		case 'G': case 'g': c = 2; break;
		case 'T': case 't': c = 3; break;
		default: return EMA_EFORMAT;
		}
		v = v << 2 | c;
	}
	*out = v;
	return 0;
}

uint64_t encode_haplotag(const char *s)
{
	// AxxCxxBxxDxx: the four two-digit numbers, A and C in the upper half, B and D in the lower (src/util.c:63-70);
	// the same int arithmetic as the reference's macros, whatever the bytes are
	auto two = [](const char *p) { return 10 * (p[0] - '0') + (p[1] - '0'); };
	const uint32_t a = (uint32_t)two(s + 1), c = (uint32_t)two(s + 4), b = (uint32_t)two(s + 7), d = (uint32_t)two(s + 10);
	return (uint64_t)(uint32_t)(a << 24 | c << 16 | b << 8 | d);
}

int fail(int rc, const std::string &m) { g_err = m; return rc; }

}  // namespace

extern "C" {

const char *ema_bucket_last_error(void) { return g_err.c_str(); }

int ema_barcode_encode(const char *bc, int bc_len, int is_haplotag, uint64_t *out)
{
	if (!bc || !out) return EMA_EARG;
	if (is_haplotag) { *out = encode_haplotag(bc); return 0; }
	if (bc_len < 1 || bc_len > 32) return EMA_EARG;
	return encode_default(bc, bc_len, out);
}

void ema_barcode_decode(uint64_t bc, int bc_len, int is_haplotag, char *out)
{
	if (is_haplotag) {      // decode_bc_haplotag, src/util.c:86-89
		char t[16];
		snprintf(t, sizeof t, "A%02uC%02uB%02uD%02u", (unsigned)(bc >> 24) & 127, (unsigned)(bc >> 16) & 127,
		         (unsigned)(bc >> 8) & 127, (unsigned)bc & 127);
		memcpy(out, t, 12);
		return;
	}
	for (int i = 0; i < bc_len; ++i) { out[i] = "ACGT"[bc & 3]; bc >>= 2; }      // decode_bc_default, src/util.c:78-84
}

void ema_bucket_free(ema_bucket *b)
{
	if (!b) return;
	free(b->group_off); free(b->bc); free(b->off); free(b->bases); free(b->quals); free(b->id_off); free(b->ids);
	if (b->dev) ema_bucket_dev_release(b->dev);      // the arrays on the device (ingest_dev.hip)
	free(b);
}

int ema_bucket_parse(const char *text, size_t len, int bc_len, int is_haplotag, int max_read_len, ema_bucket **out)
{
	if (!out) return EMA_EARG;
	EMA_CPU(EMA_CPU_READER);
	*out = nullptr;
	g_err.clear();
	if ((!text && len) || bc_len < 1 || bc_len > 32 || max_read_len < 1 || max_read_len > 4096 || (is_haplotag && bc_len != 12))
		return fail(EMA_EARG, "bad argument");
	const bool prof = getenv("EMA_INGEST_PROF") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto t0 = now();
	auto lap = [&](const char *what) { if (prof) { auto t = now(); fprintf(stderr, "[ingest] %-8s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; } };
	// 1. lines and fields, in file order: each thread takes the lines that START in its stretch of the text
	const size_t t_scan = std::max<size_t>(1, std::min<size_t>((size_t)n_threads(), len >> 20));
	const size_t per = (len + t_scan - 1) / t_scan;
	const bool coded = !is_haplotag && bc_len <= 21;      // barcodes travel as sort codes (base_code)
	struct Part { std::vector<Line> lines; std::vector<Fields> fields; std::vector<uint64_t> codes; size_t bad_line = (size_t)-1; Bad bad = kOk; };
	std::vector<Part> parts(t_scan);
	auto scan = [&](size_t k) {
		EMA_CPU(EMA_CPU_READER);
		Part &pt = parts[k];
		const size_t b = std::min(len, k * per), e = std::min(len, b + per);
		size_t at = b;
		if (k) {      // the line that straddles b belongs to the stretch before
			const char *q = (const char *)memchr(text + b - 1, '\n', len - (b - 1));
			at = q ? (size_t)(q - text) + 1 : len;
		}
		pt.lines.reserve((e - b) / 256 + 16); pt.fields.reserve((e - b) / 256 + 16);
		if (coded) pt.codes.reserve((e - b) / 256 + 16);
		while (at < e && at < len) {
			const char *s = text + at;
			const size_t room = len - at;
			Fields f{};
			Bad bad = kOk;
			const char *q = (const char *)memchr(s, '\n', room);
			const size_t ln = q ? (size_t)(q - s) + 1 : room;      // the '\n' counts, as in fgets' buffer
			size_t p = 0, fb[6] = {0}, fl[6] = {0};
			bool six = ln <= kMaxLine;
			for (int x = 0; x < 6 && six; ++x) six = next_field(s, ln, p, fb[x], fl[x]);
			if (ln > kMaxLine) bad = kLong;
			else if (!six) bad = kFew;
			else if (fl[0] != (size_t)bc_len) bad = kBcLen;
			else if (!is_haplotag && !coded && !acgt_only(s, (size_t)bc_len)) bad = kBcBase;
			else if (fl[1] == 0) bad = kIdEmpty;
			else if (fl[1] > kMaxId) bad = kIdLong;
			else if (fl[2] > (size_t)max_read_len || fl[4] > (size_t)max_read_len) bad = kReadLong;
			else if (fl[3] != fl[2] || fl[5] != fl[4]) bad = kQualLen;
			uint64_t code = 0;
			if (bad == kOk && coded) {
				int any_bad = 0;
				for (int j = 0; j < bc_len; ++j) {
					const int c = base_code((unsigned char)s[j]);
					any_bad |= c;
					code = code << 3 | (uint64_t)(c & 7);
				}
				if (any_bad < 0) bad = kBcBase;
			}
			if (bad == kOk) {
				f.id_b = (uint16_t)fb[1]; f.id_l = (uint16_t)fl[1];
				f.r1_b = (uint16_t)fb[2]; f.r1_l = (uint16_t)fl[2]; f.q1_b = (uint16_t)fb[3];
				f.r2_b = (uint16_t)fb[4]; f.r2_l = (uint16_t)fl[4]; f.q2_b = (uint16_t)fb[5];
			} else if (pt.bad == kOk) { pt.bad = bad; pt.bad_line = pt.lines.size(); }
			pt.lines.push_back(Line{(uint64_t)at, (uint32_t)std::min<size_t>(ln, UINT32_MAX)});
			pt.fields.push_back(f);
			if (coded) pt.codes.push_back(code);
			at += ln;
		}
	};
	EmaPool::get().run(t_scan, scan);
	std::vector<size_t> first(t_scan + 1, 0);
	for (size_t k = 0; k < t_scan; ++k) first[k + 1] = first[k] + parts[k].lines.size();
	const size_t n = first[t_scan];
	if (n >= UINT32_MAX) return fail(EMA_EARG, "more than 2^32 lines in one bucket");
	for (size_t k = 0; k < t_scan; ++k)
		if (parts[k].bad != kOk)
			return fail(EMA_EFORMAT, "line " + std::to_string(first[k] + parts[k].bad_line + 1) + ": " + kBadText[parts[k].bad]);
	std::vector<Line> lines(n);
	std::vector<Fields> fields(n);      // by line number
	std::vector<uint64_t> codes(coded ? n : 0);
	{
		auto gather = [&](size_t k) {
			EMA_CPU(EMA_CPU_READER);
			std::copy(parts[k].lines.begin(), parts[k].lines.end(), lines.begin() + first[k]);
			std::copy(parts[k].fields.begin(), parts[k].fields.end(), fields.begin() + first[k]);
			if (coded) std::copy(parts[k].codes.begin(), parts[k].codes.end(), codes.begin() + first[k]);
			std::vector<Line>().swap(parts[k].lines); std::vector<Fields>().swap(parts[k].fields); std::vector<uint64_t>().swap(parts[k].codes);
		};
		EmaPool::get().run(t_scan, gather);
	}
	lap("scan");
	// 2. order
	std::vector<uint32_t> order;
	if (coded) radix_order(codes, 3 * bc_len, order);
	else switch ((bc_len + 7) / 8) {
	case 1: sorted_lines<1>(text, lines, bc_len, order); break;
	case 2: sorted_lines<2>(text, lines, bc_len, order); break;
	case 3: sorted_lines<3>(text, lines, bc_len, order); break;
	default: sorted_lines<4>(text, lines, bc_len, order); break;
	}
	lap("order");
	// 3. barcodes, in sorted order
	ema_bucket *o = (ema_bucket *)calloc(1, sizeof(ema_bucket));
	if (!o) { g_err = "out of memory"; return EMA_EIO; }
	o->n_pairs = n;
	o->bc = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
	o->off = (uint32_t *)malloc((2 * n + 1) * sizeof(uint32_t));
	o->id_off = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
	if (!o->bc || !o->off || !o->id_off) { ema_bucket_free(o); g_err = "out of memory"; return EMA_EIO; }
	std::vector<size_t> bad_bc((size_t)n_threads() + 1, (size_t)-1);
	parallel_ranges(n, 1 << 12, [&](size_t k, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			if (coded) {      // two bits a base, first base lowest (encode_bc_default): the low bits of the sort code's digits
				const uint64_t code = codes[order[i]];
				uint64_t v = 0;
				for (int t = 0; t < bc_len; ++t) v = v << 2 | ((code >> (3 * t)) & 3);
				o->bc[i] = v;
				continue;
			}
			const char *s = text + lines[order[i]].start;
			if (is_haplotag) o->bc[i] = encode_haplotag(s);
			else if (encode_default(s, bc_len, &o->bc[i])) { o->bc[i] = 0; bad_bc[k] = std::min<size_t>(bad_bc[k], order[i]); }
		}
	});
	{
		const size_t worst = *std::min_element(bad_bc.begin(), bad_bc.end());
		if (worst != (size_t)-1) {
			ema_bucket_free(o);
			return fail(EMA_EFORMAT, "line " + std::to_string(worst + 1) + ": " + kBadText[kBcBase]);
		}
	}
	lap("barcodes");
	// 4. offsets
	uint64_t nb = 0, ni = 0;
	for (size_t i = 0; i < n; ++i) {
		const Fields &f = fields[order[i]];
		o->off[2 * i] = (uint32_t)nb; nb += f.r1_l;
		o->off[2 * i + 1] = (uint32_t)nb; nb += f.r2_l;
		o->id_off[i] = (uint32_t)ni; ni += f.id_l;
		if (nb > UINT32_MAX || ni > UINT32_MAX) { ema_bucket_free(o); return fail(EMA_EARG, "more than 4 GiB of bases or identifiers in one bucket; split it"); }
	}
	o->off[2 * n] = (uint32_t)nb;
	o->id_off[n] = (uint32_t)ni;
	o->bases = (char *)malloc(nb + 1);
	o->quals = (char *)malloc(nb + 1);
	o->ids = (char *)malloc(ni + 1);
	if (!o->bases || !o->quals || !o->ids) { ema_bucket_free(o); g_err = "out of memory"; return EMA_EIO; }
	lap("offsets");
	// 5. payload, lines in file order to their sorted places
	std::vector<uint32_t> rank(n);
	parallel_ranges(n, 1 << 14, [&](size_t, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) rank[order[i]] = (uint32_t)i; });
	parallel_ranges(n, 1 << 12, [&](size_t, size_t lo, size_t hi) {
		for (size_t ln = lo; ln < hi; ++ln) {
			const size_t i = rank[ln];
			const char *s = text + lines[ln].start;
			const Fields &f = fields[ln];
			memcpy(o->bases + o->off[2 * i], s + f.r1_b, f.r1_l);
			memcpy(o->quals + o->off[2 * i], s + f.q1_b, f.r1_l);
			memcpy(o->bases + o->off[2 * i + 1], s + f.r2_b, f.r2_l);
			memcpy(o->quals + o->off[2 * i + 1], s + f.q2_b, f.r2_l);
			memcpy(o->ids + o->id_off[i], s + f.id_b, f.id_l);
		}
	});
	lap("payload");
	size_t n_groups = 0;
	for (size_t i = 0; i < n; ++i) n_groups += (i == 0 || o->bc[i] != o->bc[i - 1]);
	o->n_groups = n_groups;
	o->group_off = (uint64_t *)malloc((n_groups + 1) * sizeof(uint64_t));
	if (!o->group_off) { ema_bucket_free(o); g_err = "out of memory"; return EMA_EIO; }
	size_t g = 0;
	for (size_t i = 0; i < n; ++i) if (i == 0 || o->bc[i] != o->bc[i - 1]) o->group_off[g++] = i;
	o->group_off[n_groups] = n;
	lap("groups");
	*out = o;
	return 0;
}

int ema_bucket_read(const char *path, int bc_len, int is_haplotag, int max_read_len, ema_bucket **out)
{
	if (!out) return EMA_EARG;
	EMA_CPU(EMA_CPU_READER);
	*out = nullptr;
	g_err.clear();
	if (!path) return fail(EMA_EARG, "bad argument");
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return fail(EMA_EIO, std::string(path) + ": " + strerror(errno));
	struct stat st;
	if (fstat(fd, &st) != 0) { const int e = errno; close(fd); return fail(EMA_EIO, std::string(path) + ": " + strerror(e)); }
	int rc;
	if (S_ISREG(st.st_mode)) {
		const size_t len = (size_t)st.st_size;
		if (len == 0) { close(fd); return ema_bucket_parse(nullptr, 0, bc_len, is_haplotag, max_read_len, out); }
		// The file comes in through pread() on the host's threads, each taking a stretch, into a buffer that is kept from bucket to
		// bucket.  (Round 2 mapped the file: every 4 KB of a bucket read once is a page fault, and the faults of all parsing threads
		// queue on one lock -- 100-200 ms for a 120 MB bucket against 45 ms for the parse itself, r03.)
		// One buffer per calling thread (a stream's reader thread; concurrent readers do not queue behind each other), malloc'd without
		// a zero fill and grown geometrically; it lives as long as its thread.
		struct KeptBuf { char *p = nullptr; size_t cap = 0; ~KeptBuf() { free(p); } };
		static thread_local KeptBuf kept;
		if (kept.cap < len) {
			free(kept.p);
			kept.cap = len + (len >> 3);
			kept.p = (char *)malloc(kept.cap);
			if (!kept.p) { kept.cap = 0; close(fd); return fail(EMA_EIO, std::string(path) + ": out of memory for the read buffer"); }
		}
		char *const buf = kept.p;
		std::atomic<int> bad{0};
		parallel_ranges(len, (size_t)4 << 20, [&](size_t, size_t lo, size_t hi) {
			size_t at = lo;
			while (at < hi) {
				const ssize_t got = pread(fd, buf + at, hi - at, (off_t)at);
				if (got < 0 && errno == EINTR) continue;
				if (got <= 0) { bad.store(got < 0 ? errno : EIO); return; }
				at += (size_t)got;
			}
		});
		if (bad.load()) { const int e = bad.load(); close(fd); return fail(EMA_EIO, std::string(path) + ": read: " + strerror(e)); }
		rc = ema_bucket_parse(buf, len, bc_len, is_haplotag, max_read_len, out);
	} else {      // a pipe or a device: read it whole
		std::vector<char> buf;
		char tmp[1 << 16];
		ssize_t got;
		while ((got = read(fd, tmp, sizeof tmp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
		if (got < 0) { const int e = errno; close(fd); return fail(EMA_EIO, std::string(path) + ": " + strerror(e)); }
		rc = ema_bucket_parse(buf.data(), buf.size(), bc_len, is_haplotag, max_read_len, out);
	}
	close(fd);
	return rc;
}

// ---- `ema align -1 a.fq [-2 b.fq]`: barcode-sorted FASTQ (include/ema_ingest.h; reference src/align.c:637-744, src/techs.c:5-69)
namespace {
struct FqRec { const char *id; uint32_t id_l; const char *rd; const char *ql; uint32_t rl; uint64_t bc; };

bool slurp_file(const char *path, std::vector<char> &buf, std::string &err)
{
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { err = std::string(path) + ": " + strerror(errno); return false; }
	struct stat st;
	if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
		buf.resize((size_t)st.st_size);
		std::atomic<int> bad{0};
		parallel_ranges(buf.size(), (size_t)4 << 20, [&](size_t, size_t lo, size_t hi) {
			size_t at = lo;
			while (at < hi) {
				const ssize_t got = pread(fd, buf.data() + at, hi - at, (off_t)at);
				if (got < 0 && errno == EINTR) continue;
				if (got <= 0) { bad.store(got < 0 ? errno : EIO); return; }
				at += (size_t)got;
			}
		});
		if (bad.load()) { err = std::string(path) + ": read: " + strerror(bad.load()); close(fd); return false; }
	} else {
		char tmp[1 << 16];
		ssize_t got;
		while ((got = read(fd, tmp, sizeof tmp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
		if (got < 0) { err = std::string(path) + ": " + strerror(errno); close(fd); return false; }
	}
	close(fd);
	return true;
}

// the records of one FASTQ text; false + message on a malformed one
bool parse_fastq(const char *what, const std::vector<char> &buf, int name_style, int bc_len, int is_haplotag, int max_read_len,
                 std::vector<FqRec> &out, std::string &err)
{
	const char *p = buf.data(), *end = p + buf.size();
	size_t n_rec = 0;
	auto line = [&](const char *&b, uint32_t &l) -> bool {      // next line without its newline (and without a '\r' before it)
		if (p >= end) return false;
		const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
		b = p;
		const char *e = q ? q : end;
		p = q ? q + 1 : end;
		if (e > b && e[-1] == '\r') --e;
		l = (uint32_t)(e - b);
		return true;
	};
	auto bad = [&](const char *m) { err = std::string(what) + ": record " + std::to_string(n_rec + 1) + ": " + m; return false; };
	for (;;) {
		const char *id, *rd, *sp, *ql;
		uint32_t id_l, rl, sl, qlen;
		if (!line(id, id_l)) break;
		if (id_l == 0 && p >= end) break;      // a trailing empty line
		if (!line(rd, rl) || !line(sp, sl) || !line(ql, qlen)) return bad("truncated (the reference asserts, src/align.c:644-646)");
		if (id_l < 2 || id[0] != '@') return bad("the name line does not start with '@'");
		if (id_l > kMaxId) return bad("name of 150 bytes or more (id[150], include/samrecord.h:12)");
		if (rl > (uint32_t)max_read_len) return bad("read longer than the limit");
		if (qlen != rl) return bad("quality string and read differ in length");
		FqRec r;
		r.rd = rd; r.ql = ql; r.rl = rl;
		// barcode and identifier, as the platform's extract_bc leaves them (src/techs.c:5-54)
		const char *name_end = id + id_l;                 // the identifier ends here ...
		const char *bc = nullptr;
		const char *space = (const char *)memchr(id, ' ', id_l);
		if (name_style == 2 || name_style == 3) {
			// TruSeq SLR / CPT-seq: integer barcodes.  extract_bc_truseq: atoi() of the name behind its '@', the name left whole;
			// extract_bc_cptseq: the name cut at its last ':', atoi() of what follows that ':' and two more characters (src/techs.c:56-68)
			const char *num = id + 1;
			if (name_style == 3) {
				const char *colon = nullptr;
				for (const char *c = id + id_l; c-- > id;) if (*c == ':') { colon = c; break; }
				if (!colon) return bad("no ':' before the barcode in the name (the reference asserts, src/techs.c:65)");
				name_end = colon;
				num = colon + 3 <= id + id_l ? colon + 3 : id + id_l;
			}
			// atoi: blanks, a sign, digits (the line's end stops it: the buffer the reference parses ends in '\n')
			const char *c = num, *lim = id + id_l;
			while (c < lim && is_space((unsigned char)*c)) ++c;
			bool neg = false;
			if (c < lim && (*c == '-' || *c == '+')) { neg = *c == '-'; ++c; }
			long long v = 0;
			while (c < lim && *c >= '0' && *c <= '9' && v < (1LL << 40)) { v = v * 10 + (*c - '0'); ++c; }
			if (v > 2147483647LL) return bad("barcode number beyond int (the reference's atoi is undefined there)");
			r.bc = (uint64_t)(int64_t)(int)(neg ? -v : v);      // bc_t = uint64_t of an int
			r.id = id; r.id_l = (uint32_t)(name_end - id);
			if (r.id_l < 2) return bad("empty identifier");
			out.push_back(r);
			++n_rec;
			continue;
		}
		if (name_style == 1 && space && (size_t)(id + id_l - space) >= 6 && memcmp(space, " BX:Z:", 6) == 0) {      // tellseq, Long Ranger basic format
			const char *colon = nullptr;
			for (const char *c = id + id_l; c-- > space;) if (*c == ':') { colon = c; break; }
			bc = colon + 1;
			name_end = space;
		} else {
			const char *lim = id + id_l;
			if (name_style == 1 && space) lim = space;      // tellseq: text after the first blank is dropped first
			const char *colon = nullptr;
			for (const char *c = lim; c-- > id;) if (*c == ':') { colon = c; break; }
			if (!colon) return bad("no ':' before the barcode in the name (the reference asserts, src/techs.c:8,21)");
			bc = colon + 1;
			name_end = colon;
			if (space && space < name_end) name_end = space;
		}
		if ((size_t)(id + id_l - bc) < (size_t)bc_len) return bad("barcode shorter than the platform's");
		if (is_haplotag) r.bc = encode_haplotag(bc);
		else if (encode_default(bc, bc_len, &r.bc)) return bad("barcode with a character outside ACGT (the reference asserts, src/util.c:54)");
		r.id = id; r.id_l = (uint32_t)(name_end - id);
		if (r.id_l < 2) return bad("empty identifier");
		out.push_back(r);
		++n_rec;
	}
	return true;
}
}  // namespace

int ema_fastq_read(const char *path1, const char *path2, int name_style, int bc_len, int is_haplotag, int max_read_len, ema_bucket **out)
{
	if (!out) return EMA_EARG;
	EMA_CPU(EMA_CPU_READER);
	*out = nullptr;
	g_err.clear();
	const bool numbered = name_style == 2 || name_style == 3;      // (integer barcodes: the platform's bc_len is 0)
	if (!path1 || bc_len < (numbered ? 0 : 1) || bc_len > 32 || max_read_len < 1 || max_read_len > 4096 || (is_haplotag && bc_len != 12) || name_style < 0 || name_style > 3)
		return fail(EMA_EARG, "bad argument");
	std::vector<char> t1, t2;
	std::string err;
	if (!slurp_file(path1, t1, err) || (path2 && !slurp_file(path2, t2, err))) return fail(EMA_EIO, err);
	std::vector<FqRec> r1, r2;
	if (!parse_fastq(path1, t1, name_style, bc_len, is_haplotag, max_read_len, r1, err)) return fail(EMA_EFORMAT, err);
	if (path2 && !parse_fastq(path2, t2, name_style, bc_len, is_haplotag, max_read_len, r2, err)) return fail(EMA_EFORMAT, err);
	size_t n;
	std::vector<FqRec> m1, m2;
	const std::vector<FqRec> *a = &r1, *b = &r2;
	if (!path2) {      // interleaved: mate 1, mate 2, mate 1, ...
		if (r1.size() & 1) return fail(EMA_EFORMAT, std::string(path1) + ": an odd number of records in an interleaved file");
		for (size_t i = 0; i < r1.size(); i += 2) { m1.push_back(r1[i]); m2.push_back(r1[i + 1]); }
		a = &m1; b = &m2;
	}
	n = a->size();
	if (b->size() != n) return fail(EMA_EFORMAT, "the two files hold different numbers of records (the reference asserts, src/align.c:341)");
	for (size_t i = 0; i < n; ++i) {
		const FqRec &x = (*a)[i], &y = (*b)[i];
		if (x.bc != y.bc) return fail(EMA_EFORMAT, "pair " + std::to_string(i + 1) + ": the mates carry different barcodes (the reference asserts, src/align.c:708,733)");
		if (x.id_l != y.id_l || memcmp(x.id, y.id, x.id_l) != 0) return fail(EMA_EFORMAT, "pair " + std::to_string(i + 1) + ": the mates' identifiers differ");
	}
	ema_bucket *o = (ema_bucket *)calloc(1, sizeof(ema_bucket));
	if (!o) { g_err = "out of memory"; return EMA_EIO; }
	o->n_pairs = n;
	o->bc = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
	o->off = (uint32_t *)malloc((2 * n + 1) * sizeof(uint32_t));
	o->id_off = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
	if (!o->bc || !o->off || !o->id_off) { ema_bucket_free(o); g_err = "out of memory"; return EMA_EIO; }
	uint64_t nb = 0, ni = 0;
	for (size_t i = 0; i < n; ++i) {
		o->bc[i] = (*a)[i].bc;
		o->off[2 * i] = (uint32_t)nb; nb += (*a)[i].rl;
		o->off[2 * i + 1] = (uint32_t)nb; nb += (*b)[i].rl;
		o->id_off[i] = (uint32_t)ni; ni += (*a)[i].id_l;
		if (nb > UINT32_MAX || ni > UINT32_MAX) { ema_bucket_free(o); return fail(EMA_EARG, "more than 4 GiB of bases or identifiers in one input; split it"); }
	}
	o->off[2 * n] = (uint32_t)nb; o->id_off[n] = (uint32_t)ni;
	o->bases = (char *)malloc(nb + 1); o->quals = (char *)malloc(nb + 1); o->ids = (char *)malloc(ni + 1);
	if (!o->bases || !o->quals || !o->ids) { ema_bucket_free(o); g_err = "out of memory"; return EMA_EIO; }
	parallel_ranges(n, 1 << 12, [&](size_t, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			const FqRec &x = (*a)[i], &y = (*b)[i];
			memcpy(o->bases + o->off[2 * i], x.rd, x.rl); memcpy(o->quals + o->off[2 * i], x.ql, x.rl);
			memcpy(o->bases + o->off[2 * i + 1], y.rd, y.rl); memcpy(o->quals + o->off[2 * i + 1], y.ql, y.rl);
			memcpy(o->ids + o->id_off[i], x.id, x.id_l);
		}
	});
	size_t n_groups = 0;
	for (size_t i = 0; i < n; ++i) n_groups += (i == 0 || o->bc[i] != o->bc[i - 1]);      // runs in FILE order: the input is trusted to be sorted
	o->n_groups = n_groups;
	o->group_off = (uint64_t *)malloc((n_groups + 1) * sizeof(uint64_t));
	if (!o->group_off) { ema_bucket_free(o); g_err = "out of memory"; return EMA_EIO; }
	size_t g = 0;
	for (size_t i = 0; i < n; ++i) if (i == 0 || o->bc[i] != o->bc[i - 1]) o->group_off[g++] = i;
	o->group_off[n_groups] = n;
	*out = o;
	return 0;
}

}  // extern "C"
