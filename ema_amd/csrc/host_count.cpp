// ema_amd/csrc/host_count.cpp -- `ema count` behind the C ABI of include/ema_count.h (reference cpp/count.cc:38-182).
//
// What the reference does with std::getline on std::cin, one line at a time, is one pass over a large read buffer here
// (memchr for the line ends); the two containers whose iteration order IS the output order are the reference's own kinds --
// std::unordered_map<uint32_t, int64_t> filled in whitelist order (the .ema-ncnt order is its iteration order) and
// filled in whitelist order (the .ema-ncnt order is its iteration order).  The reference's second container, a
// std::map<std::string, int64_t> whose blocks go to .ema-fcnt sorted, is a hash table of 16-byte keys here that is sorted
// (bytes compared as unsigned, as std::string compares) only when a block is written: a block's content and its boundary --
// the number of distinct keys seen since the last one -- do not depend on the container.  Counting is memory-latency work
// on two look-ups per pair; it stays sequential because the block boundaries are defined by the order of first occurrences.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <unistd.h>
#include "ema_count.h"

namespace {

thread_local std::string g_err;

const int kBcLen = 16, kQualOffset = 33, kQualBase = 34, kMinRead = 32;      // cpp/common.h:56-63

inline int code2(unsigned char c)      // hash_dna: A 0, C 1, G 2, T 3 (either case), anything else 0
{
	switch (c) { case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 0; }
}
inline int code2n(unsigned char c)     // hash_dna_n: the same with N = 4
{
	return (c == 'N' || c == 'n') ? 4 : code2(c);
}

// the lines of a stream, as std::getline sees them: a last line without '\n' counts, "\n" at the very end adds none
struct Lines {
	int fd;
	std::vector<char> buf;
	size_t lo = 0, hi = 0;
	bool eof = false;
	explicit Lines(int fd_) : fd(fd_), buf((size_t)16 << 20) {}
	bool fill()
	{
		if (eof) return false;
		if (lo > 0) { memmove(buf.data(), buf.data() + lo, hi - lo); hi -= lo; lo = 0; }
		if (hi == buf.size()) buf.resize(buf.size() * 2);
		for (;;) {
			const ssize_t n = read(fd, buf.data() + hi, buf.size() - hi);
			if (n < 0 && (errno == EINTR || errno == EAGAIN)) continue;
			if (n <= 0) { eof = true; return false; }
			hi += (size_t)n;
			return true;
		}
	}
	// next line -> (ptr, len); false at the end of the stream (then len = 0, like getline's erased string)
	bool next(const char *&p, size_t &len)
	{
		for (;;) {
			const char *nl = lo < hi ? (const char *)memchr(buf.data() + lo, '\n', hi - lo) : nullptr;
			if (nl) { p = buf.data() + lo; len = (size_t)(nl - p); lo += len + 1; return true; }
			if (!fill()) {
				if (lo < hi) { p = buf.data() + lo; len = hi - lo; lo = hi; return true; }
				p = nullptr; len = 0;
				return false;
			}
		}
	}
};

bool write_all(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n; }

struct Key16 {
	uint64_t a, b;      // the 16 bytes, each half big-endian so that integer order is byte order
	bool operator==(const Key16 &o) const { return a == o.a && b == o.b; }
	bool operator<(const Key16 &o) const { return a != o.a ? a < o.a : b < o.b; }
};
// the 16-byte keys with their counts: open addressing, linear probing, doubled at half full (a count of 0 = an empty slot).
// Its layout is free: a block is sorted when it is written.
struct FullMap {
	std::vector<Key16> keys;
	std::vector<int64_t> cnt;
	size_t n = 0, mask = 0;
	FullMap() { resize_to((size_t)1 << 16); }
	static size_t hash(const Key16 &k) { uint64_t h = k.a * 0x9E3779B97F4A7C15ULL ^ (k.b + 0x7F4A7C15ULL) * 0xD6E8FEB86659FD93ULL; return (size_t)(h ^ h >> 29); }
	void resize_to(size_t cap)
	{
		std::vector<Key16> ok; std::vector<int64_t> oc;
		ok.swap(keys); oc.swap(cnt);
		keys.assign(cap, Key16{0, 0}); cnt.assign(cap, 0); mask = cap - 1;
		for (size_t i = 0; i < oc.size(); ++i) if (oc[i]) { size_t at = hash(ok[i]) & mask; while (cnt[at]) at = (at + 1) & mask; keys[at] = ok[i]; cnt[at] = oc[i]; }
	}
	size_t size() const { return n; }
	int64_t bump(const Key16 &k)      // the count before the increment (0: a new key)
	{
		size_t at = hash(k) & mask;
		while (cnt[at] && !(keys[at] == k)) at = (at + 1) & mask;
		const int64_t before = cnt[at];
		if (!before) { keys[at] = k; ++n; }
		++cnt[at];
		if (!before && 2 * n > mask) resize_to(2 * (mask + 1));
		return before;
	}
	void clear() { std::fill(cnt.begin(), cnt.end(), 0); n = 0; }
};

bool dump_block(FullMap &full, FILE *fo)      // dump_map, cpp/count.cc:18-34: the block sorted by key
{
	const int64_t n = (int64_t)full.size();
	if (!write_all(fo, &n, 8)) return false;
	std::vector<std::pair<Key16, int64_t>> v;
	v.reserve(full.size());
	for (size_t i = 0; i < full.cnt.size(); ++i) if (full.cnt[i]) v.emplace_back(full.keys[i], full.cnt[i]);
	std::sort(v.begin(), v.end(), [](const std::pair<Key16, int64_t> &x, const std::pair<Key16, int64_t> &y) { return x.first < y.first; });
	std::vector<unsigned char> out((size_t)n * 24);
	for (size_t i = 0; i < v.size(); ++i) {
		unsigned char *o = out.data() + i * 24;
		for (int k = 0; k < 8; ++k) { o[k] = (unsigned char)(v[i].first.a >> (56 - 8 * k)); o[8 + k] = (unsigned char)(v[i].first.b >> (56 - 8 * k)); }
		memcpy(o + 16, &v[i].second, 8);
	}
	if (!out.empty() && !write_all(fo, out.data(), out.size())) return false;
	fflush(fo);
	full.clear();
	return true;
}

}  // namespace

extern "C" const char *ema_count_last_error(void) { return g_err.c_str(); }

extern "C" int ema_count_fastq(const char *known_barcodes_path, int in_fd, const char *output_prefix, size_t max_map_size,
                               int is_haplotag, ema_count_stats *st)
{
	g_err.clear();
	if (!output_prefix || in_fd < 0 || (!is_haplotag && !known_barcodes_path)) { g_err = "bad argument"; return EMA_EARG; }
	std::unordered_map<uint32_t, int64_t> counts;      // filled in the reference's order: its iteration order is the output order
	FullMap full;
	ema_count_stats s;
	memset(&s, 0, sizeof(s));
	if (!is_haplotag) {
		FILE *f = fopen(known_barcodes_path, "rb");
		if (!f) { g_err = std::string("Cannot open file ") + known_barcodes_path; return EMA_EIO; }
		Lines wl(fileno(f));
		const char *p; size_t len;
		while (wl.next(p, len)) {
			uint32_t bc = 0;
			for (int i = 0; i < kBcLen; ++i) bc = (bc << 2) | (uint32_t)code2((size_t)i < len ? (unsigned char)p[i] : 0);
			if (bc == 0) { fclose(f); g_err = "Invalid barcode AAA...AA whitelisted"; return EMA_EFORMAT; }
			counts[bc] = 0;
		}
		fclose(f);
	} else {
		for (uint32_t a = 1; a <= 96; ++a) for (uint32_t b = 1; b <= 96; ++b) for (uint32_t c = 1; c <= 96; ++c) for (uint32_t d = 1; d <= 96; ++d)
			counts[a << 24 | c << 16 | b << 8 | d] = 0;      // GenerateAllHaplotagBC, cpp/common.h:72
	}
	s.whitelist = (int64_t)counts.size();
	const std::string p_full = std::string(output_prefix) + ".ema-fcnt", p_nice = std::string(output_prefix) + ".ema-ncnt";
	FILE *f_full = nullptr;
	if (!is_haplotag) {
		f_full = fopen(p_full.c_str(), "wb");
		if (!f_full) { g_err = "Cannot open file " + p_full; return EMA_EIO; }
	}
	FILE *f_nice = fopen(p_nice.c_str(), "wb");
	if (!f_nice) { if (f_full) fclose(f_full); g_err = "Cannot open file " + p_nice; return EMA_EIO; }
	auto fail = [&](const char *what) { if (f_full) fclose(f_full); fclose(f_nice); g_err = what; return EMA_EIO; };

	// The stream is taken in blocks that end on a pair boundary (pairs are eight lines: the boundaries come from counting line
	// ends); the pairs of a block are parsed on the host's threads into {counted?, 2-bit code, N?, the 16-byte key}, and one thread
	// then makes the two table updates per pair in input order -- the order that defines the blocks of .ema-fcnt.
	int n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	if (const char *v = getenv("EMA_HOST_THREADS")) n_threads = std::max(1, std::min(64, atoi(v)));
	struct LineRef { const char *p; size_t len; };
	struct Parsed { Key16 key; uint32_t code; uint8_t process, has_n; };
	std::vector<char> blk((size_t)64 << 20);
	std::vector<size_t> nl;
	std::vector<Parsed> parsed;
	size_t have = 0;
	bool eof = false;
	std::string prev_q1;      // line 4 (mate 1's qualities) of the last pair seen: what the reference's `q` still holds when a getline fails
	while (!eof || have > 0) {
		while (!eof && have < blk.size()) {
			const ssize_t got = read(in_fd, blk.data() + have, blk.size() - have);
			if (got < 0 && (errno == EINTR || errno == EAGAIN)) continue;
			if (got <= 0) { eof = true; break; }
			have += (size_t)got;
		}
		nl.clear();
		for (const char *q0 = blk.data(), *e = blk.data() + have; q0 < e;) {
			const char *x = (const char *)memchr(q0, '\n', (size_t)(e - q0));
			if (!x) break;
			nl.push_back((size_t)(x - blk.data()));
			q0 = x + 1;
		}
		size_t n_lines = nl.size();
		const size_t tail_at = n_lines ? nl.back() + 1 : 0;
		if (eof && tail_at < have) ++n_lines;      // a last line without its line end: std::getline returns it all the same
		size_t n_rec = n_lines / 8;
		const size_t missing = (eof && n_lines % 8) ? 8 - n_lines % 8 : 0;      // the stream ends inside a pair
		if (missing) ++n_rec;
		// What the reference reads for the missing lines (cpp/count.cc:86-108): a getline that fails BEFORE the stream has hit its end
		// erases its string and then sets the end flag -- the line reads as empty -- but once the flag is set (the last line present had
		// no line end, so reading it hit the end) every further getline leaves its string untouched: the read then takes the NAME
		// line for its bases (one line present) and the PREVIOUS pair's mate-1 qualities for its own (one or two lines present).
		const bool open_tail = eof && tail_at < have;
		const size_t stale_lines = (missing && open_tail) ? n_lines % 8 : 0;      // 1 or 2: the stale strings matter; 3 and up: they do not
		if (n_rec == 0) {
			if (eof) break;
			blk.resize(blk.size() * 2);
			continue;
		}
		auto line = [&](size_t i) -> LineRef {
			if (i >= n_lines) return LineRef{"", 0};
			const size_t beg = i ? nl[i - 1] + 1 : 0;
			const size_t end = i < nl.size() ? nl[i] : have;
			return LineRef{blk.data() + beg, end - beg};
		};
		parsed.resize(n_rec);
		const size_t per = (n_rec + (size_t)n_threads - 1) / (size_t)n_threads;
		auto work = [&](int t) {
			const size_t lo = std::min(n_rec, (size_t)t * per), hi = std::min(n_rec, (size_t)(t + 1) * per);
			for (size_t rec = lo; rec < hi; ++rec) {
				const LineRef name = line(8 * rec);
				LineRef seq = line(8 * rec + 1), q = line(8 * rec + 3);
				if (stale_lines && rec + 1 == n_rec) {      // the cut-short last pair of a stream without a final line end
					if (stale_lines == 1) seq = name;
					if (stale_lines <= 2) {
						if (rec > 0) q = line(8 * rec - 5); else q = LineRef{prev_q1.data(), prev_q1.size()};
					}
				}
				Parsed &o = parsed[rec];
				bool bx = false;
				uint32_t barcode = 0;
				if (is_haplotag) {      // cpp/count.cc:91-103
					size_t at = 0;
					while (at < name.len && name.p[at] != ' ' && name.p[at] != '\t') ++at;
					if (at < name.len) {
						size_t tag = std::string::npos;
						for (size_t i = at; i + 5 <= name.len; ++i) if (memcmp(name.p + i, "BX:Z:", 5) == 0) { tag = i; break; }
						if (tag != std::string::npos && tag + 16 < name.len) {
							const char *h = name.p + tag + 5;      // substr(tag + 5, 12) of a string with at least tag + 17 characters: all twelve are there
							auto two = [&](int i) { return 10 * (h[i] - '0') + (h[i + 1] - '0'); };      // TwoCharToInt
							barcode = (uint32_t)two(1) << 24 | (uint32_t)two(4) << 16 | (uint32_t)two(7) << 8 | (uint32_t)two(10);
							bx = true;
						}
					}
				} else bx = true;
				bool process = bx && seq.len >= (size_t)kMinRead;
				bool has_n = false;
				unsigned char b[16];
				memset(b, '#', 16);      // (haplotag: the reference's key string is never written)
				if (!is_haplotag) {
					barcode = 0;
					if (process) for (int i = 0; i < kBcLen; ++i) {
						int qc = (size_t)i < q.len ? (signed char)q.p[i] : 0;      // (std::string's terminator; further out the reference reads beyond its string)
						if (qc < kQualOffset) { process = false; break; }      // "Ignoring long read--- quality score ... less than 33"
						if (qc - kQualOffset >= kQualBase) qc = kQualOffset + kQualBase - 1;
						const unsigned char base = (unsigned char)seq.p[i];
						const int qv = qc - kQualOffset < kQualBase - 1 ? qc - kQualOffset : kQualBase - 1;
						b[i] = (unsigned char)(code2n(base) * kQualBase + qv);
						barcode = (barcode << 2) | (uint32_t)code2(base);
						has_n |= base == 'N';
					}
				}
				o.process = process; o.has_n = has_n; o.code = barcode;
				o.key.a = o.key.b = 0;
				for (int i = 0; i < 8; ++i) { o.key.a = o.key.a << 8 | b[i]; o.key.b = o.key.b << 8 | b[8 + i]; }
			}
		};
		{
			std::vector<std::thread> th;
			for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
			work(0);
			for (auto &x : th) x.join();
		}
		for (size_t rec = 0; rec < n_rec; ++rec) {
			const Parsed &o = parsed[rec];
			if (!o.process) { ++s.ignored_reads; continue; }
			if (!o.has_n) {
				auto it = counts.find(o.code);
				if (it != counts.end()) { ++it->second; ++s.nice_reads; }
			}
			const int64_t cnt = full.bump(o.key);
			if (!cnt && (sizeof(std::string) + sizeof(int64_t) + 32) * full.size() >= max_map_size) {      // a new element: estimate_size of the reference's std::map<std::string, int64_t>, cpp/common.h:110-115
				if (!f_full) return fail("the barcode map outgrew max_map_size in haplotag mode (the reference writes through an unopened file there)");
				if (!dump_block(full, f_full)) return fail("fwrite failed");
				++s.full_blocks;
			}
			++s.total_reads;
		}
		if (n_rec) { const LineRef l4 = line(8 * (n_rec - 1) + 3); if (8 * (n_rec - 1) + 3 < n_lines) prev_q1.assign(l4.p, l4.len); }
		const size_t used = 8 * n_rec <= nl.size() ? nl[8 * n_rec - 1] + 1 : have;
		s.bytes += (int64_t)used + (int64_t)missing + ((eof && tail_at < have) ? 1 : 0);      // the reference's `sz`: every line's length + 1, failed getlines included
		memmove(blk.data(), blk.data() + used, have - used);
		have -= used;
	}
	int64_t nice = 0;
	for (auto &kv : counts) if (kv.second) ++nice;
	if (!write_all(f_nice, &nice, 8)) return fail("fwrite failed");
	for (auto &kv : counts) if (kv.second) {
		if (!write_all(f_nice, &kv.first, 4) || !write_all(f_nice, &kv.second, 8)) return fail("fwrite failed");
	}
	s.nice_barcodes = nice;
	if (fclose(f_nice) != 0) { f_nice = nullptr; if (f_full) fclose(f_full); g_err = "cannot write " + p_nice; return EMA_EIO; }
	if (!is_haplotag) {
		if (!dump_block(full, f_full)) { fclose(f_full); g_err = "fwrite failed"; return EMA_EIO; }
		++s.full_blocks;
		if (fclose(f_full) != 0) { g_err = "cannot write " + p_full; return EMA_EIO; }
	}
	if (st) *st = s;
	return EMA_OK;
}
