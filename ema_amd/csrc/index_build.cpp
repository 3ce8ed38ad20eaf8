// ema_amd/csrc/index_build.cpp -- FM-index builder (host, C++/OpenMP).
//
// The reference loads a pre-built bwa index with bwa_idx_load(path, BWA_IDX_ALL)
// (reference src/bwabridge.c:77-96); the tool that builds it (`bwa index`) lives
// in the un-vendored lh3/bwa submodule and does not exist in this image, so the
// engine ships its own builder.  It writes bwa's on-disk layout:
//   <fa>.pac  forward strand, 2 bit/base, first base in the high bits, + tail bytes
//   <fa>.ann  <fa>.amb  text annotations / holes (N runs replaced by lrand48()&3, srand48(11))
//   <fa>.bwt  u64 primary, u64 L2[1..4], then per 128 bases: 4 x u64 occ + 8 x u32 bases
//   <fa>.sa   u64 primary, 4 x u64 L2, u64 sa_intv(32), u64 seq_len, sampled SA
//   <fa>.fai  name, length, offset, linebases, linewidth (reference src/main.c:57-71 reads column 1)
// plus one file of our own, used by the HIP engine:
//   <fa>.fsa  'EMAFSA01', u64 seq_len, u64 width(4|8), then SA[0..seq_len] uncompressed
// (with 288 GB of HBM per GPU the suffix array is kept whole, so locating an
// occurrence is one load instead of up to 31 dependent LF steps).
//
// Text = forward strand followed by its reverse complement (seq_len = 2*l_pac),
// '$' smallest, row 0 = the empty suffix.  Construction: counting sort on the first
// K bases, then an independent comparison sort of every bucket on 2-bit packed words.

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <dlfcn.h>

namespace {

struct Contig { std::string name, anno; int64_t offset; int32_t len, n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };

// Large arrays ask for huge pages: they are gathered from / scattered into at random, and 4 KB pages make every access
// a TLB miss.
inline void want_huge_pages(void *p, size_t bytes)
{
	const uintptr_t two_mb = (uintptr_t)2 << 20, lo = ((uintptr_t)p + two_mb - 1) & ~(two_mb - 1), hi = ((uintptr_t)p + bytes) & ~(two_mb - 1);
	if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
}

struct Packed {            // big-endian 2-bit text: base t of a word sits at bits 62-2t
	std::vector<uint64_t> w;
	uint64_t n = 0;
	void init(uint64_t n_) { n = n_; w.resize((n_ >> 5) + 3); want_huge_pages(w.data(), w.size() * 8); std::fill(w.begin(), w.end(), 0); }
	inline void set(uint64_t i, unsigned c) { w[i >> 5] |= (uint64_t)c << (62 - ((i & 31) << 1)); }
	inline unsigned get(uint64_t i) const { return w[i >> 5] >> (62 - ((i & 31) << 1)) & 3; }
	inline uint64_t get32(uint64_t i) const {   // 32 bases starting at i (garbage past n is zero)
		unsigned s = (i & 31) << 1;
		uint64_t a = w[i >> 5];
		return s ? (a << s) | (w[(i >> 5) + 1] >> (64 - s)) : a;
	}
};

inline bool suffix_less(const Packed &T, uint64_t i, uint64_t j)
{
	if (i == j) return false;
	const uint64_t n = T.n;
	for (;;) {
		uint64_t ri = n - i, rj = n - j;
		uint64_t a = T.get32(i), b = T.get32(j);
		uint64_t m = ri < rj ? ri : rj;
		if (m >= 32) {
			if (a != b) return a < b;
			i += 32; j += 32;
			if (i == n) return true;      // i exhausted first: shorter suffix is smaller
			if (j == n) return false;
		} else {
			if (m == 0) return ri < rj;
			uint64_t mask = ~0ULL << (64 - 2 * m);
			if ((a & mask) != (b & mask)) return (a & mask) < (b & mask);
			return ri < rj;
		}
	}
}

// Big arrays are allocated without being filled: a serial zero-fill of 50 GB costs more than some of the phases, and
// every element is written by the phase that owns the array.
template <typename V> struct Raw {
	V *p = nullptr;
	explicit Raw(uint64_t n)
	{
		const size_t bytes = (size_t)(n ? n : 1) * sizeof(V);
		if (bytes >= ((size_t)64 << 20)) {
			void *q = nullptr;
			if (posix_memalign(&q, (size_t)2 << 20, bytes) == 0) { p = (V *)q; want_huge_pages(q, bytes); }
		} else p = (V *)malloc(bytes);
	}
	~Raw() { free(p); }
	Raw(const Raw &) = delete;
	Raw &operator=(const Raw &) = delete;
};

// header + payload into a file, the payload in 64 MB pieces from several threads (one fwrite fills the page cache at a
// couple of GB/s; the flat suffix array of a human-size genome is 50 GB)
bool write_file(const std::string &path, const void *head, size_t n_head, const void *data, size_t n_data)
{
	const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (fd < 0) return false;
	bool ok = n_head == 0 || pwrite(fd, head, n_head, 0) == (ssize_t)n_head;
	const size_t piece = (size_t)64 << 20, n_pieces = (n_data + piece - 1) / piece;
	int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(16, omp_get_max_threads())) reduction(+ : bad)
	for (int64_t k = 0; k < (int64_t)n_pieces; ++k) {
		size_t at = (size_t)k * piece;
		const size_t end = std::min(n_data, at + piece);
		while (at < end) {
			const ssize_t w = pwrite(fd, (const char *)data + at, end - at, (off_t)(n_head + at));
			if (w <= 0) { ++bad; break; }
			at += (size_t)w;
		}
	}
	return close(fd) == 0 && ok && bad == 0;
}

struct Lap {      // phase timings on stderr when EMA_INDEX_PROF is set
	bool on = getenv("EMA_INDEX_PROF") != nullptr;
	double t0 = omp_get_wtime();
	void operator()(const char *what) { if (on) { const double t = omp_get_wtime(); fprintf(stderr, "[index] %-12s %7.1f s\n", what, t - t0); t0 = t; } }
};

// Suffix array of T (sa[0..n]: row 0 = '$').  Every phase runs on all host threads:
//   1. counting sort of the positions by their first K <= 8 bases, with per-thread histograms (each thread owns a
//      stretch of the text, so the scatter is stable and needs no atomics);
//   2. each bucket on its own: (next 32 bases as one word, position) pairs sorted by the word, the rare ties settled by
//      comparing the suffixes themselves -- almost all comparisons stay inside the pair array instead of gathering
//      from the text.
template <typename I>
void build_sa(const Packed &T, I *sa)
{
	const uint64_t n = T.n;
	sa[0] = (I)n;
	int K = 1;
	while (K < 8 && (1ULL << (2 * (K + 1))) <= n / 4 + 4) ++K;
	const uint64_t nb = 1ULL << (2 * K);
	auto key_at = [&](uint64_t i) { return T.get32(i) >> (64 - 2 * K); };   // zero (=A) padded past n
	const int n_thr = omp_get_max_threads();
	const uint64_t per = (n + (uint64_t)n_thr - 1) / (uint64_t)n_thr;
	std::vector<std::vector<uint64_t>> hist((size_t)n_thr);
	std::vector<uint64_t> cnt(nb + 1, 0);
	Lap lap;
	// The text is cut into n_thr stretches ("slots"); the team works through them whatever its actual size turns out to be.
#pragma omp parallel num_threads(n_thr)
	{
		const int t0 = omp_get_thread_num(), team = omp_get_num_threads();
		for (int v = t0; v < n_thr; v += team) {
			std::vector<uint64_t> &h = hist[(size_t)v];
			h.assign(nb, 0);
			const uint64_t lo = std::min(n, (uint64_t)v * per), hi = std::min(n, lo + per);
			for (uint64_t i = lo; i < hi; ++i) ++h[key_at(i)];
		}
#pragma omp barrier
#pragma omp for schedule(static)
		for (int64_t b = 0; b < (int64_t)nb; ++b) {      // bucket totals, and each slot's share turned into its offset inside the bucket
			uint64_t run = 0;
			for (int u = 0; u < n_thr; ++u) { const uint64_t c = hist[(size_t)u][(size_t)b]; hist[(size_t)u][(size_t)b] = run; run += c; }
			cnt[(size_t)b + 1] = run;
		}
#pragma omp single
		for (uint64_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
		I *base = sa + 1;
		for (int v = t0; v < n_thr; v += team) {
			std::vector<uint64_t> &h = hist[(size_t)v];
			const uint64_t lo = std::min(n, (uint64_t)v * per), hi = std::min(n, lo + per);
			for (uint64_t i = lo; i < hi; ++i) { const uint64_t k = key_at(i); base[cnt[k] + h[k]++] = (I)i; }
		}
	}
	hist.clear(); hist.shrink_to_fit();
	lap("bucket");
	I *base = sa + 1;
	struct KP { uint64_t key; I pos; };
#pragma omp parallel
	{
		std::vector<KP> kp;
#pragma omp for schedule(dynamic, 16)
		for (int64_t b = 0; b < (int64_t)nb; ++b) {
			const uint64_t lo = cnt[(size_t)b], hi = cnt[(size_t)b + 1];
			if (hi - lo < 2) continue;
			kp.resize(hi - lo);
			for (uint64_t r = lo; r < hi; ++r) {
				const uint64_t p = (uint64_t)base[r], q = p + (uint64_t)K;
				kp[r - lo] = KP{q < n ? T.get32(q) : 0, (I)p};
			}
			std::sort(kp.begin(), kp.end(), [&](const KP &x, const KP &y) {
				if (x.key != y.key) return x.key < y.key;      // (a suffix that ends inside the word is zero-padded: a proper prefix sorts first, as '$' does)
				return suffix_less(T, (uint64_t)x.pos, (uint64_t)y.pos);
			});
			for (uint64_t r = lo; r < hi; ++r) base[r] = kp[r - lo].pos;
		}
	}
	lap("sort");
}

void fput64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }

// The suffix array on the GPU (k_sa.hip in libema_engine.so, next to this library): same rows as build_sa, plus the base before every
// row's suffix.  This library is plain host code (it also runs where there is no GPU), so the entry point is looked up at run time;
// without the engine library, without a device, with EMA_INDEX_GPU=0 or on any failure there the host's cores sort as before.
typedef int (*gpu_sa_fn)(const uint64_t *text, uint64_t n, int width, void *rows, uint8_t *prev, int verbose);
gpu_sa_fn find_gpu_sa()
{
	const char *v = getenv("EMA_INDEX_GPU");
	if (v && atoi(v) == 0) return nullptr;
	Dl_info me;
	if (!dladdr((void *)&find_gpu_sa, &me) || !me.dli_fname) return nullptr;
	std::string dir(me.dli_fname);
	const size_t slash = dir.rfind('/');
	dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
	const char *lib = getenv("EMA_ENGINE_LIB");
	void *h = dlopen((dir + "/" + (lib ? lib : "libema_engine.so")).c_str(), RTLD_NOW | RTLD_LOCAL);
	if (!h) return nullptr;
	return (gpu_sa_fn)dlsym(h, "ema_gpu_suffix_array");
}

// which suffix-array builder ran, on stderr -- never silently the slower one
void say_builder(bool on_gpu, bool gpu_entry_found)
{
	const char *v = getenv("EMA_INDEX_GPU");
	fprintf(stderr, "[ema_index_build] suffix array built %s\n", on_gpu ? "on the GPU (k_sa.hip)"
	        : (v && atoi(v) == 0) ? "on the host's threads (EMA_INDEX_GPU=0)"
	        : gpu_entry_found ? "on the host's threads: the GPU builder failed (no device, or not enough device memory)"
	        : "on the host's threads: libema_engine.so was not found next to this library");
}

template <typename I>
int write_index(const std::string &prefix, const Packed &T, const I *sa, const uint8_t *prev = nullptr)      // prev: the GPU builder's BWT symbols per row
{
	const uint64_t n = T.n;
	// BWT without the sentinel row
	Lap lap;
	uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
	Raw<uint8_t> Bbuf(n);
	uint8_t *B = Bbuf.p;
	{
		uint64_t found = 0;
#pragma omp parallel for schedule(static) reduction(max : found)
		for (int64_t r = 0; r <= (int64_t)n; ++r) if (sa[(size_t)r] == 0) found = (uint64_t)r;
		primary = found;
		uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma omp parallel for schedule(static) reduction(+ : c0, c1, c2, c3)
		for (int64_t r = 0; r <= (int64_t)n; ++r) {
			if ((uint64_t)r == primary) continue;
			const uint8_t c = prev ? prev[(size_t)r] : (uint8_t)T.get((uint64_t)sa[(size_t)r] - 1);
			B[(size_t)r - ((uint64_t)r > primary ? 1 : 0)] = c;
			c0 += c == 0; c1 += c == 1; c2 += c == 2; c3 += c == 3;
		}
		L2[1] = c0; L2[2] = c1; L2[3] = c2; L2[4] = c3;
	}
	for (int c = 0; c < 4; ++c) L2[c + 1] += L2[c];
	lap("bwt");

	{
		// per 128 bases: the four running counts (as 8 x u32) and 8 words of 16 bases; one more count block at the end
		const uint64_t n_words = (n + 15) >> 4, n_blocks = (n_words + 7) >> 3;
		Raw<uint32_t> outbuf(n_words + 8 * n_blocks + 8);
		uint32_t *out = outbuf.p;
		const int n_thr = omp_get_max_threads();
		const uint64_t per = ((n_blocks + (uint64_t)n_thr - 1) / (uint64_t)n_thr);
		std::vector<std::array<uint64_t, 4>> before((size_t)n_thr + 1);
		for (auto &x : before) x = {0, 0, 0, 0};
#pragma omp parallel num_threads(n_thr)
		{
			const int t0 = omp_get_thread_num(), team = omp_get_num_threads();
			for (int v = t0; v < n_thr; v += team) {      // slots, as in build_sa: independent of the team's actual size
				const uint64_t b_lo = std::min(n_blocks, (uint64_t)v * per), b_hi = std::min(n_blocks, b_lo + per);
				uint64_t c4[4] = {0, 0, 0, 0};
				for (uint64_t i = b_lo << 7, e = std::min(n, b_hi << 7); i < e; ++i) ++c4[B[i]];
				for (int c = 0; c < 4; ++c) before[(size_t)v + 1][c] = c4[c];
			}
#pragma omp barrier
#pragma omp single
			for (int u = 0; u < n_thr; ++u) for (int c = 0; c < 4; ++c) before[(size_t)u + 1][c] += before[(size_t)u][c];
			for (int v = t0; v < n_thr; v += team) {
				const uint64_t b_lo = std::min(n_blocks, (uint64_t)v * per), b_hi = std::min(n_blocks, b_lo + per);
				uint64_t c4[4];
				for (int c = 0; c < 4; ++c) c4[c] = before[(size_t)v][c];
				for (uint64_t blk = b_lo; blk < b_hi; ++blk) {
					uint32_t *o = out + blk * 16;
					memcpy(o, c4, 32);
					for (uint64_t wi = blk << 3, we = std::min(n_words, (blk + 1) << 3); wi < we; ++wi) {
						uint32_t word = 0;
						for (int tt = 0; tt < 16; ++tt) {
							const uint64_t i = (wi << 4) + tt;
							if (i < n) { word |= (uint32_t)B[i] << ((15 - tt) << 1); ++c4[B[i]]; }
						}
						o[8 + (wi & 7)] = word;
					}
				}
			}
		}
		const uint64_t used = n_words + 8 * n_blocks;      // the last block may hold fewer than 8 words
		uint64_t tot[4];
		for (int c = 0; c < 4; ++c) tot[c] = before[(size_t)n_thr][c];
		memcpy(out + used, tot, 32);
		uint64_t head[5] = {primary, L2[1], L2[2], L2[3], L2[4]};
		if (!write_file(prefix + ".bwt", head, sizeof head, out, (used + 8) * 4)) return -1;
	}
	lap("occ");

	const uint64_t sa_intv = 32;
	FILE *f = fopen((prefix + ".sa").c_str(), "wb");
	if (!f) return -1;
	fput64(f, primary);
	for (int c = 1; c <= 4; ++c) fput64(f, L2[c]);
	fput64(f, sa_intv);
	fput64(f, n);
	for (uint64_t r = sa_intv; r <= n; r += sa_intv) fput64(f, (uint64_t)sa[r]);
	fclose(f);

	{
		uint64_t head[3];
		memcpy(head, "EMAFSA01", 8);
		head[1] = n; head[2] = sizeof(I);
		if (!write_file(prefix + ".fsa", head, sizeof head, sa, (n + 1) * sizeof(I))) return -1;
	}
	lap("sa files");
	return 0;
}

}  // namespace

// Builds every index file next to `fasta`.  Returns 0 on success, negative on error.
extern "C" int ema_index_build(const char *fasta, int n_threads)
{
	if (n_threads > 0) omp_set_num_threads(n_threads);
	Lap lap;
	FILE *f = fopen(fasta, "rb");
	if (!f) return -1;
	std::string prefix(fasta);
	std::vector<Contig> contigs;
	std::vector<Hole> holes;
	std::vector<uint8_t> fwd;
	std::vector<std::string> fai;
	{ fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET); if (sz > 0) fwd.reserve((size_t)sz); }
	srand48(11);
	uint8_t kCode[256];
	memset(kCode, 4, sizeof kCode);
	kCode['A'] = kCode['a'] = 0; kCode['C'] = kCode['c'] = 1; kCode['G'] = kCode['g'] = 2; kCode['T'] = kCode['t'] = 3;
	{
		std::vector<char> buf(1 << 20);
		std::string line;
		int64_t file_off = 0, seq_off = 0;
		int linebases = 0, linewidth = 0;
		long cur_hole = -1;
		char last_amb = 0;
		auto flush_fai = [&]() {
			if (!contigs.empty()) {
				char tmp[8192];
				snprintf(tmp, sizeof(tmp), "%s\t%d\t%lld\t%d\t%d\n", contigs.back().name.c_str(), contigs.back().len,
				         (long long)seq_off, linebases, linewidth);
				fai.push_back(tmp);
			}
		};
		while (fgets(buf.data(), (int)buf.size(), f)) {
			size_t l = strlen(buf.data());
			int64_t line_start = file_off;
			file_off += (int64_t)l;
			size_t raw = l;
			while (l && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) --l;
			if (l && buf[0] == '>') {
				flush_fai();
				Contig c;
				std::string h(buf.data() + 1, l - 1);
				size_t sp = h.find_first_of(" \t");
				c.name = h.substr(0, sp);
				c.anno = sp == std::string::npos ? "" : h.substr(sp + 1);
				c.offset = (int64_t)fwd.size(); c.len = 0; c.n_ambs = 0;
				contigs.push_back(c);
				seq_off = file_off; linebases = 0; linewidth = 0;
				cur_hole = -1; last_amb = 0;
				(void)line_start;
				continue;
			}
			if (contigs.empty()) continue;
			if (linebases == 0 && l) { linebases = (int)l; linewidth = (int)raw; }
			// bases in bulk through a table; the rare ambiguous ones take the reference's hole bookkeeping below
			const size_t at0 = fwd.size();
			fwd.resize(at0 + l);
			uint8_t *dst = fwd.data() + at0;
			bool any_amb = false;
			for (size_t i = 0; i < l; ++i) { const uint8_t c = kCode[(unsigned char)buf[i]]; dst[i] = c; any_amb |= c > 3; }
			contigs.back().len += (int32_t)l;
			if (!any_amb) { if (l) last_amb = 0; continue; }
			for (size_t i = 0; i < l; ++i) {
				if (dst[i] < 4) { last_amb = 0; continue; }
				const char ch = buf[i];
				const int64_t pos = (int64_t)(at0 + i);
				if (cur_hole >= 0 && last_amb == ch && holes[cur_hole].offset + holes[cur_hole].len == pos) ++holes[cur_hole].len;
				else {
					holes.push_back(Hole{pos, 1, ch});
					cur_hole = (long)holes.size() - 1;
					++contigs.back().n_ambs;
				}
				last_amb = ch;
				dst[i] = (uint8_t)(lrand48() & 3);
			}
		}
		flush_fai();
	}
	fclose(f);
	const int64_t l_pac = (int64_t)fwd.size();
	if (l_pac == 0) return -2;
	lap("fasta");

	f = fopen((prefix + ".fai").c_str(), "w");
	if (!f) return -1;
	for (auto &s : fai) fputs(s.c_str(), f);
	fclose(f);

	f = fopen((prefix + ".ann").c_str(), "w");
	if (!f) return -1;
	fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)contigs.size(), 11u);
	for (auto &c : contigs) {
		fprintf(f, "%d %s", 0, c.name.c_str());
		if (!c.anno.empty()) fprintf(f, " %s\n", c.anno.c_str()); else fprintf(f, "\n");
		fprintf(f, "%lld %d %d\n", (long long)c.offset, c.len, c.n_ambs);
	}
	fclose(f);

	f = fopen((prefix + ".amb").c_str(), "w");
	if (!f) return -1;
	fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)contigs.size(), (unsigned)holes.size());
	for (auto &h : holes) fprintf(f, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
	fclose(f);

	{
		std::vector<uint8_t> pac((l_pac >> 2) + 1, 0);
#pragma omp parallel for schedule(static)
		for (int64_t by = 0; by < (l_pac + 3) >> 2; ++by) {
			uint8_t v = 0;
			for (int64_t i = by << 2; i < std::min<int64_t>(l_pac, (by << 2) + 4); ++i) v |= (uint8_t)(fwd[i] << ((~i & 3) << 1));
			pac[by] = v;
		}
		f = fopen((prefix + ".pac").c_str(), "wb");
		if (!f) return -1;
		fwrite(pac.data(), 1, (l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1), f);
		if ((l_pac & 3) == 0) fputc(0, f);
		fputc((int)(l_pac & 3), f);
		fclose(f);
	}

	Packed T;
	T.init(2 * (uint64_t)l_pac);
	{
		const uint64_t n2 = 2 * (uint64_t)l_pac, L = (uint64_t)l_pac;
#pragma omp parallel for schedule(static)
		for (int64_t wi = 0; wi < (int64_t)((n2 + 31) >> 5); ++wi) {      // each thread owns whole words
			uint64_t w = 0;
			for (int t = 0; t < 32; ++t) {
				const uint64_t p = ((uint64_t)wi << 5) + (uint64_t)t;
				if (p >= n2) break;
				const uint64_t c = p < L ? fwd[p] : 3u - fwd[n2 - 1 - p];
				w |= c << (62 - (t << 1));
			}
			T.w[(size_t)wi] = w;
		}
	}
	lap("text");
	std::vector<uint8_t>().swap(fwd);
	// EMA_INDEX_SA64=1 (tests): 8-byte suffix-array rows whatever the size, so that a small reference drives the row width
	// that only references beyond 2^32 symbols get otherwise
	const char *force64 = getenv("EMA_INDEX_SA64");
	const gpu_sa_fn gpu_sa = find_gpu_sa();
	Raw<uint8_t> prev(gpu_sa ? T.n + 1 : 1);
	const bool verbose = getenv("EMA_INDEX_PROF") != nullptr || getenv("EMA_VERBOSE") != nullptr;
	if (T.n < 0xffffff00ULL && !(force64 && atoi(force64) != 0)) {
		Raw<uint32_t> sa(T.n + 1);
		if (!sa.p) return -3;
		const bool on_gpu = gpu_sa && prev.p && gpu_sa(T.w.data(), T.n, 4, sa.p, prev.p, verbose) == 0;
		if (!on_gpu) build_sa(T, sa.p);
		say_builder(on_gpu, gpu_sa != nullptr);
		lap(on_gpu ? "sa (gpu)" : "sa (host)");
		return write_index(prefix, T, sa.p, on_gpu ? prev.p : nullptr);
	} else {
		Raw<uint64_t> sa(T.n + 1);
		if (!sa.p) return -3;
		const bool on_gpu = gpu_sa && prev.p && gpu_sa(T.w.data(), T.n, 8, sa.p, prev.p, verbose) == 0;
		if (!on_gpu) build_sa(T, sa.p);
		say_builder(on_gpu, gpu_sa != nullptr);
		lap(on_gpu ? "sa (gpu)" : "sa (host)");
		return write_index(prefix, T, sa.p, on_gpu ? prev.p : nullptr);
	}
}
