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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

namespace {

struct Contig { std::string name, anno; int64_t offset; int32_t len, n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };

struct Packed {            // big-endian 2-bit text: base t of a word sits at bits 62-2t
	std::vector<uint64_t> w;
	uint64_t n = 0;
	void init(uint64_t n_) { n = n_; w.assign((n_ >> 5) + 3, 0); }
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

template <typename I>
void build_sa(const Packed &T, std::vector<I> &sa)   // sa[0..n]: row 0 = '$'
{
	const uint64_t n = T.n;
	sa.assign(n + 1, 0);
	sa[0] = (I)n;
	int K = 1;
	while (K < 12 && (1ULL << (2 * (K + 1))) <= n / 4 + 4) ++K;
	const uint64_t nb = 1ULL << (2 * K);
	std::vector<uint64_t> cnt(nb + 1, 0);
	auto key_at = [&](uint64_t i) { return T.get32(i) >> (64 - 2 * K); };   // zero (=A) padded past n
	for (uint64_t i = 0; i < n; ++i) ++cnt[key_at(i) + 1];
	for (uint64_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
	{
		std::vector<uint64_t> pos(cnt.begin(), cnt.end() - 1);
		for (uint64_t i = 0; i < n; ++i) sa[1 + pos[key_at(i)]++] = (I)i;
	}
	I *base = sa.data() + 1;
#pragma omp parallel for schedule(dynamic, 4096)
	for (int64_t b = 0; b < (int64_t)nb; ++b) {
		uint64_t lo = cnt[b], hi = cnt[b + 1];
		if (hi - lo > 1)
			std::sort(base + lo, base + hi, [&](I x, I y) { return suffix_less(T, x, y); });
	}
}

void fput64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }

template <typename I>
int write_index(const std::string &prefix, const Packed &T, const std::vector<I> &sa)
{
	const uint64_t n = T.n;
	// BWT without the sentinel row
	uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
	std::vector<uint8_t> B(n);
	{
		uint64_t k = 0;
		for (uint64_t r = 0; r <= n; ++r) {
			uint64_t p = sa[r];
			if (p == 0) { primary = r; continue; }
			B[k++] = (uint8_t)T.get(p - 1);
		}
	}
	for (uint64_t i = 0; i < n; ++i) ++L2[B[i] + 1];
	for (int c = 0; c < 4; ++c) L2[c + 1] += L2[c];

	FILE *f = fopen((prefix + ".bwt").c_str(), "wb");
	if (!f) return -1;
	fput64(f, primary);
	for (int c = 1; c <= 4; ++c) fput64(f, L2[c]);
	{
		uint64_t c4[4] = {0, 0, 0, 0};
		const uint64_t n_words = (n + 15) >> 4;
		std::vector<uint32_t> out;
		out.reserve(n_words * 2 + 64);
		for (uint64_t wi = 0; wi < n_words; ++wi) {
			if ((wi & 7) == 0) { uint32_t tmp[8]; memcpy(tmp, c4, 32); out.insert(out.end(), tmp, tmp + 8); }
			uint32_t word = 0;
			for (int t = 0; t < 16; ++t) {
				uint64_t i = (wi << 4) + t;
				if (i < n) { word |= (uint32_t)B[i] << ((15 - t) << 1); ++c4[B[i]]; }
			}
			out.push_back(word);
		}
		{ uint32_t tmp[8]; memcpy(tmp, c4, 32); out.insert(out.end(), tmp, tmp + 8); }
		fwrite(out.data(), 4, out.size(), f);
	}
	fclose(f);

	const uint64_t sa_intv = 32;
	f = fopen((prefix + ".sa").c_str(), "wb");
	if (!f) return -1;
	fput64(f, primary);
	for (int c = 1; c <= 4; ++c) fput64(f, L2[c]);
	fput64(f, sa_intv);
	fput64(f, n);
	for (uint64_t r = sa_intv; r <= n; r += sa_intv) fput64(f, (uint64_t)sa[r]);
	fclose(f);

	f = fopen((prefix + ".fsa").c_str(), "wb");
	if (!f) return -1;
	fwrite("EMAFSA01", 1, 8, f);
	fput64(f, n);
	fput64(f, sizeof(I));
	fwrite(sa.data(), sizeof(I), n + 1, f);
	fclose(f);
	return 0;
}

}  // namespace

// Builds every index file next to `fasta`.  Returns 0 on success, negative on error.
extern "C" int ema_index_build(const char *fasta, int n_threads)
{
	if (n_threads > 0) omp_set_num_threads(n_threads);
	FILE *f = fopen(fasta, "rb");
	if (!f) return -1;
	std::string prefix(fasta);
	std::vector<Contig> contigs;
	std::vector<Hole> holes;
	std::vector<uint8_t> fwd;
	std::vector<std::string> fai;
	srand48(11);
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
			for (size_t i = 0; i < l; ++i) {
				char ch = buf[i];
				int c;
				switch (ch) {
				case 'A': case 'a': c = 0; break;
				case 'C': case 'c': c = 1; break;
				case 'G': case 'g': c = 2; break;
				case 'T': case 't': c = 3; break;
				default: c = 4;
				}
				if (c >= 4) {
					if (cur_hole >= 0 && last_amb == ch && holes[cur_hole].offset + holes[cur_hole].len == (int64_t)fwd.size()) ++holes[cur_hole].len;
					else {
						holes.push_back(Hole{(int64_t)fwd.size(), 1, ch});
						cur_hole = (long)holes.size() - 1;
						++contigs.back().n_ambs;
					}
					last_amb = ch;
					c = (int)(lrand48() & 3);
				} else last_amb = 0;
				fwd.push_back((uint8_t)c);
				++contigs.back().len;
			}
		}
		flush_fai();
	}
	fclose(f);
	const int64_t l_pac = (int64_t)fwd.size();
	if (l_pac == 0) return -2;

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
		for (int64_t i = 0; i < l_pac; ++i) pac[i >> 2] |= fwd[i] << ((~i & 3) << 1);
		f = fopen((prefix + ".pac").c_str(), "wb");
		if (!f) return -1;
		fwrite(pac.data(), 1, (l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1), f);
		if ((l_pac & 3) == 0) fputc(0, f);
		fputc((int)(l_pac & 3), f);
		fclose(f);
	}

	Packed T;
	T.init(2 * (uint64_t)l_pac);
	for (int64_t i = 0; i < l_pac; ++i) {
		T.set((uint64_t)i, fwd[i]);
		T.set((uint64_t)(2 * l_pac - 1 - i), 3 - fwd[i]);
	}
	std::vector<uint8_t>().swap(fwd);
	if (T.n < 0xffffff00ULL) {
		std::vector<uint32_t> sa;
		build_sa(T, sa);
		return write_index(prefix, T, sa);
	} else {
		std::vector<uint64_t> sa;
		build_sa(T, sa);
		return write_index(prefix, T, sa);
	}
}
