// ema_amd/csrc/host_index.cpp -- see host_index.h.
#include "host_index.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>

namespace {
bool slurp(const std::string &path, std::vector<uint8_t> &buf)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf.resize(n);
	bool ok = n == 0 || fread(buf.data(), 1, n, f) == (size_t)n;
	fclose(f);
	return ok;
}
}  // namespace

// The coarse contig table (dev_types.h, DevIndex::ctg_tab) of contig offsets ctg_off[0..n] (ctg_off[n] = l_pac)
void host_contig_table(const std::vector<int64_t> &ctg_off, std::vector<int32_t> &tab, int &shift)
{
	const int64_t l_pac = ctg_off.empty() ? 0 : ctg_off.back();
	const int32_t last = (int32_t)ctg_off.size() - 2;
	shift = 0;
	while ((l_pac >> shift) >= (1 << 16)) ++shift;
	const int64_t nb = (l_pac >> shift) + 1;
	tab.assign((size_t)nb + 1, 0);
	int32_t c = 0;
	for (int64_t b = 0; b < nb; ++b) {
		const int64_t pos = std::min<int64_t>(b << shift, l_pac - 1);
		while (c < last && pos >= ctg_off[(size_t)c + 1]) ++c;
		tab[(size_t)b] = c;
	}
	tab[(size_t)nb] = last < 0 ? 0 : last;
}

DevIndex HostIndex::view() const
{
	DevIndex d;
	d.occ = occ.data();
	d.sa = sa_bytes.data();
	d.pac = pac.data();
	d.ctg_off = ctg_off.data();
	d.ctg_alt = ctg_alt.empty() ? nullptr : ctg_alt.data();
	d.ctg_tab = ctg_tab.data(); d.ctg_shift = ctg_shift;
	d.primary = primary; d.seq_len = seq_len;
	for (int i = 0; i < 5; ++i) d.L2[i] = L2[i];
	d.l_pac = l_pac;
	d.n_seqs = (int32_t)contigs.size();
	d.sa_width = sa_width;
	d.n_super = n_super; d.kmer_k = kmer_k; d.kmer_wide = kmer_wide.empty() ? nullptr : kmer_wide.data(); d.kmer_narrow = kmer_narrow.empty() ? nullptr : kmer_narrow.data();
	d.text2 = text2.empty() ? nullptr : text2.data();
	memcpy(d.occ_super, occ_super, sizeof(occ_super));
	return d;
}

// bwa's bwt_sa() for every row: walk LF (bwt_invPsi) until a sampled row, count the steps.  Row 0 (the sentinel's suffix, which no
// interval ever contains) gets seq_len, as ema_index_build writes it.
void host_expand_sa(HostIndex &ix)
{
	const uint64_t n = ix.seq_len, mask = (uint64_t)ix.sa_intv - 1;
	int shift = 0;
	while (((uint64_t)1 << shift) < (uint64_t)ix.sa_intv) ++shift;
	ix.sa_bytes.assign(ix.sa_size, 0);
	auto inv_psi = [&](uint64_t k) -> uint64_t {
		if (k == ix.primary) return 0;
		const uint64_t p = k - (k > ix.primary ? 1 : 0);
		const OccBlock &b = ix.occ[p >> 6];
		const int r = (int)(p & 63);
		const unsigned c = (unsigned)((b.bases[0] >> r) & 1u) | (unsigned)((b.bases[1] >> r) & 1u) << 1;      // two bit planes (dev_types.h)
		const uint64_t m = (2ULL << r) - 1, lo = c & 1 ? b.bases[0] : ~b.bases[0], hi = c & 2 ? b.bases[1] : ~b.bases[1];
		uint64_t cnt = b.cnt[c] + (uint64_t)__builtin_popcountll(lo & hi & m);
		const uint64_t sb = p >> EMA_OCC_SUPER_SHIFT;
		if (sb > 0) cnt += ix.occ_super[sb - 1][c];
		return ix.L2[c] + cnt;
	};
	unsigned n_thr = std::thread::hardware_concurrency();
	n_thr = n_thr < 1 ? 1 : n_thr > 64 ? 64 : n_thr;
	if (n < 65536) n_thr = 1;
	auto work = [&](unsigned t) {
		const uint64_t per = (n + 1 + n_thr - 1) / n_thr, lo = std::min(n + 1, t * per), hi = std::min(n + 1, lo + per);
		for (uint64_t r = lo; r < hi; ++r) {
			uint64_t k = r, steps = 0;
			while (k & mask) { k = inv_psi(k); ++steps; }
			uint64_t v = steps + ix.sa_sampled[k >> shift];
			if (r == 0) v = n;
			if (ix.sa_width == 4) ((uint32_t *)ix.sa_bytes.data())[r] = (uint32_t)v;
			else ((uint64_t *)ix.sa_bytes.data())[r] = v;
		}
	};
	std::vector<std::thread> th;
	for (unsigned t = 1; t < n_thr; ++t) th.emplace_back(work, t);
	work(0);
	for (auto &x : th) x.join();
}

std::string host_index_load(const std::string &prefix, HostIndex &ix, bool with_sa)
{
	std::vector<uint8_t> raw;
	// ---- .bwt (bwa layout): u64 primary, u64 L2[1..4], 16-word blocks {4 x u64 occ, 8 x u32 bases}
	if (!slurp(prefix + ".bwt", raw) || raw.size() < 40) return "cannot read " + prefix + ".bwt";
	const uint64_t *h = (const uint64_t *)raw.data();
	ix.primary = h[0];
	ix.L2[0] = 0;
	for (int c = 0; c < 4; ++c) ix.L2[c + 1] = h[1 + c];
	ix.seq_len = ix.L2[4];
	{
		const uint32_t *w = (const uint32_t *)(raw.data() + 40);
		const uint64_t n_words = (raw.size() - 40) / 4;
		const uint64_t n_blocks = (ix.seq_len + 127) >> 7;      // bwa's 128-symbol blocks; two device blocks each
		ix.n_super = (int)((ix.seq_len >> EMA_OCC_SUPER_SHIFT) + 1);
		if (ix.n_super > EMA_OCC_MAX_SUPER) return "reference too long for the device rank structure (2^33 BWT symbols)";
		// every 128-symbol block carries its own running counts, so blocks convert independently (all host threads);
		// each checks that its counts plus its symbols give the next block's counts (the totals after the last one)
		const uint64_t n_data = (ix.seq_len + 15) >> 4;
		if (n_words < n_data + 8 * n_blocks + 8) return "truncated .bwt";
		OccBlock zero; memset(&zero, 0, sizeof(zero));
		ix.occ.assign((n_blocks + 1) * 2, zero);
		auto counts_at = [&](uint64_t blk, uint64_t out[4]) {      // running counts before block blk (blk == n_blocks: totals)
			memcpy(out, w + (blk < n_blocks ? blk * 16 : n_data + 8 * n_blocks), 32);
		};
		for (int sb = 1; sb < ix.n_super; ++sb)      // superblock boundaries are multiples of 128 symbols
			counts_at(((uint64_t)sb << EMA_OCC_SUPER_SHIFT) >> 7, ix.occ_super[sb - 1]);
		unsigned n_thr = std::thread::hardware_concurrency();
		n_thr = n_thr < 1 ? 1 : n_thr > 64 ? 64 : n_thr;
		if (n_blocks < 4096) n_thr = 1;
		std::vector<int> bad(n_thr, 0);
		auto work = [&](unsigned t) {
			const uint64_t per = (n_blocks + n_thr - 1) / n_thr, b_lo = std::min(n_blocks, t * per), b_hi = std::min(n_blocks, b_lo + per);
			for (uint64_t b = b_lo; b < b_hi; ++b) {
				uint64_t running[4], next[4], super[4] = {0, 0, 0, 0};
				counts_at(b, running);
				const uint64_t sb = (b << 7) >> EMA_OCC_SUPER_SHIFT;
				if (sb > 0) memcpy(super, ix.occ_super[sb - 1], 32);
				for (int half = 0; half < 2; ++half) {
					const uint64_t pos0 = (b << 7) + (uint64_t)half * 64;
					if (pos0 >= ix.seq_len) break;
					OccBlock &o = ix.occ[pos0 >> 6];
					for (int c = 0; c < 4; ++c) o.cnt[c] = (uint32_t)(running[c] - super[c]);
					for (int t2 = 0; t2 < 64; ++t2) {
						const uint64_t pos = pos0 + (uint64_t)t2;
						if (pos >= ix.seq_len) break;
						const int tt = half * 64 + t2;
						const unsigned sym = w[b * 16 + 8 + (tt >> 4)] >> ((~tt & 15) << 1) & 3;
						o.bases[0] |= (uint64_t)(sym & 1) << t2; o.bases[1] |= (uint64_t)(sym >> 1) << t2;      // two bit planes (dev_types.h)
						++running[sym];
					}
				}
				counts_at(b + 1, next);
				if (memcmp(running, next, 32) != 0) { bad[t] = 1; return; }
			}
		};
		{
			std::vector<std::thread> th;
			for (unsigned t = 1; t < n_thr; ++t) th.emplace_back(work, t);
			work(0);
			for (auto &x : th) x.join();
		}
		for (int x : bad) if (x) return "inconsistent occ counters in .bwt";
		uint64_t tot[4];
		counts_at(n_blocks, tot);
		for (int c = 0; c < 4; ++c)
			if (tot[c] != ix.L2[c + 1] - ix.L2[c]) return "symbol totals disagree with L2 in .bwt";
	}
	std::vector<uint8_t>().swap(raw);
	// ---- .fsa: whole suffix array (the engine streams it to the device itself: with_sa == false reads the header only)
	{
		FILE *f = fopen((prefix + ".fsa").c_str(), "rb");
		uint8_t head[24];
		if (!f) {
			// a stock bwa index: bwa's sampled suffix array (bwt_dump_sa: primary, L2[1..4], sa_intv, seq_len, then SA[sa_intv],
			// SA[2 sa_intv], ...), expanded to the flat one by LF-mapping -- here for with_sa, on the device by the engine
			std::vector<uint8_t> sraw;
			if (!slurp(prefix + ".sa", sraw) || sraw.size() < 56) return "cannot read " + prefix + ".fsa nor " + prefix + ".sa (suffix array)";
			const uint64_t *h = (const uint64_t *)sraw.data();
			if (h[0] != ix.primary || h[1] != ix.L2[1] || h[2] != ix.L2[2] || h[3] != ix.L2[3] || h[4] != ix.L2[4] || h[6] != ix.seq_len)
				return prefix + ".sa does not belong to " + prefix + ".bwt";
			const uint64_t intv = h[5];
			if (intv == 0 || (intv & (intv - 1)) != 0 || intv > 1024) return "bad sampling interval in " + prefix + ".sa";
			const uint64_t n_sa = (ix.seq_len + intv) / intv;
			if (sraw.size() != 56 + (n_sa - 1) * 8) return "truncated " + prefix + ".sa";
			ix.sa_intv = (int)intv;
			ix.sa_sampled.resize(n_sa);
			ix.sa_sampled[0] = ~(uint64_t)0;
			memcpy(ix.sa_sampled.data() + 1, sraw.data() + 56, (n_sa - 1) * 8);
			const char *force64 = getenv("EMA_INDEX_SA64");
			ix.sa_width = (ix.seq_len < 0xffffff00ULL && !(force64 && atoi(force64) != 0)) ? 4 : 8;      // as ema_index_build chooses
			ix.sa_path.clear();
			ix.sa_file_off = 0;
			ix.sa_size = (ix.seq_len + 1) * (uint64_t)ix.sa_width;
			if (with_sa) host_expand_sa(ix);
		} else {
		if (fread(head, 1, 24, f) != 24 || memcmp(head, "EMAFSA01", 8) != 0) {
			fclose(f);
			return "cannot read " + prefix + ".fsa (flat suffix array; rebuild the index with ema_index_build)";
		}
		uint64_t n, width;
		memcpy(&n, head + 8, 8);
		memcpy(&width, head + 16, 8);
		fseek(f, 0, SEEK_END);
		const uint64_t size = (uint64_t)ftell(f);
		if (n != ix.seq_len || (width != 4 && width != 8) || size != 24 + (n + 1) * width) { fclose(f); return "bad .fsa header"; }
		ix.sa_width = (int)width;
		ix.sa_path = prefix + ".fsa";
		ix.sa_file_off = 24;
		ix.sa_size = (n + 1) * width;
		if (with_sa) {
			ix.sa_bytes.resize(ix.sa_size);
			fseek(f, 24, SEEK_SET);
			if (fread(ix.sa_bytes.data(), 1, ix.sa_size, f) != ix.sa_size) { fclose(f); return "cannot read " + prefix + ".fsa"; }
		}
		fclose(f);
		}
	}
	// ---- .pac
	if (!slurp(prefix + ".pac", ix.pac)) return "cannot read " + prefix + ".pac";
	ix.pac.resize(ix.pac.size() + 8, 0);
	// ---- .ann
	{
		FILE *f = fopen((prefix + ".ann").c_str(), "r");
		if (!f) return "cannot read " + prefix + ".ann";
		char line[8192];
		long long l_pac; int n_seqs; unsigned seed;
		if (!fgets(line, sizeof(line), f) || sscanf(line, "%lld %d %u", &l_pac, &n_seqs, &seed) != 3) { fclose(f); return "bad .ann"; }
		ix.l_pac = l_pac;
		for (int i = 0; i < n_seqs; ++i) {
			HostContig c;
			unsigned gi; char name[4096]; long long off; int len, nambs;
			if (!fgets(line, sizeof(line), f) || sscanf(line, "%u %4095s", &gi, name) != 2) { fclose(f); return "bad .ann"; }
			c.name = name;
			if (!fgets(line, sizeof(line), f) || sscanf(line, "%lld %d %d", &off, &len, &nambs) != 3) { fclose(f); return "bad .ann"; }
			c.offset = off; c.len = len; c.is_alt = 0;
			ix.contigs.push_back(c);
		}
		fclose(f);
	}
	// ---- .alt (optional; bwa_idx_load_from_disk): every line not starting with '@' names an ALT contig by its first field
	if (FILE *f = fopen((prefix + ".alt").c_str(), "r")) {
		char line[8192];
		bool any = false;
		while (fgets(line, sizeof(line), f)) {
			if (line[0] == '@') continue;
			char *e = line;
			while (*e && *e != '\t' && *e != ' ' && *e != '\n' && *e != '\r') ++e;
			*e = 0;
			if (!line[0]) continue;
			for (auto &c : ix.contigs)
				if (c.name == line) { c.is_alt = 1; any = true; break; }
		}
		fclose(f);
		if (any) for (auto &c : ix.contigs) ix.ctg_alt.push_back((uint8_t)c.is_alt);
	}
	if ((uint64_t)ix.l_pac * 2 != ix.seq_len) return "l_pac disagrees with seq_len";
	ix.ctg_off.clear();
	for (auto &c : ix.contigs) ix.ctg_off.push_back(c.offset);
	ix.ctg_off.push_back(ix.l_pac);
	host_contig_table(ix.ctg_off, ix.ctg_tab, ix.ctg_shift);
	return "";
}
