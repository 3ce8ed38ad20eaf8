// ema_amd/csrc/host_index.cpp -- see host_index.h.
#include "host_index.h"
#include <cstdio>
#include <cstring>

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

DevIndex HostIndex::view() const
{
	DevIndex d;
	d.occ = occ.data();
	d.sa = sa_bytes.data();
	d.pac = pac.data();
	d.ctg_off = ctg_off.data();
	d.primary = primary; d.seq_len = seq_len;
	for (int i = 0; i < 5; ++i) d.L2[i] = L2[i];
	d.l_pac = l_pac;
	d.n_seqs = (int32_t)contigs.size();
	d.sa_width = sa_width;
	d.n_super = n_super; d.pad_ = 0;
	memcpy(d.occ_super, occ_super, sizeof(occ_super));
	return d;
}

std::string host_index_load(const std::string &prefix, HostIndex &ix)
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
		OccBlock zero; memset(&zero, 0, sizeof(zero));
		ix.occ.assign((n_blocks + 1) * 2, zero);
		uint64_t running[4] = {0, 0, 0, 0}, super[4] = {0, 0, 0, 0};
		for (uint64_t b = 0; b < n_blocks; ++b) {
			uint64_t base = b * 16;
			if (base + 8 > n_words) return "truncated .bwt";
			uint64_t cnt[4];
			memcpy(cnt, w + base, 32);
			for (int c = 0; c < 4; ++c)
				if (cnt[c] != running[c]) return "inconsistent occ counters in .bwt";
			for (int t = 0; t < 128; ++t) {
				uint64_t pos = (b << 7) + t;
				if (pos >= ix.seq_len) break;
				if ((t & 63) == 0) {      // a device block starts here
					if ((pos & (((uint64_t)1 << EMA_OCC_SUPER_SHIFT) - 1)) == 0) {
						const uint64_t sb = pos >> EMA_OCC_SUPER_SHIFT;
						for (int c = 0; c < 4; ++c) { super[c] = running[c]; if (sb > 0) ix.occ_super[sb - 1][c] = running[c]; }
					}
					for (int c = 0; c < 4; ++c) ix.occ[pos >> 6].cnt[c] = (uint32_t)(running[c] - super[c]);
				}
				uint64_t wi = base + 8 + (t >> 4);
				if (wi >= n_words) return "truncated .bwt";
				unsigned sym = w[wi] >> ((~t & 15) << 1) & 3;
				ix.occ[pos >> 6].bases[(t & 63) >> 5] |= (uint64_t)sym << ((t & 31) << 1);
				++running[sym];
			}
		}
		for (int c = 0; c < 4; ++c)
			if (running[c] != ix.L2[c + 1] - ix.L2[c]) return "symbol totals disagree with L2 in .bwt";
	}
	// ---- .fsa: whole suffix array
	if (!slurp(prefix + ".fsa", raw) || raw.size() < 24 || memcmp(raw.data(), "EMAFSA01", 8) != 0)
		return "cannot read " + prefix + ".fsa (flat suffix array; rebuild the index with ema_index_build)";
	{
		uint64_t n, width;
		memcpy(&n, raw.data() + 8, 8);
		memcpy(&width, raw.data() + 16, 8);
		if (n != ix.seq_len || (width != 4 && width != 8) || raw.size() != 24 + (n + 1) * width) return "bad .fsa header";
		ix.sa_width = (int)width;
		ix.sa_bytes.assign(raw.begin() + 24, raw.end());
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
	if ((uint64_t)ix.l_pac * 2 != ix.seq_len) return "l_pac disagrees with seq_len";
	ix.ctg_off.clear();
	for (auto &c : ix.contigs) ix.ctg_off.push_back(c.offset);
	ix.ctg_off.push_back(ix.l_pac);
	return "";
}
