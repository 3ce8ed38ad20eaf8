// ema_amd/csrc/synth_genome.cpp -- the synthetic reference of bench.py ("synthetic reference"), generated natively:
// i.i.d. bases at 41 % GC, then the repeat structure that exercises the max_occ paths -- a 300 bp interspersed family (10-15 %
// divergence), a 6 kb family (5-20 %, members truncated), segmental duplications (10-100 kb at 1-2 %).  The same model as
// ema_amd/synth.py's make_genome (which the tests keep using for their small references: numpy's generator, 45 s at 3.1 Gbp),
// with its own counter-based generator so that the result depends on (seed, lengths) only, not on the thread count.
// Bench / test tooling: not on the product path.
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>

namespace {
struct Rng {      // splitmix64 stream
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed) {}
	uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
	double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
	uint64_t below(uint64_t n) { return n ? (uint64_t)(((unsigned __int128)next() * n) >> 64) : 0; }
};
inline uint8_t base_of(uint32_t r16) { return r16 < 19333 ? 0 : r16 < 32768 ? 1 : r16 < 46203 ? 2 : 3; }      // 0.295 / 0.205 / 0.205 / 0.295

void mutate_into(Rng &rng, uint8_t *dst, const uint8_t *src, int64_t n, double div, bool rc)
{
	const uint64_t thr = (uint64_t)(div * 18446744073709551615.0);
	for (int64_t i = 0; i < n; ++i) {
		uint8_t b = rc ? (uint8_t)(3 - src[n - 1 - i]) : src[i];
		const uint64_t r = rng.next();
		if (r < thr) b = (uint8_t)((b + 1 + (r >> 7) % 3) & 3);
		dst[i] = b;
	}
}
}  // namespace

// g[0 .. total): bases 0..3.  Returns 0.
extern "C" int ema_synth_genome(uint8_t *g, int64_t total, uint64_t seed, double short_rep, double long_rep, double segdup, int n_threads)
{
	if (!g || total <= 0) return -1;
	if (n_threads < 1) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
	const int64_t chunk = (int64_t)1 << 22, n_chunks = (total + chunk - 1) / chunk;
	auto fill = [&](int t) {
		for (int64_t c = t; c < n_chunks; c += n_threads) {
			Rng rng(seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(c + 1)));
			const int64_t lo = c * chunk, hi = std::min(total, lo + chunk);
			int64_t i = lo;
			for (; i + 4 <= hi; i += 4) {
				const uint64_t r = rng.next();
				g[i] = base_of((uint32_t)(r & 0xffff)); g[i + 1] = base_of((uint32_t)(r >> 16 & 0xffff));
				g[i + 2] = base_of((uint32_t)(r >> 32 & 0xffff)); g[i + 3] = base_of((uint32_t)(r >> 48));
			}
			for (; i < hi; ++i) g[i] = base_of((uint32_t)(rng.next() & 0xffff));
		}
	};
	{
		std::vector<std::thread> th;
		for (int t = 1; t < n_threads; ++t) th.emplace_back(fill, t);
		fill(0);
		for (auto &x : th) x.join();
	}
	if (total <= 2 * 6000) return 0;
	Rng rng(seed * 0x2545F4914F6CDD1DULL + 0x1234567);
	std::vector<uint8_t> fam_short(300), fam_long(6000), tmp;
	for (auto &b : fam_short) b = (uint8_t)(rng.next() & 3);
	for (auto &b : fam_long) b = (uint8_t)(rng.next() & 3);
	auto inject = [&](const std::vector<uint8_t> &cons, double frac, double dlo, double dhi) {
		const int64_t n_copies = (int64_t)((double)total * frac / (double)cons.size());
		for (int64_t k = 0; k < n_copies; ++k) {
			int64_t L = (int64_t)cons.size();
			if (L > 1000) L = 500 + (int64_t)rng.below((uint64_t)(cons.size() - 500 + 1));      // long family members are usually truncated
			const int64_t s0 = (int64_t)rng.below((uint64_t)(cons.size() - L + 1));
			const double div = dlo + (dhi - dlo) * rng.uni();
			const bool rc = rng.next() & 1;
			const int64_t p = (int64_t)rng.below((uint64_t)(total - L));
			mutate_into(rng, g + p, cons.data() + s0, L, div, rc);
		}
	};
	inject(fam_short, short_rep, 0.10, 0.15);
	inject(fam_long, long_rep, 0.05, 0.20);
	const int64_t n_sd = segdup > 0 ? std::max<int64_t>(1, (int64_t)((double)total * segdup / 30000.0)) : 0;
	for (int64_t k = 0; k < n_sd; ++k) {
		const int64_t lo = std::min<int64_t>(10000, total / 8), hi = std::min<int64_t>(100000, total / 4);
		const int64_t L = lo + (int64_t)rng.below((uint64_t)(hi - lo + 1));
		const int64_t src = (int64_t)rng.below((uint64_t)(total - L)), dst = (int64_t)rng.below((uint64_t)(total - L));
		tmp.assign(g + src, g + src + L);
		mutate_into(rng, g + dst, tmp.data(), L, 0.01 + 0.01 * rng.uni(), false);
	}
	return 0;
}
