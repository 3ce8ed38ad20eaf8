#include "host_pool.h"
#include <cstdio>
#include <vector>
#include <numeric>
int main()
{
	EmaPool &p = EmaPool::get();
	long total = 0;
	for (int rep = 0; rep < 200; ++rep) {
		std::vector<long> out(64, 0);
		p.run(64, [&](size_t i) {
			std::vector<long> in(8, 0);
			EmaPool::get().run(8, [&](size_t j) { long s = 0; for (int k = 0; k < 1000; ++k) s += (long)(i * j + k) % 7; in[j] = s; });      // a pass inside a piece
			out[i] = std::accumulate(in.begin(), in.end(), 0L);
		});
		total += std::accumulate(out.begin(), out.end(), 0L);
	}
	// several callers at once
	std::vector<std::thread> th;
	std::vector<long> sums(6, 0);
	for (int t = 0; t < 6; ++t) th.emplace_back([&, t] { for (int rep = 0; rep < 100; ++rep) { std::vector<long> o(40, 0); EmaPool::get().run(40, [&](size_t i) { o[i] = (long)i * t; }); sums[t] += std::accumulate(o.begin(), o.end(), 0L); } });
	for (auto &x : th) x.join();
	printf("%ld %ld %d threads\n", total, sums[5], p.size());
	return 0;
}
