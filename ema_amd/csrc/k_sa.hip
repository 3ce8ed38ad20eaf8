// ema_amd/csrc/k_sa.hip -- suffix array of the index text on the GPU ("6.2 G suffixes fit one MI355X").
//
// The FM-index `ema align` loads (reference src/bwabridge.c:77-96: bwa_idx_load) is built from the suffix array of the text
// forward strand + reverse complement.  ema_index_build (index_build.cpp) sorts it on the host's cores: 54 of its 100 s at the
// GRCh38 scale on the CPUs a GPU box grants, and 31 s more gathering the BWT from it (r03g).  This file does both on the device
// and hands the host the finished rows -- the SAME rows: the order is the total order of the suffixes ('$' smallest, a suffix that
// is a prefix of another one first), so the files the builder writes from them are byte-identical (tests/test_gpu_sa.py).
//
// Method.  The text is 2 bit/base in 64-bit words, so 32 bases of any suffix are one unaligned 64-bit read.
//   * chunks: the suffixes starting with each of the 16 two-base prefixes are sorted on their own (a 6.2 G-row array is done in
//     pieces of ~390 M rows: ~13 GB of sort buffers at a time);
//   * the chunk's positions are collected in DESCENDING order and sorted by their first 32 bases with a stable radix sort
//     (rocPRIM): rows with equal keys keep that order, which is what makes a suffix that ENDS inside the compared stretch
//     (zero = 'A' padded) come before a longer one that continues with A's -- the proper-prefix rule, for free;
//   * refinement: rows still tied (equal keys, neither exhausted) form runs; every round the tied rows alone are gathered, keyed by
//     their NEXT 32 bases and sorted stably by (run, key) -- two radix sorts, least significant first -- and written back into
//     their run's slots, until no two rows of a run are tied.  A random genome is done after two rounds for almost every row;
//     repeat families and segmental duplications keep ~1 % of the rows busy for a few dozen rounds; exact repeats of length L
//     need L / 32 rounds over their own rows only.
// Also produced per row: the base preceding the suffix (the BWT symbol), so that the host does not gather it.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

// 32 bases starting at i (base t of a word at bits 62 - 2t; zero past n: the array is zero-padded by two words)
__device__ __forceinline__ uint64_t sa_get32(const uint64_t *w, uint64_t n, uint64_t i)
{
	if (i >= n) return 0;
	const unsigned s = (unsigned)(i & 31) << 1;
	const uint64_t a = w[i >> 5];
	return s ? (a << s) | (w[(i >> 5) + 1] >> (64 - s)) : a;
}

struct RevIndex {      // i -> position top - i: positions in descending order
	uint64_t top;
	__device__ uint64_t operator()(uint64_t i) const { return top - i; }
};
struct InChunk {       // does the suffix at p start with the chunk's two bases?  (the last suffix is padded with an 'A')
	const uint64_t *w; uint64_t n; unsigned c;
	__device__ bool operator()(const uint64_t &p) const { return (unsigned)(sa_get32(w, n, p) >> 60) == c; }
};
struct IsSet { __device__ bool operator()(const uint8_t &f) const { return f != 0; } };
struct MaxOp { __device__ uint32_t operator()(const uint32_t &a, const uint32_t &b) const { return a > b ? a : b; } };

__global__ void k_count_chunks(const uint64_t *w, uint64_t n, unsigned long long *cnt)
{
	__shared__ unsigned long long local[16];
	if (threadIdx.x < 16) local[threadIdx.x] = 0;
	__syncthreads();
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x)
		atomicAdd(&local[(unsigned)(sa_get32(w, n, p) >> 60)], 1ULL);
	__syncthreads();
	if (threadIdx.x < 16) atomicAdd(&cnt[threadIdx.x], local[threadIdx.x]);
}

__global__ void k_keys(const uint64_t *w, uint64_t n, const uint64_t *pos, uint64_t depth, uint64_t *key, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) key[k] = sa_get32(w, n, pos[k] + depth);
}

__global__ void k_iota(uint32_t *a, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) a[k] = (uint32_t)k;
}

// head[k]: row k starts a run (first row, another run or key than its predecessor, or the predecessor is exhausted at `depth`);
// headidx[k] = k at a head, 0 elsewhere (its running maximum is the run's id).  run == nullptr: one run (the first round).
__global__ void k_heads(const uint64_t *pos, const uint64_t *key, const uint32_t *run, uint64_t n, uint64_t depth, uint8_t *head, uint32_t *headidx, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= m) return;
	const bool h = k == 0 || (run && run[k] != run[k - 1]) || key[k] != key[k - 1] || pos[k - 1] + depth >= n;
	head[k] = h ? 1 : 0;
	headidx[k] = h ? (uint32_t)k : 0;
}

// act[k]: row k is tied with a neighbour of its run and has bases left at `depth`
__global__ void k_active(const uint64_t *pos, const uint8_t *head, uint64_t n, uint64_t depth, uint8_t *act, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= m) return;
	const bool live = pos[k] + depth < n;
	act[k] = (live && (!head[k] || (k + 1 < m && !head[k + 1]))) ? 1 : 0;
}

template <typename T>
__global__ void k_gather(const T *src, const uint32_t *idx, T *dst, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) dst[k] = src[idx[k]];
}

__global__ void k_scatter_pos(const uint64_t *pos, const uint32_t *slot, uint64_t *dst, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) dst[slot[k]] = pos[k];
}

// the chunk's rows as the host wants them: suffix array rows of `width` bytes and the base before each suffix (4: none, row of suffix 0)
__global__ void k_rows_out(const uint64_t *w, const uint64_t *pos, int width, void *rows, uint8_t *prev, size_t m)
{
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= m) return;
	const uint64_t p = pos[k];
	if (width == 4) reinterpret_cast<uint32_t *>(rows)[k] = (uint32_t)p; else reinterpret_cast<uint64_t *>(rows)[k] = p;
	prev[k] = p ? (uint8_t)(w[(p - 1) >> 5] >> (62 - (((p - 1) & 31) << 1)) & 3) : 4;
}

struct Bufs {      // everything the chunks share, sized for the largest one
	void *tmp = nullptr; size_t tmp_bytes = 0;
	uint64_t *pos[2] = {nullptr, nullptr}, *key[2] = {nullptr, nullptr};      // the chunk (G = pos[cur]), and sort partners
	uint64_t *apos[2] = {nullptr, nullptr}, *akey[2] = {nullptr, nullptr};    // the tied rows
	uint32_t *slot[2] = {nullptr, nullptr}, *run[2] = {nullptr, nullptr}, *idx[2] = {nullptr, nullptr}, *headidx = nullptr;
	uint8_t *head = nullptr, *act = nullptr, *prev = nullptr;
	void *rows = nullptr;
	size_t *d_count = nullptr;
};

#define SA_CHECK(call) do { hipError_t rc_ = (call); if (rc_ != hipSuccess) { fprintf(stderr, "[gpu-sa] %s: %s\n", #call, hipGetErrorString(rc_)); return 1; } } while (0)

inline unsigned grid_for(size_t m) { return (unsigned)((m + 255) / 256); }

}  // namespace

// text: the 2-bit text as index_build.cpp packs it ((n >> 5) + 3 words, zero past n); rows: (n + 1) x width bytes, row 0 = n;
// prev: n + 1 bytes, the base before each row's suffix (4 for the row of suffix 0, which has none).  0 on success; 1 = no device, no
// memory or a runtime error (the caller then sorts on the host).
extern "C" int ema_gpu_suffix_array(const uint64_t *text, uint64_t n, int width, void *rows, uint8_t *prev, int verbose)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return 1;
	if (n < 64 || (width != 4 && width != 8) || (width == 4 && n >= (1ULL << 32))) return 1;
	hipEvent_t ev0, ev1;
	SA_CHECK(hipEventCreate(&ev0)); SA_CHECK(hipEventCreate(&ev1));
	SA_CHECK(hipEventRecord(ev0, nullptr));
	const size_t n_words = (size_t)(n >> 5) + 3;
	uint64_t *d_text = nullptr;
	SA_CHECK(hipMalloc(&d_text, n_words * 8));
	SA_CHECK(hipMemcpy(d_text, text, n_words * 8, hipMemcpyHostToDevice));
	unsigned long long *d_cnt = nullptr, cnt[16];
	SA_CHECK(hipMalloc(&d_cnt, 16 * 8));
	SA_CHECK(hipMemset(d_cnt, 0, 16 * 8));
	hipLaunchKernelGGL(k_count_chunks, dim3(4096), dim3(256), 0, nullptr, d_text, n, d_cnt);
	SA_CHECK(hipMemcpy(cnt, d_cnt, 16 * 8, hipMemcpyDeviceToHost));
	size_t m_max = 0;
	for (int c = 0; c < 16; ++c) m_max = std::max<size_t>(m_max, (size_t)cnt[c]);
	if (m_max >= (1ULL << 32)) { fprintf(stderr, "[gpu-sa] a two-base chunk of %zu rows exceeds 32-bit slots\n", m_max); (void)hipFree(d_text); (void)hipFree(d_cnt); return 1; }

	Bufs B;
	const size_t range = (size_t)1 << 30;      // positions examined per select call
	{   // temporary storage: the largest request of the operations below
		size_t need = 0, b = 0;
		auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RevIndex{0});
		(void)rocprim::select(nullptr, b, in, (uint64_t *)nullptr, (size_t *)nullptr, range, InChunk{d_text, n, 0}); need = std::max(need, b);
		(void)rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, m_max, 0u, 64u); need = std::max(need, b);
		(void)rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, m_max, 0u, 64u); need = std::max(need, b);
		(void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, m_max, 0u, 32u); need = std::max(need, b);
		(void)rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, m_max, MaxOp()); need = std::max(need, b);
		(void)rocprim::select(nullptr, b, (uint64_t *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr, (size_t *)nullptr, m_max); need = std::max(need, b);
		B.tmp_bytes = 2 * need + (1 << 20);
	}
	auto alloc_all = [&]() -> int {
		SA_CHECK(hipMalloc(&B.tmp, B.tmp_bytes));
		for (int i = 0; i < 2; ++i) {
			SA_CHECK(hipMalloc(&B.pos[i], (m_max + 1) * 8)); SA_CHECK(hipMalloc(&B.key[i], (m_max + 1) * 8));
			SA_CHECK(hipMalloc(&B.apos[i], (m_max + 1) * 8)); SA_CHECK(hipMalloc(&B.akey[i], (m_max + 1) * 8));
			SA_CHECK(hipMalloc(&B.slot[i], (m_max + 1) * 4)); SA_CHECK(hipMalloc(&B.run[i], (m_max + 1) * 4)); SA_CHECK(hipMalloc(&B.idx[i], (m_max + 1) * 4));
		}
		SA_CHECK(hipMalloc(&B.headidx, (m_max + 1) * 4));
		SA_CHECK(hipMalloc(&B.head, m_max + 1)); SA_CHECK(hipMalloc(&B.act, m_max + 1)); SA_CHECK(hipMalloc(&B.prev, m_max + 1));
		SA_CHECK(hipMalloc(&B.rows, (m_max + 1) * (size_t)width));
		SA_CHECK(hipMalloc(&B.d_count, 16));
		return 0;
	};
	auto free_all = [&] {
		(void)hipFree(B.tmp);
		for (int i = 0; i < 2; ++i) { (void)hipFree(B.pos[i]); (void)hipFree(B.key[i]); (void)hipFree(B.apos[i]); (void)hipFree(B.akey[i]); (void)hipFree(B.slot[i]); (void)hipFree(B.run[i]); (void)hipFree(B.idx[i]); }
		(void)hipFree(B.headidx); (void)hipFree(B.head); (void)hipFree(B.act); (void)hipFree(B.prev); (void)hipFree(B.rows); (void)hipFree(B.d_count);
		(void)hipFree(d_text); (void)hipFree(d_cnt);
	};
	if (alloc_all()) { free_all(); return 1; }

	int rc = 0;
	size_t row0 = 1, total_rounds = 0, total_tied = 0;
	if (width == 4) reinterpret_cast<uint32_t *>(rows)[0] = (uint32_t)n; else reinterpret_cast<uint64_t *>(rows)[0] = n;
	prev[0] = (uint8_t)(text[(n - 1) >> 5] >> (62 - (((n - 1) & 31) << 1)) & 3);      // row 0 is the empty suffix: the base before it is the text's last
	auto run_chunk = [&](unsigned c) -> int {
		const size_t m = (size_t)cnt[c];
		if (!m) return 0;
		// 1. the chunk's positions, descending
		size_t got = 0;
		for (uint64_t hi = n; hi > 0;) {
			const uint64_t len = std::min<uint64_t>(hi, range);
			auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RevIndex{hi - 1});
			size_t b = B.tmp_bytes;
			SA_CHECK(rocprim::select(B.tmp, b, in, B.pos[0] + got, B.d_count, (size_t)len, InChunk{d_text, n, c}));
			size_t k = 0;
			SA_CHECK(hipMemcpy(&k, B.d_count, sizeof(size_t), hipMemcpyDeviceToHost));
			got += k;
			hi -= len;
		}
		if (got != m) { fprintf(stderr, "[gpu-sa] chunk %u: %zu positions collected, %zu counted\n", c, got, m); return 1; }
		// 2. first 32 bases as the key; stable sort (the two leading bases are the chunk's: 60 bits)
		hipLaunchKernelGGL(k_keys, dim3(grid_for(m)), dim3(256), 0, nullptr, d_text, n, B.pos[0], (uint64_t)0, B.key[0], m);
		{
			size_t b = B.tmp_bytes;
			SA_CHECK(rocprim::radix_sort_pairs(B.tmp, b, B.key[0], B.key[1], B.pos[0], B.pos[1], m, 0u, 60u));
		}
		uint64_t *G = B.pos[1];      // the chunk's rows, final where no longer tied
		// 3. refinement: the tied rows, round by round.  Current set: (slot, pos, key, run) x cur_n, in row order.
		const uint64_t *c_pos = G, *c_key = B.key[1];
		const uint32_t *c_run = nullptr, *c_slot = nullptr;      // first round: the whole chunk, slot k = k, one run
		size_t cur_n = m;
		int flip = 0;
		for (uint64_t depth = 32; cur_n > 1; depth += 32) {
			hipLaunchKernelGGL(k_heads, dim3(grid_for(cur_n)), dim3(256), 0, nullptr, c_pos, c_key, c_run, n, depth, B.head, B.headidx, cur_n);
			hipLaunchKernelGGL(k_active, dim3(grid_for(cur_n)), dim3(256), 0, nullptr, c_pos, B.head, n, depth, B.act, cur_n);
			size_t b = B.tmp_bytes;
			SA_CHECK(rocprim::inclusive_scan(B.tmp, b, B.headidx, B.headidx, cur_n, MaxOp()));      // run id = index of the run's head
			// the tied rows, in order: their slot in the chunk, position, run
			uint32_t *n_slot = B.slot[flip], *n_run = B.run[flip];
			uint64_t *n_pos = B.apos[flip];
			size_t n_act = 0;
			b = B.tmp_bytes;
			SA_CHECK(rocprim::select(B.tmp, b, c_pos, B.act, n_pos, B.d_count, cur_n));
			SA_CHECK(hipMemcpy(&n_act, B.d_count, sizeof(size_t), hipMemcpyDeviceToHost));
			if (n_act == 0) break;
			b = B.tmp_bytes;
			SA_CHECK(rocprim::select(B.tmp, b, B.headidx, B.act, n_run, B.d_count, cur_n));
			b = B.tmp_bytes;
			if (c_slot) SA_CHECK(rocprim::select(B.tmp, b, c_slot, B.act, n_slot, B.d_count, cur_n));
			else SA_CHECK(rocprim::select(B.tmp, b, rocprim::make_counting_iterator<uint32_t>(0), B.act, n_slot, B.d_count, cur_n));
			// key = the next 32 bases; stable sort by (run, key): by key first, then by run
			uint64_t *k_in = B.akey[flip], *k_out = B.akey[flip ^ 1];
			hipLaunchKernelGGL(k_keys, dim3(grid_for(n_act)), dim3(256), 0, nullptr, d_text, n, n_pos, depth, k_in, n_act);
			hipLaunchKernelGGL(k_iota, dim3(grid_for(n_act)), dim3(256), 0, nullptr, B.idx[0], n_act);
			b = B.tmp_bytes;
			SA_CHECK(rocprim::radix_sort_pairs(B.tmp, b, k_in, k_out, B.idx[0], B.idx[1], n_act, 0u, 64u));
			// rows in key order: run ids, then the second sort (by run; ids are indices into the current set: < cur_n)
			hipLaunchKernelGGL(k_gather<uint32_t>, dim3(grid_for(n_act)), dim3(256), 0, nullptr, n_run, B.idx[1], B.run[flip ^ 1], n_act);
			hipLaunchKernelGGL(k_iota, dim3(grid_for(n_act)), dim3(256), 0, nullptr, B.idx[0], n_act);
			unsigned bits = 1;
			while (bits < 32 && ((size_t)1 << bits) < cur_n) ++bits;
			b = B.tmp_bytes;
			SA_CHECK(rocprim::radix_sort_pairs(B.tmp, b, B.run[flip ^ 1], n_run, B.idx[0], B.headidx /* as scratch: idx of sort 2 */, n_act, 0u, bits));
			// compose: final row k takes the row that sort 1 put at idx2[k]
			uint32_t *idx2 = B.headidx;
			hipLaunchKernelGGL(k_gather<uint32_t>, dim3(grid_for(n_act)), dim3(256), 0, nullptr, B.idx[1], idx2, B.idx[0], n_act);      // idx[0][k] = original index
			hipLaunchKernelGGL(k_gather<uint64_t>, dim3(grid_for(n_act)), dim3(256), 0, nullptr, n_pos, B.idx[0], B.apos[flip ^ 1], n_act);
			hipLaunchKernelGGL(k_gather<uint64_t>, dim3(grid_for(n_act)), dim3(256), 0, nullptr, k_out, idx2, k_in, n_act);      // keys in final order
			hipLaunchKernelGGL(k_scatter_pos, dim3(grid_for(n_act)), dim3(256), 0, nullptr, B.apos[flip ^ 1], n_slot, G, n_act);
			SA_CHECK(hipGetLastError());
			c_pos = B.apos[flip ^ 1]; c_key = k_in; c_run = n_run; c_slot = n_slot;
			cur_n = n_act;
			total_tied += n_act; ++total_rounds;
			// next round writes slot / run / apos[flip'] with flip' such that it does not overwrite what it reads: c_slot, c_run live in
			// [flip], c_pos in apos[flip ^ 1], c_key in akey[flip] -- so the next round's outputs go to slot / run / apos [flip ^ 1] ... but
			// apos[flip ^ 1] holds c_pos: copy it aside first
			SA_CHECK(hipMemcpyAsync(B.pos[0], c_pos, cur_n * 8, hipMemcpyDeviceToDevice, nullptr));      // pos[0] is free after the first sort
			SA_CHECK(hipMemcpyAsync(B.key[0], c_key, cur_n * 8, hipMemcpyDeviceToDevice, nullptr));
			c_pos = B.pos[0]; c_key = B.key[0];
			flip ^= 1;
		}
		// 4. the chunk's rows to the host
		hipLaunchKernelGGL(k_rows_out, dim3(grid_for(m)), dim3(256), 0, nullptr, d_text, G, width, B.rows, B.prev, m);
		SA_CHECK(hipGetLastError());
		SA_CHECK(hipMemcpy((uint8_t *)rows + row0 * (size_t)width, B.rows, m * (size_t)width, hipMemcpyDeviceToHost));
		SA_CHECK(hipMemcpy(prev + row0, B.prev, m, hipMemcpyDeviceToHost));
		row0 += m;
		return 0;
	};
	for (unsigned c = 0; c < 16 && !rc; ++c) rc = run_chunk(c);
	if (!rc && row0 != n + 1) { fprintf(stderr, "[gpu-sa] %zu rows written, %llu expected\n", row0, (unsigned long long)(n + 1)); rc = 1; }
	(void)hipEventRecord(ev1, nullptr);
	(void)hipEventSynchronize(ev1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, ev0, ev1);
	if (verbose && !rc) fprintf(stderr, "[gpu-sa] %llu suffixes in %.2f s on the device: 16 chunks (largest %zu rows), %zu refinement rounds over %zu tied rows in all\n",
	                            (unsigned long long)n, ms * 1e-3, m_max, total_rounds, total_tied);
	(void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
	free_all();
	return rc;
}
