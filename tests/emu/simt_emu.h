// tests/emu/simt_emu.h -- a tiny lock-step SIMT interpreter for debugging the HIP kernels on a
// machine without a GPU.  TEST INFRASTRUCTURE: it compiles the unmodified kernel sources
// (ema_amd/csrc/k_*.hip) as plain C++ and runs every 64-lane wavefront as 64 ucontext fibers
// scheduled round-robin; each cross-lane primitive (__shfl*, __ballot, ...) is one scheduling
// round.  It models a wavefront only: no __syncthreads, no LDS sharing between waves, no
// memory model.  It also checks convergence: all live lanes of a wave must reach the same
// cross-lane call site in the same round, otherwise it aborts (that is a kernel bug on the GPU too).
//
// The results produced through this harness are NOT parity evidence -- the `-m gpu` tests are.
#pragma once
#include <ucontext.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

namespace emu {
struct Lane {
	ucontext_t ctx;
	std::vector<char> stack;
	bool done = false;
	long site = 0;
	unsigned par = 0;
};
struct Wave {
	Lane lane[64];
	ucontext_t sched;
	uint64_t slot[2][64];
	int cur = 0;
	unsigned wave_base = 0;
	dim3 bid, bdim, gdim;
	std::function<void()> body;
};
inline Wave *W = nullptr;
inline dim3 tid() { return dim3(W->wave_base + (unsigned)W->cur); }
inline void yield(long site) {
	Lane &l = W->lane[W->cur];
	l.site = site;
	swapcontext(&l.ctx, &W->sched);
}
inline void trampoline() {
	W->body();
	W->lane[W->cur].done = true;
	swapcontext(&W->lane[W->cur].ctx, &W->sched);
}
inline void run_wave_at(Wave &w, unsigned base, unsigned n_lanes) {
	W = &w;
	w.wave_base = base;
	for (unsigned l = 0; l < 64; ++l) {
		Lane &L = w.lane[l];
		L.done = l >= n_lanes; L.par = 0; L.site = 0;
		if (L.done) continue;
		if (L.stack.empty()) L.stack.resize(256 * 1024);
		getcontext(&L.ctx);
		L.ctx.uc_stack.ss_sp = L.stack.data();
		L.ctx.uc_stack.ss_size = L.stack.size();
		L.ctx.uc_link = nullptr;
		makecontext(&L.ctx, (void (*)())trampoline, 0);
	}
	for (;;) {
		bool any = false;
		long site = 0; bool have = false;
		for (unsigned l = 0; l < 64; ++l) {
			Lane &L = w.lane[l];
			if (L.done) continue;
			w.cur = (int)l;
			swapcontext(&w.sched, &L.ctx);
			if (!L.done) {
				any = true;
				if (!have) { site = L.site; have = true; }
				else if (site != L.site) {
					fprintf(stderr, "simt_emu: divergent cross-lane call: lane %u is at source line %ld, earlier lanes at line %ld\n", l, L.site, site);
					abort();
				}
			}
		}
		if (!any) break;
	}
}
inline unsigned lane_id() { return (unsigned)W->cur; }
template <typename T> inline T exch(T v, int src, long site) {
	static_assert(sizeof(T) <= 8, "shuffle width");
	Lane &L = W->lane[W->cur];
	unsigned p = L.par & 1; ++L.par;
	uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
	W->slot[p][W->cur] = raw;
	yield(site);
	uint64_t r = W->slot[p][src & 63];
	T out; memcpy(&out, &r, sizeof(T));
	return out;
}
}  // namespace emu

#define threadIdx (emu::tid())
#define blockIdx (emu::W->bid)
#define blockDim (emu::W->bdim)
#define gridDim (emu::W->gdim)
#define warpSize 64

template <typename T> __attribute__((noinline)) T __shfl(T v, int src, int width = 64, int line_ = __builtin_LINE()) {
	int l = (int)emu::lane_id();
	int s = (l & ~(width - 1)) | (src & (width - 1));
	return emu::exch(v, s, line_);
}
template <typename T> __attribute__((noinline)) T __shfl_xor(T v, int mask, int width = 64, int line_ = __builtin_LINE()) {
	int l = (int)emu::lane_id();
	int s = l ^ mask;
	if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
	return emu::exch(v, s, line_);
}
template <typename T> __attribute__((noinline)) T __shfl_up(T v, unsigned d, int width = 64, int line_ = __builtin_LINE()) {
	int l = (int)emu::lane_id();
	int s = l - (int)d;
	if (s < (l & ~(width - 1))) s = l;
	return emu::exch(v, s, line_);
}
template <typename T> __attribute__((noinline)) T __shfl_down(T v, unsigned d, int width = 64, int line_ = __builtin_LINE()) {
	int l = (int)emu::lane_id();
	int s = l + (int)d;
	if (s > (l | (width - 1))) s = l;
	return emu::exch(v, s, line_);
}
__attribute__((noinline)) inline unsigned long long __ballot(int pred, int line_ = __builtin_LINE()) {
	emu::Lane &L = emu::W->lane[emu::W->cur];
	unsigned p = L.par & 1; ++L.par;
	emu::W->slot[p][emu::W->cur] = pred ? 1 : 0;
	emu::yield(line_);
	unsigned long long m = 0;
	for (int i = 0; i < 64; ++i)
		if (!emu::W->lane[i].done && emu::W->slot[p][i]) m |= 1ULL << i;
	return m;
}
__attribute__((noinline)) inline int __any(int pred, int line_ = __builtin_LINE()) { return __ballot(pred, line_) != 0; }
__attribute__((noinline)) inline int __all(int pred, int line_ = __builtin_LINE()) {
	unsigned long long live = __ballot(1, line_);
	return __ballot(pred, line_) == live;
}
__attribute__((noinline)) inline void __builtin_amdgcn_wave_barrier(int line_ = __builtin_LINE()) { emu::yield(line_); }
// readfirstlane: value of the first live lane; the interpreter also checks that the value really is wave-uniform
__attribute__((noinline)) inline int __builtin_amdgcn_readfirstlane(int v, int line_ = __builtin_LINE()) {
	emu::Lane &L = emu::W->lane[emu::W->cur];
	unsigned p = L.par & 1; ++L.par;
	emu::W->slot[p][emu::W->cur] = (uint64_t)(uint32_t)v;
	emu::yield(line_);
	int first = -1;
	for (int i = 0; i < 64; ++i) if (!emu::W->lane[i].done) { first = i; break; }
	const int r = (int)(uint32_t)emu::W->slot[p][first];
	if (r != v) { fprintf(stderr, "simt_emu: value marked wave-uniform at line %d differs between lanes (%d vs %d)\n", line_, v, r); abort(); }
	return r;
}
// DPP move: row_shr:n (0x110+n), row_bcast:15 (0x142), row_bcast:31 (0x143), wave_shr:1 (0x138), wave_shl:1 (0x130); bound_ctrl = false
__attribute__((noinline)) inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool, int line_ = __builtin_LINE()) {
	emu::Lane &L = emu::W->lane[emu::W->cur];
	unsigned p = L.par & 1; ++L.par;
	emu::W->slot[p][emu::W->cur] = (uint64_t)(uint32_t)v;
	emu::yield(line_);
	const int lane = emu::W->cur, row = lane >> 4, bank = (lane & 15) >> 2;
	if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
	int src = -1;
	if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if ((lane & 15) >= n) src = lane - n; }
	else if (ctrl == 0x142) { if (row >= 1) src = row * 16 - 1; }
	else if (ctrl == 0x143) { if (row >= 2) src = 31; }
	else if (ctrl == 0x138) { if (lane >= 1) src = lane - 1; }
	else if (ctrl == 0x130) { if (lane < 63) src = lane + 1; }      // wave_shl:1
	else { fprintf(stderr, "simt_emu: unsupported DPP control 0x%x\n", ctrl); abort(); }
	if (src < 0 || emu::W->lane[src].done) return old;
	return (int)(uint32_t)emu::W->slot[p][src];
}
__attribute__((noinline)) inline int __builtin_amdgcn_readlane(int v, int src, int line_ = __builtin_LINE()) {
	emu::Lane &L = emu::W->lane[emu::W->cur];
	unsigned p = L.par & 1; ++L.par;
	emu::W->slot[p][emu::W->cur] = (uint64_t)(uint32_t)v;
	emu::yield(line_);
	return (int)(uint32_t)emu::W->slot[p][src & 63];
}
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
inline void __builtin_amdgcn_fence(int, const char *) {}
inline void __builtin_amdgcn_s_waitcnt(int) {}
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <typename T> inline void __hip_atomic_store(T *p, T v, int, int) { *p = v; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline void __syncthreads() { fprintf(stderr, "simt_emu: __syncthreads is not modelled\n"); abort(); }
inline void __threadfence_block() {}
inline void __threadfence() {}
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicSub(T *p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
using std::min;
using std::max;

struct uint4 { unsigned x, y, z, w; };
struct ulong2 { unsigned long long x, y; };

template <typename K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 1; return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                            \
	do {                                                                                     \
		dim3 g_ = (grid); dim3 b_ = (block);                                                           \
		static emu::Wave wave_;                                                              \
		for (unsigned bx_ = 0; bx_ < g_.x; ++bx_)                                            \
			for (unsigned w0_ = 0; w0_ < b_.x; w0_ += 64) {                                  \
				wave_.bid = dim3(bx_); wave_.bdim = b_; wave_.gdim = g_;                     \
				wave_.body = [&]() { kern(__VA_ARGS__); };                                   \
				emu::run_wave_at(wave_, w0_, std::min(64u, b_.x - w0_));                     \
			}                                                                                \
	} while (0)
