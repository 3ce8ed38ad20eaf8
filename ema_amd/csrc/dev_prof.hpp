// ema_amd/csrc/dev_prof.hpp -- the wave-per-item kernels' clocks for `make prof-lib` (-DEMA_K34_PROF): a development build of the
// library (ema_amd/libema_engine_prof.so, tools/gpu_k34_profile.py) in which K3b / K3t / K3r and K4b / K4t / K4r keep eight phase
// clocks in scalar registers, as K2b's PROF 2 build does (k_align.hip).  Without the macro every line here compiles to nothing:
// the product's kernels carry no trace of it.
#ifndef EMA_DEV_PROF_HPP
#define EMA_DEV_PROF_HPP
#ifdef EMA_K34_PROF
struct EmaLp {
	unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, mark = 0;
	unsigned long long items = 0;
	__device__ __forceinline__ void start() { t0 = mark = __builtin_amdgcn_s_memtime(); }
	__device__ __forceinline__ void upto(int k) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t[k] += n - mark; mark = n; }
	// slots of one mode: phases 0..7, lifetimes, work items, wavefronts
	__device__ __forceinline__ void flush(unsigned long long *o)
	{
		for (int k = 0; k < 8; ++k) atomicAdd(o + k, t[k]);
		atomicAdd(o + 8, __builtin_amdgcn_s_memtime() - t0); atomicAdd(o + 9, items); atomicAdd(o + 10, 1ULL);
	}
};
#define EMA_LP_DECL(lp) EmaLp lp; lp.start()
#define EMA_LP_UPTO(lp, k) (lp).upto(k)
#define EMA_LP_ITEM(lp) (++(lp).items)
#define EMA_LP_FLUSH(lp, out) do { if (ema_lane() == 0) (lp).flush(out); } while (0)
#else
struct EmaLp {};
#define EMA_LP_DECL(lp) EmaLp lp; (void)lp
#define EMA_LP_UPTO(lp, k) do {} while (0)
#define EMA_LP_ITEM(lp) do {} while (0)
#define EMA_LP_FLUSH(lp, out) do {} while (0)
#endif
#endif
