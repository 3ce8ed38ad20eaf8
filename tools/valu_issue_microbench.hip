// tools/valu_issue_microbench.hip -- how many wave64 integer VALU instructions a SIMD of this chip issues per clock, measured: the peak
// `bench.py`'s roofline_k2b divides by (the guide's "one wave64 VALU instruction per 2 cycles" against this
// repository's own "one per four clocks" had never been reconciled).  Dev aid, built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_mb tools/valu_issue_microbench.hip && /tmp/valu_mb
// For 1, 2, 4, 6, 8 waves per SIMD (one 256-thread block = four waves = one per SIMD; blocks per CU = waves per SIMD) and four
// instruction kinds -- v_add_u32, v_bcnt_u32_b32, v_lshlrev_b64 and a 1:1 mix of v_add_u32 and v_and_b32 on independent registers --
// a loop of 8 x 64 dependent-chain-free instructions per lane runs N times; rate = instructions x waves / time, per SIMD and clock
// (clock from s_memtime / s_memrealtime inside the kernel) and chip-wide in G wave-instructions/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) k_valu(int iters, unsigned long long *clk, unsigned *sink)
{
	unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int i = 0; i < iters; ++i) {
		if (KIND == 0) {      // 64 x v_add_u32 on eight independent chains
			REP8(asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
			                  "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
			                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));)
		} else if (KIND == 1) {      // 64 x v_bcnt_u32_b32
			REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n\tv_bcnt_u32_b32 %1, %8, %1\n\tv_bcnt_u32_b32 %2, %8, %2\n\tv_bcnt_u32_b32 %3, %8, %3\n\t"
			                  "v_bcnt_u32_b32 %4, %8, %4\n\tv_bcnt_u32_b32 %5, %8, %5\n\tv_bcnt_u32_b32 %6, %8, %6\n\tv_bcnt_u32_b32 %7, %8, %7"
			                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));)
		} else if (KIND == 2) {      // 64 x v_lshlrev_b64 (counted as 64 instructions)
			REP8(asm volatile("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\t"
			                  "v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3"
			                  : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));)
		} else {      // 32 x v_add_u32 + 32 x v_and_b32, interleaved
			REP8(asm volatile("v_add_u32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\t"
			                  "v_add_u32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
			                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i | 0xff00));)
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
	if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)b0 ^ (unsigned)b1 ^ (unsigned)b2 ^ (unsigned)b3) == 0x12345678u) sink[0] = a0;
}

template <int KIND> static void run(const char *name, int n_cu)
{
	unsigned long long *clk; unsigned *sink;
	hipMalloc(&clk, 16); hipMalloc(&sink, 4);
	const int iters = 20000;
	for (int wps : {1, 2, 4, 6, 8}) {
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipLaunchKernelGGL(k_valu<KIND>, dim3(n_cu * wps), dim3(256), 0, 0, 100, clk, sink);      // warm-up
		hipDeviceSynchronize();
		hipEventRecord(e0);
		hipLaunchKernelGGL(k_valu<KIND>, dim3(n_cu * wps), dim3(256), 0, 0, iters, clk, sink);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
		const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;      // s_memrealtime ticks at 100 MHz
		const double inst_per_wave = 64.0 * iters;
		const double waves = (double)n_cu * 4 * wps;
		const double per_simd_clk = inst_per_wave * wps / (double)h[0];      // one block's four waves sit on the four SIMDs: wps waves per SIMD
		printf("%-26s %d waves/SIMD: %.3f wave-instructions per SIMD and clock (1 per %.2f clocks); chip %.0f G wave-instructions/s at %.2f GHz (%.2f ms)\n",
		       name, wps, per_simd_clk, 1.0 / per_simd_clk, inst_per_wave * waves / (ms * 1e-3) / 1e9, ghz, ms);
	}
	hipFree(clk); hipFree(sink);
}

int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
	run<0>("v_add_u32", p.multiProcessorCount);
	run<1>("v_bcnt_u32_b32", p.multiProcessorCount);
	run<2>("v_lshlrev_b64", p.multiProcessorCount);
	run<3>("v_add_u32 + v_and_b32", p.multiProcessorCount);
	return 0;
}
