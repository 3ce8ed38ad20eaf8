// tools/gather_microbench.hip -- what the chip gives for K1's access pattern: dependent random 32-byte gathers (two 16-byte
// loads of one aligned 32-byte block per lane, like one rank query) over a table far beyond the Infinity Cache.
// Dev aid, built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_mb tools/gather_microbench.hip && /tmp/gather_mb
// Prints G gathers/s and GB/s (32 B counted per gather) by table size, waves per CU and independent gathers per lane and step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

struct Blk { uint32_t c[4]; uint64_t b[2]; };

template <int ILP>
__global__ void __launch_bounds__(256) k_gather(const Blk *__restrict__ tab, uint64_t n_blk, int steps, uint64_t *out)
{
	uint64_t x[ILP];
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
	for (int j = 0; j < ILP; ++j) x[j] = (tid * ILP + j) * 0x9E3779B97F4A7C15ULL;
	uint64_t acc = 0;
	for (int s = 0; s < steps; ++s) {
		uint4 h[ILP]; ulong2 v[ILP];
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			const Blk *p = tab + (x[j] % n_blk);
			h[j] = *reinterpret_cast<const uint4 *>(p);
			v[j] = *(reinterpret_cast<const ulong2 *>(p) + 1);
		}
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			const uint64_t m = (uint64_t)h[j].x + h[j].y + h[j].z + h[j].w + __popcll(v[j].x) + __popcll(v[j].y);
			acc += m;
			x[j] = (x[j] ^ m) * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL;      // next address depends on the loaded data
			x[j] ^= x[j] >> 29;
		}
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int ILP> double run(const Blk *tab, uint64_t n_blk, int blocks, int steps, uint64_t *out)
{
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k_gather<ILP>, dim3(blocks), dim3(256), 0, 0, tab, n_blk, 8, out);
	hipEventRecord(a, 0);
	hipLaunchKernelGGL(k_gather<ILP>, dim3(blocks), dim3(256), 0, 0, tab, n_blk, steps, out);
	hipEventRecord(b, 0);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	return (double)blocks * 256 * ILP * steps / (ms * 1e-3);
}

int main()
{
	const uint64_t max_blk = (uint64_t)6 << 30 >> 5;      // 6 GB of 32-byte blocks
	Blk *tab = nullptr; uint64_t *out = nullptr;
	if (hipMalloc(&tab, max_blk * sizeof(Blk)) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
	hipMemset(tab, 0x5a, max_blk * sizeof(Blk));
	hipDeviceSynchronize();
	const double sizes_gb[] = {0.0625, 0.25, 3.1, 6.0};
	printf("table_GB waves_per_CU ilp  Ggathers/s  GB/s(32B)\n");
	for (double gb : sizes_gb) {
		const uint64_t n_blk = (uint64_t)(gb * (1 << 30)) >> 5;
		for (int bpc : {1, 2, 4, 8}) {      // 256-thread blocks per CU = 4 waves each
			const int blocks = 256 * bpc;
			const double r1 = run<1>(tab, n_blk, blocks, 512, out);
			const double r2 = run<2>(tab, n_blk, blocks, 256, out);
			const double r4 = run<4>(tab, n_blk, blocks, 128, out);
			printf("%7.3f %5d %3d %10.2f %9.1f\n", gb, bpc * 4, 1, r1 / 1e9, r1 * 32 / 1e9);
			printf("%7.3f %5d %3d %10.2f %9.1f\n", gb, bpc * 4, 2, r2 / 1e9, r2 * 32 / 1e9);
			printf("%7.3f %5d %3d %10.2f %9.1f\n", gb, bpc * 4, 4, r4 / 1e9, r4 * 32 / 1e9);
			fflush(stdout);
		}
	}
	return 0;
}
