// ema_amd/csrc/host_cpuacct.h -- CPU seconds of the host stages, by stage (diagnostics: ema_host_cpu_seconds, include/ema_stream.h).
// The stages overlap in time and each spreads over short-lived worker threads, so wall clocks and the process's rusage cannot say
// which stage the host's cores went to; every worker body and every stage entry point opens a scope that charges ITS THREAD's CPU
// clock to the stage (the outermost scope of a thread counts; a worker lambda run inline by the caller is covered by the caller's).
#ifndef EMA_HOST_CPUACCT_H
#define EMA_HOST_CPUACCT_H
#include <atomic>
#include <stdint.h>
#include <time.h>

enum { EMA_CPU_READER = 0, EMA_CPU_STAGE, EMA_CPU_FETCH, EMA_CPU_APPEND, EMA_CPU_CLOUDS, EMA_CPU_FORMAT, EMA_CPU_OTHER, EMA_CPU_N };

inline std::atomic<uint64_t> ema_cpu_ns[EMA_CPU_N];
inline thread_local int ema_cpu_depth = 0;
inline thread_local int ema_cpu_current = EMA_CPU_OTHER;      // the stage of this thread's outermost scope: helpers that fan out pass it on

struct EmaCpuScope {
	int stage;
	uint64_t t0 = 0;
	static uint64_t now()
	{
		struct timespec ts;
		clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
		return (uint64_t)ts.tv_sec * 1000000000ULL + (uint64_t)ts.tv_nsec;
	}
	explicit EmaCpuScope(int s) : stage(s) { if (ema_cpu_depth++ == 0) { t0 = now(); ema_cpu_current = s; } }
	~EmaCpuScope() { if (--ema_cpu_depth == 0) ema_cpu_ns[stage].fetch_add(now() - t0, std::memory_order_relaxed); }
	EmaCpuScope(const EmaCpuScope &) = delete;
	EmaCpuScope &operator=(const EmaCpuScope &) = delete;
};
#define EMA_CPU(stage) EmaCpuScope ema_cpu_scope_(stage)
#endif
