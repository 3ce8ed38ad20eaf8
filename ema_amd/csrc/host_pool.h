// ema_amd/csrc/host_pool.h -- the host stages' worker threads: ONE pool per process, as many threads as the host grants
// (EMA_HOST_THREADS, default min(32, hardware threads)).
//
// Until round 3 every data-parallel pass of every stage (reader, staging, result assembly, append, clouds, formatter) started its
// own std::threads, up to that many each.  In a stream the stages overlap: five of them at once meant five times as many runnable
// threads as CPUs, each running in short slices on a cold cache -- the formatter cost four times as many CPU-seconds per line
// inside ema_stream_sam as alone (r03: 0.72 us against 0.18).  Now a pass hands its pieces to the pool and the calling thread
// works on them too, so a pass started from inside another pass's piece cannot wait for a thread that is waiting for it.
#ifndef EMA_HOST_POOL_H
#define EMA_HOST_POOL_H
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

class EmaPool {
public:
	static EmaPool &get()
	{
		static EmaPool *p = new EmaPool();      // never destroyed: its threads sleep until the process ends
		return *p;
	}
	int size() const { return n_; }      // threads a pass can count on, the caller included
	// fn(i) for every i in [0, n), on the pool's threads and on the caller; returns when all have run
	template <typename F> void run(size_t n, F &&fn)
	{
		if (n == 0) return;
		if (n == 1 || n_ <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
		auto job = std::make_shared<Job>();
		job->n = n;
		job->fn = [&fn](size_t i) { fn(i); };
		{
			std::lock_guard<std::mutex> lk(mu_);
			jobs_.push_back(job);
		}
		cv_.notify_all();
		work_on(*job);
		std::unique_lock<std::mutex> lk(job->mu);
		job->cv.wait(lk, [&] { return job->done.load() == job->n; });
	}

private:
	struct Job {
		std::function<void(size_t)> fn;
		size_t n = 0;
		std::atomic<size_t> next{0}, done{0};
		std::mutex mu;
		std::condition_variable cv;
	};
	int n_ = 1;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<std::shared_ptr<Job>> jobs_;

	EmaPool()
	{
		const char *v = getenv("EMA_HOST_THREADS");
		int t = v ? atoi(v) : (int)std::thread::hardware_concurrency();
		n_ = t < 1 ? 1 : t > 32 ? 32 : t;
		for (int k = 1; k < n_; ++k) std::thread([this] { worker(); }).detach();
	}
	static void work_on(Job &j)
	{
		for (;;) {
			const size_t i = j.next.fetch_add(1);
			if (i >= j.n) return;
			j.fn(i);
			if (j.done.fetch_add(1) + 1 == j.n) { std::lock_guard<std::mutex> lk(j.mu); j.cv.notify_all(); }
		}
	}
	void worker()
	{
		for (;;) {
			std::shared_ptr<Job> job;
			{
				std::unique_lock<std::mutex> lk(mu_);
				for (;;) {
					while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->n) jobs_.pop_front();      // all pieces handed out
					if (!jobs_.empty()) { job = jobs_.front(); break; }
					cv_.wait(lk);
				}
			}
			work_on(*job);
		}
	}
};
#endif
