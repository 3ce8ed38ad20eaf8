// ema_amd/csrc/host_stream.cpp -- the bucket loop around the hot path (include/ema_stream.h): reader -> engine -> append
// stage over a list of barcode buckets, pipelined on the host, results delivered in input order.
//
// Reference side: `ema align -x` walks its bucket files one after another (reference src/main.c:396-406), each through
// find_clouds_and_align(), which reads the whole bucket (src/align.c:258) and runs append_alignments() per pair
// (src/align.c:307-349).  The three steps here are the library's own C-ABI calls -- ema_bucket_read, ema_engine_stage /
// run / sync / fetch (or ema_engine_align_pairs for a bucket beyond the batch capacity), ema_batch_append_alignments --
// so this file adds scheduling only: one reader thread parsing ahead (the parser itself uses the host's cores), one
// worker per set of batch buffers (the engine and its peer), and the caller's thread handing buckets to the sink in
// order.  While one worker stages or fetches (PCIe + host cores), the other's kernels have the GPU.
#include <condition_variable>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include "ema_stream.h"
#include "host_cpuacct.h"
#include "host_pool.h"
#include "dev_bucket.h"

const char *ema_tuning_get(const char *key);      // engine.hip: the library's tuning string (ema_engine_set_tuning / EMA_TUNING)

namespace {

thread_local std::string g_err;

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

enum { ST_EMPTY = 0, ST_LOADED, ST_TAKEN, ST_DONE };

struct Item {
	ema_bucket *bk = nullptr;
	const char *bases = nullptr;
	const uint32_t *off = nullptr;
	size_t n_pairs = 0;
	ema_batch_out *b = nullptr;
	ema_aln_out *a = nullptr;
	int rc = EMA_OK, state = ST_EMPTY;
	int slot = -1;                 // >= 0: the batch is already staged in this input slot of its worker's engine
	std::string err;
	ema_bucket_stats st;
};

struct Stream {
	ema_engine_t *eng[2] = {nullptr, nullptr};
	int n_eng = 1;
	size_t cap = 0;
	ema_engine_opts eopts;
	ema_stream_opts o;
	std::vector<Item> items;
	const char *const *paths = nullptr;
	std::mutex mu, eng_mu[2], stage_mu;
	std::condition_variable cv;
	size_t delivered = 0, pairs_read = 0, pairs_delivered = 0;
	bool stop = false;
	bool dev_reader = false;     // buckets are read by ema_bucket_read_device: reads and qualities stay on `device` (ema_stream_sam)
	int device = 0;
	bool sink_kept = false;      // set by a sink of this file that took the bucket's objects over (they are not freed on its return)
	bool trace = getenv("EMA_STREAM_TRACE") != nullptr;      // when what happened, on stderr
	double t_start = now_s();
};

void fill_stats(Item &it)
{
	ema_bucket_stats &s = it.st;
	s.pairs = it.n_pairs;
	s.barcode_groups = it.bk ? it.bk->n_groups : 0;
	if (it.b) {
		const size_t nr = 2 * it.b->n_pairs;
		s.candidates = it.b->cand_off[nr];
		s.redone_pairs = it.b->n_redone;
		uint64_t with = 0; int32_t flags = 0;
		for (size_t r = 0; r < nr; ++r) { with += it.b->cand_off[r + 1] > it.b->cand_off[r]; flags |= it.b->status[r]; }
		s.reads_with_candidates = with;
		s.capacity_flags = flags;
	}
	if (it.a) {
		s.records = it.a->n;
		uint64_t uniq = 0;
		for (size_t i = 0; i < it.a->n; ++i) {
			const ema_aln_rec &r = it.a->rec[i];
			uniq += r.unique;
			const int q = r.mapq;
			++s.mapq_hist[q <= 0 ? 0 : q < 10 ? 1 : q < 20 ? 2 : q < 30 ? 3 : q < 40 ? 4 : q < 60 ? 5 : 6];
		}
		s.unique_records = uniq;
	}
}

// one bucket through engine + append stage on worker `w`'s set of buffers
void align_item(Stream &S, int w, Item &it)
{
	const double t0 = now_s();
	ema_engine_t *g = S.eng[w];
	if (it.n_pairs > S.cap) {      // beyond one batch: the engine's own two-set piece pipeline, which needs both sets
		std::unique_lock<std::mutex> l0(S.eng_mu[0], std::defer_lock), l1(S.eng_mu[1], std::defer_lock);
		std::lock(l0, l1);
		it.rc = ema_engine_align_pairs(S.eng[0], it.bases, it.off, it.n_pairs, &it.b);
		if (it.rc != EMA_OK) it.err = ema_engine_strerror(S.eng[0]);
	} else {
		std::lock_guard<std::mutex> l(S.eng_mu[w]);
		int rc = EMA_OK;
		if (it.slot < 0) {
			rc = ema_engine_stage(g, it.bases, it.off, it.n_pairs);
			if (rc == EMA_OK) rc = ema_engine_run(g);
		} else rc = ema_engine_run_slot(g, it.slot);
		if (rc == EMA_OK) rc = ema_engine_sync(g);
		ema_engine_timing tm;
		if (rc == EMA_OK && ema_engine_last_timing(g, &tm) == EMA_OK) {
			it.st.seed_ms = tm.seed_ms; it.st.extend_ms = tm.extend_ms; it.st.rescue_ms = tm.rescue_ms; it.st.final_ms = tm.final_ms;
			it.st.full_tier_ms = tm.full_tier_ms;
		}
		if (rc == EMA_OK) rc = ema_engine_fetch(g, &it.b);
		it.rc = rc;
		if (rc != EMA_OK) it.err = ema_engine_strerror(g);
	}
	const double t1 = now_s();
	it.st.align_s = t1 - t0;
	if ((it.rc == EMA_OK || it.rc == EMA_ELIMIT) && it.b) {
		const int rc = ema_batch_append_alignments(it.b, it.off, &S.eopts, S.o.error_rate, &it.a);
		if (rc != EMA_OK) { it.rc = rc; it.err = "ema_batch_append_alignments failed"; }
	}
	it.st.append_s = now_s() - t1;
	fill_stats(it);
	it.st.rc = it.rc;
}

// worker w takes buckets w, w + n_eng, ... (alternate buckets on alternate sets of batch buffers)
void worker(Stream &S, int w)
{
	for (size_t k = (size_t)w; k < S.items.size(); k += (size_t)S.n_eng) {
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.stop || S.items[k].state == ST_LOADED; });
			if (S.stop) return;
			S.items[k].state = ST_TAKEN;
		}
		Item &it = S.items[k];
		if (it.rc == EMA_OK) align_item(S, w, it);      // a bucket the reader failed on goes straight through
		{
			std::lock_guard<std::mutex> lk(S.mu);
			it.state = ST_DONE;
		}
		S.cv.notify_all();
	}
}

// a device-resident bucket (ema_bucket_read_device) gets host copies of its reads and qualities as well
int host_resident(ema_bucket *bk)
{
	if (!bk->dev || bk->bases) return EMA_OK;
	const size_t nb = bk->off[2 * bk->n_pairs];
	bk->bases = (char *)malloc(nb + 1); bk->quals = (char *)malloc(nb + 1);
	int rc = bk->bases && bk->quals ? ema_bucket_dev_fetch(bk, bk->bases, bk->quals) : EMA_EIO;
	if (rc != EMA_OK) { free(bk->bases); free(bk->quals); bk->bases = bk->quals = nullptr; }      // (never leave an unfilled copy behind: the next call would take it for the reads)
	return rc;
}

void reader(Stream &S, size_t first, size_t step)      // buckets first, first + step, ...
{
	// How far ahead of the delivery: read_ahead buckets plus the ones inside the engine's pipeline -- or, with small buckets, as many
	// as make up that many full batches (the stream lays them end to end in one pass; a bound in buckets would starve it)
	const size_t depth = (size_t)(S.o.read_ahead > 0 ? S.o.read_ahead : 2) + (S.n_eng == 2 ? 2 : 3);
	const size_t pair_budget = depth * S.cap;
	for (size_t k = first; k < S.items.size(); k += step) {
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.stop || k < S.delivered + depth || (k < S.delivered + 512 && S.pairs_read - S.pairs_delivered < pair_budget); });
			if (S.stop) return;
		}
		Item &it = S.items[k];
		const double t0 = now_s();
		const int rc = S.o.fastq_input
			? ema_fastq_read(S.paths[k], S.o.paths2 ? S.o.paths2[k] : nullptr, S.o.fastq_name_style, S.o.bc_len, S.o.is_haplotag, S.o.max_read_len, &it.bk)
			: S.dev_reader ? ema_bucket_read_device(S.paths[k], S.o.bc_len, S.o.is_haplotag, S.o.max_read_len, S.device, &it.bk)
			: ema_bucket_read(S.paths[k], S.o.bc_len, S.o.is_haplotag, S.o.max_read_len, &it.bk);
		it.st.read_s = now_s() - t0;
		if (rc != EMA_OK) { it.rc = rc; it.err = *ema_bucket_last_error() ? ema_bucket_last_error() : ema_bucket_dev_last_error(); it.st.rc = rc; }
		else {
			int hrc = EMA_OK;
			if (it.bk->dev && it.bk->n_pairs > S.cap) hrc = host_resident(it.bk);      // beyond one batch: ema_engine_align_pairs takes host arrays
			if (hrc != EMA_OK) { it.rc = hrc; it.err = "a bucket's reads could not be copied back from the device"; it.st.rc = hrc; }
			else { it.bases = it.bk->bases; it.off = it.bk->off; it.n_pairs = it.bk->n_pairs; }
		}
		{
			std::lock_guard<std::mutex> lk(S.mu);
			it.state = ST_LOADED;
			S.pairs_read += it.n_pairs;
		}
		S.cv.notify_all();
	}
}

// The default schedule: ONE set of batch buffers, asynchronous passes (ema_engine_run_async, up to EMA_MAX_INFLIGHT in flight).
// A stager thread converts and uploads the next passes' input into free input slots while the engine thread queues a pass,
// fetches the oldest one and runs its buckets' append stage while the younger ones compute.
// One pass = one or SEVERAL consecutive buckets: the engine is built for batches of a million pairs (every kernel ends in a tail
// of long reads, and the full-capacity tier costs its ~0.1 s whatever the batch size), so small buckets that are waiting
// anyway are laid end to end in one batch and the results cut apart again.  A bucket beyond the batch capacity drains the
// pipeline and goes through ema_engine_align_pairs.
const int kInSlots = 4;      // input slots 1..4: slot 0 stays with ema_engine_stage / ema_engine_align_pairs (the big-bucket path)
const size_t kMaxGroup = 64;

struct Pass { size_t first = 0, len = 0; int staged = 0; };      // items [first, first + len); staged: 1 yes, -1 failed
struct AsyncState {
	std::vector<Pass> passes;     // in order, made by the stager
	bool all_made = false;
	size_t n_run = 0;             // passes queued so far (the stager may fill slot p % kInSlots once pass p - kInSlots has been queued)
};

// pairs [p0, p0 + n) of a fetched pass as a batch: a VIEW into the pass's arrays (engine.hip: no candidate or CIGAR is copied;
// round 2 copied them, 0.4 KB a pair through freshly faulted pages); the pass is freed with its last view
extern "C" ema_batch_out *ema_batch_view(void **share, ema_batch_out *whole, size_t p0, size_t n);
extern "C" void ema_batch_share_release(void *share);

void stager(Stream &S, AsyncState &A)
{
	const size_t n = S.items.size();
	size_t k = 0;
	std::vector<char> bases;
	std::vector<uint32_t> off;
	while (k < n) {
		Pass ps;
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.stop || (S.items[k].state == ST_LOADED && A.passes.size() < A.n_run + (size_t)kInSlots); });
			if (S.stop) return;
			// the buckets already waiting behind this one come along while they fit the batch; a pass that would be less than half
			// a batch waits a moment (at most 60 ms) for the reader, if it is still at work on the next bucket
			const double t_wait0 = now_s();
			for (;;) {
				ps.first = k; ps.len = 1;
				const Item &h = S.items[k];
				size_t sum = h.n_pairs;
				bool more_coming = false;
				if (h.rc == EMA_OK && h.slot < 0 && h.n_pairs <= S.cap) {
					uint64_t bytes = h.off ? h.off[2 * h.n_pairs] : 0;
					while (k + ps.len < n && ps.len < kMaxGroup) {
						const Item &x = S.items[k + ps.len];
						if (x.state == ST_EMPTY && S.paths) { more_coming = true; break; }
						if (x.state != ST_LOADED || x.rc != EMA_OK || x.slot >= 0 || sum + x.n_pairs > S.cap) break;
						bytes += x.off ? x.off[2 * x.n_pairs] : 0;
						if (bytes > 0xfffffff0ULL) break;
						sum += x.n_pairs; ++ps.len;
					}
				}
				if (!more_coming || 2 * sum >= S.cap || now_s() - t_wait0 > 0.06 || S.stop) break;
				S.cv.wait_for(lk, std::chrono::milliseconds(4));
			}
			if (S.stop) return;
		}
		Item &it = S.items[k];
		int ok = 1;
		// buckets whose reads are on the device already go into the slot there (no host copy, no upload); a pass that mixes them with
		// host-resident ones (a bucket the device reader handed to the host reader) takes the host path, with copies fetched back
		size_t n_dev = 0, n_used = 0;
		for (size_t j = 0; j < ps.len; ++j) { const Item &x = S.items[k + j]; if (x.rc == EMA_OK && x.n_pairs) { ++n_used; n_dev += x.bk && x.bk->dev && !x.bk->bases; } }
		if (it.rc == EMA_OK && it.slot < 0 && it.n_pairs <= S.cap && n_dev && n_dev < n_used)
			for (size_t j = 0; j < ps.len; ++j) {
				Item &x = S.items[k + j];
				if (!(x.bk && x.bk->dev)) continue;
				const int hr = host_resident(x.bk);
				if (hr == EMA_OK) x.bases = x.bk->bases;
				else {      // the whole pass fails, as it does when the reader's own fetch fails (no copy to lay end to end)
					for (size_t j2 = 0; j2 < ps.len; ++j2) { S.items[k + j2].rc = hr; S.items[k + j2].err = "cannot fetch a device-resident bucket's reads back to the host"; }
					ok = -1;
					break;
				}
			}
		if (it.rc == EMA_OK && it.slot < 0 && it.n_pairs <= S.cap && n_dev && n_dev == n_used) {
			std::vector<const ema_bucket *> bks;
			size_t np = 0;
			bool too_long = false;
			for (size_t j = 0; j < ps.len; ++j) {
				const Item &x = S.items[k + j];
				if (!x.n_pairs) continue;
				bks.push_back(x.bk); np += x.n_pairs;
				for (size_t r = 0; r < 2 * x.n_pairs; ++r) too_long |= x.off[r + 1] - x.off[r] > (uint32_t)ema_engine_max_read_len();
			}
			std::lock_guard<std::mutex> hold(S.stage_mu);
			const int rc = too_long ? EMA_ELIMIT : ema_engine_stage_async_dev(S.eng[0], 1 + (int)(A.passes.size() % (size_t)kInSlots), bks.data(), bks.size());
			if (S.trace) fprintf(stderr, "[stream] stager: pass %zu = buckets %zu..%zu (%zu pairs): staged on the device by %.3f (s since start)\n",
			                     A.passes.size(), k, k + ps.len - 1, np, now_s() - S.t_start);
			if (rc != EMA_OK) { for (size_t j = 0; j < ps.len; ++j) { S.items[k + j].rc = rc; S.items[k + j].err = too_long ? "read longer than EMA_MAX_READ" : ema_engine_strerror(S.eng[0]); } ok = -1; }
		} else if (it.rc == EMA_OK && it.slot < 0 && it.n_pairs <= S.cap) {
			const char *pb = it.bases; const uint32_t *po = it.off; size_t np = it.n_pairs;
			if (ps.len > 1) {      // end to end in one batch
				size_t tot_p = 0, tot_b = 0;
				for (size_t j = 0; j < ps.len; ++j) { const Item &x = S.items[k + j]; tot_p += x.n_pairs; tot_b += x.n_pairs ? x.off[2 * x.n_pairs] : 0; }
				bases.resize(tot_b + 1); off.resize(2 * tot_p + 1);
				size_t at_r = 0; uint32_t at_b = 0;
				off[0] = 0;
				for (size_t j = 0; j < ps.len; ++j) {
					const Item &x = S.items[k + j];
					if (!x.n_pairs) continue;
					const uint32_t nb = x.off[2 * x.n_pairs];
					memcpy(bases.data() + at_b, x.bases, nb);
					for (size_t r = 1; r <= 2 * x.n_pairs; ++r) off[at_r + r] = at_b + x.off[r];
					at_r += 2 * x.n_pairs; at_b += nb;
				}
				pb = bases.data(); po = off.data(); np = tot_p;
			}
			const double t_cat = now_s();
			std::lock_guard<std::mutex> hold(S.stage_mu);      // the engine's host-side staging buffers: one user at a time
			const int rc = ema_engine_stage_async(S.eng[0], 1 + (int)(A.passes.size() % (size_t)kInSlots), pb, po, np);
			if (S.trace) fprintf(stderr, "[stream] stager: pass %zu = buckets %zu..%zu (%zu pairs): laid end to end by %.3f, staged by %.3f (s since start)\n",
			                     A.passes.size(), k, k + ps.len - 1, np, t_cat - S.t_start, now_s() - S.t_start);
			if (rc != EMA_OK) { for (size_t j = 0; j < ps.len; ++j) { S.items[k + j].rc = rc; S.items[k + j].err = ema_engine_strerror(S.eng[0]); } ok = -1; }
		}
		ps.staged = ok;
		k += ps.len;
		{
			std::lock_guard<std::mutex> lk(S.mu);
			A.passes.push_back(ps);
			if (k >= n) A.all_made = true;
		}
		S.cv.notify_all();
	}
	{
		std::lock_guard<std::mutex> lk(S.mu);
		A.all_made = true;
	}
	S.cv.notify_all();
}

void finish_item(Stream &S, Item &it, double t0)
{
	const double t1 = now_s();
	it.st.align_s = t1 - t0;
	if ((it.rc == EMA_OK || it.rc == EMA_ELIMIT) && it.b) {
		const int rc = ema_batch_append_alignments(it.b, it.off, &S.eopts, S.o.error_rate, &it.a);
		if (rc != EMA_OK) { it.rc = rc; it.err = "ema_batch_append_alignments failed"; }
	}
	it.st.append_s = now_s() - t1;
	fill_stats(it);
	it.st.rc = it.rc;
	{
		std::lock_guard<std::mutex> lk(S.mu);
		it.state = ST_DONE;
	}
	S.cv.notify_all();
}

void async_engine_thread(Stream &S, AsyncState &A)
{
	ema_engine_t *g = S.eng[0];
	std::vector<int> ticket;
	std::vector<double> t_queued;
	size_t p_run = 0, p_fetch = 0;
	auto pass_at = [&](size_t p) { std::lock_guard<std::mutex> lk(S.mu); return A.passes[p]; };
	auto fetch_one = [&] {
		const Pass ps = pass_at(p_fetch);
		Item &head = S.items[ps.first];
		ema_batch_out *b = nullptr;
		int rc = EMA_OK; std::string err;
		ema_engine_timing tm; bool have_tm = false;
		const double t_f0 = now_s();
		if (ticket[p_fetch] >= 0) {
			rc = ema_engine_fetch_ticket(g, ticket[p_fetch], &b);
			if (rc != EMA_OK) err = ema_engine_strerror(g);
			have_tm = ema_engine_last_timing(g, &tm) == EMA_OK;
		}
		const double t_f1 = now_s();
		// the buckets of a shared pass are cut out of the fetched batch and run through the append stage side by side, one thread
		// each (round 2 did them one after another on this thread: 0.35 us a pair, most of it first-touch page faults of the
		// buckets' own arrays -- the engine thread spent more time here than waiting for the GPU, r03e)
		std::vector<size_t> first_pair(ps.len + 1, 0);
		for (size_t j = 0; j < ps.len; ++j) first_pair[j + 1] = first_pair[j] + S.items[ps.first + j].n_pairs;
		const bool fetched = ticket[p_fetch] >= 0;
		const double t_q = t_queued[p_fetch];
		std::vector<ema_batch_out *> views(ps.len, nullptr);
		void *share = nullptr;
		const bool shared_pass = fetched && b && ps.len > 1;
		if (shared_pass) {
			for (size_t j = 0; j < ps.len; ++j) views[j] = ema_batch_view(&share, b, first_pair[j], S.items[ps.first + j].n_pairs);
			if (share) b = nullptr;      // the views own the pass now (this thread's reference is released below)
		}
		auto one = [&](size_t j) {
			Item &it = S.items[ps.first + j];
			if (fetched) {
				it.rc = rc; it.err = err;
				if (have_tm) {
					it.st.seed_ms = tm.seed_ms; it.st.extend_ms = tm.extend_ms; it.st.rescue_ms = tm.rescue_ms; it.st.final_ms = tm.final_ms;
					it.st.full_tier_ms = tm.full_tier_ms;
				}
				if (shared_pass) {
					it.b = views[j];
					if (!it.b) { it.rc = EMA_EDEVICE; it.err = "out of host memory cutting a batch into its buckets"; }
					else if (rc == EMA_ELIMIT) {      // the capacity flag is the batch's: does this bucket hold a flagged read?
						int32_t flags = 0;
						for (size_t r = 0; r < 2 * it.b->n_pairs; ++r) flags |= it.b->status[r];
						if (!flags) { it.rc = EMA_OK; it.err.clear(); }
					}
				}
			}
			finish_item(S, it, t_q);
		};
		if (fetched && b && ps.len == 1) { S.items[ps.first].b = b; b = nullptr; }

		if (shared_pass) {
			EmaPool::get().run(ps.len, one);
		} else for (size_t j = 0; j < ps.len; ++j) one(j);
		if (share) ema_batch_share_release(share);
		if (b) ema_batch_free(b);
		(void)head;
		if (S.trace) fprintf(stderr, "[stream] engine thread: pass %zu queued at %.3f, fetch %.3f..%.3f, cut + append stage done by %.3f\n", p_fetch,
		                     t_queued[p_fetch] - S.t_start, t_f0 - S.t_start, t_f1 - S.t_start, now_s() - S.t_start);
		++p_fetch;
	};
	for (;;) {
		bool can_run = false, done = false;
		{
			std::unique_lock<std::mutex> lk(S.mu);
			if (S.stop) break;
			// queue the next pass if it is staged and fewer than EMA_MAX_INFLIGHT are in flight; otherwise fetch the oldest; otherwise wait
			S.cv.wait(lk, [&] { return S.stop || p_fetch < p_run || p_run < A.passes.size() || (A.all_made && p_fetch >= A.passes.size()); });
			if (S.stop) break;
			done = A.all_made && p_fetch >= A.passes.size() && p_run >= A.passes.size();
			can_run = p_run < A.passes.size() && p_run - p_fetch < (size_t)EMA_MAX_INFLIGHT;
		}
		if (done) break;
		if (can_run) {
			const Pass ps = pass_at(p_run);
			Item &it = S.items[ps.first];
			ticket.push_back(-1); t_queued.push_back(now_s());
			if (ps.staged < 0 || it.rc != EMA_OK) {      // reader or stager failed: passes straight through, in order
			} else if (it.n_pairs > S.cap) {      // beyond one batch: drain, then the engine's own piece pipeline
				while (p_fetch < p_run) fetch_one();
				std::lock_guard<std::mutex> hold(S.stage_mu);
				it.rc = ema_engine_align_pairs(g, it.bases, it.off, it.n_pairs, &it.b);
				if (it.rc != EMA_OK) it.err = ema_engine_strerror(g);
			} else {
				const int slot = it.slot >= 0 ? it.slot : 1 + (int)(p_run % (size_t)kInSlots);
				int tk = -1;
				const int rc = ema_engine_run_async(g, slot, &tk);
				if (rc != EMA_OK) { for (size_t j = 0; j < ps.len; ++j) { S.items[ps.first + j].rc = rc; S.items[ps.first + j].err = ema_engine_strerror(g); } }
				else ticket[p_run] = tk;
			}
			{
				std::lock_guard<std::mutex> lk(S.mu);
				for (size_t j = 0; j < ps.len; ++j) S.items[ps.first + j].state = ST_TAKEN;
				++p_run; A.n_run = p_run;
			}
			S.cv.notify_all();
		} else if (p_fetch < p_run) fetch_one();
	}
	// on an early stop, passes still in flight are fetched and dropped so that the engine is reusable
	while (p_fetch < p_run) {
		if (ticket[p_fetch] >= 0) { ema_batch_out *b = nullptr; (void)ema_engine_fetch_ticket(g, ticket[p_fetch], &b); if (b) ema_batch_free(b); }
		++p_fetch;
	}
}

void release(Item &it)
{
	if (it.a) ema_aln_free(it.a);
	if (it.b) ema_batch_free(it.b);
	if (it.bk) ema_bucket_free(it.bk);
	it.a = nullptr; it.b = nullptr; it.bk = nullptr;
}

int run_stream(ema_engine_t *e, Stream &S, ema_stream_sink sink, void *user, ema_bucket_stats *stats)
{
	S.eng[0] = e;
	const bool two_sets = S.o.n_engines == 2;      // the older schedule: alternate batches on two sets of batch buffers, one pass each
	S.n_eng = 1;
	if (two_sets) {
		S.eng[1] = ema_engine_peer(e);
		if (S.eng[1]) S.n_eng = 2;      // else no room for a second set of batch buffers
	}
	S.cap = ema_engine_batch_capacity(e);
	ema_engine_get_opts(e, &S.eopts);
	for (size_t k = 0; k < S.items.size(); ++k)      // resident batches: which slot of which set (see ema_stream_resident)
		if (S.items[k].slot >= 0) S.items[k].slot = (int)((k / (size_t)S.n_eng) % (size_t)S.items[k].slot);
	std::vector<std::thread> th;
	AsyncState A;
	// (two device readers taking turns were measured, r05p: 5.9-6.1 M pairs/s files -> SAM against 6.3 M with one -- the stream is bound
	// by the GPU's passes, and two readers only contend for the copy engines)
	// [r6] ... with buckets of 262 K pairs.  With BASELINE configs[2]'s 50-100 K-pair buckets a bucket's fixed costs -- four waits for small
	// kernels on a busy device -- are most of its 10 ms and ONE reader delivers 5.3 M pairs/s, the stream's bound: two readers there
	// (500 buckets of 52 K pairs, files -> SAM: 5.0-5.4 -> 6.4-6.5 M pairs/s; three: the same) -- PROVIDED their kernels go to the device's
	// highest-priority queue (ingest_dev.hip): on ordinary queues two readers contend and gain nothing (profiles/r06_sam_leg_ab.txt).
	// By the first file's size; tuning knob stream_readers overrides.
	size_t n_readers = 1;
	if (S.paths && !S.items.empty() && S.dev_reader) {
		struct stat sb;
		if (stat(S.paths[0], &sb) == 0 && sb.st_size < (off_t)48 << 20) n_readers = 2;
	}
	if (const char *v = ema_tuning_get("stream_readers")) n_readers = (size_t)std::max(1, std::min(4, atoi(v)));
	if (S.trace) fprintf(stderr, "[stream] %zu reader thread(s), device reader %d\n", S.paths ? n_readers : (size_t)0, (int)S.dev_reader);
	if (S.paths) for (size_t t = 0; t < n_readers; ++t) th.emplace_back(reader, std::ref(S), t, n_readers);
	if (S.n_eng == 2) {
		for (int w = 0; w < S.n_eng; ++w) th.emplace_back(worker, std::ref(S), w);
	} else {
		th.emplace_back(stager, std::ref(S), std::ref(A));
		th.emplace_back(async_engine_thread, std::ref(S), std::ref(A));
	}
	int result = EMA_OK;
	for (size_t k = 0; k < S.items.size(); ++k) {
		Item &it = S.items[k];
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return it.state == ST_DONE; });
		}
		if (stats) stats[k] = it.st;
		int rc = it.rc;
		const double t_s0 = now_s();
		if (rc == EMA_OK || rc == EMA_ELIMIT) {
			if (rc == EMA_ELIMIT) { result = EMA_ELIMIT; g_err = it.err; }
			const int src = sink ? sink(user, k, it.bk, it.b, it.a) : 0;
			if (src != 0) { rc = src; g_err = "stopped by the sink"; }
			else rc = EMA_OK;
		} else g_err = (S.paths ? std::string(S.paths[k]) + ": " : std::string()) + it.err;
		if (S.trace) fprintf(stderr, "[stream] sink: bucket %zu %.3f..%.3f\n", k, t_s0 - S.t_start, now_s() - S.t_start);
		if (S.sink_kept) { it.bk = nullptr; it.b = nullptr; it.a = nullptr; S.sink_kept = false; }      // (ema_stream_sam's sink: its writer frees them)
		release(it);
		{
			std::lock_guard<std::mutex> lk(S.mu);
			++S.delivered; S.pairs_delivered += it.n_pairs;
			if (rc != EMA_OK) S.stop = true;
		}
		S.cv.notify_all();
		if (rc != EMA_OK) { result = rc; break; }
	}
	{
		std::lock_guard<std::mutex> lk(S.mu);
		S.stop = true;
	}
	S.cv.notify_all();
	for (auto &t : th) t.join();
	for (auto &it : S.items) release(it);
	return result;
}

}  // namespace

extern "C" {

void ema_stream_opts_default(ema_stream_opts *o)
{
	if (!o) return;
	o->bc_len = 16; o->is_haplotag = 0; o->max_read_len = 255; o->error_rate = 0.001; o->n_engines = 0; o->read_ahead = 0;
	o->fastq_input = 0; o->fastq_name_style = 0; o->paths2 = nullptr;
}

const char *ema_stream_last_error(void) { return g_err.c_str(); }

void ema_host_cpu_seconds(double out[6], int reset)
{
	for (int i = 0; i < 6; ++i) {
		const uint64_t ns = reset ? ema_cpu_ns[i].exchange(0) : ema_cpu_ns[i].load();
		if (out) out[i] = (double)ns * 1e-9;
	}
}

int ema_stream_buckets(ema_engine_t *e, const char *const *paths, size_t n, const ema_stream_opts *o, ema_stream_sink sink,
                       void *user, ema_bucket_stats *stats)
{
	g_err.clear();
	if (!e || (!paths && n)) { g_err = "bad argument"; return EMA_EARG; }
	Stream S;
	if (o) S.o = *o; else ema_stream_opts_default(&S.o);
	S.paths = paths;
	S.items.resize(n);
	for (auto &it : S.items) memset(&it.st, 0, sizeof(it.st));
	return run_stream(e, S, sink, user, stats);
}

int ema_stream_batches(ema_engine_t *e, const char *const *bases, const uint32_t *const *off, const size_t *n_pairs, size_t n,
                       const ema_stream_opts *o, ema_stream_sink sink, void *user, ema_bucket_stats *stats)
{
	g_err.clear();
	if (!e || ((!bases || !off || !n_pairs) && n)) { g_err = "bad argument"; return EMA_EARG; }
	Stream S;
	if (o) S.o = *o; else ema_stream_opts_default(&S.o);
	S.items.resize(n);
	for (size_t k = 0; k < n; ++k) {
		Item &it = S.items[k];
		memset(&it.st, 0, sizeof(it.st));
		it.bases = bases[k]; it.off = off[k]; it.n_pairs = n_pairs[k];
		it.state = ST_LOADED;
	}
	return run_stream(e, S, sink, user, stats);
}

namespace {
// ema_stream_sam's sink hands every bucket to a three-stage host pipeline of its own, so that the engine thread goes straight back to
// fetching the next pass: clouds / EM / duplicates on cloud threads (barcode groups on the host's threads inside), then the formatter +
// write on a writer thread, in bucket order.  The lines point into the bucket, the batch and the selection, so the sink takes those over
// from the stream and the writer frees them.  (Round 2 ran the cloud stage inside the sink, on the engine thread: 78 of the ~130 ms a
// 262 K-pair bucket spent there.)
// [r6] How many cloud threads: ONE when the cloud counter runs on from bucket to bucket (`-x`: bucket k + 1 starts where bucket k
// ended) or when -d draws from the process's one rand() stream; otherwise -- every bucket its own `ema align -s` process, cloud
// numbers from first_cloud_id, -d draws (if any) from the bucket's own stream, seed + bucket index -- up to three buckets at a time:
// the part of -d that must run on one thread per bucket (the groups with a bad cloud, in order) then overlaps with other buckets'.
struct CloudJob { size_t k, seq; ema_bucket *bk; ema_batch_out *b; ema_aln_out *a; };
struct WriteJob { size_t k; ema_clouds_out *sel; ema_bucket *bk; ema_batch_out *b; ema_aln_out *a; bool skip; };
struct SamSink {
	ema_engine_t *e;
	ema_sam_run_opts o;
	int fd;
	ema_sam_stats *sstats;
	std::vector<const char *> names;
	int32_t next_cloud_id;
	std::string err;
	Stream *stream = nullptr;
	ema_sam_dev_t *dev = nullptr;      // the formatter on the device (k_sam.hip); NULL: ema_sam_write on the host's threads
	std::mutex mu;
	std::condition_variable cv;
	std::deque<CloudJob> cloud_jobs;           // waiting for a cloud thread, in bucket order
	std::map<size_t, WriteJob> jobs;           // through the cloud stage, by sequence number: the writer takes next_write
	size_t n_seq = 0, next_write = 0, in_clouds = 0;
	int n_clouders = 1, clouders_left = 0;
	bool closing = false;
	int write_rc = EMA_OK, cloud_rc = EMA_OK;
};

void free_job(ema_clouds_out *sel, ema_bucket *bk, ema_batch_out *b, ema_aln_out *a)
{
	if (sel) ema_clouds_free(sel);
	if (a) ema_aln_free(a);
	if (b) ema_batch_free(b);
	if (bk) ema_bucket_free(bk);
}

void sam_writer(SamSink &S)
{
	for (;;) {
		WriteJob j;
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.jobs.count(S.next_write) || (S.clouders_left == 0 && S.jobs.empty()); });
			if (!S.jobs.count(S.next_write)) return;
			j = S.jobs[S.next_write];
		}
		size_t n_bytes = 0;
		const double t0 = now_s();
		int rc = EMA_OK;
		if (S.write_rc == EMA_OK && !j.skip) {
			if (S.dev) {
				const ema_clouds_out *sl = j.sel;
				rc = ema_sam_dev_write(S.dev, S.fd, j.bk, j.b->cigar ? j.b->cigar + sl->cigar_lo : nullptr, sl->cigar_lo, sl->cigar_hi, sl->descs, sl->n_descs,
				                       sl->xas, sl->n_xas, sl->sel_at, sl->n_sel, &S.o.sam, &n_bytes);
				if (rc == EMA_EIO && *ema_sam_dev_last_error()) { std::lock_guard<std::mutex> lk(S.mu); if (S.err.empty()) S.err = std::string("SAM formatter on the device: ") + ema_sam_dev_last_error(); }
			} else rc = ema_sam_write(S.fd, j.sel->lines, j.sel->n_lines, &S.o.sam, &n_bytes);
		}
		if (S.sstats && !j.skip) S.sstats[j.k].write_s = now_s() - t0;
		free_job(j.sel, j.bk, j.b, j.a);
		{
			std::lock_guard<std::mutex> lk(S.mu);
			if (rc != EMA_OK && S.write_rc == EMA_OK) S.write_rc = rc;
			S.jobs.erase(S.next_write);
			++S.next_write;
		}
		S.cv.notify_all();
	}
}

void sam_clouds(SamSink &S)
{
	for (;;) {
		CloudJob c;
		bool failed;
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.closing || !S.cloud_jobs.empty(); });
			if (S.cloud_jobs.empty()) { --S.clouders_left; lk.unlock(); S.cv.notify_all(); return; }
			c = S.cloud_jobs.front();
			S.cloud_jobs.pop_front();
			++S.in_clouds;
			failed = S.cloud_rc != EMA_OK || S.write_rc != EMA_OK;      // read under the lock the writer stores write_rc under
		}
		S.cv.notify_all();
		ema_clouds_out *sel = nullptr;
		int rc = failed ? EMA_ESTATE : EMA_OK;      // after a failure: drain, freeing what arrives
		if (rc == EMA_OK) {
			ema_cloud_opts co = S.o.clouds;
			if (S.o.continue_cloud_ids) co.first_cloud_id = S.next_cloud_id;      // (one cloud thread then: nobody else touches it)
			else if (co.density_opt && co.seed_private) co.seed += (uint32_t)c.k;  // the bucket's own stream
			co.emit = S.dev ? 1 : 0;      // the compact records for the device's formatter, or the lines for the host's
			rc = ema_clouds_select(c.bk, c.b, c.a, S.names.data(), (int32_t)S.names.size(), &co, &sel);
			if (rc == EMA_OK) {
				if (S.o.continue_cloud_ids) S.next_cloud_id = sel->next_cloud_id;
				if (S.sstats) S.sstats[c.k] = sel->stats;
			}
		}
		std::unique_lock<std::mutex> lk(S.mu);
		if (rc != EMA_OK && !failed && S.cloud_rc == EMA_OK && S.write_rc == EMA_OK) { S.cloud_rc = rc; S.err = "ema_clouds_select failed"; }
		// at most two buckets' worth of objects wait for the writer beside the one it needs next
		S.cv.wait(lk, [&] { return c.seq == S.next_write || S.jobs.size() < 2; });
		S.jobs[c.seq] = WriteJob{c.k, sel, c.bk, c.b, c.a, rc != EMA_OK};
		--S.in_clouds;
		lk.unlock();
		S.cv.notify_all();
	}
}

int sam_sink(void *user, size_t k, const ema_bucket *bk, const ema_batch_out *b, const ema_aln_out *a)
{
	SamSink &S = *(SamSink *)user;
	std::unique_lock<std::mutex> lk(S.mu);
	S.cv.wait(lk, [&] { return S.cloud_jobs.size() < 2; });      // at most two buckets wait for a cloud thread
	if (S.cloud_rc != EMA_OK) return S.cloud_rc;
	if (S.write_rc != EMA_OK) { S.err = "ema_sam_write failed"; return S.write_rc; }
	S.cloud_jobs.push_back(CloudJob{k, S.n_seq++, const_cast<ema_bucket *>(bk), const_cast<ema_batch_out *>(b), const_cast<ema_aln_out *>(a)});
	lk.unlock();
	S.cv.notify_all();
	S.stream->sink_kept = true;      // the pipeline frees them
	return 0;
}
}  // namespace

void ema_sam_run_opts_default(ema_sam_run_opts *o)
{
	if (!o) return;
	ema_stream_opts_default(&o->stream);
	ema_cloud_opts_default(&o->clouds);
	ema_sam_opts_default(&o->sam);
	o->continue_cloud_ids = 0;
}

int ema_sam_run_opts_platform(const char *name, ema_sam_run_opts *o)
{
	if (!name || !o) return EMA_EARG;
	// reference src/techs.c:74-135: {name, bc_len, many_clouds, dist_thresh, error_rate}
	static const struct { const char *name; int bc_len, haplotag, many_clouds; uint32_t dist_thresh; double error_rate; } tab[] = {
		{"haplotag", 12, 1, 0, 50000, 0.001}, {"10x", 16, 0, 0, 50000, 0.001}, {"tru", 0, 0, 1, 15000, 0.001},
		{"cpt", 0, 0, 1, 3500, 0.01}, {"dbs", 20, 0, 0, 50000, 0.001}, {"tellseq", 18, 0, 0, 50000, 0.001}};
	for (const auto &t : tab) if (strcmp(name, t.name) == 0) {
		ema_sam_run_opts_default(o);
		o->stream.bc_len = t.bc_len; o->stream.is_haplotag = t.haplotag; o->stream.error_rate = t.error_rate;
		o->clouds.dist_thresh = t.dist_thresh; o->clouds.many_clouds = t.many_clouds;
		if (strcmp(name, "cpt") == 0) {      // its own density model for -d (src/techs.c:107-109); every other platform has the default one
			static const double cpt[9] = {0.6, 0.01, 0.15, 0.001, 0.05, 0.001, 0.02, 0.001, 0.01};
			o->clouds.n_density_probs = 9;
			for (int i = 0; i < 9; ++i) o->clouds.density_probs[i] = cpt[i];
		}
		o->sam.bc_len = t.bc_len; o->sam.is_haplotag = t.haplotag;
		// how `-1 / -2` input names its barcodes (ema_fastq_read): after the last ':' | a " BX:Z:" comment | the name's leading number | ":xxNNN"
		o->stream.fastq_name_style = strcmp(name, "tellseq") == 0 ? 1 : strcmp(name, "tru") == 0 ? 2 : strcmp(name, "cpt") == 0 ? 3 : 0;
		return EMA_OK;
	}
	g_err = std::string("unknown platform ") + name;
	return EMA_EARG;
}

int ema_stream_sam(ema_engine_t *e, const char *const *paths, size_t n, const ema_sam_run_opts *o, int fd, ema_bucket_stats *bstats,
                   ema_sam_stats *sstats)
{
	g_err.clear();
	if (!e || (!paths && n)) { g_err = "bad argument"; return EMA_EARG; }
	SamSink S;
	S.e = e; S.fd = fd; S.sstats = sstats;
	if (o) S.o = *o; else ema_sam_run_opts_default(&S.o);
	S.next_cloud_id = S.o.clouds.first_cloud_id;
	S.o.sam.bc_len = S.o.stream.bc_len; S.o.sam.is_haplotag = S.o.stream.is_haplotag;      // one platform for reader and writer
	const int nc = ema_engine_n_contigs(e);
	for (int i = 0; i < nc; ++i) S.names.push_back(ema_engine_contig_name(e, i));
	Stream T;
	T.o = S.o.stream;
	T.paths = paths;
	T.items.resize(n);
	for (auto &it : T.items) memset(&it.st, 0, sizeof(it.st));
	S.stream = &T;
	{   // SAM text from the device's kernels unless tuned off ("sam_device_format=0": the host formatter, ema_sam_write)
		const char *v = ema_tuning_get("sam_device_format");
		if (!(v && atoi(v) == 0)) {
			const int drc = ema_sam_dev_open(ema_engine_device(e), S.names.data(), (int32_t)S.names.size(), &S.dev);
			if (drc != EMA_OK) { g_err = std::string("SAM formatter on the device: ") + ema_sam_dev_last_error(); return drc; }
			// ... and then the reader's parse, sort and gather are kernels too, the reads staying where the formatter finds them
			// ("sam_device_reader=0": the host reader; -1 / -2 FASTQ input and two-engine streams use the host reader as well)
			v = ema_tuning_get("sam_device_reader");
			T.dev_reader = !(v && atoi(v) == 0) && !T.o.fastq_input && T.o.n_engines != 2;
			T.device = ema_engine_device(e);
		}
	}
	{   // cloud threads (see SamSink): buckets are independent of each other unless the cloud counter or the rand() stream runs through them
		const bool chained = S.o.continue_cloud_ids != 0 || (S.o.clouds.density_opt && !S.o.clouds.seed_private);
		const char *v = ema_tuning_get("sam_cloud_threads");
		S.n_clouders = chained ? 1 : v ? std::max(1, std::min(8, atoi(v))) : 3;
		S.clouders_left = S.n_clouders;
	}
	std::thread writer(sam_writer, std::ref(S));
	std::vector<std::thread> clouders;
	for (int t = 0; t < S.n_clouders; ++t) clouders.emplace_back(sam_clouds, std::ref(S));
	int rc = run_stream(e, T, sam_sink, &S, bstats);
	{
		std::lock_guard<std::mutex> lk(S.mu);
		S.closing = true;
	}
	S.cv.notify_all();
	for (auto &t : clouders) t.join();
	writer.join();
	if (S.dev) ema_sam_dev_close(S.dev);
	if (rc == EMA_OK && S.cloud_rc != EMA_OK) rc = S.cloud_rc;
	if (rc == EMA_OK && S.write_rc != EMA_OK) { rc = S.write_rc; S.err = "ema_sam_write failed"; }
	if (rc != EMA_OK && !S.err.empty()) g_err = S.err;
	return rc;
}

int ema_stream_resident(ema_engine_t *e, const uint32_t *const *off, const size_t *n_pairs, size_t n, int slots_per_set,
                        const ema_stream_opts *o, ema_stream_sink sink, void *user, ema_bucket_stats *stats)
{
	g_err.clear();
	if (!e || ((!off || !n_pairs) && n) || slots_per_set < 1 || slots_per_set > EMA_MAX_SLOTS) { g_err = "bad argument"; return EMA_EARG; }
	Stream S;
	if (o) S.o = *o; else ema_stream_opts_default(&S.o);
	S.items.resize(n);
	for (size_t k = 0; k < n; ++k) {
		Item &it = S.items[k];
		memset(&it.st, 0, sizeof(it.st));
		it.off = off[k]; it.n_pairs = n_pairs[k];
		it.slot = slots_per_set;      // turned into the slot index once the number of sets is known (run_stream)
		it.state = ST_LOADED;
	}
	return run_stream(e, S, sink, user, stats);
}

}  // extern "C"
