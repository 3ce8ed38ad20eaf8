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
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "ema_stream.h"

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
	size_t delivered = 0;
	bool stop = false;
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

void reader(Stream &S)
{
	const size_t depth = (size_t)(S.o.read_ahead > 0 ? S.o.read_ahead : 2) + (S.n_eng == 2 ? 2 : 3);      // + the buckets inside the engine's pipeline
	for (size_t k = 0; k < S.items.size(); ++k) {
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.stop || k < S.delivered + depth; });
			if (S.stop) return;
		}
		Item &it = S.items[k];
		const double t0 = now_s();
		const int rc = ema_bucket_read(S.paths[k], S.o.bc_len, S.o.is_haplotag, S.o.max_read_len, &it.bk);
		it.st.read_s = now_s() - t0;
		if (rc != EMA_OK) { it.rc = rc; it.err = ema_bucket_last_error(); it.st.rc = rc; }
		else { it.bases = it.bk->bases; it.off = it.bk->off; it.n_pairs = it.bk->n_pairs; }
		{
			std::lock_guard<std::mutex> lk(S.mu);
			it.state = ST_LOADED;
		}
		S.cv.notify_all();
	}
}

// The default schedule: ONE set of batch buffers, passes queued two deep (ema_engine_run_async).  A stager thread converts and
// uploads batch k+1.. into free input slots while this thread queues pass k, then fetches pass k-1 and runs its append stage
// while pass k computes.  A bucket beyond the batch capacity drains the pipeline and goes through ema_engine_align_pairs.
const int kInSlots = 4;      // input slots 1..4: slot 0 stays with ema_engine_stage / ema_engine_align_pairs (the big-bucket path)

struct AsyncState {
	std::vector<int> staged;      // per item: 0 no, 1 staged, -1 staging failed
	size_t n_run = 0;             // passes queued so far (the stager may fill slot k % kInSlots once pass k - kInSlots has been queued)
};

void stager(Stream &S, AsyncState &A)
{
	for (size_t k = 0; k < S.items.size(); ++k) {
		Item &it = S.items[k];
		{
			std::unique_lock<std::mutex> lk(S.mu);
			S.cv.wait(lk, [&] { return S.stop || (it.state == ST_LOADED && k < A.n_run + (size_t)kInSlots); });
			if (S.stop) return;
		}
		int ok = 1;
		if (it.rc == EMA_OK && it.slot < 0 && it.n_pairs <= S.cap) {
			std::lock_guard<std::mutex> hold(S.stage_mu);      // the engine's host-side staging buffers: one user at a time
			const int rc = ema_engine_stage_async(S.eng[0], 1 + (int)(k % (size_t)kInSlots), it.bases, it.off, it.n_pairs);
			if (rc != EMA_OK) { it.rc = rc; it.err = ema_engine_strerror(S.eng[0]); ok = -1; }
		}
		{
			std::lock_guard<std::mutex> lk(S.mu);
			A.staged[k] = ok;
		}
		S.cv.notify_all();
	}
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
	const size_t n = S.items.size();
	std::vector<int> ticket(n, -1);
	std::vector<double> t_queued(n, 0.0);
	size_t k_run = 0, k_fetch = 0;
	auto fetch_one = [&] {
		Item &it = S.items[k_fetch];
		if (ticket[k_fetch] >= 0) {
			const int rc = ema_engine_fetch_ticket(g, ticket[k_fetch], &it.b);
			it.rc = rc;
			if (rc != EMA_OK) it.err = ema_engine_strerror(g);
			ema_engine_timing tm;
			if (ema_engine_last_timing(g, &tm) == EMA_OK) {
				it.st.seed_ms = tm.seed_ms; it.st.extend_ms = tm.extend_ms; it.st.rescue_ms = tm.rescue_ms; it.st.final_ms = tm.final_ms;
				it.st.full_tier_ms = tm.full_tier_ms;
			}
		}
		finish_item(S, it, t_queued[k_fetch]);
		++k_fetch;
	};
	while (k_fetch < n) {
		bool can_run = false;
		{
			std::unique_lock<std::mutex> lk(S.mu);
			if (S.stop) break;
			// queue the next pass if its batch is staged and fewer than EMA_MAX_INFLIGHT are in flight; otherwise fetch the oldest; otherwise wait
			S.cv.wait(lk, [&] { return S.stop || k_fetch < k_run || (k_run < n && A.staged[k_run] != 0); });
			if (S.stop) break;
			can_run = k_run < n && A.staged[k_run] != 0 && k_run - k_fetch < (size_t)EMA_MAX_INFLIGHT;
		}
		if (can_run) {
			Item &it = S.items[k_run];
			t_queued[k_run] = now_s();
			if (it.rc != EMA_OK) {      // reader or stager failed: passes straight through, in order
			} else if (it.n_pairs > S.cap) {      // beyond one batch: drain, then the engine's own piece pipeline
				while (k_fetch < k_run) fetch_one();
				std::lock_guard<std::mutex> hold(S.stage_mu);
				it.rc = ema_engine_align_pairs(g, it.bases, it.off, it.n_pairs, &it.b);
				if (it.rc != EMA_OK) it.err = ema_engine_strerror(g);
			} else {
				const int slot = it.slot >= 0 ? it.slot : 1 + (int)(k_run % (size_t)kInSlots);
				int tk = -1;
				const int rc = ema_engine_run_async(g, slot, &tk);
				if (rc != EMA_OK) { it.rc = rc; it.err = ema_engine_strerror(g); } else ticket[k_run] = tk;
			}
			{
				std::lock_guard<std::mutex> lk(S.mu);
				it.state = ST_TAKEN;
				++k_run; A.n_run = k_run;
			}
			S.cv.notify_all();
		} else if (k_fetch < k_run) fetch_one();
	}
	// on an early stop, passes still in flight are fetched and dropped so that the engine is reusable
	while (k_fetch < k_run) {
		Item &it = S.items[k_fetch];
		if (ticket[k_fetch] >= 0 && !it.b) { ema_batch_out *b = nullptr; (void)ema_engine_fetch_ticket(g, ticket[k_fetch], &b); if (b) ema_batch_free(b); }
		++k_fetch;
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
	A.staged.assign(S.items.size(), 0);
	if (S.paths) th.emplace_back(reader, std::ref(S));
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
		if (rc == EMA_OK || rc == EMA_ELIMIT) {
			if (rc == EMA_ELIMIT) { result = EMA_ELIMIT; g_err = it.err; }
			const int src = sink ? sink(user, k, it.bk, it.b, it.a) : 0;
			if (src != 0) { rc = src; g_err = "stopped by the sink"; }
			else rc = EMA_OK;
		} else g_err = (S.paths ? std::string(S.paths[k]) + ": " : std::string()) + it.err;
		release(it);
		{
			std::lock_guard<std::mutex> lk(S.mu);
			++S.delivered;
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
}

const char *ema_stream_last_error(void) { return g_err.c_str(); }

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
struct SamSink {
	ema_engine_t *e;
	ema_sam_run_opts o;
	int fd;
	ema_sam_stats *sstats;
	std::vector<const char *> names;
	int32_t next_cloud_id;
	std::string err;
};

int sam_sink(void *user, size_t k, const ema_bucket *bk, const ema_batch_out *b, const ema_aln_out *a)
{
	SamSink &S = *(SamSink *)user;
	ema_cloud_opts co = S.o.clouds;
	if (S.o.continue_cloud_ids) co.first_cloud_id = S.next_cloud_id;
	ema_clouds_out *sel = nullptr;
	int rc = ema_clouds_select(bk, b, a, S.names.data(), (int32_t)S.names.size(), &co, &sel);
	if (rc != EMA_OK) { S.err = "ema_clouds_select failed"; if (sel) ema_clouds_free(sel); return rc; }
	S.next_cloud_id = sel->next_cloud_id;
	if (S.sstats) S.sstats[k] = sel->stats;
	size_t n_bytes = 0;
	const double t0 = now_s();
	rc = ema_sam_write(S.fd, sel->lines, sel->n_lines, &S.o.sam, &n_bytes);
	if (S.sstats) S.sstats[k].write_s = now_s() - t0;
	ema_clouds_free(sel);
	if (rc != EMA_OK) { S.err = "ema_sam_write failed"; return rc; }
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
	const int rc = ema_stream_buckets(e, paths, n, &S.o.stream, sam_sink, &S, bstats);
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
