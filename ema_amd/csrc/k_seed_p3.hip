// ema_amd/csrc/k_seed_p3.hip -- K1c: pass 3 of mem_collect_intv (the LAST-like seeds, bwa's bwt_seed_strategy1) as a kernel of
// its own, one lane per read, behind K1's passes 1 and 2.
//
// Replaces (un-vendored bwa, reached from reference src/bwabridge.c:236-237 -> mem_align1_core -> mem_collect_intv): the loop
//     while (x < len) if (seq[x] < 4) { x = bwt_seed_strategy1(bwt, len, seq, x, min_seed_len, max_mem_intv, &m); if (m.x[2] > 0) push(m); } else ++x;
// i.e. from every start x a forward extension base by base until the match is at least min_seed_len + 1 bases long and occurs
// fewer than max_mem_intv times; that match is a seed interval and the next start is the base behind it.
//
// Why apart from K1.  K1's lane machine (k_seed.hip) has twenty-two states, and a wavefront executes the code of every state that
// one of its 64 lanes is in: ~1,300 vector and ~1,000 scalar instructions per tick whatever the tick achieves (r04: 17-20 K clocks).
// Pass 3 is 13 % of K1's lane-ticks and needs three of those states -- next start, extend, result -- no working lists, no
// backward rows, no text: as its own machine a tick is a few hundred instructions.  K1 (built without those states) ends a read
// after pass 2 and leaves the extends it used in DevOpts::seed_ext; this kernel appends pass 3's intervals to the read's list
// (the consumer orders the list by (start, end): the order of discovery is not part of the result) and counts on from there, so
// the lean budget gives up exactly the reads it gave up before.  Same requests as K1's phase B: a rank query = the two 32-byte
// blocks of k' - 1 and k' - 1 + size; a string of at most kmer_k bases = one table entry (and, at the table's last level, the
// entry of its reverse complement: the next rank query needs k'); the first min(kmer_k, min_seed_len) bases of a seed in one look-up.
// Bound: instruction issue + the same dependent gathers per extend as K1.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"

namespace {
__device__ __forceinline__ uint32_t p3_rev_groups(uint32_t v)      // the order of the 16 two-bit groups of v reversed (k_seed.hip, seed_rev_groups)
{
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	return (v >> 16) | (v << 16);
}
}  // namespace

// reads / intv / n_intv / status / long list: as for ema_k_seed_t (k_seed.hip); ext: the extends passes 1 and 2 used, per read.
// counter: zero on entry.  A read K1 flagged (status != 0) is left as it is.
__global__ void __launch_bounds__(256)
ema_k_seed_p3(DevIndex ix, DevOpts opt, const uint32_t *__restrict__ qpack, const uint32_t *__restrict__ off, int n_reads,
              const int *__restrict__ n_pairs_dev, const int *__restrict__ map, Intv *__restrict__ intv, int *__restrict__ n_intv,
              int *__restrict__ status, const int32_t *__restrict__ ext, int *__restrict__ counter, int *__restrict__ long_list,
              int *__restrict__ n_long, int long_cap)
{
	__shared__ uint32_t lds_q[4][16 * 64];      // 2-bit read codes, 16 words per lane, lane-interleaved
	__shared__ uint32_t lds_n[4][8 * 64];       // N mask, 8 words per lane
	__shared__ uint64_t lds_rt[4][EMA_RANK_TABLE];      // ema_extend_blocks' bases, one copy per wavefront
	const int lane = (int)(threadIdx.x & 63), wib = ema_uni((int)(threadIdx.x >> 6));
	const uint32_t *qw = lds_q[wib] + lane, *nm = lds_n[wib] + lane;
	const uint64_t *rt = lds_rt[wib];
	ema_rank_table_init(ix, lds_rt[wib]);
	const int n_tasks = ema_work_count(n_reads, n_pairs_dev, 2);
	const int kk = ix.kmer_k;
	const int jump = kk > 0 ? (kk < opt.min_seed_len ? kk : opt.min_seed_len) : 0;
	auto q = [&](int p_) -> int {
		const int code = (qw[(p_ >> 4) << 6] >> ((p_ & 15) << 1)) & 3;
		return ((nm[(p_ >> 5) << 6] >> (p_ & 31)) & 1) ? 4 : code;
	};
	enum { P_IDLE = 0, P_NEXT, P_EXT, P_RES };
	int pc = P_IDLE, read = -1, len = 0, x = 0, i = 0, n_out = 0, st = 0, n_ext = 0;
	uint64_t c0 = 0, c1 = 0, c2 = 0, r0 = 0, r1 = 0, r2 = 0;
	uint32_t c_code = 0, r_code = 0, req_code = 0, req_len = 0;
	int has_req = 0, req_c = 0;      // 1: rank query, 2: table look-up
	size_t out_base = 0;
	bool exhausted = false;
	for (;;) {
		// ---- phase A: the lane's control program until it has a request
		while (!has_req && !exhausted) {
			if (pc == P_RES) {      // bwt_seed_strategy1's loop body after the extend
				if (r2 < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
					if (r2 > 0) {
						if (n_out >= opt.intv_cap) st |= EMA_ST_INTV_OVERFLOW;
						else {
							Intv e; e.x0 = r0; e.x1 = kk ? 0 : r1; e.x2 = r2; e.info = (uint64_t)(uint32_t)x << 32 | (uint32_t)(i + 1);
							intv[out_base + n_out] = e;
							++n_out;
						}
					}
					x = i + 1; pc = P_NEXT;
				} else {
					c0 = r0; c1 = r1; c2 = r2; c_code = r_code;
					if (++i == len) { x = len; pc = P_IDLE; } else pc = P_EXT;
				}
			}
			if (pc == P_IDLE) {      // the read is finished (or none yet): its totals out, the next read in
				if (read >= 0) {
					n_intv[read] = n_out; status[read] = st;
					if ((st & EMA_ST_LONG) && long_list) {
						const int at = atomicAdd(n_long, 1);
						if (at < long_cap) long_list[at] = read;
					}
				}
				read = atomicAdd(counter, 1);
				if (read >= n_tasks) { read = -1; exhausted = true; break; }
				st = status[read];
				if (st) { read = -1; continue; }      // given up by K1 (budget, capacity): the full-capacity tier redoes the pair
				const int in_read = ema_in_read(map, read);
				len = (int)(off[in_read + 1] - off[in_read]);
				const uint4 *pw = reinterpret_cast<const uint4 *>(qpack + (size_t)in_read * 24);
				const uint4 a = pw[0], b = pw[1], c = pw[2], d = pw[3], m0 = pw[4], m1 = pw[5];
				uint32_t *qd = lds_q[wib] + lane, *nd = lds_n[wib] + lane;
				qd[0 << 6] = a.x; qd[1 << 6] = a.y; qd[2 << 6] = a.z; qd[3 << 6] = a.w;
				qd[4 << 6] = b.x; qd[5 << 6] = b.y; qd[6 << 6] = b.z; qd[7 << 6] = b.w;
				qd[8 << 6] = c.x; qd[9 << 6] = c.y; qd[10 << 6] = c.z; qd[11 << 6] = c.w;
				qd[12 << 6] = d.x; qd[13 << 6] = d.y; qd[14 << 6] = d.z; qd[15 << 6] = d.w;
				nd[0 << 6] = m0.x; nd[1 << 6] = m0.y; nd[2 << 6] = m0.z; nd[3 << 6] = m0.w;
				nd[4 << 6] = m1.x; nd[5 << 6] = m1.y; nd[6 << 6] = m1.z; nd[7 << 6] = m1.w;
				out_base = (size_t)read * opt.intv_cap;
				n_out = n_intv[read]; n_ext = ext[read]; x = 0;
				pc = len >= opt.min_seed_len ? P_NEXT : P_IDLE;      // mem_chain: no seeds for a read shorter than min_seed_len
				if (pc == P_IDLE) { read = -1; continue; }           // (nothing to add: its totals stand as K1 left them)
			}
			if (pc == P_NEXT) {      // the next start
				while (x < len && q(x) > 3) ++x;
				if (x >= len) { pc = P_IDLE; continue; }
				if (jump > 0) {      // the first `jump` bases in one look-up (nothing is tested before the match is min_seed_len + 1 long)
					if (len - x <= opt.min_seed_len) { x = len; pc = P_IDLE; continue; }      // no seed fits any more
					const int wn = x >> 5, wq = x >> 4;
					const uint64_t nn = (uint64_t)(wn < 7 ? nm[(wn + 1) << 6] : 0u) << 32 | nm[wn << 6];
					const uint32_t nbits = (uint32_t)(nn >> (x & 31)) & ((1u << jump) - 1u);
					if (nbits) { x += __ffs(nbits); continue; }      // an ambiguous base ends the attempt; the next one starts behind it
					if (++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = P_IDLE; continue; }
					const uint64_t qq = (uint64_t)(wq < 15 ? qw[(wq + 1) << 6] : 0u) << 32 | qw[wq << 6];
					req_code = p3_rev_groups((uint32_t)(qq >> ((x & 15) << 1))) >> (32 - 2 * jump);
					req_len = (uint32_t)jump; req_c = 0; has_req = 2;
					i = x + jump - 1;
					pc = P_RES;
					continue;
				}
				{
					const int b = q(x);
					c0 = ix.L2[b] + 1; c2 = ix.L2[b + 1] - ix.L2[b]; c1 = ix.L2[3 - b] + 1; c_code = (uint32_t)b;
				}
				i = x + 1;
				if (i >= len) { x = len; pc = P_IDLE; continue; }
				pc = P_EXT;
			}
			if (pc == P_EXT) {      // one more base to the right
				const int b = (i >= 0 && i < len) ? q(i) : 4;
				if (b < 4 && ++n_ext > opt.seed_budget) { st |= EMA_ST_LONG; pc = P_IDLE; }      // too long for this tier
				else if (b < 4) {
					has_req = 1; req_c = 3 - b;
					const int rl = i + 1 - x;      // q[x .. i]
					if (rl <= kk) { has_req = 2; req_len = (uint32_t)rl; req_code = (c_code << 2) | (uint32_t)b; }
					pc = P_RES;
				} else { x = i + 1; pc = P_NEXT; }
			}
		}
		if (!__ballot(!exhausted)) break;
		// ---- phase B: the tick's loads, issued together (k_seed.hip, phase B: the rank and table branches, forward extension)
		if (has_req) {
			const bool tab = has_req == 2;
			const uint64_t x_nb = c1, x_b = c0, sz = c2;
			const uint64_t pk = x_nb - 1, pl = x_nb - 1 + sz;
			const uint64_t qk = pk - (pk >= ix.primary ? 1 : 0), ql = pl - (pl >= ix.primary ? 1 : 0);      // '$' is not stored
			const bool want_rc = tab && (int)req_len == kk;
			const uint4 *p0, *p1, *p2, *p3;
			if (tab) {
				const int L = (int)req_len;
				const uint32_t rcode = want_rc ? ema_kmer_revcomp(req_code, kk) : req_code;
				if (L <= EMA_KMER_WIDE) {
					const size_t base = ema_kmer_base_wide(L);
					p0 = reinterpret_cast<const uint4 *>(ix.kmer_wide + 2 * (base + req_code));
					p2 = reinterpret_cast<const uint4 *>(ix.kmer_wide + 2 * (base + rcode));
				} else {
					const size_t base = ema_kmer_base_narrow(L);
					p0 = reinterpret_cast<const uint4 *>(ix.kmer_narrow + base + req_code);
					p2 = reinterpret_cast<const uint4 *>(ix.kmer_narrow + base + rcode);
				}
				p1 = p0; p3 = p2;
			} else {
				p0 = reinterpret_cast<const uint4 *>(ix.occ + (qk >> 6));
				p2 = reinterpret_cast<const uint4 *>(ix.occ + (ql >> 6));
				p1 = p0 + 1; p3 = p2 + 1;
			}
			const uint4 a0 = *p0, a1 = *p1, b0 = *p2, b1 = *p3;
			if (tab) {
				const uint64_t ea = (uint64_t)a0.y << 32 | a0.x, eb = (uint64_t)b0.y << 32 | b0.x;
				if ((int)req_len <= EMA_KMER_WIDE) { r0 = ea; r2 = (uint64_t)a0.w << 32 | a0.z; r1 = want_rc ? eb : 0; }
				else { r0 = ea & 0xFFFFFFFFFFULL; r2 = ea >> 40; r1 = want_rc ? (eb & 0xFFFFFFFFFFULL) : 0; }
				r_code = req_code;
			} else {
				uint64_t o_nb; uint32_t o_size, n_gt;
				ema_extend_blocks(ix, rt, qk, ql, a0, a1, b0, b1, req_c & 3, o_nb, o_size, n_gt);
				r0 = x_b + ((x_nb <= ix.primary && x_nb + sz - 1 >= ix.primary) ? 1 : 0) + n_gt; r1 = o_nb; r2 = o_size;
				r_code = 0;
			}
			has_req = 0;
		}
	}
}

extern "C" void ema_launch_seed_p3(const DevIndex *ix, const DevOpts *opt, const uint32_t *qpack, const uint32_t *off, int n_reads, const int *n_pairs_dev,
                                   const int *map, Intv *intv, int *n_intv, int *status, const int32_t *ext, int *counter, int *long_list, int *n_long,
                                   int long_cap, int n_blocks, hipStream_t stream)
{
	if (opt->max_mem_intv <= 0 || n_reads <= 0) return;      // (mem_collect_intv runs pass 3 only then)
	hipLaunchKernelGGL(ema_k_seed_p3, dim3(n_blocks), dim3(256), 0, stream, *ix, *opt, qpack, off, n_reads, n_pairs_dev, map, intv, n_intv, status, ext,
	                   counter, long_list, n_long, long_cap);
}
