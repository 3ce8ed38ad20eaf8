// ema_amd/csrc/host_sam.cpp -- SAM record formatter (include/ema_sam.h; SURVEY 8f rank 1, the writer part).
//
// print_sam_record() (reference src/samrecord.c:104-284) writes one line through ~30 fprintf/fputc calls under the
// output lock; here a batch of lines is formatted by the host's cores, each thread appending to its own buffer with
// hand-rolled integer output, and the pieces are laid end to end, so the caller issues one large write per batch
// (~2 records x ~450 B per pair is GB/s of text at the engine's rate).  The text is the reference's, byte for
// byte; "%.5g" goes through snprintf so that the gamma field cannot differ.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>
#include <poll.h>
#include <cerrno>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "ema_sam.h"
#include "host_cpuacct.h"
#include "host_fmt.h"
#include "host_pool.h"

namespace {

enum { kPaired = 1, kProper = 2, kUnmapped = 4, kMateUnmapped = 8, kReversed = 16, kMateReversed = 32, k1st = 64, k2nd = 128,
       kDup = 1024 };      // reference include/samrecord.h:73-81

// Append-only text buffer: room for a whole line is made once (line_bound), then bytes go through a bare cursor.
struct Out {
	std::vector<char> buf;
	size_t n = 0;
	char *p = nullptr;
	void room(size_t more)
	{
		if (n + more > buf.size()) buf.resize(std::max(buf.size() * 2, n + more + (1 << 16)));
		p = buf.data() + n;
	}
	void done() { n = (size_t)(p - buf.data()); }
	void ch(char c) { *p++ = c; }
	void str(const char *q) { while (*q) *p++ = *q++; }
	void mem(const char *q, size_t k) { memcpy(p, q, k); p += k; }
	void u64(uint64_t v)
	{
		char t[24]; int k = 0;
		do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
		while (k) *p++ = t[--k];
	}
	void i64(int64_t v) { if (v < 0) { *p++ = '-'; u64((uint64_t)0 - (uint64_t)v); } else u64((uint64_t)v); }
};

// what a call's lines share: lengths of the option strings (the RG identifier ends at its first whitespace, src/samrecord.c)
struct Shared { size_t rg_len = 0, bx_len = 0; };

const struct CompTable {
	char t[256];
	CompTable() { memset(t, 0, sizeof t); t['A'] = 'T'; t['C'] = 'G'; t['G'] = 'C'; t['T'] = 'A'; t['N'] = 'N'; }
	char operator[](unsigned char c) const { return t[c]; }
} kComp;

size_t line_bound(const ema_sam_rec *rec, const ema_sam_rec *mate, const Shared &sh, size_t ident_len, size_t chrom_len, size_t mchrom_len)
{
	size_t b = 512 + ident_len + sh.rg_len + sh.bx_len + chrom_len + mchrom_len;
	b += 2 * (size_t)(rec ? rec->read_len : mate->mate_read_len);
	if (rec) {
		b += 12 * (size_t)rec->n_cigar;
		for (size_t i = 0; i < rec->n_alts; ++i) b += strlen(rec->alts[i].chrom) + 12 * (size_t)rec->alts[i].n_cigar + 48;
	}
	return b;
}

// A reversed read and its qualities, sixteen bytes at a time where the CPU has PSHUFB (every x86-64 of the last fifteen years;
// checked at run time): the byte order through one shuffle, the complement through a second one on the low nibbles of
// A C G T N (1 3 7 4 14: all different), and a third one that maps the result back, which only the five valid bytes survive.
// rc(), src/samrecord.c:86-102; false = a byte outside ACGTN.
bool revcomp_scalar(char *dst, const char *src, int n)
{
	for (int i = n - 1; i >= 0; --i) {
		const char c = kComp[(unsigned char)src[i]];
		if (!c) return false;
		*dst++ = c;
	}
	return true;
}
void reverse_scalar(char *dst, const char *src, int n) { for (int i = n - 1; i >= 0; --i) *dst++ = src[i]; }

#if defined(__x86_64__)
__attribute__((target("ssse3"))) bool revcomp_ssse3(char *dst, const char *src, int n)
{
	const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
	const __m128i comp = _mm_setr_epi8(0, 'T', 0, 'G', 'A', 0, 0, 'C', 0, 0, 0, 0, 0, 0, 'N', 0);      // by low nibble: A=1 C=3 T=4 G=7 N=14
	const __m128i low = _mm_set1_epi8(0x0f);
	int i = n;
	__m128i bad = _mm_setzero_si128();
	while (i >= 16) {
		i -= 16;
		const __m128i x = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(src + i)), rev);
		const __m128i y = _mm_shuffle_epi8(comp, _mm_and_si128(x, low));
		const __m128i back = _mm_shuffle_epi8(comp, _mm_and_si128(y, low));      // the complement of the complement: x again, if x is valid
		bad = _mm_or_si128(bad, _mm_xor_si128(back, x));
		bad = _mm_or_si128(bad, _mm_cmpeq_epi8(y, _mm_setzero_si128()));      // no complement (a NUL among them: its round trip is 0 too)
		_mm_storeu_si128((__m128i *)dst, y);
		dst += 16;
	}
	if (_mm_movemask_epi8(_mm_cmpeq_epi8(bad, _mm_setzero_si128())) != 0xffff) return false;
	return revcomp_scalar(dst, src, i);
}
__attribute__((target("ssse3"))) void reverse_ssse3(char *dst, const char *src, int n)
{
	const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
	int i = n;
	while (i >= 16) {
		i -= 16;
		_mm_storeu_si128((__m128i *)dst, _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(src + i)), rev));
		dst += 16;
	}
	reverse_scalar(dst, src, i);
}
const bool kHaveSsse3 = __builtin_cpu_supports("ssse3");
#else
const bool kHaveSsse3 = false;
bool revcomp_ssse3(char *dst, const char *src, int n) { return revcomp_scalar(dst, src, n); }
void reverse_ssse3(char *dst, const char *src, int n) { reverse_scalar(dst, src, n); }
#endif

int ref_len(int n_cigar, const uint32_t *cigar)      // get_rlen, src/samrecord.c:75-84
{
	int l = 0;
	for (int k = 0; k < n_cigar; ++k) { const int op = (int)(cigar[k] & 0xf); if (op == 0 || op == 2) l += (int)(cigar[k] >> 4); }
	return l;
}

void put_cigar(Out &o, const uint32_t *cigar, int n)      // hard clips shown as soft: "MIDSS"
{
	for (int i = 0; i < n; ++i) { o.u64(cigar[i] >> 4); o.ch("MIDSS"[cigar[i] & 0xf]); }
}

bool is_pair(const ema_sam_rec *r1, const ema_sam_rec *r2, const ema_sam_opts &opt)      // src/align.c:27-40
{
	if (r1->rev == r2->rev || r1->chrom_id != r2->chrom_id) return false;
	if (r2->rev) { const ema_sam_rec *t = r2; r2 = r1; r1 = t; }
	const int64_t d = (int64_t)(uint32_t)(r1->pos - r2->pos);      // the reference subtracts two uint32_t: never negative
	return opt.insert_min <= d && d <= opt.insert_max;
}

void decode_bc(uint64_t bc, const ema_sam_opts &opt, char *out)      // src/util.c:78-95
{
	if (opt.is_haplotag) {
		snprintf(out, 40, "A%02uC%02uB%02uD%02u", (unsigned)(bc >> 24) & 127, (unsigned)(bc >> 16) & 127, (unsigned)(bc >> 8) & 127,
		         (unsigned)bc & 127);
		return;
	}
	for (int i = 0; i < opt.bc_len; ++i) { out[i] = "ACGT"[bc & 3]; bc >>= 2; }
	out[opt.bc_len] = 0;
}

// one line; false: a base outside ACGTN in a reversed read
bool put_line(Out &o, const ema_sam_rec *rec, const ema_sam_rec *mate, const ema_sam_opts &opt, const Shared &sh)
{
	const size_t ident_len = strlen(rec ? rec->ident : mate->ident), chrom_len = rec ? strlen(rec->chrom) : 1, mchrom_len = mate ? strlen(mate->chrom) : 0;
	o.room(line_bound(rec, mate, sh, ident_len, chrom_len, mchrom_len));
	int flag = kPaired, mapq = 0, read_len;
	const char *ident, *chrom = "*", *read, *qual;
	uint32_t pos = 0;
	uint64_t bc;
	if (rec) {
		ident = rec->ident; chrom = rec->chrom; pos = rec->pos; read_len = rec->read_len; bc = rec->bc; read = rec->read; qual = rec->qual;
		const int gamma_mapq = rec->gamma <= 0.999999 ? (int)(-10 * std::log10(1 - rec->gamma)) : 60;
		mapq = gamma_mapq < rec->score_mapq ? gamma_mapq : rec->score_mapq;
		mapq = mapq < rec->mapq ? mapq : rec->mapq;
		mapq = mapq > 0 ? mapq : 0;
		mapq = mapq < 60 ? mapq : 60;
		if (rec->rev) flag |= kReversed;
		if (rec->duplicate) flag |= kDup;
		flag |= rec->mate == 0 ? k1st : k2nd;
	} else {
		ident = mate->ident; read_len = mate->mate_read_len; bc = mate->bc; read = mate->mate_read; qual = mate->mate_qual;
		flag |= kUnmapped;
		flag |= mate->mate == 0 ? k2nd : k1st;
	}
	if (mate) {
		if (rec && is_pair(rec, mate, opt)) flag |= kProper;
		if (mate->rev) flag |= kMateReversed;
	} else flag |= kMateUnmapped;
	o.mem(ident, ident_len); o.ch('\t'); o.i64(flag); o.ch('\t'); o.mem(chrom, chrom_len); o.ch('\t'); o.u64(pos); o.ch('\t'); o.i64(mapq); o.ch('\t');
	if (rec) put_cigar(o, rec->cigar, rec->n_cigar); else o.ch('*');
	if (mate) {
		const bool same_chrom = rec && mate->chrom_id == rec->chrom_id;
		o.ch('\t');
		if (same_chrom) o.ch('='); else o.mem(mate->chrom, mchrom_len);
		o.ch('\t'); o.i64((int)mate->pos);      // "%d" of a uint32_t
		if (same_chrom) {
			const int64_t p0 = rec->aln_pos + (rec->aln_rev ? ref_len(rec->n_cigar, rec->cigar) - 1 : 0);
			const int64_t p1 = mate->aln_pos + (mate->aln_rev ? ref_len(mate->n_cigar, mate->cigar) - 1 : 0);
			o.ch('\t');
			if (mate->n_cigar == 0 || rec->n_cigar == 0) o.ch('0');
			else o.i64(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else o.str("\t0");
	} else o.str("\t*\t0\t0");
	o.ch('\t');
	if (rec && rec->rev) {
		if (!(kHaveSsse3 ? revcomp_ssse3(o.p, read, read_len) : revcomp_scalar(o.p, read, read_len))) return false;
		o.p += read_len;
		o.ch('\t');
		if (kHaveSsse3) reverse_ssse3(o.p, qual, read_len); else reverse_scalar(o.p, qual, read_len);
		o.p += read_len;
	} else {
		o.mem(read, (size_t)read_len); o.ch('\t'); o.mem(qual, (size_t)read_len);
	}
	char bc_str[48];
	decode_bc(bc, opt, bc_str);
	if (rec) {
		char g[48];
		if (rec->gamma == 1.0) { g[0] = '1'; g[1] = 0; }      // "%.5g" of the two values most records carry, without the library call
		else if (rec->gamma == 0.0) { g[0] = '0'; g[1] = 0; }
		else if (const int k5 = ema_fmt_g5(rec->gamma, g)) g[k5] = 0;      // host_fmt.h: exact, without the library call
		else snprintf(g, sizeof g, "%.5g", rec->gamma);
		o.str("\tNM:i:"); o.i64(rec->edit_dist); o.str("\tBX:Z:"); o.str(bc_str);
		if (!opt.is_haplotag) { o.ch('-'); o.mem(opt.bx_index, sh.bx_len); }
		o.str("\tXG:f:"); o.str(g); o.str("\tMI:i:"); o.i64(rec->cloud_id); o.str("\tXF:i:"); o.i64(rec->cloud_bad);
	} else {
		o.str("\tBX:Z:"); o.str(bc_str);
		if (!opt.is_haplotag) o.str("-1");      // the literal suffix, not bx_index (src/samrecord.c:255)
	}
	if (opt.rg_id) {
		o.str("\tRG:Z:");
		o.mem(opt.rg_id, sh.rg_len);
	}
	if (rec && rec->n_alts > 0) {
		o.str("\tXA:Z:");
		for (size_t i = 0; i < rec->n_alts; ++i) {
			const ema_sam_alt &a = rec->alts[i];
			o.str(a.chrom); o.ch(','); o.ch(a.rev ? '-' : '+'); o.i64((int)a.pos); o.ch(',');
			put_cigar(o, a.cigar, a.n_cigar);
			o.ch(','); o.i64(a.edit_dist); o.ch(';');
		}
	}
	o.ch('\n');
	o.done();
	return true;
}

}  // namespace

extern "C" {

void ema_sam_opts_default(ema_sam_opts *o)
{
	if (!o) return;
	o->rg_id = nullptr; o->bx_index = "1"; o->bc_len = 16; o->is_haplotag = 0; o->insert_min = -35; o->insert_max = 750;
}

void ema_sam_free(char *text) { free(text); }

// lines[0..n) formatted by the host's threads into one buffer per thread, in order
static int format_parts(const ema_sam_line *lines, size_t n, const ema_sam_opts *opt, std::vector<Out> &parts)
{
	if ((!lines && n) || !opt || !opt->bx_index || opt->bc_len < 0 || opt->bc_len > 32 || (opt->is_haplotag && opt->bc_len != 12)) return EMA_EARG;
	for (size_t i = 0; i < n; ++i) if (!lines[i].rec && !lines[i].mate) return EMA_EARG;
	const size_t t = n < 4096 ? 1 : (size_t)EmaPool::get().size(), per = (n + t - 1) / t;      // host_pool.h
	// the threads' buffers are the caller's thread's from call to call (grown, never handed back: a fresh 30 MB vector per call is
	// zero-filled and page-faulted before the first byte is formatted)
	if (parts.size() < t) parts.resize(t);
	for (auto &pt : parts) { pt.n = 0; pt.p = nullptr; }
	std::vector<int> bad(t, 0);
	Shared sh;
	sh.bx_len = strlen(opt->bx_index);
	if (opt->rg_id) for (size_t i = 0; opt->rg_id[i] && !(opt->rg_id[i] == ' ' || (opt->rg_id[i] >= '\t' && opt->rg_id[i] <= '\r')); ++i) sh.rg_len = i + 1;
	auto work = [&](size_t k) {
		EMA_CPU(EMA_CPU_FORMAT);
		const size_t lo = std::min(n, k * per), hi = std::min(n, lo + per);
		if (parts[k].buf.size() < (hi - lo) * 400 + (1 << 16)) parts[k].buf.resize((hi - lo) * 400 + (1 << 16));
		for (size_t i = lo; i < hi; ++i) if (!put_line(parts[k], lines[i].rec, lines[i].mate, *opt, sh)) { bad[k] = 1; return; }
	};
	EmaPool::get().run(t, work);
	for (int b : bad) if (b) return EMA_EFORMAT;
	return 0;
}

int ema_sam_format(const ema_sam_line *lines, size_t n, const ema_sam_opts *opt, char **text, size_t *n_bytes)
{
	if (!text || !n_bytes) return EMA_EARG;
	EMA_CPU(EMA_CPU_FORMAT);
	*text = nullptr; *n_bytes = 0;
	static thread_local std::vector<Out> parts_tls;      // kept from call to call (see format_parts)
	std::vector<Out> &parts = parts_tls;                  // (a lambda run by another thread would name ITS instance of a thread_local)
	const int rc = format_parts(lines, n, opt, parts);
	if (rc) return rc;
	const size_t t = parts.size();
	size_t total = 0;
	std::vector<size_t> at(t + 1, 0);
	for (size_t k = 0; k < t; ++k) { total += parts[k].n; at[k + 1] = total; }
	char *buf = (char *)malloc(total + 1);
	if (!buf) return EMA_EARG;
	{
		auto copy = [&](size_t k) { EMA_CPU(EMA_CPU_FORMAT); memcpy(buf + at[k], parts[k].buf.data(), parts[k].n); };
		EmaPool::get().run(t, copy);
	}
	*text = buf; *n_bytes = total;
	return 0;
}

int ema_sam_write(int fd, const ema_sam_line *lines, size_t n, const ema_sam_opts *opt, size_t *n_bytes)
{
	EMA_CPU(EMA_CPU_FORMAT);
	if (n_bytes) *n_bytes = 0;
	static thread_local std::vector<Out> parts_tls;      // kept from call to call (see format_parts)
	std::vector<Out> &parts = parts_tls;                  // (a lambda run by another thread would name ITS instance of a thread_local)
	const int rc = format_parts(lines, n, opt, parts);
	if (rc) return rc;
	// an interrupted or momentarily refused write is retried; on a real failure *n_bytes says how much text is on the fd
	// (a descriptor that stays unwritable for 120 polls of one second in a row is a failure too: EMA_EIO, not a hang)
	size_t total = 0;
	int refused = 0;
	for (const Out &o : parts) {
		size_t at = 0;
		while (at < o.n) {
			const ssize_t w = write(fd, o.buf.data() + at, o.n - at);
			if (w < 0 && errno == EINTR) continue;
			if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK) && ++refused <= 120) {
				struct pollfd pf; pf.fd = fd; pf.events = POLLOUT; pf.revents = 0;
				(void)poll(&pf, 1, 1000);
				continue;
			}
			if (w > 0) refused = 0;
			if (w <= 0) { if (n_bytes) *n_bytes = total + at; return EMA_EIO; }
			at += (size_t)w;
		}
		total += o.n;
	}
	if (n_bytes) *n_bytes = total;
	return 0;
}

int ema_sam_header(const char *const *contig_names, const int32_t *contig_lens, int32_t n_contigs, const char *rg_line,
                   const char *version, int argc, const char *const *argv, char **text, size_t *n_bytes)
{
	if (!text || !n_bytes) return EMA_EARG;
	*text = nullptr; *n_bytes = 0;
	if (n_contigs < 0 || (n_contigs && (!contig_names || !contig_lens)) || !version || argc < 1 || !argv) return EMA_EARG;
	std::string s = "@HD\tVN:1.3\tSO:unsorted\n";
	for (int32_t i = 0; i < n_contigs; ++i) {
		s += "@SQ\tSN:"; s += contig_names[i]; s += "\tLN:"; s += std::to_string(contig_lens[i]); s += '\n';
	}
	if (rg_line) { s += rg_line; s += '\n'; }
	s += "@PG\tID:ema\tPN:ema\tVN:"; s += version; s += "\tCL:"; s += argv[0];
	for (int i = 1; i < argc; ++i) { s += ' '; s += argv[i]; }
	s += '\n';
	char *buf = (char *)malloc(s.size() + 1);
	if (!buf) return EMA_EARG;
	memcpy(buf, s.data(), s.size());
	*text = buf; *n_bytes = s.size();
	return 0;
}

}  // extern "C"
