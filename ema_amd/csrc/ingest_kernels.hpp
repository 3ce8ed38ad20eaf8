// ema_amd/csrc/ingest_kernels.hpp -- the bucket reader's kernels (driver: ingest_dev.hip, which explains the passes; the host SIMT
// interpreter of tests/emu/ runs the same source on the CPU: tests/test_ingest_device_emu.py).
#ifndef EMA_INGEST_KERNELS_HPP
#define EMA_INGEST_KERNELS_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace {

const uint32_t kMaxLine = 4999;      // fgets(buf, 5000) (src/align.c:762,768)
const uint32_t kMaxId = 149;         // id[150] (include/samrecord.h:12)

struct Fields { uint16_t id_b, id_l, r1_b, r1_l, q1_b, r2_b, r2_l, q2_b; };      // byte offsets within the line (host_ingest.cpp's)

__device__ __forceinline__ bool is_space(unsigned c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

// line i of the text: [start, start + len), the '\n' counted (as in fgets' buffer)
__device__ __forceinline__ void line_of(const uint32_t *nl, uint32_t n_nl, uint32_t text_len, uint32_t i, uint32_t &start, uint32_t &len)
{
	start = i ? nl[i - 1] + 1 : 0;
	len = (i < n_nl ? nl[i] + 1 : text_len) - start;
}

__device__ __forceinline__ uint32_t field_end(const char *s, uint32_t at, uint32_t len)
{
	while (at + 8 <= len) {      // eight bytes at a time: a word with no byte below 0x21 holds no separator
		uint64_t x;
		__builtin_memcpy(&x, s + at, 8);
		const uint64_t low = (x - 0x2121212121212121ULL) & ~x & 0x8080808080808080ULL;
		if (!low) { at += 8; continue; }
		at += (uint32_t)(__ffsll((long long)low) - 1) >> 3;
		if (!s[at] || is_space((unsigned char)s[at])) return at;
		++at;      // some other control byte: part of the field
	}
	while (at < len && s[at] && !is_space((unsigned char)s[at])) ++at;
	return at;
}
__device__ __forceinline__ bool next_field(const char *s, uint32_t len, uint32_t &at, uint32_t &b, uint32_t &l)
{
	if (at > len) return false;
	b = at;
	at = field_end(s, at, len);
	l = at - b;
	at = (at < len && s[at]) ? at + 1 : len + 1;
	return true;
}

// bytes of w equal to zero, exactly: 0x80 in each such byte
__device__ __forceinline__ uint32_t zero_bytes(uint32_t w) { return ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w | 0x7F7F7F7Fu); }

__global__ void __launch_bounds__(256)
ema_k_ing_count(const char *__restrict__ text, uint32_t len, unsigned long long *__restrict__ n_nl, int *__restrict__ irregular)
{
	const uint32_t at = (blockIdx.x * 256u + threadIdx.x) * 16u;      // (the text starts on a 256-byte boundary: sixteen bytes are one aligned load)
	int c = 0, z = 0;
	if (at + 16u <= len) {
		const uint4 v = *(const uint4 *)(text + at);
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int k = 0; k < 4; ++k) { c += __popc(zero_bytes(w[k] ^ 0x0A0A0A0Au)); z |= zero_bytes(w[k]) != 0; }
	} else if (at < len) {
		for (uint32_t k = at; k < len; ++k) { c += text[k] == '\n'; z |= text[k] == 0; }
	}
	if (z) atomicOr(irregular, 1);      // a NUL ends the reference's C strings early: the host reader's business
	// one atomic per wave, not per thread with a newline: a quarter of a million additions to ONE address took the kernel 2.4 ms
	for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
	if ((threadIdx.x & 63u) == 0 && c) atomicAdd(n_nl, (unsigned long long)c);
}

struct IsNewline {
	const char *text;
	__device__ bool operator()(uint32_t i) const { return text[i] == '\n'; }
};

__global__ void __launch_bounds__(256)
ema_k_ing_parse(const char *__restrict__ text, uint32_t text_len, const uint32_t *__restrict__ nl, uint32_t n_nl, uint32_t n_lines, int bc_len,
                int is_haplotag, uint32_t max_read_len, Fields *__restrict__ fields, uint64_t *__restrict__ codes, uint32_t *__restrict__ codes_lo,
                uint32_t *__restrict__ idx, int *__restrict__ irregular)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_lines) return;
	uint32_t start, ln;
	line_of(nl, n_nl, text_len, i, start, ln);
	const char *s = text + start;
	Fields f;
	memset(&f, 0, sizeof f);
	uint32_t p = 0, fb[6] = {0, 0, 0, 0, 0, 0}, fl[6] = {0, 0, 0, 0, 0, 0};
	bool six = ln <= kMaxLine;
#pragma unroll
	for (int x = 0; x < 6; ++x) if (six) six = next_field(s, ln, p, fb[x], fl[x]);
	bool bad = !six || fl[0] != (uint32_t)bc_len || fl[1] == 0 || fl[1] > kMaxId || fl[2] > max_read_len || fl[4] > max_read_len ||
	           fl[3] != fl[2] || fl[5] != fl[4];
	uint64_t code = 0;
	uint32_t code_lo = 0;
	if (!bad && is_haplotag) {
		// AxxCxxBxxDxx (bc_len 12): the reference sorts the LINES by strncmp over these twelve bytes whatever they are -- the key is the
		// bytes themselves, big-endian: eight in `code`, four in `code_lo` (two stable sorts, least significant first)
		for (int j = 0; j < 8; ++j) code = code << 8 | (uint64_t)(unsigned char)s[j];
		for (int j = 8; j < 12; ++j) code_lo = code_lo << 8 | (uint32_t)(unsigned char)s[j];
	} else if (!bad) {
		for (int j = 0; j < bc_len; ++j) {
			const char ch = s[j];
			int c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'a' ? 4 : ch == 'c' ? 5 : ch == 'g' ? 6 : ch == 't' ? 7 : -1;
			if (c < 0) { bad = true; c = 0; }
			code = code << 3 | (uint64_t)c;
		}
	}
	if (bad) { atomicOr(irregular, 2); code = 0; code_lo = 0; }      // (the host reader takes the bucket; the line still gets its -- empty -- entries: the passes
	else {                                               //  queued behind this one index by them before the flag is looked at)
		f.id_b = (uint16_t)fb[1]; f.id_l = (uint16_t)fl[1];
		f.r1_b = (uint16_t)fb[2]; f.r1_l = (uint16_t)fl[2]; f.q1_b = (uint16_t)fb[3];
		f.r2_b = (uint16_t)fb[4]; f.r2_l = (uint16_t)fl[4]; f.q2_b = (uint16_t)fb[5];
	}
	fields[i] = f; codes[i] = code; idx[i] = i;
	if (is_haplotag) codes_lo[i] = code_lo;
}

// lengths in sorted order, interleaved for one prefix sum each: rlen[2i], rlen[2i+1] (reads), ilen[i] (name); one zero past the end
__global__ void __launch_bounds__(256)
ema_k_ing_lens(const uint32_t *__restrict__ order, const Fields *__restrict__ fields, uint32_t n, uint32_t *__restrict__ rlen, uint32_t *__restrict__ ilen)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i > n) return;
	if (i == n) { rlen[2 * n] = 0; ilen[n] = 0; return; }
	const Fields f = fields[order[i]];
	rlen[2 * i] = f.r1_l; rlen[2 * i + 1] = f.r2_l; ilen[i] = f.id_l;
}

// the high key words in the order the first (low-word) sort left the lines in
__global__ void __launch_bounds__(256)
ema_k_ing_pick(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ order1, uint32_t n, uint64_t *__restrict__ out)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) out[i] = codes[order1[i]];
}

// encode_bc_haplotag (reference src/util.c:63-70): the four two-digit numbers, the reference's int arithmetic whatever the bytes are
__device__ __forceinline__ uint64_t encode_haplotag(const char *s)
{
	const int a = 10 * (s[1] - '0') + (s[2] - '0'), c = 10 * (s[4] - '0') + (s[5] - '0'), b = 10 * (s[7] - '0') + (s[8] - '0'), d = 10 * (s[10] - '0') + (s[11] - '0');
	return (uint64_t)(uint32_t)((uint32_t)a << 24 | (uint32_t)c << 16 | (uint32_t)b << 8 | (uint32_t)d);
}

__device__ __forceinline__ void copy_bytes(char *__restrict__ dst, const char *__restrict__ src, uint32_t n)
{
	uint32_t k = 0;
	for (; k + 4 <= n; k += 4) { uint32_t v; __builtin_memcpy(&v, src + k, 4); __builtin_memcpy(dst + k, &v, 4); }      // (words at any alignment)
	for (; k < n; ++k) dst[k] = src[k];
}

__global__ void __launch_bounds__(256)
ema_k_ing_gather(const char *__restrict__ text, uint32_t text_len, const uint32_t *__restrict__ nl, uint32_t n_nl, const uint32_t *__restrict__ order,
                 const Fields *__restrict__ fields, const uint64_t *__restrict__ codes_sorted, uint32_t n, int bc_len, int is_haplotag,
                 const uint32_t *__restrict__ off,
                 const uint32_t *__restrict__ id_off, char *__restrict__ bases, char *__restrict__ quals, char *__restrict__ ids, uint64_t *__restrict__ bc)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t line = order[i];
	uint32_t start, ln;
	line_of(nl, n_nl, text_len, line, start, ln);
	const char *s = text + start;
	const Fields f = fields[line];
	copy_bytes(bases + off[2 * i], s + f.r1_b, f.r1_l);
	copy_bytes(quals + off[2 * i], s + f.q1_b, f.r1_l);
	copy_bytes(bases + off[2 * i + 1], s + f.r2_b, f.r2_l);
	copy_bytes(quals + off[2 * i + 1], s + f.q2_b, f.r2_l);
	copy_bytes(ids + id_off[i], s + f.id_b, f.id_l);
	// two bits a base, first base lowest (encode_bc_default, src/util.c:41-61): the low bits of the sort code's digits
	if (is_haplotag) { bc[i] = encode_haplotag(s); return; }
	const uint64_t code = codes_sorted[i];
	uint64_t v = 0;
	for (int t = 0; t < bc_len; ++t) v = v << 2 | ((code >> (3 * t)) & 3u);
	bc[i] = v;
}

}  // namespace
#endif
