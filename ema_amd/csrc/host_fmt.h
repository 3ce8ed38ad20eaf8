// ema_amd/csrc/host_fmt.h -- number formatting shared by the host formatter (host_sam.cpp) and the cloud stage (host_clouds.cpp).
#ifndef EMA_HOST_FMT_H
#define EMA_HOST_FMT_H
#include <cmath>
#include <cstdint>
#include <cstring>

// "%.5g" of a double in [1e-10, 1] without the library call: the decimal expansion of mant * 2^e2 is exact in 128-bit integers
// there (53 + 47 bits), so the five significant digits are rounded half-to-even on the true value, as glibc's printf does.
// Returns the length, 0: outside the range (the caller falls back on snprintf).
static inline int ema_fmt_g5(double g, char *out)
{
	if (!(g >= 1e-10 && g <= 1.0)) return 0;
	uint64_t bits;
	memcpy(&bits, &g, 8);
	const uint64_t mant = (bits & ((1ull << 52) - 1)) | (1ull << 52);
	const int s = 1075 - (int)((bits >> 52) & 0x7ff);      // g = mant / 2^s, 52 <= s <= 86
	static const uint64_t p10[16] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
	                                 10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull, 1000000000000000ull};
	int X = (int)floor(log10(g));      // may be one off next to a power of ten: corrected by the digit count
	if (X > 0) X = 0;
	if (X < -10) X = -10;
	unsigned __int128 q, rem, half = (unsigned __int128)1 << (s - 1);
	for (int tries = 0;; ++tries) {
		const unsigned __int128 N = (unsigned __int128)mant * p10[4 - X];
		q = N >> s; rem = N & (((unsigned __int128)1 << s) - 1);
		if (q < 10000 && X > -11 && tries < 3) { --X; if (4 - X > 15) return 0; continue; }
		if (q >= 100000 && X < 0 && tries < 3) { ++X; continue; }
		break;
	}
	if (q < 10000 || q >= 100000) return 0;
	uint32_t d = (uint32_t)q;
	if (rem > half || (rem == half && (d & 1))) ++d;
	if (d == 100000) { d = 10000; ++X; }
	char dig[5];
	for (int i = 4; i >= 0; --i) { dig[i] = (char)('0' + d % 10); d /= 10; }
	int nd = 5;
	while (nd > 1 && dig[nd - 1] == '0') --nd;      // %g drops trailing zeros
	char *p = out;
	if (X >= -4) {      // fixed notation
		if (X == 0) { *p++ = dig[0]; if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; ++i) *p++ = dig[i]; } }
		else if (X > 0) return 0;
		else { *p++ = '0'; *p++ = '.'; for (int i = 0; i < -X - 1; ++i) *p++ = '0'; for (int i = 0; i < nd; ++i) *p++ = dig[i]; }
	} else {
		*p++ = dig[0];
		if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; ++i) *p++ = dig[i]; }
		*p++ = 'e'; *p++ = '-';
		const int ax = -X;
		*p++ = (char)('0' + ax / 10); *p++ = (char)('0' + ax % 10);
	}
	return (int)(p - out);
}
#endif
