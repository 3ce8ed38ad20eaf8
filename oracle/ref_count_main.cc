// oracle/ref_count_main.cc -- TEST INFRASTRUCTURE.  The three lines that turn the reference's own `ema count`
// (/root/reference/cpp/count.cc + cpp/format.cc, compiled where they lie by oracle/Makefile, target `ref`) into a program:
//   oracle/_ref/ref_count WHITELIST PREFIX MAX_MAP_SIZE IS_HAPLOTAG < interleaved.fastq
// writes PREFIX.ema-fcnt and PREFIX.ema-ncnt exactly as `ema count` does (reference src/main.c:239 passes 1 GB).
// tests/test_count.py compares include/ema_count.h's product with it (and with tests/golden/count_vectors.json, which
// tests/golden/make_count_vectors.py wrote from it, where the reference tree is absent).
#include <cstdlib>
extern "C" void count(const char *known_barcodes_path, const char *output_prefix, const size_t max_map_size, const int is_haplotag);
int main(int argc, char **argv)
{
	if (argc < 5) return 2;
	count(argv[1], argv[2], (size_t)atoll(argv[3]), atoi(argv[4]));
	return 0;
}
