// oracle/ref_preproc_main.cc -- TEST INFRASTRUCTURE.  main() for the reference's own `ema preproc`
// (/root/reference/cpp/correct.cc + cpp/format.cc, compiled where they lie by oracle/Makefile, target `ref`):
//   oracle/_ref/ref_preproc WHITELIST OUT_DIR DO_H2 BUFFER_SIZE DO_BX THREADS BUCKETS IS_HAPLOTAG X.ema-ncnt... < interleaved.fastq
// writes OUT_DIR/ema-bin-NNN and OUT_DIR/ema-nobc exactly as `ema preproc` does (reference src/main.c:201 passes a 10 MB buffer).
// tests/test_preproc.py compares include/ema_preproc.h's product with it (and with tests/golden/preproc_vectors.json, which
// tests/golden/make_preproc_vectors.py wrote from it, where the reference tree is absent).
#include <cstdlib>
extern "C" void correct(const char *known_barcodes_path, const char **input_prefix, const int input_prefix_size, const char *output_dir,
                        const char do_h2, const size_t buffer_size, const char do_bx_format, const int nthreads, const int nbuckets,
                        const int is_haplotag);
int main(int argc, char **argv)
{
	if (argc < 10) return 2;
	correct(argv[1], (const char **)(argv + 9), argc - 9, argv[2], (char)atoi(argv[3]), (size_t)atoll(argv[4]), (char)atoi(argv[5]), atoi(argv[6]),
	        atoi(argv[7]), atoi(argv[8]));
	return 0;
}
