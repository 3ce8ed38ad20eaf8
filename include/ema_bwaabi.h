/* include/ema_bwaabi.h -- the libbwa link surface of EMA, as exported by libema_bwaabi.so on top of the MI355X engine.
 *
 * The reference links exactly nine symbols from -lbwa (reference src/bwabridge.c:13-17,79,155; src/align.c:184,190;
 * `nm -u` of its objects):
 *
 *     bwa_idx_load  bwa_idx_destroy  mem_opt_init  mem_align1_core  mem_chain  mem_matesw  mem_reg2aln  bns_fetch_seq
 *     nst_nt4_table (data)
 *
 * and reads a handful of struct fields directly (mem_opt_t.{max_occ,a,b,min_seed_len,mapQ_coef_len,mapQ_coef_fac},
 * mem_alnreg_t, mem_aln_t through the cast at src/bwabridge.c:159-168, bntseq_t.{l_pac,n_seqs,anns[]}, bwaidx_t.{bwt,bns,pac},
 * mem_pestat_t) -- so the struct LAYOUTS below are part of the ABI.  They restate the public declarations of lh3/bwa 0.7.x
 * (bwt.h, bntseq.h, bwa.h, bwamem.h; the submodule is absent from the reference tree): a header
 * this repository owns, not a copy of bwa's.  A build of the reference that puts this file where it expects
 * "bwa/bwamem.h" etc. and links -lema_bwaabi instead of -lbwa runs its unmodified bridge on the GPU, one call at a time
 * (the batched boundary of include/ema_engine.h is the fast path).  The same header lets
 * tools/bwa_dump.c be compiled once against this library and once against a real bwa checkout (tools/diff_vs_bwa.sh): the
 * differential tester that can pin the oracle when bwa's source is available.
 *
 * Every function goes through the engine's C ABI (include/ema_engine.h); nothing here computes an alignment on the CPU and
 * nothing links oracle/.  Calls are serialised per index (the reference calls them from OpenMP threads).
 */
#ifndef EMA_BWAABI_H
#define EMA_BWAABI_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <assert.h>      /* bwa's bntseq.h brings it in, and reference src/techs.c:8,21,41 relies on that */

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t bwtint_t;

/* bwt.h */
typedef struct {
	bwtint_t primary;
	bwtint_t L2[5];
	bwtint_t seq_len;
	bwtint_t bwt_size;
	uint32_t *bwt;          /* not populated: the rank structure lives in HBM */
	uint32_t cnt_table[256];
	int sa_intv;
	bwtint_t n_sa;
	bwtint_t *sa;           /* not populated */
} bwt_t;

/* bntseq.h */
typedef struct {
	int64_t offset;
	int32_t len;
	int32_t n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} bntann1_t;

typedef struct {
	int64_t offset;
	int32_t len;
	char amb;
} bntamb1_t;

typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	bntann1_t *anns;
	int32_t n_holes;
	bntamb1_t *ambs;
	FILE *fp_pac;
} bntseq_t;

extern unsigned char nst_nt4_table[256];

/* bwa.h */
#define BWA_IDX_BWT 0x1
#define BWA_IDX_BNS 0x2
#define BWA_IDX_PAC 0x4
#define BWA_IDX_ALL 0x7

typedef struct {
	bwt_t *bwt;
	bntseq_t *bns;
	uint8_t *pac;
	int is_shm;
	int64_t l_mem;
	uint8_t *mem;
} bwaidx_t;

/* bwamem.h */
#define MEM_MAPQ_COEF 30.0
#define MEM_MAPQ_MAX 60
#define MEM_F_PE 0x2
#define MEM_F_NOPAIRING 0x4
#define MEM_F_ALL 0x8
#define MEM_F_NO_MULTI 0x10
#define MEM_F_NO_RESCUE 0x20

typedef struct {
	int a, b;
	int o_del, e_del;
	int o_ins, e_ins;
	int pen_unpaired;
	int pen_clip5, pen_clip3;
	int w;
	int zdrop;
	uint64_t max_mem_intv;
	int T;
	int flag;
	int min_seed_len;
	int min_chain_weight;
	int max_chain_extend;
	float split_factor;
	int split_width;
	int max_occ;
	int max_chain_gap;
	int n_threads;
	int chunk_size;
	float mask_level;
	float drop_ratio;
	float XA_drop_ratio;
	float mask_level_redun;
	float mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins;
	int max_matesw;
	int max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} mem_opt_t;

typedef struct {
	int64_t rb, re;
	int qb, qe;
	int rid;
	int score;
	int truesc;
	int sub;
	int alt_sc;
	int csub;
	int sub_n;
	int w;
	int seedcov;
	int secondary;
	int secondary_all;
	int seedlen0;
	int n_comp:30, is_alt:2;
	float frac_rep;
	uint64_t hash;
} mem_alnreg_t;

typedef struct { size_t n, m; mem_alnreg_t *a; } mem_alnreg_v;

typedef struct {
	int low, high;
	int failed;
	double avg, std;
} mem_pestat_t;

typedef struct {
	int64_t pos;
	int rid;
	int flag;
	uint32_t is_rev:1, is_alt:1, mapq:8, NM:22;
	int n_cigar;
	uint32_t *cigar;
	char *XA;
	int score, sub, alt_sc;
} mem_aln_t;

/* The chain types are private to bwa's bwamem.c; the reference restates them itself (include/bwabridge.h:25-41, after it has
 * included the four bwa headers).  The shims under include/bwa_compat/ therefore define EMA_BWAABI_REFERENCE_BUILD, which
 * leaves them -- and the prototype of mem_chain, which the reference declares at src/bwabridge.c:14 -- to the includer:
 * tests/test_refhost_build.py compiles every unmodified reference source this way. */
#ifndef EMA_BWAABI_REFERENCE_BUILD
typedef struct {
	int64_t rbeg;
	int32_t qbeg, len;
	int score;
} mem_seed_t;

typedef struct {
	int n, m, first, rid;
	uint32_t w:29, kept:2, is_alt:1;
	float frac_rep;
	int64_t pos;
	mem_seed_t *seeds;
} mem_chain_t;

typedef struct { size_t n, m; mem_chain_t *a; } mem_chain_v;
#endif

/* --- the nine link symbols (nst_nt4_table above) --- */

/* Opens the engine on the GPU named by EMA_DEVICE (default 0) with <hint>.{bwt,fsa,pac,ann} (the index layout
 * ema_index_build writes); NULL on failure, with a message on stderr.  reference src/bwabridge.c:79 */
bwaidx_t *bwa_idx_load(const char *hint, int which);
void bwa_idx_destroy(bwaidx_t *idx);

/* malloc'd options with bwa's defaults (max_occ = 500; the reference then sets 3000, src/align.c:184-185) */
mem_opt_t *mem_opt_init(void);

/* Regions of one read: seeding, chaining, extension, dedup/patch (K1 + K2).  seq: l_seq nt4 codes or ASCII, converted to
 * nt4 in place as bwa does; buf is ignored.  Result array is free()-able.  reference src/bwabridge.c:173,236,237 */
mem_alnreg_v mem_align1_core(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int l_seq, char *seq, void *buf);

/* Referenced by the reference's dead bridge entry points only (src/bwabridge.c:122,192; no caller in src/): exported so
 * that the link succeeds; returns an empty vector. */
#ifndef EMA_BWAABI_REFERENCE_BUILD
mem_chain_v mem_chain(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, int len, const uint8_t *seq, void *buf);
#endif

/* One rescue attempt of the mate ms (nt4) around region a; ma is updated in place (realloc).  Supports the insert model the
 * reference passes (src/bwabridge.c:216-227): only pes[1] (FR) not failed; any other returns 0 without aligning.
 * Returns the number of alignments run.  reference src/bwabridge.c:267,281 */
int mem_matesw(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], const mem_alnreg_t *a,
               int l_ms, const uint8_t *ms, mem_alnreg_v *ma);

/* Final alignment of one region: position, strand, CIGAR (malloc'd, BAM packing), NM, mapq (bwa's mem_approx_mapq_se).
 * seq: ASCII or nt4.  XA is NULL.  reference src/bwabridge.c:304 */
mem_aln_t mem_reg2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_seq, const char *seq, const mem_alnreg_t *ar);

/* Reference bases [*beg, *end) in forward-reverse coordinates, clamped to the contig holding mid (host copy of .pac);
 * malloc'd nt4 codes.  reference src/bwabridge.c:17 (declared; its callers are dead code) */
uint8_t *bns_fetch_seq(const bntseq_t *bns, const uint8_t *pac, int64_t *beg, int64_t mid, int64_t *end, int *rid);

/* not part of bwa: sizes of the ABI structs as this library was compiled, for layout checks from other languages
 * (0 mem_opt_t, 1 mem_alnreg_t, 2 mem_aln_t, 3 mem_pestat_t, 4 bntann1_t, 5 bntseq_t, 6 bwaidx_t, 7 bwt_t, 8 mem_chain_t) */
size_t ema_bwaabi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif
