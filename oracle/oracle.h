/*
 * oracle/oracle.h -- CPU oracle for the `ema align` seed-and-extend hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path
 * (ema_amd/) never links, imports or calls anything in this directory.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the un-vendored git
 * submodule lh3/bwa (reference .gitmodules:2-4; pinned commit unknown,
 * >= 0.7.11 by the signatures at src/bwabridge.c:13-17, plausibly 0.7.17).
 * Its source is absent from /root/reference, the reference ships no tests,
 * fixtures or golden vectors for this path, and the reference binary cannot be
 * built here.  This is therefore a restatement of bwa-mem's published
 * algorithm (Li 2013, arXiv:1303.3997, and the behaviour of bwa 0.7.x:
 * bwt.c, bwamem.c, bwamem_pair.c, ksw.c, bwa.c, bntseq.c, ksort.h, kbtree.h),
 * anchored on the reference's own call sites:
 *   src/bwabridge.c:204-299  bwa_mem_mate_sw      -> orc_mate_sw()
 *   src/bwabridge.c:301-311  bwa_smith_waterman   -> orc_reg2aln()
 *   src/bwabridge.c:313-339  interpret_align      -> (fields kept raw in orc_reg_t)
 *   src/align.c:180-186      bwa_init             -> orc_idx_load(), orc_opt_init()
 *   src/align.c:986-1061     append_alignments    -> orc_align_pair() (candidate part), orc_append_alignments() (filters, mapq, scores)
 * The later parts restate code that IS in the reference tree -- oracle/ingest.c (read_special_fastq and its helpers),
 * oracle/clouds.c (clouds, EM, duplicates), oracle/sam.c (print_sam_record) -- and since round 3 they are PINNED: every
 * unmodified reference source (src/ *.c, cpp/ *.cc) compiles against include/bwa_compat/ and links over this oracle's nine libbwa
 * symbols (oracle/bwaface.c -> oracle/_ref/ema_refhost, `make ref`, build container only); the SAM that binary writes is
 * committed under tests/golden/sam/ with the script that made it, and these restatements reproduce it byte for byte
 * (tests/test_golden_sam.py) -- except `-d`'s random draws and the `-1/-2` FASTQ reader, which only the product restates.
 * The bwa half stays unpinned: self-consistency is checked in tests/ against brute-force models (suffix array search,
 * exhaustive DP re-scoring); nothing here is checked against real bwa output (tools/diff_vs_bwa.sh is the way out for
 * whoever has a bwa checkout).
 */
#ifndef EMA_ORACLE_H
#define EMA_ORACLE_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- options: bwa's mem_opt_t subset used on this path (mem_opt_init(),
 * then max_occ = 3000 at reference src/align.c:185) ---- */
typedef struct {
	int a, b, o_del, e_del, o_ins, e_ins;
	int pen_unpaired, pen_clip5, pen_clip3;
	int w, zdrop;
	uint64_t max_mem_intv;
	int T, flag, min_seed_len, min_chain_weight, max_chain_extend;
	float split_factor;
	int split_width, max_occ, max_chain_gap;
	int n_threads, chunk_size;
	float mask_level, drop_ratio, XA_drop_ratio, mask_level_redun;
	float mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins, max_matesw, max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} orc_opt_t;

/* ---- index: bwa on-disk layout (.bwt .sa .pac .ann .amb) ---- */
typedef struct {
	int64_t offset;
	int32_t len, n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} orc_ann_t;

typedef struct {
	uint64_t primary, L2[5], seq_len, bwt_size;
	uint32_t *bwt;            /* 64-byte blocks: 4 x u64 counts + 128 bases */
	int sa_intv;
	uint64_t n_sa, *sa;
	int64_t l_pac;
	int32_t n_seqs;
	orc_ann_t *anns;
	uint8_t *pac;
} orc_idx_t;

/* bwa's mem_alnreg_t */
typedef struct {
	int64_t rb, re;
	int qb, qe;
	int rid;
	int score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0;
	int n_comp, is_alt;
	float frac_rep;
	uint64_t hash;
} orc_reg_t;

typedef struct { size_t n, m; orc_reg_t *a; } orc_reg_v;

/* bwa's mem_aln_t (the part the reference reads, src/bwabridge.c:359-379) */
typedef struct {
	int64_t pos;
	int rid, flag;
	int is_rev, is_alt, mapq, NM;
	int n_cigar;
	uint32_t *cigar;
	int score, sub, alt_sc;
} orc_aln_t;

typedef struct { uint64_t x[3], info; } orc_intv_t;
typedef struct { size_t n, m; orc_intv_t *a; } orc_intv_v;

typedef struct { int64_t rbeg; int32_t qbeg, len; int score; } orc_seed_t;
typedef struct {
	int n, m, first, rid;
	uint32_t w, kept, is_alt;
	float frac_rep;
	int64_t pos;
	orc_seed_t *seeds;
} orc_chain_t;
typedef struct { size_t n, m; orc_chain_t *a; } orc_chain_v;

/* instrumentation for the roofline's algorithmic byte count */
typedef struct {
	uint64_t n_ext;     /* bwt_extend calls (2 x 64-B block reads each) */
	uint64_t n_lf;      /* LF steps inside bwt_sa */
	uint64_t n_occ;     /* located occurrences (bwt_sa calls) */
	uint64_t w_ref;     /* reference bases fetched (chain windows, rescue, global) */
	uint64_t n_regs;    /* regions returned */
	uint64_t n_cigar;   /* cigar ops returned */
	uint64_t l_read;    /* read bases */
	uint64_t cells_ext, cells_local, cells_global; /* DP cells */
	uint64_t rows_ext, rows_local, rows_global, n_ext_calls, n_local_calls, n_global_calls; /* DP rows / calls */
} orc_stats_t;
extern __thread orc_stats_t orc_stats;   /* per thread */
void orc_stats_get(orc_stats_t *out);
void orc_stats_reset(void);

/* CPU baseline: aligns n_pairs pairs (ASCII reads, 2n+1 offsets) with n_threads OpenMP threads and returns the
 * wall seconds; *n_cand receives the number of candidates produced (keeps the work observable). */
double orc_bench_pairs(const orc_opt_t *opt, const orc_idx_t *idx, const char *bases, const uint32_t *off, size_t n_pairs,
                       int n_threads, uint64_t *n_cand);

void orc_opt_init(orc_opt_t *o);                 /* mem_opt_init() + max_occ=3000 */
orc_idx_t *orc_idx_load(const char *prefix);
void orc_idx_destroy(orc_idx_t *idx);

/* FM primitives (bwt.c) */
void orc_occ4(const orc_idx_t *b, uint64_t k, uint64_t cnt[4]);
uint64_t orc_occ(const orc_idx_t *b, uint64_t k, int c);
void orc_extend(const orc_idx_t *b, const orc_intv_t *ik, orc_intv_t ok[4], int is_back);
uint64_t orc_sa(const orc_idx_t *b, uint64_t k);
int orc_smem1a(const orc_idx_t *b, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv,
               orc_intv_v *mem, orc_intv_v *tmp0, orc_intv_v *tmp1);
int orc_seed_strategy1(const orc_idx_t *b, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem);
void orc_collect_intv(const orc_opt_t *opt, const orc_idx_t *b, int len, const uint8_t *seq, orc_intv_v *mem);

/* bntseq.c */
int orc_pos2rid(const orc_idx_t *idx, int64_t pos_f);
int orc_intv2rid(const orc_idx_t *idx, int64_t rb, int64_t re);
uint8_t *orc_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len);
uint8_t *orc_fetch_seq(const orc_idx_t *idx, int64_t *beg, int64_t mid, int64_t *end, int *rid);

/* chaining (bwamem.c) */
orc_chain_v orc_chain(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq);
int orc_chain_flt(const orc_opt_t *opt, int n_chn, orc_chain_t *a);

/* DP (ksw.c) */
int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                    int *qle, int *tle, int *gtle, int *gscore, int *max_off);
int orc_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t **cigar);
typedef struct { int score, te, qe, score2, te2, tb, qb; } orc_kswr_t;
#define ORC_KSW_XBYTE  0x10000
#define ORC_KSW_XSTOP  0x20000
#define ORC_KSW_XSUBO  0x40000
#define ORC_KSW_XSTART 0x80000
orc_kswr_t orc_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins, int xtra);

/* extension + post-processing (bwamem.c) */
void orc_chain2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query,
                   const orc_chain_t *c, orc_reg_v *av);
int orc_sort_dedup_patch(const orc_opt_t *opt, const orc_idx_t *idx, uint8_t *query, int n, orc_reg_t *a);
orc_reg_v orc_align1_core(const orc_opt_t *opt, const orc_idx_t *idx, int l_seq, uint8_t *seq);

/* pairing + final alignment (bwamem_pair.c, bwamem.c, bwa.c) */
typedef struct { int low, high, failed; double avg, std; } orc_pestat_t;
int orc_matesw(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], const orc_reg_t *a,
               int l_ms, const uint8_t *ms, orc_reg_v *ma);
uint32_t *orc_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac,
                         const uint8_t *pac, int l_query, uint8_t *query, int64_t rb, int64_t re,
                         int *score, int *n_cigar, int *NM);
orc_aln_t orc_reg2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query, const orc_reg_t *ar);

/* the reference's bridge (src/bwabridge.c:204-299): both mates, rescue both ways */
void orc_mate_sw(const orc_opt_t *opt, const orc_idx_t *idx, const char *read1, int len1, const char *read2, int len2,
                 int score_delta, orc_reg_v *r1, orc_reg_v *r2);

/* Flat per-pair result used by the parity tests: all regions of both mates
 * after rescue, each with its final alignment (reference src/align.c:1010-1013,
 * 1035-1038 call bwa_smith_waterman for every region). */
typedef struct {
	orc_reg_t reg;
	int64_t pos;
	int is_rev, NM, n_cigar;
	uint32_t cigar_off;       /* into the pool */
	int aln_score, aln_sub;
} orc_cand_t;

typedef struct {
	size_t n1, n2;            /* candidates of mate 1 / mate 2 */
	orc_cand_t *c;            /* n1 + n2 entries */
	size_t n_pool;
	uint32_t *pool;
} orc_pair_out_t;

void orc_align_pair(const orc_opt_t *opt, const orc_idx_t *idx, const char *read1, int len1,
                    const char *read2, int len2, orc_pair_out_t *out);
void orc_pair_out_free(orc_pair_out_t *out);
/* per-read digests of the candidate lists of a whole batch (2 * n_pairs words), n_threads OpenMP threads; returns seconds */
double orc_digest_pairs(const orc_opt_t *opt, const orc_idx_t *idx, const char *bases, const uint32_t *off, size_t n_pairs,
                        int n_threads, uint64_t *digest);

/* the host stage behind the bridge calls (reference src/align.c:846-911, 959-1061): filters, mapq, likelihoods */
int orc_append_alignments(const orc_opt_t *opt, const orc_pair_out_t *p, int len1, int len2, double error_rate, int *which,
                          int *clip_, int *dist_, int *mapq_, int *score_mapq_, int *unique_, double *score_);

extern const unsigned char orc_nt4_table[256];

/* klib ks_introsort order for keys; used by tests to pin the sort restatement */
void orc_introsort_u64(size_t n, uint64_t *a);

/* ---- bucket reader (oracle/ingest.c; reference src/align.c:751-843, src/util.c:11-21,41-95) ---- */
#define ORC_MAX_READ_LEN 255      /* the reference's MAX_READ_LEN is 200 (include/align.h:61); the engine takes 255 */
typedef struct {                  /* FASTQRecord, reference include/samrecord.h:9-15 */
	uint64_t bc;
	unsigned short rlen;
	char id[150];
	char read[ORC_MAX_READ_LEN + 2];
	char qual[ORC_MAX_READ_LEN + 2];
} orc_fastq_rec_t;

void orc_copy_until_space(char *dest, char **src);
uint64_t orc_encode_bc(const char *bc, int bc_len, int is_haplotag);
void orc_decode_bc(uint64_t bc, int bc_len, int is_haplotag, char *out);
/* *r1, *r2: n + 1 records each (the last one the sentinel), freed by the caller with free() */
int orc_read_special_fastq(const char *path, int bc_len, int is_haplotag, orc_fastq_rec_t **r1, orc_fastq_rec_t **r2, size_t *n);
size_t orc_next_group(const orc_fastq_rec_t *recs, size_t at);

/* ---- SAM record formatter (oracle/sam.c; reference src/samrecord.c:75-284, src/align.c:27-40).  The structs have the
 * layout of include/ema_sam.h's, so that a test can hand the same arrays to both sides. ---- */
typedef struct { const char *chrom; uint32_t pos; int32_t edit_dist, rev, n_cigar; const uint32_t *cigar; } orc_sam_alt_t;
typedef struct {
	const char *ident, *chrom;
	uint32_t chrom_id, pos;
	int32_t mapq, score_mapq;
	double gamma;
	uint8_t mate, rev, duplicate, pad_;
	int32_t cloud_id, cloud_bad;
	uint64_t bc;
	const char *read, *qual;
	int32_t read_len, mate_read_len;
	const char *mate_read, *mate_qual;
	int64_t aln_pos;
	int32_t aln_rev, edit_dist, n_cigar, pad2_;
	const uint32_t *cigar;
	const orc_sam_alt_t *alts;
	size_t n_alts;
} orc_sam_rec_t;
typedef struct { const orc_sam_rec_t *rec, *mate; } orc_sam_line_t;
typedef struct { const char *rg_id, *bx_index; int32_t bc_len, is_haplotag, insert_min, insert_max; } orc_sam_opts_t;
int orc_sam_format(const orc_sam_line_t *lines, size_t n, const orc_sam_opts_t *o, char **text, size_t *n_bytes);
int orc_sam_header(const char *const *names, const int32_t *lens, int32_t n, const char *rg, const char *version, int pg_argc,
                   const char *const *pg_argv, char **text, size_t *n_bytes);

/* ---- cloud / EM / duplicate stage (oracle/clouds.c; reference src/align.c:347-608, src/samdict.c) ---- */
typedef struct orc_crec {         /* the fields of SAMRecord (reference include/samrecord.h:21-56) this stage reads and writes */
	uint64_t bc;
	uint32_t chrom, pos;          /* chrom index; 1-based position (alignment_to_sam_rec, src/align.c:922-923) */
	char ident[256];
	double score;                 /* alignment log-likelihood (score_alignment) */
	uint32_t mate, rev;
	uint32_t orig;                /* index in append_alignments' order */
	uint32_t hash, mate_hash;
	uint8_t hashed, mate_hashed, active, duplicate, visited, pad_[3];
	/* results */
	double gamma;
	int32_t cloud_id, cloud_bad, alt;      /* alt: `orig` of the record the XA entry is copied from, or -1 */
	int32_t clip_edit_dist;       /* input, read by -d only (SAMRecord.clip_edit_dist, src/align.c:938); sits in what was padding */
	struct orc_crec *sel_mate;
} orc_crec_t;
/* -d (reference src/split.c, called at src/align.c:396-397): off by default.  probs: the platform's read-density model
 * (src/techs.c: density_probs, n of them).  The optimiser draws from libc's rand(), which the reference seeds from time() once per
 * process; orc_clouds_reseed(seed) is srand(seed). */
void orc_clouds_set_density(int apply_opt, int n_probs, const double *probs);
void orc_clouds_reseed(unsigned seed);
size_t orc_clouds_group(orc_crec_t *recs, size_t n, size_t n_pairs, uint32_t dist_thresh, int many_clouds, int *cloud_id, int *order);

#ifdef __cplusplus
}
#endif
#endif
