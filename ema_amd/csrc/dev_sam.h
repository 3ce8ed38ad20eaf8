// ema_amd/csrc/dev_sam.h -- what the SAM formatter's kernels (k_sam.hip) are handed: device pointers to one bucket's text, the batch's
// CIGAR operations, the compact records of include/ema_sam.h and the strings the lines share.
#ifndef EMA_DEV_SAM_H
#define EMA_DEV_SAM_H

#include <stdint.h>
#include "ema_sam.h"

struct SamJob {
	const char *bases, *quals;      // ema_bucket: read r at [off[r], off[r+1])
	const uint32_t *off;
	const char *ids;                // names back to back, pair p at [id_off[p], id_off[p+1]), the first byte ('@') not printed
	const uint32_t *id_off;
	const uint64_t *bc;             // encoded barcode per pair
	const uint32_t *cigar;          // operation cigar_lo of the batch's array onwards
	const ema_sam_desc *desc;
	const ema_sam_xa *xa;
	const uint32_t *sel_at;
	const char *names;              // contig names back to back, contig i at [name_off[i], name_off[i+1])
	const uint32_t *name_off;
	const char *rg, *bx;            // RG identifier (rg_len bytes), bx_index (bx_len bytes)
	uint32_t n_lines;               // 2 x selected pairs
	uint32_t cigar_lo;
	int32_t has_rg, rg_len, bx_len, bc_len, is_haplotag, insert_min, insert_max;
};

// the prefix sum works in wave-sized chunks: line i's text starts at chunk_base[i / 64] + local[i]

void ema_launch_sam_len(const SamJob &j, uint32_t *local, uint32_t *chunk_tot, hipStream_t st);
void ema_launch_sam_tops(uint32_t n_chunks, const uint32_t *chunk_tot, uint64_t *chunk_base, uint64_t *total, hipStream_t st);
void ema_launch_sam_write(const SamJob &j, const uint32_t *local, const uint64_t *chunk_base, char *text, int *bad, hipStream_t st);

#endif
