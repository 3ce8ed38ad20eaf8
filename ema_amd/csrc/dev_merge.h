// ema_amd/csrc/dev_merge.h -- argument record of the batch-layout kernels (k_pack.hip, ema_launch_merge), shared with engine.hip.
#ifndef EMA_DEV_MERGE_H
#define EMA_DEV_MERGE_H
#include <stdint.h>
#include "ema_engine.h"

struct MergeParts {      // by value: the packed sets of up to 16 slices and, last, the full tier's
	const uint64_t *c_off[17], *g_off[17];
	const int *status[17];
	const ema_cand_t *cand[17];
	const uint32_t *cig[17];
	int first_read[17], n_reads[17];      // batch reads [first_read, first_read + n_reads) belong to slice k (not used for the last part)
	int n_parts;                          // slices + 1
	uint64_t cand_cap[17], cig_cap[17];   // capacities of each part's own packed arrays: a part whose pack overflowed is not read past them
};


#endif
