// ema_amd/csrc/dev_bucket.h -- a bucket's arrays in device memory (ema_bucket.dev, made by ema_bucket_read_device, ingest_dev.hip):
// what the engine's staging (ema_engine_stage_async_dev) and the SAM formatter (sam_dev.hip) take instead of the host arrays.
#ifndef EMA_DEV_BUCKET_H
#define EMA_DEV_BUCKET_H
#include <stddef.h>
#include <stdint.h>
#include "ema_ingest.h"

typedef struct ema_bucket_dev {
	int device;
	const char *bases, *quals, *ids;      /* as ema_bucket's, in device memory */
	const uint32_t *off, *id_off;
	const uint64_t *bc;
	size_t n_pairs, n_bases, n_ids;
} ema_bucket_dev;

#ifdef __cplusplus
extern "C" {
#endif
const ema_bucket_dev *ema_bucket_dev_view(const ema_bucket *bk);      /* NULL: the bucket lives on the host only */
void ema_bucket_dev_release(void *dev);                               /* ema_bucket_free's part */
#ifdef __cplusplus
}
#endif
#endif
