/* include/bwa_compat/bwa/bwamem.h -- for building the unmodified reference against libema_bwaabi.so:
 * the reference includes "bwa/bwamem.h" (include/bwabridge.h:9-12); every declaration it uses is in ema_bwaabi.h. */
#ifndef EMA_BWAABI_REFERENCE_BUILD
#define EMA_BWAABI_REFERENCE_BUILD 1      /* the reference owns mem_seed_t / mem_chain_t / mem_chain_v (include/bwabridge.h:25-41) */
#endif
#include "../../ema_bwaabi.h"
