/* include/bwa_compat/bwa/bwa.h -- for building the unmodified reference against libema_bwaabi.so:
 * the reference includes "bwa/bwa.h" (include/bwabridge.h:9-12); every declaration it uses is in ema_bwaabi.h. */
#include "../../ema_bwaabi.h"
