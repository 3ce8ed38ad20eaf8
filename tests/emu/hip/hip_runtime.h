// tests/emu/hip/hip_runtime.h -- stand-in used ONLY by the host-side SIMT harness in tests/emu.
// The product is compiled by hipcc against the real <hip/hip_runtime.h>.
#pragma once
#include "../simt_emu.h"
