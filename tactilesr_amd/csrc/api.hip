// ABI bookkeeping for libtactilesr_hip.so (see include/tactilesr_hip.h).
#include "tsr_common.h"

#ifndef TSR_VARIANT_BUILD
#define TSR_VARIANT_BUILD 0
#endif

extern "C" int tsr_abi_version(void) { return 23; }

// 0 for the shipped library (`python -m tactilesr_amd.build`).  tools/build_variant.py (kernel A/B experiments, built with
// extra -D flags) always defines TSR_VARIANT_BUILD=1: such a library reports itself here, `_lib.load()` refuses it unless
// TSR_ALLOW_VARIANT=1 and bench.py writes the flag into its JSON line, so an experiment's numbers cannot pass for results.
extern "C" int tsr_build_flags(void) { return TSR_VARIANT_BUILD; }
