// ABI bookkeeping for libtactilesr_hip.so (see include/tactilesr_hip.h).
#include "tsr_common.h"

extern "C" int tsr_abi_version(void) { return 14; }
