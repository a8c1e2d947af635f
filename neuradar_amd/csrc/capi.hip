// ABI bookkeeping entry points of libneuradar_hip.so (see include/neuradar_hip.h).
#include "nr_common.h"

extern "C" int nr_abi_version(void) { return NR_ABI_VERSION; }
extern "C" const char* nr_target_arch(void) { return "gfx950"; }
