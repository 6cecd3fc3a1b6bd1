// capi.hip -- library identification and shared helpers of the C ABI (include/orcai_hip.h).
#include <hip/hip_runtime.h>

#include "orcai_hip.h"

extern "C" const char* orcai_version(void) { return "orcai_hip 0.1.0 gfx950"; }
