// Version probe of the C ABI (include/mmf.h).
#include "mmf_common.h"

extern "C" int mmf_version(void) { return MMF_ABI_VERSION; }
