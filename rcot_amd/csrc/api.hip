#include "../../include/rcot_hip.h"
extern "C" int rcot_abi_version(void) { return RCOT_ABI_VERSION; }
