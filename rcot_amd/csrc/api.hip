#include "../../include/rcot_hip.h"
extern "C" int rcot_abi_version(void) { return 2; }
