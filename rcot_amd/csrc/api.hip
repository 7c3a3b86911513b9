#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "common.h"
#include "../../include/rcot_hip.h"

namespace {
thread_local char g_kernel_name[192] = "";
thread_local unsigned g_kernel_seq = 0;
}  // namespace

namespace rcot {
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel_name, sizeof(g_kernel_name), fmt, ap);
    va_end(ap);
    ++g_kernel_seq;
}
}  // namespace rcot

extern "C" int rcot_abi_version(void) { return RCOT_ABI_VERSION; }

extern "C" int rcot_last_kernel(char* out, int n) {
    if (out && n > 0) {
        strncpy(out, g_kernel_name, (size_t)n - 1);
        out[n - 1] = 0;
    }
    return (int)g_kernel_seq;
}
