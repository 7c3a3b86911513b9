#include <cxxabi.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "common.h"
#include "../../include/rcot_hip.h"

namespace {
thread_local char g_kernel_name[192] = "";
thread_local unsigned g_kernel_seq = 0;

struct ProfRec { const void* fn; hipEvent_t e0, e1; };
thread_local bool g_prof = false;
thread_local std::vector<ProfRec> g_recs;
thread_local std::vector<hipEvent_t> g_pool;       // events are created once and reused by later collections
thread_local size_t g_pool_used = 0;
}  // namespace

namespace rcot {
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel_name, sizeof(g_kernel_name), fmt, ap);
    va_end(ap);
    ++g_kernel_seq;
}
bool prof_on() { return g_prof; }
void prof_slot(const void* fn, hipEvent_t* e0, hipEvent_t* e1) {
    while (g_pool.size() < g_pool_used + 2) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        g_pool.push_back(e);
    }
    *e0 = g_pool[g_pool_used++];
    *e1 = g_pool[g_pool_used++];
    g_recs.push_back(ProfRec{fn, *e0, *e1});
}
}  // namespace rcot

extern "C" int rcot_profile_begin(void) {
    g_recs.clear();
    g_pool_used = 0;
    g_prof = true;
    return RCOT_OK;
}

// The caller has synchronised the device.  Writes "kernel symbol|launches|total ms" lines, largest total first, into out[0..n) and
// returns the number of launches collected (or RCOT_EINVAL).
extern "C" int rcot_profile_end(char* out, int n) {
    g_prof = false;
    if (!out || n <= 0) return RCOT_EINVAL;
    struct Agg { long calls = 0; double ms = 0.0; };
    std::map<const void*, Agg> agg;
    long dropped = 0;
    for (const ProfRec& r : g_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void)hipGetLastError(); ++dropped; continue; }
        Agg& a = agg[r.fn];
        a.calls += 1;
        a.ms += ms;
    }
    std::vector<std::pair<const void*, Agg>> rows(agg.begin(), agg.end());
    std::sort(rows.begin(), rows.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
    std::string text;
    for (const auto& kv : rows) {
        const char* nm = hipKernelNameRefByPtr(kv.first, nullptr);
        int st = 0;
        char* dm = nm ? abi::__cxa_demangle(nm, nullptr, nullptr, &st) : nullptr;      // (the runtime hands out the mangled symbol)
        char line[1024];
        snprintf(line, sizeof(line), "%.900s|%ld|%.6f\n", (dm && st == 0) ? dm : (nm ? nm : "?"), kv.second.calls, kv.second.ms);
        free(dm);
        text += line;
    }
    if (dropped) {
        char line[96];
        snprintf(line, sizeof(line), "#launches whose events could not be read|%ld|0\n", dropped);
        text += line;
    }
    strncpy(out, text.c_str(), (size_t)n - 1);
    out[n - 1] = 0;
    const int launches = (int)g_recs.size();
    g_recs.clear();
    return launches;
}

extern "C" int rcot_abi_version(void) { return RCOT_ABI_VERSION; }

extern "C" int rcot_last_kernel(char* out, int n) {
    if (out && n > 0) {
        strncpy(out, g_kernel_name, (size_t)n - 1);
        out[n - 1] = 0;
    }
    return (int)g_kernel_seq;
}
