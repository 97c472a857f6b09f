// Error channel and version of the C ABI (include/p4p_hip.h).
#include "common.h"
#include "../../include/p4p_hip.h"
#include <string.h>

static thread_local char g_err[512] = "";

void p4p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* p4p_last_error(void) { return g_err; }
extern "C" int p4p_abi_version(void) { return P4P_ABI_VERSION; }

// ---- optional launch profiler: HIP events around exactly one kernel launch (bench.py's roofline) --------
// kind 1 = weight-gradient GEMM main kernel, kind 2 = gemm_tn main kernel, kind 3 = attention forward, kind 4 = attention
// backward (its two or three launches in one bracket).  Events sit on the launch stream.
#include <vector>
struct ProfRec { hipEvent_t a, b; double flops; };
static int g_prof_kind = 0;
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;

extern "C" int p4p_prof_enable(int kind) {
    g_prof_kind = kind;
    g_prof_used = 0;
    return 0;
}

bool p4p_prof_begin(int kind, double flops, hipStream_t st) {
    if (kind != g_prof_kind) return false;
    if (g_prof_used == g_prof.size()) {
        if (g_prof.size() >= 8192) return false;
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
        g_prof.push_back(r);
    }
    g_prof[g_prof_used].flops = flops;
    hipEventRecord(g_prof[g_prof_used].a, st);
    return true;
}

void p4p_prof_end(hipStream_t st) {
    hipEventRecord(g_prof[g_prof_used].b, st);
    ++g_prof_used;
}

// Synchronises on the recorded events; out[0] = total ms, out[1] = launches, out[2] = total FLOPs.
extern "C" int p4p_prof_collect(double* out) {
    double ms = 0.0, fl = 0.0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        float t = 0.f;
        P4P_HIP(hipEventSynchronize(g_prof[i].b));
        P4P_HIP(hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b));
        ms += t;
        fl += g_prof[i].flops;
    }
    out[0] = ms; out[1] = (double)g_prof_used; out[2] = fl;
    g_prof_kind = 0;
    g_prof_used = 0;
    return 0;
}

// ---- CU-masked streams: a stream whose kernels may only occupy the compute units whose bit is set in `mask`
// (hipExtStreamCreateWithCUMask).  Two such streams with disjoint masks PARTITION the chip: a memory-bound kernel on one
// and an MFMA-bound kernel on the other then really overlap (two unmasked streams only time-slice whole CUs, because a
// 512-thread / 256-VGPR workgroup fills a CU).  words = number of 32-bit mask words (8 for 256 CUs).
extern "C" int p4p_stream_create_cu_mask(const uint32_t* mask, int words, void** out) {
    P4P_CHECK(mask && out && words >= 1, "p4p_stream_create_cu_mask: bad arguments");
    hipStream_t st;
    P4P_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
    *out = (void*)st;
    return 0;
}
extern "C" int p4p_stream_destroy(void* stream) {
    P4P_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
extern "C" int p4p_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}
