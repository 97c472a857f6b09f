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
// ---- persistent kernels and the compute units other work needs -----------------------------------------------------
// The persistent tile loop of gemm_tn4p.h and the walking attention backward kernels (attention_fused.h) launch ONE workgroup
// per CU, each with a static share of the work and all of the CU's LDS.  A concurrent kernel that holds r CUs for a long time
// (RCCL's all-reduce workgroups during a data-parallel backward) makes r of those workgroups start only when others have
// finished -- with their whole share still to do, the launch takes up to twice as long.  p4p_set_reserved_cus(r):
//   r = 0 (default, world size 1): grid = all CUs;   r > 0: grid = CUs - r (rounded up to a multiple of 8 so that a workgroup's
//   tiles stay on one XCD), the tile loops stride by the grid size;   r < 0: no persistent kernels at all (a workgroup per tile /
//   pair: the hardware dispatcher balances whatever CUs are free).  ddp.build_reducer sets it when world > 1.
static int g_reserved_cus = 0;
extern "C" int p4p_set_reserved_cus(int r) {
    int o = g_reserved_cus;
    g_reserved_cus = r < 0 ? -1 : (r + 7) / 8 * 8;
    return o;
}
int p4p_persistent_grid(void) {              // workgroups of a persistent launch on the current device; 0: do not launch persistent kernels
    static int ncu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!ncu[dev] && hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (g_reserved_cus < 0) return 0;
    const int g = ncu[dev] - g_reserved_cus;
    return g >= 64 ? g : 64;                 // never fewer than a quarter of an MI355X
}
extern "C" int p4p_persistent_grid_size(void) { return p4p_persistent_grid(); }

// A stand-in for a long-lived foreign kernel (tools/contention_rehearsal.py): `workgroups` single-wave workgroups that each hold
// `lds_bytes` of LDS (> 80 KiB: one per CU, and no 160-KiB workgroup of ours fits beside it) and sleep until `microseconds` of
// s_memrealtime (100 MHz) have passed -- a bounded wait every wave reaches the end of, whatever else happens.
__global__ void occupy_cus_kernel(long long ticks, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char occ_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    occ_lds[threadIdx.x] = (char)threadIdx.x;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && occ_lds[threadIdx.x] == 77 && ticks < 0) *sink = 1;          // keeps the LDS store alive; never true
}
extern "C" int p4p_debug_occupy_cus(int workgroups, int lds_bytes, long long microseconds, void* stream) {
    P4P_CHECK(workgroups >= 1 && workgroups <= 256 && lds_bytes >= 64 && lds_bytes <= 160 * 1024 && microseconds >= 1 &&
              microseconds <= 5000000, "p4p_debug_occupy_cus: bad arguments (workgroups=%d, lds=%d, us=%lld)", workgroups, lds_bytes, microseconds);
    static bool attr_set = false;
    if (!attr_set) {
        P4P_HIP(hipFuncSetAttribute((const void*)occupy_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(workgroups), dim3(64), lds_bytes, (hipStream_t)stream, microseconds * 100, (int*)nullptr);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}
