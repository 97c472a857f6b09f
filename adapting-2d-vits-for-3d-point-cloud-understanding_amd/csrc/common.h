// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the Pix4Point hot path.
// Wave = 64 lanes everywhere in this tree; nothing here is meant to compile for another target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

typedef unsigned long long u64;
typedef unsigned int u32;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error reporting across the C ABI -------------------------------------------------------
void p4p_set_error(const char* fmt, ...);
#define P4P_CHECK(cond, ...)                         \
    do {                                             \
        if (!(cond)) {                               \
            p4p_set_error(__VA_ARGS__);              \
            return 1;                                \
        }                                            \
    } while (0)
#define P4P_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            p4p_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define P4P_LAUNCH_CHECK() P4P_HIP(hipGetLastError())

// ---- bf16 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(__bf16 x) { return (float)x; }
__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }   // v_cvt_pk_bf16_f32: RNE, NaN safe

__device__ __forceinline__ u32 pack_bf16x2(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<u32*>(&v);
}

// ---- DPP cross-lane moves (gfx9 family) -----------------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E; row_half_mirror = 0x141; row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ u32 dpp_u32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, dpp_u32<CTRL>(__builtin_bit_cast(u32, v)));
}
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    u32 lo = dpp_u32<CTRL>((u32)v), hi = dpp_u32<CTRL>((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// All-lanes max / min of a u64 over the 64-lane wave; result is wave-uniform.
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
    u64 o;
    o = dpp_u64<0xB1>(v);  v = o > v ? o : v;
    o = dpp_u64<0x4E>(v);  v = o > v ? o : v;
    o = dpp_u64<0x141>(v); v = o > v ? o : v;
    o = dpp_u64<0x140>(v); v = o > v ? o : v;
    u64 a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
    a = a > b ? a : b;
    c = c > d ? c : d;
    return a > c ? a : c;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
    u64 o;
    o = dpp_u64<0xB1>(v);  v = o < v ? o : v;
    o = dpp_u64<0x4E>(v);  v = o < v ? o : v;
    o = dpp_u64<0x141>(v); v = o < v ? o : v;
    o = dpp_u64<0x140>(v); v = o < v ? o : v;
    u64 a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
    a = a < b ? a : b;
    c = c < d ? c : d;
    return a < c ? a : c;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// Exact, unfused fp32 squared distance in the op order the reference's torch.sum produces:
// ((dx*dx + dy*dy) + dz*dz).  __fmul_rn/__fadd_rn are never contracted into FMA.
__device__ __forceinline__ float sqdist3(float x, float y, float z, float cx, float cy, float cz) {
    float dx = __fsub_rn(x, cx), dy = __fsub_rn(y, cy), dz = __fsub_rn(z, cz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// persistent launches (capi.cpp): workgroups = CUs of the current device - p4p_set_reserved_cus(); 0 = no persistent kernels
int p4p_persistent_grid(void);
extern "C" int p4p_device_cu_count(void);

// launch profiler (capi.cpp): brackets one kernel launch with HIP events when enabled for `kind`
bool p4p_prof_begin(int kind, double flops, hipStream_t st);
void p4p_prof_end(hipStream_t st);
