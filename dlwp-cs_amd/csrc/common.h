// Shared host/device helpers for libdlwpcs (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dlwpcs.h"

namespace dlwpcs {

// thread-local last-error message (C ABI never throws)
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
const char *last_error();

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DLWPCS_OK;
}

// opt-in launch profiler (prof.cpp)
bool prof_enabled();
int prof_begin(const char *tag, double flops, double bytes, hipStream_t s);
void prof_end(int idx, hipStream_t s);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// exact floor(n/d) for n,d < 2^16 as __umulhi(n, magic)
static inline uint32_t div_magic(uint32_t d) { return (uint32_t)((0x100000000ull + d - 1) / d); }

// MI355X: 8 XCDs, block b lands on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch").  Bijective remap that
// gives every XCD a contiguous range of logical tiles so neighbouring tiles (same sample / face: shared halo rows,
// same weights) hit the same 4 MiB L2.  Placement only affects speed, never results.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
    const uint32_t NX = 8;
    uint32_t xcd = bid % NX, idx = bid / NX;
    uint32_t q = nblk / NX, r = nblk % NX;
    uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// keras ReLU(negative_slope=alpha, max_value=vmax)  (Azure/train_cs.py:199)
__device__ __forceinline__ float act_leaky_clip(float x, float alpha, float vmax) {
    return x >= 0.f ? fminf(x, vmax) : alpha * x;
}
// derivative expressed through the saved OUTPUT y: y<0 <=> x<0 (slope alpha); 0<y<vmax (slope 1); else 0
__device__ __forceinline__ float act_leaky_clip_grad_from_y(float y, float alpha, float vmax) {
    return y < 0.f ? alpha : ((y > 0.f && y < vmax) ? 1.f : 0.f);
}

// ---- bf16 storage (DLWPCS_BF16): raw 16-bit patterns in memory, fp32 in registers ------------------------------
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// round-to-nearest-even, one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}
static inline size_t dtype_size(int dtype) { return dtype == DLWPCS_BF16 ? 2 : 4; }
static inline bool dtype_ok(int dtype) { return dtype == DLWPCS_F32 || dtype == DLWPCS_BF16; }

}  // namespace dlwpcs
