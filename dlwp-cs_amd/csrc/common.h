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
// Every tag the library can hand to prof_begin is registered when the library is LOADED (dlwpcs_prof_known_tag enumerates them
// without a device): bench.py joins rocprofv3's per-kernel counter records on these names, so a tag that is not the kernel's name
// as the code object spells it silently loses the counters (round 4: a tag built with 12 of a kernel's 13 template arguments).
// tests/test_abi.py compares the registry with the kernel symbols of libdlwpcs.so.
int prof_register_tag(const char *tag);
// KTag<Name, T, Vs...>::tag(): "name<T, v1, v2, ...>" with bools as true / false -- the instantiation's name as the demangler
// prints it -- built once and registered by a static initialiser of the instantiation (i.e. at load time, whether or not the kernel
// is ever launched).  Name: a type with `static const char *str()`; T: float / unsigned short (bf16 storage) / void (no type argument).
template <typename T> struct KTagType { static const char *str() { return nullptr; } };
template <> struct KTagType<float> { static const char *str() { return "float"; } };
template <> struct KTagType<unsigned short> { static const char *str() { return "unsigned short"; } };
template <typename Name, typename T, auto... Vs> struct KTag {
    static void fmt(char *&p, char *end, bool v) { p += snprintf(p, end - p, "%s", v ? "true" : "false"); }
    static void fmt(char *&p, char *end, int v) { p += snprintf(p, end - p, "%d", v); }
    static const char *str() {
        static char buf[200];
        if (!buf[0]) {
            char *p = buf, *end = buf + sizeof(buf);
            p += snprintf(p, end - p, "%s<", Name::str());
            bool first = true;
            if (KTagType<T>::str()) { p += snprintf(p, end - p, "%s", KTagType<T>::str()); first = false; }
            ((p += snprintf(p, end - p, "%s", first ? "" : ", "), fmt(p, end, Vs), first = false), ...);
            snprintf(p, end - p, ">");
        }
        return buf;
    }
    static const int reg;
    static const char *tag() { return reg >= 0 ? str() : str(); }      // (odr-use of reg: the registration is instantiated)
};
template <typename Name, typename T, auto... Vs> const int KTag<Name, T, Vs...>::reg = prof_register_tag(KTag<Name, T, Vs...>::str());

// dlwpcs_dgrad_gather_plan buffer (halo_table.cpp): [inverse table 6*N*N*4][header][T 6*M*M][border cells 6*(4N-4)*8][triples 6*2*3]
constexpr int DGG_HEADER = 8;
constexpr int DGG_CELL = 8;
constexpr int32_t DGG_MAGIC = 0x44474731;
size_t dgrad_gather_plan_ints(int N);
int dgrad_gather_wids(int N, int32_t out[36]);

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

// ---- output stores -------------------------------------------------------------------------------------------------
// The big outputs (activations, gradients, weight-gradient partials) are stored WRITE-THROUGH (sc1) through a buffer
// descriptor: a kernel that leaves B dirty bytes in the L2s pays ~B / 6 TB/s at its end before the next kernel of the stream
// may start (MI355X_MICROARCH.md, "boundary"), and these kernels write 14-28 MB each, so plain stores put 2-4 us of
// write-back behind every launch.  Lanes with nothing to store pass ST_SKIP: an offset beyond num_records is dropped by the
// hardware's range check -- no branch around the store.  -DDLWPCS_WT_STORES=0 builds plain stores for A/B runs.
#ifndef DLWPCS_WT_STORES
#define DLWPCS_WT_STORES 1
#endif
constexpr int ST_AUX = DLWPCS_WT_STORES ? 16 : 0;           // aux bit 4 = sc1
constexpr uint32_t ST_SKIP = 0xffffffffu;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// raw buffer (stride 0) over [base, base + bytes): byte offsets, range-checked
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, base ? bytes : 0u, 0x00020000);
}
__device__ __forceinline__ void bst128(const uint4 &v, rsrc_t r, uint32_t byte_off) {
    const u32x4 q = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, byte_off, 0, ST_AUX);
}
__device__ __forceinline__ void bst64(const uint2 &v, rsrc_t r, uint32_t byte_off) {
    const u32x2 q = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(q, r, byte_off, 0, ST_AUX);
}
__device__ __forceinline__ void bst32(uint32_t v, rsrc_t r, uint32_t byte_off) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, ST_AUX);
}
// Second stage of the fused head's loss reduction (256 threads): the workgroup partials [nblocks][2] {sum of squares, sum of
// absolute errors}, added in a fixed order in double precision -> loss_out = {weight * mse, mae}.  A launch of its own
// (head_stage2_kernel) or the last workgroup of the weight-gradient reduction (dlwpcs_wgrad_batch_adam_tail): same code, same bits.
__device__ __forceinline__ void loss_stage2_body(const float *__restrict__ partial, float *__restrict__ loss_out, int nblocks,
                                                 float inv_n, float weight, int overwrite) {
    __shared__ double s_sq[256], s_ab[256];
    double sq = 0.0, ab = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { sq += partial[2 * i]; ab += partial[2 * i + 1]; }
    s_sq[threadIdx.x] = sq; s_ab[threadIdx.x] = ab;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s_sq[threadIdx.x] += s_sq[threadIdx.x + s]; s_ab[threadIdx.x] += s_ab[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l0 = (float)(s_sq[0] * inv_n) * weight, l1 = (float)(s_ab[0] * inv_n);
        loss_out[0] = overwrite ? l0 : loss_out[0] + l0;
        loss_out[1] = overwrite ? l1 : loss_out[1] + l1;
    }
}
__device__ __forceinline__ void bstv(const uint4 &v, rsrc_t r, uint32_t o) { bst128(v, r, o); }
__device__ __forceinline__ void bstv(const uint2 &v, rsrc_t r, uint32_t o) { bst64(v, r, o); }
__device__ __forceinline__ void bstv(uint32_t v, rsrc_t r, uint32_t o) { bst32(v, r, o); }

// keras ReLU(negative_slope=alpha, max_value=vmax)  (Azure/train_cs.py:199)
__device__ __forceinline__ float act_leaky_clip(float x, float alpha, float vmax) {
    return x >= 0.f ? fminf(x, vmax) : alpha * x;
}
// derivative expressed through the saved OUTPUT y: y<0 <=> x<0 (slope alpha); 0<y<vmax (slope 1); else 0
__device__ __forceinline__ float act_leaky_clip_grad_from_y(float y, float alpha, float vmax) {
    return y < 0.f ? alpha : ((y > 0.f && y < vmax) ? 1.f : 0.f);
}

// TF2.1-keras Adam, one element; contraction off so that every kernel built on it produces the same bits
__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p = p - lr_t * m / (sqrtf(v) + eps);
}
__device__ __forceinline__ float adam_lr_t(float lr, float b1, float b2, float t) {
    return lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
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
