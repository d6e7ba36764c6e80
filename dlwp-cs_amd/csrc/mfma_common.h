// Device / host helpers shared by the matrix-core convolution kernels (conv_mfma.hip) and the batched weight-gradient
// kernel (wgrad_batch.hip): register vectors of VW channels, zero-selects, the MFMA fragment update, the LDS transpose read.
#pragma once
#include <string.h>
#include "common.h"

namespace dlwpcs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// VW consecutive channels of element type T as one register vector
template <typename T, int VW> struct VecT;
template <> struct VecT<float, 1> { typedef float type; };
template <> struct VecT<float, 2> { typedef float2 type; };
template <> struct VecT<float, 4> { typedef float4 type; };
template <> struct VecT<bf16_t, 1> { typedef uint16_t type; };
template <> struct VecT<bf16_t, 2> { typedef uint32_t type; };
template <> struct VecT<bf16_t, 4> { typedef uint2 type; };
template <> struct VecT<bf16_t, 8> { typedef uint4 type; };

// v *= act'(y): fp32 vectors
__device__ __forceinline__ void vmask(float &v, const float &y, float a, float m) { v *= act_leaky_clip_grad_from_y(y, a, m); }
__device__ __forceinline__ void vmask(float2 &v, const float2 &y, float a, float m) {
    v.x *= act_leaky_clip_grad_from_y(y.x, a, m); v.y *= act_leaky_clip_grad_from_y(y.y, a, m);
}
__device__ __forceinline__ void vmask(float4 &v, const float4 &y, float a, float m) {
    v.x *= act_leaky_clip_grad_from_y(y.x, a, m); v.y *= act_leaky_clip_grad_from_y(y.y, a, m);
    v.z *= act_leaky_clip_grad_from_y(y.z, a, m); v.w *= act_leaky_clip_grad_from_y(y.w, a, m);
}
// bf16 vectors (raw bit patterns): the product is rounded back to bf16 (dz is a bf16 tensor in this mode)
__device__ __forceinline__ uint32_t bmask2(uint32_t v, uint32_t y, float a, float m) {
    return f2bf2(bf_lo(v) * act_leaky_clip_grad_from_y(bf_lo(y), a, m), bf_hi(v) * act_leaky_clip_grad_from_y(bf_hi(y), a, m));
}
__device__ __forceinline__ void vmask(uint16_t &v, const uint16_t &y, float a, float m) { v = f2bf(bf2f(v) * act_leaky_clip_grad_from_y(bf2f(y), a, m)); }
__device__ __forceinline__ void vmask(uint32_t &v, const uint32_t &y, float a, float m) { v = bmask2(v, y, a, m); }
__device__ __forceinline__ void vmask(uint2 &v, const uint2 &y, float a, float m) { v.x = bmask2(v.x, y.x, a, m); v.y = bmask2(v.y, y.y, a, m); }
__device__ __forceinline__ void vmask(uint4 &v, const uint4 &y, float a, float m) {
    v.x = bmask2(v.x, y.x, a, m); v.y = bmask2(v.y, y.y, a, m); v.z = bmask2(v.z, y.z, a, m); v.w = bmask2(v.w, y.w, a, m);
}
// The same product on packed bf16 pairs with 16-bit SIMD-in-register integer ops (11 VALU instructions per pair instead of ~17):
// thr1 = (bits of the smallest bf16 >= vmax) - 1 in both halves (0 when vmax == 0), see bf16_mask_threshold().
//   y < 0            -> alpha * v     (sign bit; y = -0.0 counts as negative here)
//   0 < y < vmax     -> v             (1 <= bits(y) < thr, as an unsigned saturating subtraction)
//   otherwise        -> 0
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bmask2_pk(uint32_t v, uint32_t y, float alpha, uint32_t thr1) {
    const uint32_t m_neg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, y) >> 15);
    const u16x2 one = {1, 1}, zero = {0, 0};
    const u16x2 t = __builtin_bit_cast(u16x2, y) - one;
    const u16x2 c = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, thr1), t);
    const uint32_t m_pos = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, zero - c) >> 15);
    const uint32_t za = f2bf2(bf_lo(v) * alpha, bf_hi(v) * alpha);
    return (v & m_pos) | (za & m_neg);
}
__device__ __forceinline__ void vmask_pk(uint4 &v, const uint4 &y, float a, uint32_t thr1) {
    v.x = bmask2_pk(v.x, y.x, a, thr1); v.y = bmask2_pk(v.y, y.y, a, thr1);
    v.z = bmask2_pk(v.z, y.z, a, thr1); v.w = bmask2_pk(v.w, y.w, a, thr1);
}
// host: thr1 of bmask2_pk for a max_value >= 0 (+inf: every finite positive value passes)
static inline uint32_t bf16_mask_threshold(float vmax) {
    uint32_t u;
    memcpy(&u, &vmax, 4);
    uint32_t thr = u >> 16;
    if (u & 0xffffu) thr += 1;                      // not representable: the next bf16 above
    if (vmax != vmax || thr > 0x7f80u) thr = 0x7f80u;
    const uint32_t t1 = thr ? thr - 1 : 0;
    return t1 | (t1 << 16);
}

__device__ __forceinline__ float vsel(bool c, float v) { return c ? v : 0.f; }
__device__ __forceinline__ float2 vsel(bool c, float2 v) { return c ? v : make_float2(0.f, 0.f); }
__device__ __forceinline__ float4 vsel(bool c, float4 v) { return c ? v : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ uint16_t vsel(bool c, uint16_t v) { return c ? v : (uint16_t)0; }
__device__ __forceinline__ uint32_t vsel(bool c, uint32_t v) { return c ? v : 0u; }
__device__ __forceinline__ uint2 vsel(bool c, uint2 v) { return c ? v : make_uint2(0u, 0u); }
__device__ __forceinline__ uint4 vsel(bool c, uint4 v) { return c ? v : make_uint4(0u, 0u, 0u, 0u); }
typedef uint4 uint4_a4 __attribute__((aligned(4)));
typedef uint2 uint2_a4 __attribute__((aligned(4)));
// drop the first `sh` dwords of a 16-B vector (zeros shift in), see TAIL8
__device__ __forceinline__ uint4 vshl_dwords(uint4 v, int sh) {
    if (sh == 1) return make_uint4(v.y, v.z, v.w, 0u);
    if (sh == 2) return make_uint4(v.z, v.w, 0u, 0u);
    if (sh == 3) return make_uint4(v.w, 0u, 0u, 0u);
    return v;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one (M tile, N tile) update from a 16-B A and a 16-B B fragment:
//   fp32: 4 x v_mfma_f32_32x32x2_f32  (K = 8 channels per fragment pair)
//   bf16: 1 x v_mfma_f32_32x32x16_bf16 (K = 16 channels per fragment pair)
template <typename T> __device__ __forceinline__ void frag_mma(f32x16 &acc, const uint4 &a, const uint4 &b);
template <> __device__ __forceinline__ void frag_mma<float>(f32x16 &acc, const uint4 &a, const uint4 &b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void frag_mma<bf16_t>(f32x16 &acc, const uint4 &a, const uint4 &b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <typename T> struct MmaPerFrag;
template <> struct MmaPerFrag<float> { static constexpr int N = 4; };
template <> struct MmaPerFrag<bf16_t> { static constexpr int N = 1; };


typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const char *p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
    return __builtin_bit_cast(uint2, v);
}


// rows of the face touched by `pix` consecutive flat pixels whose first pixel is a multiple of `pix`
static inline int tile_rows_for(int pix, int No) {
    if (pix % No == 0) return pix / No;
    int r = (pix + No - 2) / No + 1;
    return r > No ? No : r;
}


}  // namespace dlwpcs
