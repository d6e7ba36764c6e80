// HBM-bound helper kernels of the DLWP-CS hot path (gfx950): stand-alone halo gather / inverse gather, activation,
// 2x2 pooling / upsampling, concat / split, layout converters, batch gather (feed), loss and Adam.  All are pure streaming
// kernels: channels_last rows are moved 16 B per lane whenever the channel count allows (fp32 and bf16 storage vectors
// with an fp32 register image), grid-stride loops capped at ~2048 workgroups (cdna_hip_programming.md, Guideline 11/13).
// No atomics anywhere -> bitwise deterministic.
#include "common.h"

namespace dlwpcs {

static inline dim3 stream_grid(size_t work_items, int block = 256) {
    size_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

// ---- storage vectors and their fp32 register image ------------------------------------------------------------
// A kernel instantiated on a storage type V moves VT<V>::N consecutive channels per lane; arithmetic happens on the
// fp32 image Acc<N> and is rounded once on store (bf16: round-to-nearest-even), so the fp32 instantiations are
// bit-identical to plain float code.
struct H2 { uint32_t u; };                  // 2 x bf16
struct alignas(16) H8 { uint4 u; };         // 8 x bf16
template <int N> struct Acc { float v[N]; };
template <int N> __device__ __forceinline__ Acc<N> vadd(Acc<N> a, const Acc<N> &b) {
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] += b.v[i];
    return a;
}
template <int N> __device__ __forceinline__ Acc<N> vscale(Acc<N> a, float s) {
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] *= s;
    return a;
}
// a[i] *= act'(y[i]) of keras ReLU(negative_slope = alpha, max_value = vmax), evaluated from the activation's OUTPUT y
template <int N> __device__ __forceinline__ Acc<N> vmaskacc(Acc<N> a, const Acc<N> &y, float alpha, float vmax) {
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] *= act_leaky_clip_grad_from_y(y.v[i], alpha, vmax);
    return a;
}
template <int N> __device__ __forceinline__ Acc<N> vzero() {
    Acc<N> a;
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] = 0.f;
    return a;
}
template <typename V> struct VT;
template <> struct VT<float> {
    static constexpr int N = 1;
    static __device__ __forceinline__ Acc<1> ld(const float *p) { Acc<1> a; a.v[0] = *p; return a; }
    static __device__ __forceinline__ void st(float *p, const Acc<1> &a) { *p = a.v[0]; }
};
template <> struct VT<float4> {
    static constexpr int N = 4;
    static __device__ __forceinline__ Acc<4> ld(const float4 *p) { const float4 q = *p; Acc<4> a; a.v[0] = q.x; a.v[1] = q.y; a.v[2] = q.z; a.v[3] = q.w; return a; }
    static __device__ __forceinline__ void st(float4 *p, const Acc<4> &a) { *p = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }
};
template <> struct VT<bf16_t> {
    static constexpr int N = 1;
    static __device__ __forceinline__ Acc<1> ld(const bf16_t *p) { Acc<1> a; a.v[0] = bf2f(*p); return a; }
    static __device__ __forceinline__ void st(bf16_t *p, const Acc<1> &a) { *p = f2bf(a.v[0]); }
};
template <> struct VT<H2> {
    static constexpr int N = 2;
    static __device__ __forceinline__ Acc<2> ld(const H2 *p) { const uint32_t u = p->u; Acc<2> a; a.v[0] = bf_lo(u); a.v[1] = bf_hi(u); return a; }
    static __device__ __forceinline__ void st(H2 *p, const Acc<2> &a) { p->u = f2bf2(a.v[0], a.v[1]); }
};
template <> struct VT<H8> {
    static constexpr int N = 8;
    static __device__ __forceinline__ Acc<8> ld(const H8 *p) {
        const uint4 q = p->u; Acc<8> a;
        a.v[0] = bf_lo(q.x); a.v[1] = bf_hi(q.x); a.v[2] = bf_lo(q.y); a.v[3] = bf_hi(q.y);
        a.v[4] = bf_lo(q.z); a.v[5] = bf_hi(q.z); a.v[6] = bf_lo(q.w); a.v[7] = bf_hi(q.w);
        return a;
    }
    static __device__ __forceinline__ void st(H8 *p, const Acc<8> &a) {
        p->u = make_uint4(f2bf2(a.v[0], a.v[1]), f2bf2(a.v[2], a.v[3]), f2bf2(a.v[4], a.v[5]), f2bf2(a.v[6], a.v[7]));
    }
};

struct alignas(16) F8 { float4 a, b; };     // 8 x fp32
template <> struct VT<F8> {
    static constexpr int N = 8;
    static __device__ __forceinline__ Acc<8> ld(const F8 *p) {
        const float4 a = p->a, b = p->b; Acc<8> r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        return r;
    }
    static __device__ __forceinline__ void st(F8 *p, const Acc<8> &r) {
        p->a = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]); p->b = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
    }
};

// f(tag, width): widest storage vector that divides every channel count in play (g = their gcd-like common divisor)
template <typename F> static inline void dispatch_vec(int dtype, int g, F &&f) {
    if (dtype == DLWPCS_BF16) {
        if (g % 8 == 0) f(H8{}, 8); else if (g % 2 == 0) f(H2{}, 2); else f(bf16_t{}, 1);
    } else {
        if (g % 4 == 0) f(float4{}, 4); else f(float{}, 1);
    }
}
// pure data movement: widest power-of-two byte vector dividing `row_bytes`
template <typename F> static inline void dispatch_mover(size_t row_bytes, F &&f) {
    if (row_bytes % 16 == 0) f(uint4{}, 16); else if (row_bytes % 4 == 0) f(uint32_t{}, 4); else f(uint16_t{}, 2);
}

// ---------------------------------------------------------------------------------------------------------------
// CubeSpherePadding2D forward: y[b][dst][c] = x[b][T[dst]][c]           (DLWP/custom.py:1082-1308 as one gather)
// ---------------------------------------------------------------------------------------------------------------
template <typename V>
__global__ void __launch_bounds__(256) pad_fwd_kernel(const V *__restrict__ x, V *__restrict__ y,
                                                      const int32_t *__restrict__ table, size_t total, int CV,
                                                      int src_cells, int dst_cells) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        const size_t pix = e / CV;
        const int dst = (int)(pix % dst_cells);
        const size_t b = pix / dst_cells;
        const int src = table[dst];
        y[e] = x[(b * src_cells + src) * CV + cv];
    }
}

// backward: dx[b][src] = dy[b][identity(src)] + sum_k dy[b][inv[src][k]]        (<= 5 terms, fixed order)
template <typename V>
__global__ void __launch_bounds__(256) pad_bwd_kernel(const V *__restrict__ dy, V *__restrict__ dx,
                                                      const int32_t *__restrict__ inv, size_t total, int CV,
                                                      int N, int p) {
    const int M = N + 2 * p;
    const int src_cells = 6 * N * N, dst_cells = 6 * M * M;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        const size_t pix = e / CV;
        const int src = (int)(pix % src_cells);
        const size_t b = pix / src_cells;
        const int xx = src % N, yy = (src / N) % N, f = src / (N * N);
        const V *base = dy + b * dst_cells * (size_t)CV + cv;
        auto acc = VT<V>::ld(base + (size_t)((f * M + yy + p) * M + xx + p) * CV);
        const bool border = (yy < p) | (yy >= N - p) | (xx < p) | (xx >= N - p);
        if (border) {
            const int4 t = *reinterpret_cast<const int4 *>(inv + (size_t)src * 4);
            if (t.x >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.x * CV));
            if (t.y >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.y * CV));
            if (t.z >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.z * CV));
            if (t.w >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.w * CV));
        }
        VT<V>::st(dx + e, acc);
    }
}

// Adjoint of the fused conv loader: dxpad (B,6,M,M,CT) -> gradient of ONE source (channel window [choff, choff+CS))
// on the N grid, or on the N/2 grid with the 2x2 block sum of the nearest-upsample adjoint (up != 0).  p = 1.
// msrc != nullptr (pre-masked gradients): the result is multiplied by act'(msrc), msrc = the source itself (same shape as dsrc)
template <typename V>
__device__ __forceinline__ void pad_bwd_src_body(const V *__restrict__ dxpad, V *__restrict__ dsrc,
                                                 const int32_t *__restrict__ inv, size_t total, int CSV, int CTV,
                                                 int choffV, int N, int up, unsigned block, unsigned nblocks,
                                                 const V *__restrict__ msrc = nullptr, float m_alpha = 0.f, float m_vmax = 0.f) {
    const int p = 1;
    const int M = N + 2 * p;
    const int No = up ? N / 2 : N;
    const int out_cells = 6 * No * No, dst_cells = 6 * M * M;
    for (size_t e = (size_t)block * blockDim.x + threadIdx.x; e < total; e += (size_t)nblocks * blockDim.x) {
        const int cv = (int)(e % CSV);
        const size_t pix = e / CSV;
        const int cell = (int)(pix % out_cells);
        const size_t b = pix / out_cells;
        const int xo = cell % No, yo = (cell / No) % No, f = cell / (No * No);
        const V *base = dxpad + b * dst_cells * (size_t)CTV + choffV + cv;
        auto acc = vzero<VT<V>::N>();
        const int reps = up ? 2 : 1;
        for (int uy = 0; uy < reps; ++uy)
            for (int ux = 0; ux < reps; ++ux) {
                const int yy = up ? 2 * yo + uy : yo, xx = up ? 2 * xo + ux : xo;
                acc = vadd(acc, VT<V>::ld(base + (size_t)((f * M + yy + p) * M + xx + p) * CTV));
                const bool border = (yy < p) | (yy >= N - p) | (xx < p) | (xx >= N - p);
                if (border) {
                    const int src = (f * N + yy) * N + xx;
                    const int4 t = *reinterpret_cast<const int4 *>(inv + (size_t)src * 4);
                    if (t.x >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.x * CTV));
                    if (t.y >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.y * CTV));
                    if (t.z >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.z * CTV));
                    if (t.w >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.w * CTV));
                }
            }
        if (msrc) acc = vmaskacc(acc, VT<V>::ld(msrc + e), m_alpha, m_vmax);
        VT<V>::st(dsrc + e, acc);
    }
}

template <typename V>
__global__ void __launch_bounds__(256) pad_bwd_src_kernel(const V *__restrict__ dxpad, V *__restrict__ dsrc,
                                                          const int32_t *__restrict__ inv, size_t total, int CSV,
                                                          int CTV, int choffV, int N, int up, const V *__restrict__ msrc,
                                                          float m_alpha, float m_vmax) {
    pad_bwd_src_body<V>(dxpad, dsrc, inv, total, CSV, CTV, choffV, N, up, blockIdx.x, gridDim.x, msrc, m_alpha, m_vmax);
}

// Border fix-up after a data-gradient kernel that wrote the INTERIOR cells of dxpad straight into dsrc (conv_mfma.hip,
// direct mode): only the halo ring of dxpad was materialised; every border cell of the source (row/column 0 or N-1) still
// lacks the <= 4 ring cells that gathered from it.  One thread per (sample, border cell, channel vector); p = 1.
// msrc != nullptr (pre-masked gradients): the data-gradient kernel stored act'(msrc) * interior already; the ring cells are
// summed first and multiplied by act'(msrc) before they are added.
template <typename V>
__device__ __forceinline__ void pad_ring_fix_body(const V *__restrict__ dxpad, V *__restrict__ dsrc,
                                                  const int32_t *__restrict__ inv, size_t total, int CSV, int CTV,
                                                  int choffV, int N, unsigned block, unsigned nblocks,
                                                  const V *__restrict__ msrc = nullptr, float m_alpha = 0.f, float m_vmax = 0.f) {
    const int M = N + 2;
    const int nb = N > 1 ? 4 * N - 4 : 1;              // border cells per face
    const int dst_cells = 6 * M * M;
    for (size_t e = (size_t)block * blockDim.x + threadIdx.x; e < total; e += (size_t)nblocks * blockDim.x) {
        const int cv = (int)(e % CSV);
        size_t r = e / CSV;
        const int k = (int)(r % nb); r /= nb;
        const int f = (int)(r % 6);
        const size_t b = r / 6;
        int yy, xx;                                    // k-th border cell: top row, bottom row, then the two side columns
        if (k < N) { yy = 0; xx = k; }
        else if (k < 2 * N) { yy = N - 1; xx = k - N; }
        else if (k < 3 * N - 2) { yy = k - 2 * N + 1; xx = 0; }
        else { yy = k - (3 * N - 2) + 1; xx = N - 1; }
        const int src = (f * N + yy) * N + xx;
        const V *base = dxpad + b * dst_cells * (size_t)CTV + choffV + cv;
        const size_t didx = (b * 6 * N * N + src) * (size_t)CSV + cv;
        V *dst = dsrc + didx;
        const int4 t = *reinterpret_cast<const int4 *>(inv + (size_t)src * 4);
        if (msrc) {
            auto ring = vzero<VT<V>::N>();
            if (t.x >= 0) ring = vadd(ring, VT<V>::ld(base + (size_t)t.x * CTV));
            if (t.y >= 0) ring = vadd(ring, VT<V>::ld(base + (size_t)t.y * CTV));
            if (t.z >= 0) ring = vadd(ring, VT<V>::ld(base + (size_t)t.z * CTV));
            if (t.w >= 0) ring = vadd(ring, VT<V>::ld(base + (size_t)t.w * CTV));
            VT<V>::st(dst, vadd(VT<V>::ld(dst), vmaskacc(ring, VT<V>::ld(msrc + didx), m_alpha, m_vmax)));
            continue;
        }
        auto acc = VT<V>::ld(dst);
        if (t.x >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.x * CTV));
        if (t.y >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.y * CTV));
        if (t.z >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.z * CTV));
        if (t.w >= 0) acc = vadd(acc, VT<V>::ld(base + (size_t)t.w * CTV));
        VT<V>::st(dst, acc);
    }
}

template <typename V>
__global__ void __launch_bounds__(256) pad_ring_fix_kernel(const V *__restrict__ dxpad, V *__restrict__ dsrc,
                                                           const int32_t *__restrict__ inv, size_t total, int CSV,
                                                           int CTV, int choffV, int N, const V *__restrict__ msrc,
                                                           float m_alpha, float m_vmax) {
    pad_ring_fix_body<V>(dxpad, dsrc, inv, total, CSV, CTV, choffV, N, blockIdx.x, gridDim.x, msrc, m_alpha, m_vmax);
}

// Both sources of a fused decoder convolution in ONE launch (each launch costs ~4-5 us of floor): workgroups [0, nb0) route
// the upsampled source 0 (inverse halo gather + 2x2 block sum), the rest apply the ring fix-up to the directly written
// source 1.
template <typename V>
__global__ void __launch_bounds__(256) src_pair_kernel(const V *__restrict__ dxpad, V *__restrict__ dsrc0, V *__restrict__ dsrc1,
                                                       const int32_t *__restrict__ inv, size_t total0, size_t total1,
                                                       int CS0V, int CS1V, int CTV, int N, int up0, unsigned nb0,
                                                       const V *__restrict__ m0, const V *__restrict__ m1, float m_alpha,
                                                       float m_vmax) {
    if (blockIdx.x < nb0) pad_bwd_src_body<V>(dxpad, dsrc0, inv, total0, CS0V, CTV, 0, N, up0, blockIdx.x, nb0, m0, m_alpha, m_vmax);
    else pad_ring_fix_body<V>(dxpad, dsrc1, inv, total1, CS1V, CTV, CS0V, N, blockIdx.x - nb0, gridDim.x - nb0, m1, m_alpha, m_vmax);
}

// gradient of one source of a halo==0 convolution input (no padding): channel window copy, optional 2x2 sum
template <typename V>
__global__ void __launch_bounds__(256) window_src_kernel(const V *__restrict__ dxv, V *__restrict__ dsrc, size_t total,
                                                         int CSV, int CTV, int choffV, int N, int up,
                                                         const V *__restrict__ msrc = nullptr, float m_alpha = 0.f, float m_vmax = 0.f) {
    const int No = up ? N / 2 : N;
    const int out_cells = 6 * No * No;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CSV);
        const size_t pix = e / CSV;
        const int cell = (int)(pix % out_cells);
        const size_t b = pix / out_cells;
        const int xo = cell % No, yo = (cell / No) % No, f = cell / (No * No);
        const V *base = dxv + b * (size_t)6 * N * N * CTV + choffV + cv;
        auto acc = vzero<VT<V>::N>();
        const int reps = up ? 2 : 1;
        for (int uy = 0; uy < reps; ++uy)
            for (int ux = 0; ux < reps; ++ux) {
                const int yy = up ? 2 * yo + uy : yo, xx = up ? 2 * xo + ux : xo;
                acc = vadd(acc, VT<V>::ld(base + (size_t)((f * N + yy) * N + xx) * CTV));
            }
        if (msrc) acc = vmaskacc(acc, VT<V>::ld(msrc + e), m_alpha, m_vmax);     // pre-masked gradients: x act'(the source itself)
        VT<V>::st(dsrc + e, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// activation: keras ReLU(negative_slope, max_value)
// ---------------------------------------------------------------------------------------------------------------
template <typename V, typename S>
__global__ void __launch_bounds__(256) act_fwd_kernel(const S *__restrict__ x, S *__restrict__ y, size_t n,
                                                      float alpha, float vmax) {
    constexpr int W = VT<V>::N;
    const size_t nv = n / W;
    const V *xv = reinterpret_cast<const V *>(x);
    V *yv = reinterpret_cast<V *>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        auto v = VT<V>::ld(xv + i);
#pragma unroll
        for (int k = 0; k < W; ++k) v.v[k] = act_leaky_clip(v.v[k], alpha, vmax);
        VT<V>::st(yv + i, v);
    }
    for (size_t i = nv * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        auto v = VT<S>::ld(x + i);
        v.v[0] = act_leaky_clip(v.v[0], alpha, vmax);
        VT<S>::st(y + i, v);
    }
}

template <typename V, typename S>
__global__ void __launch_bounds__(256) act_bwd_kernel(const S *dy, const S *__restrict__ y, S *dx, size_t n, float alpha,
                                                      float vmax) {
    constexpr int W = VT<V>::N;
    const size_t nv = n / W;
    const V *gv = reinterpret_cast<const V *>(dy);
    const V *yv = reinterpret_cast<const V *>(y);
    V *ov = reinterpret_cast<V *>(dx);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        auto g = VT<V>::ld(gv + i);
        const auto v = VT<V>::ld(yv + i);
#pragma unroll
        for (int k = 0; k < W; ++k) g.v[k] *= act_leaky_clip_grad_from_y(v.v[k], alpha, vmax);
        VT<V>::st(ov + i, g);
    }
    for (size_t i = nv * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        auto g = VT<S>::ld(dy + i);
        g.v[0] *= act_leaky_clip_grad_from_y(VT<S>::ld(y + i).v[0], alpha, vmax);
        VT<S>::st(dx + i, g);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// AveragePooling3D((1,2,2)) / UpSampling3D((1,2,2)), channels_last      (Azure/train_cs.py:197-198)
// `planes` = B*6; x is (planes, N, N, C)
// ---------------------------------------------------------------------------------------------------------------
template <typename V>
__global__ void __launch_bounds__(256) avgpool2_fwd_kernel(const V *__restrict__ x, V *__restrict__ y, size_t total,
                                                           int CV, int N) {
    const int No = N / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        size_t pix = e / CV;
        const int xo = (int)(pix % No); pix /= No;
        const int yo = (int)(pix % No);
        const size_t plane = pix / No;
        const V *r0 = x + ((plane * N + 2 * yo) * N + 2 * xo) * CV + cv;
        const V *r1 = r0 + (size_t)N * CV;
        VT<V>::st(y + e, vscale(vadd(vadd(VT<V>::ld(r0), VT<V>::ld(r0 + CV)), vadd(VT<V>::ld(r1), VT<V>::ld(r1 + CV))), 0.25f));
    }
}

// dx (planes,N,N,C) = 0.25 * dy (planes,N/2,N/2,C) spread over each 2x2 block
template <typename V>
__global__ void __launch_bounds__(256) avgpool2_bwd_kernel(const V *__restrict__ dy, V *__restrict__ dx, size_t total,
                                                           int CV, int N) {
    const int No = N / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        size_t pix = e / CV;
        const int xx = (int)(pix % N); pix /= N;
        const int yy = (int)(pix % N);
        const size_t plane = pix / N;
        VT<V>::st(dx + e, vscale(VT<V>::ld(dy + ((plane * No + yy / 2) * No + xx / 2) * CV + cv), 0.25f));
    }
}

// dx = dskip + 0.25 * dy spread over each 2x2 block: the pooled tensor's other consumer (a U-Net skip connection)
// contributes dskip; one pass instead of avgpool2_bwd + an elementwise add (fp32 sum, one rounding)
template <typename V>
__global__ void __launch_bounds__(256) avgpool2_bwd_add_kernel(const V *__restrict__ dy, const V *__restrict__ dskip,
                                                               V *__restrict__ dx, size_t total, int CV, int N) {
    const int No = N / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        size_t pix = e / CV;
        const int xx = (int)(pix % N); pix /= N;
        const int yy = (int)(pix % N);
        const size_t plane = pix / N;
        VT<V>::st(dx + e, vadd(VT<V>::ld(dskip + e),
                               vscale(VT<V>::ld(dy + ((plane * No + yy / 2) * No + xx / 2) * CV + cv), 0.25f)));
    }
}

// pre-masked gradients: dx = act'(m) * (dskip + 0.25 * dy spread), m = the pooled tensor itself (output of the activated layer
// that produced it); dskip may be null (no skip connection).
// ring != nullptr: dy is the gradient of the POOLED tensor as the data-gradient kernel of its consumer wrote it directly
// (interior contributions only, DLWPCS_CONV_DEFER_RING0); the halo-ring cells that gathered from a border cell of the pooled
// grid are added here (ring = that call's padded gradient (B,6,No+2,No+2,CTV vectors), channel window at choffV; inv = inverse
// halo table of the No grid) -- the fix-up launch of its own is gone.
template <typename V>
__global__ void __launch_bounds__(256) avgpool2_bwd_masked_kernel(const V *__restrict__ dy, const V *__restrict__ dskip,
                                                                  const V *__restrict__ m, V *__restrict__ dx, size_t total,
                                                                  int CV, int N, float m_alpha, float m_vmax,
                                                                  const V *__restrict__ ring, const int32_t *__restrict__ inv,
                                                                  int CTV, int choffV) {
    const int No = N / 2, Mo = No + 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        size_t pix = e / CV;
        const int xx = (int)(pix % N); pix /= N;
        const int yy = (int)(pix % N);
        const size_t plane = pix / N;
        const int yp = yy / 2, xp = xx / 2;
        // the table entry first: the ring loads that depend on it are then in flight together with the streaming loads below
        const bool border = ring != nullptr && ((yp == 0) | (yp == No - 1) | (xp == 0) | (xp == No - 1));
        const size_t b = plane / 6;
        const int f = (int)(plane - b * 6);
        int4 t = make_int4(-1, -1, -1, -1);
        if (border) t = *reinterpret_cast<const int4 *>(inv + (size_t)((f * No + yp) * No + xp) * 4);
        auto gp = VT<V>::ld(dy + ((plane * No + yp) * No + xp) * CV + cv);
        auto sk = vzero<VT<V>::N>();
        if (dskip) sk = VT<V>::ld(dskip + e);
        auto mm = vzero<VT<V>::N>();
#ifdef DLWPCS_ABL_MASK8
        if (m) mm = VT<V>::ld(m + (e >> 3));
#else
        if (m) mm = VT<V>::ld(m + e);
#endif
        if (border) {
            const V *base = ring + b * (size_t)(6 * Mo * Mo) * CTV + choffV + cv;
            auto rs = vzero<VT<V>::N>();
            if (t.x >= 0) rs = vadd(rs, VT<V>::ld(base + (size_t)t.x * CTV));
            if (t.y >= 0) rs = vadd(rs, VT<V>::ld(base + (size_t)t.y * CTV));
            if (t.z >= 0) rs = vadd(rs, VT<V>::ld(base + (size_t)t.z * CTV));
            if (t.w >= 0) rs = vadd(rs, VT<V>::ld(base + (size_t)t.w * CTV));
            // like the fix-up kernel: the bf16 value it would have stored (interior + ring, rounded) is what gets pooled
            gp = vadd(gp, rs);
            if constexpr (sizeof(V) == 2 * VT<V>::N) {
#pragma unroll
                for (int i = 0; i < VT<V>::N; ++i) gp.v[i] = bf2f(f2bf(gp.v[i]));
            }
        }
        auto g = vscale(gp, 0.25f);
        if (dskip) g = vadd(sk, g);
        if (m) g = vmaskacc(g, mm, m_alpha, m_vmax);
        VT<V>::st(dx + e, g);
    }
}

// y (planes,2N,2N,C) = nearest(x (planes,N,N,C))
template <typename V>
__global__ void __launch_bounds__(256) upsample2_fwd_kernel(const V *__restrict__ x, V *__restrict__ y, size_t total,
                                                            int CV, int N) {
    const int No = 2 * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        size_t pix = e / CV;
        const int xx = (int)(pix % No); pix /= No;
        const int yy = (int)(pix % No);
        const size_t plane = pix / No;
        y[e] = x[((plane * N + yy / 2) * N + xx / 2) * CV + cv];
    }
}

// dx (planes,N,N,C) = sum of each 2x2 block of dy (planes,2N,2N,C)
template <typename V>
__global__ void __launch_bounds__(256) upsample2_bwd_kernel(const V *__restrict__ dy, V *__restrict__ dx, size_t total,
                                                            int CV, int N) {
    const int Ni = 2 * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        size_t pix = e / CV;
        const int xo = (int)(pix % N); pix /= N;
        const int yo = (int)(pix % N);
        const size_t plane = pix / N;
        const V *r0 = dy + ((plane * Ni + 2 * yo) * Ni + 2 * xo) * CV + cv;
        const V *r1 = r0 + (size_t)Ni * CV;
        VT<V>::st(dx + e, vadd(vadd(VT<V>::ld(r0), VT<V>::ld(r0 + CV)), vadd(VT<V>::ld(r1), VT<V>::ld(r1 + CV))));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// concat / split along channels, add
// ---------------------------------------------------------------------------------------------------------------
template <typename V>
__global__ void __launch_bounds__(256) concat2_kernel(const V *__restrict__ a, const V *__restrict__ b,
                                                      V *__restrict__ y, size_t total, int CaV, int CbV) {
    const int CV = CaV + CbV;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        const size_t row = e / CV;
        y[e] = cv < CaV ? a[row * CaV + cv] : b[row * CbV + (cv - CaV)];
    }
}

// Rollout state re-injection (reference DLWP/model/extensions.py:281-299 + the Reshape / Permute / Concatenate chain of
// Azure/train_cs.py:401-406 as ONE pass): the next main input of a forecast step is the model's last output with the known
// forcing (insolation) appended as the last channels of every time step,
//   out[b][s][n*(V+E) + j] = j < V ? state[b][s][n*V + j] : extra[b][n][s][j - V]            (n < T time steps)
// state (B,S,T*V) and extra (B,T,S,E) are channels_last tensors of the same element type W (raw 2- or 4-byte words).
template <typename W>
__global__ void __launch_bounds__(256) state_repack_kernel(const W *__restrict__ state, const W *__restrict__ extra,
                                                           W *__restrict__ out, size_t total, unsigned S, unsigned T,
                                                           unsigned V, unsigned E) {
    const unsigned VE = V + E, C = T * VE;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned c = (unsigned)(e % C);
        const size_t pix = e / C;                      // b * S + s
        const unsigned n = c / VE, j = c - n * VE;
        if (j < V) out[e] = state[pix * (size_t)(T * V) + n * V + j];
        else {
            const size_t b = pix / S, sp = pix - b * S;
            out[e] = extra[((b * T + n) * S + sp) * E + (j - V)];
        }
    }
}

// Channel padding of a channels_last tensor (dlwpcs_conv_desc.c0_valid): y[row][c] = c < C ? x[row][c] : 0 for c < Cp, and
// its adjoint (slice).  Element type W = raw 2- or 4-byte word.
template <typename W>
__global__ void __launch_bounds__(256) pad_channels_kernel(const W *__restrict__ x, W *__restrict__ y, size_t total, unsigned C,
                                                           unsigned Cp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Cp;
        const unsigned c = (unsigned)(e - row * Cp);
        y[e] = c < C ? x[row * C + c] : (W)0;
    }
}
template <typename W>
__global__ void __launch_bounds__(256) slice_channels_kernel(const W *__restrict__ y, W *__restrict__ x, size_t total, unsigned C,
                                                             unsigned Cp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / C;
        const unsigned c = (unsigned)(e - row * C);
        x[e] = y[row * Cp + c];
    }
}

template <typename V>
__global__ void __launch_bounds__(256) split2_kernel(const V *__restrict__ y, V *__restrict__ a, V *__restrict__ b,
                                                     size_t total, int CaV, int CbV) {
    const int CV = CaV + CbV;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        const size_t row = e / CV;
        const V v = y[e];
        if (cv < CaV) { if (a) a[row * CaV + cv] = v; }
        else if (b) b[row * CbV + (cv - CaV)] = v;
    }
}

template <typename V, typename S>
__global__ void __launch_bounds__(256) add_kernel(const S *__restrict__ a, const S *__restrict__ b,
                                                  S *__restrict__ y, size_t n) {
    constexpr int W = VT<V>::N;
    const size_t nv = n / W;
    const V *av = reinterpret_cast<const V *>(a);
    const V *bv = reinterpret_cast<const V *>(b);
    V *yv = reinterpret_cast<V *>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x)
        VT<V>::st(yv + i, vadd(VT<V>::ld(av + i), VT<V>::ld(bv + i)));
    for (size_t i = nv * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        VT<S>::st(y + i, vadd(VT<S>::ld(a + i), VT<S>::ld(b + i)));
}

// ---------------------------------------------------------------------------------------------------------------
// (B, C, S) <-> (B, S, C) through a padded 32x32 LDS tile (coalesced on both sides)
// ---------------------------------------------------------------------------------------------------------------
template <typename S>
__global__ void __launch_bounds__(256) transpose_kernel(const S *__restrict__ x, S *__restrict__ y,
                                                        int R, size_t Ccols) {
    // x: (batch, R, Ccols) -> y: (batch, Ccols, R); pure data movement on the raw element bits
    __shared__ S tile[32][33];
    const size_t b = blockIdx.z;
    const size_t c0 = (size_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const S *xb = x + b * (size_t)R * Ccols;
    S *yb = y + b * (size_t)R * Ccols;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k;
        const size_t c = c0 + tx;
        tile[k][tx] = (r < R && c < Ccols) ? xb[(size_t)r * Ccols + c] : (S)0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const size_t c = c0 + k;
        const int r = r0 + tx;
        if (r < R && c < Ccols) yb[c * R + r] = tile[tx][k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Batch assembly (reference ArrayDataGenerator.generate, DLWP/model/generators.py:872-984): the whole
// (time, variable, space) data array lives in HBM and a training batch is ONE gather
//     out[b][s][c_off + n*c_stride + j] = array[samples[b] + t_off + n*t_stride][var_idx[j]][s]        (channels_last)
//     out[b][c_off + n*c_stride + j][s] = ...                                                          (channels_first)
// instead of host fancy-indexing + transpose + upload.  Source rows are contiguous in s, the channels_last destination
// is contiguous in c: 64 pixels x all gathered channels go through a padded LDS tile so both sides are coalesced.
// ---------------------------------------------------------------------------------------------------------------
template <typename OT>
__global__ void __launch_bounds__(256) batch_gather_cl_kernel(const float *__restrict__ array, size_t S, int V,
                                                              const int32_t *__restrict__ samples,
                                                              const int32_t *__restrict__ var_idx, int nv, int n_steps,
                                                              int t_off, int t_stride, OT *__restrict__ out, int Ctot,
                                                              int c_off, int c_stride) {
    extern __shared__ float tile[];                 // [nch][65]
    const int nch = n_steps * nv;
    const size_t s0 = (size_t)blockIdx.x * 64;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long t0 = (long)samples[b] + t_off;
    for (int cc = w; cc < nch; cc += 4) {
        const int n = cc / nv, j = cc - n * nv;
        const float *src = array + ((size_t)(t0 + (long)n * t_stride) * V + var_idx[j]) * S;
        tile[cc * 65 + lane] = (s0 + lane < S) ? src[s0 + lane] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * nch; idx += 256) {
        const int px = idx / nch, cc = idx - px * nch;
        if (s0 + px >= S) break;
        const int n = cc / nv, j = cc - n * nv;
        Acc<1> v; v.v[0] = tile[cc * 65 + px];
        VT<OT>::st(out + ((size_t)b * S + s0 + px) * Ctot + c_off + n * c_stride + j, v);
    }
}

// The same for the common case "the gathered channels ARE the output row" (c_off = 0, c_stride = nv, Ctot = n_steps * nv; S % 4 == 0,
// 16-B aligned array rows): a workgroup moves 256 pixels x all channels -- 16-B loads (1 KB contiguous per wave and channel row
// instead of 256 B), and the tile's 256 x nch outputs are ONE contiguous block of `out`, written as 16-B vectors.  Round 6: the two
// gathers of a generator-fed training step took 26 us each (1.2-1.9 TB/s); they are what a step fed by the HBM-resident
// ArrayDataGenerator costs beyond the step itself.
template <typename OT>
__global__ void __launch_bounds__(256) batch_gather_cl_rows_kernel(const float *__restrict__ array, size_t S, int V,
                                                                   const int32_t *__restrict__ samples,
                                                                   const int32_t *__restrict__ var_idx, int nv, int n_steps,
                                                                   int t_off, int t_stride, OT *__restrict__ out) {
    extern __shared__ float tile[];                 // [nch][257]
    const int nch = n_steps * nv;
    const size_t s0 = (size_t)blockIdx.x * 256;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long t0 = (long)samples[b] + t_off;
    const int npx = (int)(S - s0 < 256 ? S - s0 : 256);             // (a multiple of 4)
    for (int cc = w; cc < nch; cc += 4) {
        const int n = cc / nv, j = cc - n * nv;
        const float *src = array + ((size_t)(t0 + (long)n * t_stride) * V + var_idx[j]) * S + s0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane * 4 < npx) v = *reinterpret_cast<const float4 *>(src + lane * 4);
        float *t = tile + cc * 257 + lane * 4;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
    constexpr int EV = 16 / (int)sizeof(OT);        // output elements per 16-B store: 8 bf16 / 4 fp32
    const int total = npx * nch;                    // (a multiple of EV: npx % 4 == 0, and bf16 rows need nch % 2 == 0 -- host)
    OT *dst = out + ((size_t)b * S + s0) * nch;
    for (int e0 = threadIdx.x * EV; e0 < total; e0 += 256 * EV) {
        int px = e0 / nch, cc = e0 - px * nch;
        float v[EV];
#pragma unroll
        for (int k = 0; k < EV; ++k) {
            v[k] = tile[cc * 257 + px];
            if (++cc == nch) { cc = 0; ++px; }
        }
        if constexpr (sizeof(OT) == 2) {
            *reinterpret_cast<uint4 *>(dst + e0) = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
        } else {
            *reinterpret_cast<float4 *>(dst + e0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <typename OT>
__global__ void __launch_bounds__(256) batch_gather_cf_kernel(const float *__restrict__ array, size_t S, int V,
                                                              const int32_t *__restrict__ samples,
                                                              const int32_t *__restrict__ var_idx, int nv, int n_steps,
                                                              int t_off, int t_stride, OT *__restrict__ out, int Ctot,
                                                              int c_off, int c_stride) {
    (void)n_steps;                                  // grid.y enumerates the n_steps * nv gathered channels
    const int cc = blockIdx.y, b = blockIdx.z;
    const int n = cc / nv, j = cc - n * nv;
    const float *src = array + ((size_t)((long)samples[b] + t_off + (long)n * t_stride) * V + var_idx[j]) * S;
    OT *dst = out + ((size_t)b * Ctot + c_off + n * c_stride + j) * S;
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (size_t)gridDim.x * blockDim.x) {
        Acc<1> v; v.v[0] = src[s];
        VT<OT>::st(dst + s, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// keras 'mse' (+ 'mae' metric) with gradient, two-stage fixed-order reduction          (Azure/train_cs.py:424-430)
// ---------------------------------------------------------------------------------------------------------------
constexpr int MSE_BLOCKS = 1024;

template <typename S, typename TT>
__global__ void __launch_bounds__(256) mse_stage1_kernel(const S *__restrict__ y, const TT *__restrict__ t,
                                                         S *__restrict__ dy, float *__restrict__ partial, size_t n,
                                                         float gscale) {
    float sq = 0.f, ab = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = VT<S>::ld(y + i).v[0] - VT<TT>::ld(t + i).v[0];
        sq += d * d;
        ab += fabsf(d);
        if (dy) { Acc<1> g; g.v[0] = gscale * d; VT<S>::st(dy + i, g); }
    }
    __shared__ float s_sq[256], s_ab[256];
    s_sq[threadIdx.x] = sq; s_ab[threadIdx.x] = ab;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s_sq[threadIdx.x] += s_sq[threadIdx.x + s]; s_ab[threadIdx.x] += s_ab[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sq[0]; partial[2 * blockIdx.x + 1] = s_ab[0]; }
}

// 8 elements per lane and iteration (n % 8 == 0): YV / TV = storage vectors of 8 predictions / 8 targets
template <typename YV, typename TV8>
__global__ void __launch_bounds__(256) mse_stage1_vec_kernel(const YV *__restrict__ y, const TV8 *__restrict__ t,
                                                             YV *__restrict__ dy, float *__restrict__ partial, size_t n8,
                                                             float gscale) {
    float sq = 0.f, ab = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const Acc<8> yv = VT<YV>::ld(y + i);
        const Acc<8> tv = VT<TV8>::ld(t + i);
        Acc<8> g;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = yv.v[k] - tv.v[k];
            sq += d * d;
            ab += fabsf(d);
            g.v[k] = gscale * d;
        }
        if (dy) VT<YV>::st(dy + i, g);
    }
    __shared__ float s_sq[256], s_ab[256];
    s_sq[threadIdx.x] = sq; s_ab[threadIdx.x] = ab;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s_sq[threadIdx.x] += s_sq[threadIdx.x + s]; s_ab[threadIdx.x] += s_ab[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sq[0]; partial[2 * blockIdx.x + 1] = s_ab[0]; }
}

__global__ void __launch_bounds__(256) mse_stage2_kernel(const float *__restrict__ partial, float *__restrict__ loss_out,
                                                         int nblocks, float inv_n, float weight, int overwrite) {
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

// ---------------------------------------------------------------------------------------------------------------
// TF2.1-keras Adam on flat buffers
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, size_t n,
                                                   const int32_t *__restrict__ step, float lr, float b1, float b2,
                                                   float eps, float gscale) {
    const float t = (float)(*step + 1);
    const float lr_t = adam_lr_t(lr, b1, b2, t);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i] * gscale, mi, vi, lr_t, b1, b2, eps);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}
__global__ void step_inc_kernel(int32_t *step) { *step += 1; }

// One launch: state = {t - 1, ticket}.  Every workgroup reads t - 1 when it starts; the last one to FINISH (ticket ==
// gridDim - 1: all others have read it by then) increments it and clears the ticket.  ZERO: g is cleared once consumed.
// hyper != nullptr (dlwpcs_adam_step_dev): {lr, beta1, beta2, eps, grad_scale} are read from device memory, so a captured
// hipGraph honours learning-rate changes between replays (by-value kernel arguments are frozen at capture time).
template <bool ZERO, int VEC>
__global__ void __launch_bounds__(256) adam_fused_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, size_t n, int32_t *state, float lr, float b1,
                                                         float b2, float eps, float gscale, const float *__restrict__ hyper) {
    typedef float VT4 __attribute__((ext_vector_type(VEC)));
    if (hyper != nullptr) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; gscale = hyper[4]; }
    const int32_t t0 = *(volatile int32_t *)state;
    const float t = (float)(t0 + 1);
    const float lr_t = adam_lr_t(lr, b1, b2, t);
    const size_t nv = n / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        VT4 gi = reinterpret_cast<const VT4 *>(g)[i], mi = reinterpret_cast<const VT4 *>(m)[i];
        VT4 vi = reinterpret_cast<const VT4 *>(v)[i], pi = reinterpret_cast<const VT4 *>(p)[i];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float pk = pi[k], mk = mi[k], vk = vi[k];
            adam_elem(pk, gi[k] * gscale, mk, vk, lr_t, b1, b2, eps);
            pi[k] = pk; mi[k] = mk; vi[k] = vk;
        }
        reinterpret_cast<VT4 *>(m)[i] = mi; reinterpret_cast<VT4 *>(v)[i] = vi; reinterpret_cast<VT4 *>(p)[i] = pi;
        if (ZERO) reinterpret_cast<VT4 *>(g)[i] = (VT4)0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = atomicAdd(state + 1, 1);
        if (done == (int)gridDim.x - 1) {
            state[1] = 0;
            state[0] = t0 + 1;
        }
    }
}

}  // namespace dlwpcs

using namespace dlwpcs;

#define REQUIRE(cond, ...) do { if (!(cond)) return fail(DLWPCS_E_INVALID, __VA_ARGS__); } while (0)
#define REQUIRE_DTYPE(dt, who) do { if (!dtype_ok(dt)) return fail(DLWPCS_E_UNSUPPORTED, who ": dtype %d not built", (int)(dt)); } while (0)

extern "C" int dlwpcs_pad_fwd(const void *x, void *y, int B, int N, int C, int p, int dtype,
                              const int32_t *table_dev, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "pad_fwd");
    REQUIRE(x && y && table_dev, "pad_fwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1 && p >= 0 && p <= N, "pad_fwd: bad shape B=%d N=%d C=%d p=%d", B, N, C, p);
    if (B == 0) return DLWPCS_OK;
    const int M = N + 2 * p;
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = (size_t)C * dtype_size(dtype);
    dispatch_mover(row_bytes, [&](auto tag, int w) {
        using V = decltype(tag);
        const int CV = (int)(row_bytes / w);
        const size_t total = (size_t)B * 6 * M * M * CV;
        hipLaunchKernelGGL(pad_fwd_kernel<V>, stream_grid(total), dim3(256), 0, s, (const V *)x, (V *)y, table_dev, total,
                           CV, 6 * N * N, 6 * M * M);
    });
    return check_launch("pad_fwd");
}

extern "C" int dlwpcs_pad_bwd(const void *dy, void *dx, int B, int N, int C, int p, int dtype,
                              const int32_t *inv_table_dev, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "pad_bwd");
    REQUIRE(dy && dx && inv_table_dev, "pad_bwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1 && p >= 0 && p <= N, "pad_bwd: bad shape B=%d N=%d C=%d p=%d", B, N, C, p);
    if (B == 0) return DLWPCS_OK;
    hipStream_t s = (hipStream_t)stream;
    dispatch_vec(dtype, C, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total = (size_t)B * 6 * N * N * (C / w);
        hipLaunchKernelGGL(pad_bwd_kernel<V>, stream_grid(total), dim3(256), 0, s, (const V *)dy, (V *)dx, inv_table_dev,
                           total, C / w, N, p);
    });
    return check_launch("pad_bwd");
}

namespace dlwpcs {
// used by conv_bwd_data: gradient of one virtual-input source out of the (padded or plain) virtual-input gradient
int launch_mask_inplace(void *dx, const void *m, size_t n, float alpha, float vmax, int dtype, hipStream_t s);

// msrc (halo only): the source itself, the gradient is multiplied by act'(msrc); returns through *masked whether it was
int launch_src_grad(const void *dxv, void *dsrc, const int32_t *inv, int B, int N, int CT, int choff, int CS, int up,
                    int halo, int dtype, hipStream_t s, const void *msrc, float m_alpha, float m_vmax, int *masked) {
    if (masked) *masked = msrc ? 1 : 0;
    const int No = up ? N / 2 : N;
    // common divisor of the three channel counts decides the vector width
    int g = 8;
    while (g > 1 && (CT % g || choff % g || CS % g)) g >>= 1;
    dispatch_vec(dtype, g, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total = (size_t)B * 6 * No * No * (CS / w);
        if (halo)
            hipLaunchKernelGGL(pad_bwd_src_kernel<V>, stream_grid(total), dim3(256), 0, s, (const V *)dxv, (V *)dsrc, inv,
                               total, CS / w, CT / w, choff / w, N, up, (const V *)msrc, m_alpha, m_vmax);
        else
            hipLaunchKernelGGL(window_src_kernel<V>, stream_grid(total), dim3(256), 0, s, (const V *)dxv, (V *)dsrc, total,
                               CS / w, CT / w, choff / w, N, up, (const V *)msrc, m_alpha, m_vmax);
    });
    return check_launch("src_grad");
}

// source 0 through the full inverse gather (upsampled source), source 1 through the ring fix-up, one launch
int launch_src_pair(const void *dxv, void *dsrc0, void *dsrc1, const int32_t *inv, int B, int N, int C0, int C1, int up0,
                    int dtype, hipStream_t s, const void *m0, const void *m1, float m_alpha, float m_vmax) {
    const int CT = C0 + C1;
    int g = 8;
    while (g > 1 && (C0 % g || C1 % g)) g >>= 1;
    const int No = up0 ? N / 2 : N;
    const int nb = N > 1 ? 4 * N - 4 : 1;
    dispatch_vec(dtype, g, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total0 = (size_t)B * 6 * No * No * (C0 / w), total1 = (size_t)B * 6 * nb * (C1 / w);
        const unsigned nb0 = stream_grid(total0).x, nb1 = stream_grid(total1).x;
        hipLaunchKernelGGL(src_pair_kernel<V>, dim3(nb0 + nb1), dim3(256), 0, s, (const V *)dxv, (V *)dsrc0, (V *)dsrc1, inv,
                           total0, total1, C0 / w, C1 / w, CT / w, N, up0, nb0, (const V *)m0, (const V *)m1, m_alpha, m_vmax);
    });
    return check_launch("src_pair");
}

// border fix-up of a source whose interior gradient was written directly by the data-gradient kernel (halo, p = 1, no
// upsampling): adds the halo-ring cells of dxv that gathered from each border cell
int launch_ring_fix(const void *dxv, void *dsrc, const int32_t *inv, int B, int N, int CT, int choff, int CS, int dtype,
                    hipStream_t s, const void *msrc, float m_alpha, float m_vmax) {
    int g = 8;
    while (g > 1 && (CT % g || choff % g || CS % g)) g >>= 1;
    const int nb = N > 1 ? 4 * N - 4 : 1;
    dispatch_vec(dtype, g, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total = (size_t)B * 6 * nb * (CS / w);
        hipLaunchKernelGGL(pad_ring_fix_kernel<V>, stream_grid(total), dim3(256), 0, s, (const V *)dxv, (V *)dsrc, inv, total,
                           CS / w, CT / w, choff / w, N, (const V *)msrc, m_alpha, m_vmax);
    });
    return check_launch("ring_fix");
}

// dx *= act'(m), in place (the fallback of the pre-masked gradient convention where no kernel fuses the multiply)
int launch_mask_inplace(void *dx, const void *m, size_t n, float alpha, float vmax, int dtype, hipStream_t s) {
    if (n == 0) return DLWPCS_OK;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL((act_bwd_kernel<H8, bf16_t>), stream_grid(n / 8 + 1), dim3(256), 0, s, (const bf16_t *)dx,
                           (const bf16_t *)m, (bf16_t *)dx, n, alpha, vmax);
    else
        hipLaunchKernelGGL((act_bwd_kernel<float4, float>), stream_grid(n / 4 + 1), dim3(256), 0, s, (const float *)dx,
                           (const float *)m, (float *)dx, n, alpha, vmax);
    return check_launch("mask_inplace");
}
}  // namespace dlwpcs

extern "C" int dlwpcs_act_fwd(const void *x, void *y, size_t n, int act, float alpha, float vmax, int dtype,
                              dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "act_fwd");
    REQUIRE(x && y, "act_fwd: null pointer");
    REQUIRE(act == DLWPCS_ACT_LEAKY_CLIP, "act_fwd: unknown activation %d", act);
    if (n == 0) return DLWPCS_OK;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL((act_fwd_kernel<H8, bf16_t>), stream_grid(n / 8 + 1), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t *)x, (bf16_t *)y, n, alpha, vmax);
    else
        hipLaunchKernelGGL((act_fwd_kernel<float4, float>), stream_grid(n / 4 + 1), dim3(256), 0, (hipStream_t)stream,
                           (const float *)x, (float *)y, n, alpha, vmax);
    return check_launch("act_fwd");
}

extern "C" int dlwpcs_act_bwd(const void *dy, const void *y, void *dx, size_t n, int act, float alpha, float vmax,
                              int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "act_bwd");
    REQUIRE(dy && y && dx, "act_bwd: null pointer");
    REQUIRE(act == DLWPCS_ACT_LEAKY_CLIP, "act_bwd: unknown activation %d", act);
    if (n == 0) return DLWPCS_OK;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL((act_bwd_kernel<H8, bf16_t>), stream_grid(n / 8 + 1), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t *)dy, (const bf16_t *)y, (bf16_t *)dx, n, alpha, vmax);
    else
        hipLaunchKernelGGL((act_bwd_kernel<float4, float>), stream_grid(n / 4 + 1), dim3(256), 0, (hipStream_t)stream,
                           (const float *)dy, (const float *)y, (float *)dx, n, alpha, vmax);
    return check_launch("act_bwd");
}

#define POOL_LAUNCH(KERNEL, IN, OUT, TOTAL_PIX, NARG)                                                                  \
    dispatch_vec(dtype, C, [&](auto tag, int w) {                                                                      \
        using V = decltype(tag);                                                                                       \
        const size_t total = (size_t)(TOTAL_PIX) * (C / w);                                                            \
        hipLaunchKernelGGL(KERNEL<V>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const V *)(IN),          \
                           (V *)(OUT), total, C / w, NARG);                                                            \
    });

extern "C" int dlwpcs_avgpool2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "avgpool2_fwd");
    REQUIRE(x && y, "avgpool2_fwd: null pointer");
    REQUIRE(B >= 0 && N >= 2 && N % 2 == 0 && C >= 1, "avgpool2_fwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(avgpool2_fwd_kernel, x, y, (size_t)B * 6 * (N / 2) * (N / 2), N)
    return check_launch("avgpool2_fwd");
}
extern "C" int dlwpcs_avgpool2_bwd(const void *dy, void *dx, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "avgpool2_bwd");
    REQUIRE(dy && dx, "avgpool2_bwd: null pointer");
    REQUIRE(B >= 0 && N >= 2 && N % 2 == 0 && C >= 1, "avgpool2_bwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(avgpool2_bwd_kernel, dy, dx, (size_t)B * 6 * N * N, N)
    return check_launch("avgpool2_bwd");
}
extern "C" int dlwpcs_avgpool2_bwd_add(const void *dy, const void *dskip, void *dx, int B, int N, int C, int dtype,
                                       dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "avgpool2_bwd_add");
    REQUIRE(dy && dskip && dx, "avgpool2_bwd_add: null pointer");
    REQUIRE(B >= 0 && N >= 2 && N % 2 == 0 && C >= 1, "avgpool2_bwd_add: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    dispatch_vec(dtype, C, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total = (size_t)B * 6 * N * N * (C / w);
        hipLaunchKernelGGL(avgpool2_bwd_add_kernel<V>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const V *)dy,
                           (const V *)dskip, (V *)dx, total, C / w, N);
    });
    return check_launch("avgpool2_bwd_add");
}
static int avgpool2_bwd_masked_impl(const void *dy, const void *dskip, const void *m, void *dx, int B, int N, int C,
                                    float m_alpha, float m_vmax, int dtype, const void *ring, const int32_t *inv, int ring_channels,
                                    int ring_choff, dlwpcs_stream_t stream);

extern "C" int dlwpcs_avgpool2_bwd_masked(const void *dy, const void *dskip, const void *m, void *dx, int B, int N, int C,
                                          float m_alpha, float m_vmax, int dtype, dlwpcs_stream_t stream) {
    return avgpool2_bwd_masked_impl(dy, dskip, m, dx, B, N, C, m_alpha, m_vmax, dtype, nullptr, nullptr, 0, 0, stream);
}

extern "C" int dlwpcs_avgpool2_bwd_ring(const void *dy, const void *dskip, const void *m, void *dx, int B, int N, int C,
                                        float m_alpha, float m_vmax, int dtype, const void *ring, const int32_t *inv_half_dev,
                                        int ring_channels, int ring_choff, dlwpcs_stream_t stream) {
    REQUIRE(ring && inv_half_dev, "avgpool2_bwd_ring: null ring / inverse table");
    REQUIRE(ring_channels >= C && ring_choff >= 0 && ring_choff + C <= ring_channels, "avgpool2_bwd_ring: channel window %d + %d of %d",
            ring_choff, C, ring_channels);
    return avgpool2_bwd_masked_impl(dy, dskip, m, dx, B, N, C, m_alpha, m_vmax, dtype, ring, inv_half_dev, ring_channels, ring_choff,
                                    stream);
}

static int avgpool2_bwd_masked_impl(const void *dy, const void *dskip, const void *m, void *dx, int B, int N, int C,
                                    float m_alpha, float m_vmax, int dtype, const void *ring, const int32_t *inv, int ring_channels,
                                    int ring_choff, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "avgpool2_bwd_masked");
    REQUIRE(dy && dx && (m || ring), "avgpool2_bwd_masked: null pointer");
    REQUIRE(B >= 0 && N >= 2 && N % 2 == 0 && C >= 1, "avgpool2_bwd_masked: bad shape B=%d N=%d C=%d", B, N, C);
    REQUIRE(m_alpha >= 0.f && m_vmax >= 0.f, "avgpool2_bwd_masked: activation needs negative_slope >= 0 and max_value >= 0");
    if (B == 0) return DLWPCS_OK;
    int g = 8;
    while (g > 1 && (C % g || (ring && (ring_channels % g || ring_choff % g)))) g >>= 1;
    dispatch_vec(dtype, g, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total = (size_t)B * 6 * N * N * (C / w);
        hipLaunchKernelGGL(avgpool2_bwd_masked_kernel<V>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const V *)dy,
                           (const V *)dskip, (const V *)m, (V *)dx, total, C / w, N, m_alpha, m_vmax, (const V *)ring, inv,
                           ring_channels / w, ring_choff / w);
    });
    return check_launch("avgpool2_bwd_masked");
}
extern "C" int dlwpcs_upsample2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "upsample2_fwd");
    REQUIRE(x && y, "upsample2_fwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1, "upsample2_fwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    const size_t row_bytes = (size_t)C * dtype_size(dtype);
    dispatch_mover(row_bytes, [&](auto tag, int w) {
        using V = decltype(tag);
        const int CV = (int)(row_bytes / w);
        const size_t total = (size_t)B * 6 * (2 * N) * (2 * N) * CV;
        hipLaunchKernelGGL(upsample2_fwd_kernel<V>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const V *)x,
                           (V *)y, total, CV, N);
    });
    return check_launch("upsample2_fwd");
}
extern "C" int dlwpcs_upsample2_bwd(const void *dy, void *dx, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "upsample2_bwd");
    REQUIRE(dy && dx, "upsample2_bwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1, "upsample2_bwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(upsample2_bwd_kernel, dy, dx, (size_t)B * 6 * N * N, N)
    return check_launch("upsample2_bwd");
}

extern "C" int dlwpcs_concat2(const void *a, const void *b, void *y, size_t rows, int Ca, int Cb, int dtype,
                              dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "concat2");
    REQUIRE(a && b && y && Ca >= 1 && Cb >= 1, "concat2: bad arguments");
    if (rows == 0) return DLWPCS_OK;
    const size_t ba = (size_t)Ca * dtype_size(dtype), bb = (size_t)Cb * dtype_size(dtype);
    dispatch_mover(ba | bb, [&](auto tag, int w) {      // (ba | bb) % w == 0  <=>  both divisible (w a power of two)
        using V = decltype(tag);
        const size_t total = rows * ((ba + bb) / w);
        hipLaunchKernelGGL(concat2_kernel<V>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const V *)a,
                           (const V *)b, (V *)y, total, (int)(ba / w), (int)(bb / w));
    });
    return check_launch("concat2");
}

extern "C" int dlwpcs_state_repack(const void *state, const void *extra, void *out, int B, size_t S, int T, int V, int E,
                                   int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "state_repack");
    REQUIRE(state && extra && out, "state_repack: null pointer");
    REQUIRE(B >= 0 && S >= 1 && T >= 1 && V >= 1 && E >= 1, "state_repack: bad shape B=%d S=%zu T=%d V=%d E=%d", B, S, T, V, E);
    REQUIRE(S < (1ull << 32), "state_repack: S too large");
    if (B == 0) return DLWPCS_OK;
    const size_t total = (size_t)B * S * T * (V + E);
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(state_repack_kernel<uint16_t>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)state, (const uint16_t *)extra, (uint16_t *)out, total, (unsigned)S, (unsigned)T,
                           (unsigned)V, (unsigned)E);
    else
        hipLaunchKernelGGL(state_repack_kernel<uint32_t>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)state, (const uint32_t *)extra, (uint32_t *)out, total, (unsigned)S, (unsigned)T,
                           (unsigned)V, (unsigned)E);
    return check_launch("state_repack");
}

extern "C" int dlwpcs_pad_channels(const void *x, void *y, size_t rows, int C, int Cp, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "pad_channels");
    REQUIRE(x && y && C >= 1 && Cp >= C, "pad_channels: bad arguments C=%d Cp=%d", C, Cp);
    if (rows == 0) return DLWPCS_OK;
    const size_t total = rows * (size_t)Cp;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(pad_channels_kernel<uint16_t>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (uint16_t *)y, total, (unsigned)C, (unsigned)Cp);
    else
        hipLaunchKernelGGL(pad_channels_kernel<uint32_t>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)x, (uint32_t *)y, total, (unsigned)C, (unsigned)Cp);
    return check_launch("pad_channels");
}

extern "C" int dlwpcs_slice_channels(const void *y, void *x, size_t rows, int Cp, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "slice_channels");
    REQUIRE(x && y && C >= 1 && Cp >= C, "slice_channels: bad arguments C=%d Cp=%d", C, Cp);
    if (rows == 0) return DLWPCS_OK;
    const size_t total = rows * (size_t)C;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(slice_channels_kernel<uint16_t>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)y, (uint16_t *)x, total, (unsigned)C, (unsigned)Cp);
    else
        hipLaunchKernelGGL(slice_channels_kernel<uint32_t>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)y, (uint32_t *)x, total, (unsigned)C, (unsigned)Cp);
    return check_launch("slice_channels");
}

extern "C" int dlwpcs_split2(const void *y, void *a, void *b, size_t rows, int Ca, int Cb, int dtype,
                             dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "split2");
    REQUIRE(y && (a || b) && Ca >= 1 && Cb >= 1, "split2: bad arguments");
    if (rows == 0) return DLWPCS_OK;
    const size_t ba = (size_t)Ca * dtype_size(dtype), bb = (size_t)Cb * dtype_size(dtype);
    dispatch_mover(ba | bb, [&](auto tag, int w) {
        using V = decltype(tag);
        const size_t total = rows * ((ba + bb) / w);
        hipLaunchKernelGGL(split2_kernel<V>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const V *)y, (V *)a,
                           (V *)b, total, (int)(ba / w), (int)(bb / w));
    });
    return check_launch("split2");
}

template <typename S>
static int launch_transpose_t(const void *x, void *y, dim3 grid, int R, size_t Ccols, hipStream_t s, const char *who) {
    hipLaunchKernelGGL(transpose_kernel<S>, grid, dim3(256), 0, s, (const S *)x, (S *)y, R, Ccols);
    return check_launch(who);
}

static int launch_transpose(const void *x, void *y, int batch, size_t R, size_t Ccols, int dtype, hipStream_t s,
                            const char *who) {
    if (batch == 0 || R == 0 || Ccols == 0) return DLWPCS_OK;
    if (R > 0x7fffffffu) return fail(DLWPCS_E_INVALID, "%s: too many rows", who);
    dim3 grid((unsigned)((Ccols + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)batch);
    if (grid.y > 65535 || grid.z > 65535) return fail(DLWPCS_E_UNSUPPORTED, "%s: grid too large", who);
    return dtype == DLWPCS_BF16 ? launch_transpose_t<uint16_t>(x, y, grid, (int)R, Ccols, s, who)
                                : launch_transpose_t<uint32_t>(x, y, grid, (int)R, Ccols, s, who);
}

extern "C" int dlwpcs_cf_to_cl(const void *x, void *y, int B, int C, size_t S, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "cf_to_cl");
    REQUIRE(x && y && B >= 0 && C >= 1, "cf_to_cl: bad arguments");
    return launch_transpose(x, y, B, (size_t)C, S, dtype, (hipStream_t)stream, "cf_to_cl");   // (B,C,S) -> (B,S,C)
}
extern "C" int dlwpcs_cl_to_cf(const void *x, void *y, int B, int C, size_t S, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "cl_to_cf");
    REQUIRE(x && y && B >= 0 && C >= 1, "cl_to_cf: bad arguments");
    return launch_transpose(x, y, B, S, (size_t)C, dtype, (hipStream_t)stream, "cl_to_cf");   // (B,S,C) -> (B,C,S)
}

extern "C" int dlwpcs_add(const void *a, const void *b, void *y, size_t n, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "add");
    REQUIRE(a && b && y, "add: null pointer");
    if (n == 0) return DLWPCS_OK;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL((add_kernel<H8, bf16_t>), stream_grid(n / 8 + 1), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t *)a, (const bf16_t *)b, (bf16_t *)y, n);
    else
        hipLaunchKernelGGL((add_kernel<float4, float>), stream_grid(n / 4 + 1), dim3(256), 0, (hipStream_t)stream,
                           (const float *)a, (const float *)b, (float *)y, n);
    return check_launch("add");
}

// ---------------------------------------------------------------------------------------------------------------
// Weight regularizers and constraints of CubeSphereConv2D (DLWP/custom.py:837-842, 898-914: passed to add_weight): keras
// regularizers.L1L2 -- penalty l1 * sum|w| + l2 * sum w^2 added to the loss, its gradient l1 * sign(w) + 2 * l2 * w to the weight's
// gradient -- and keras constraints MaxNorm / NonNeg / UnitNorm / MinMaxNorm applied to the weight after every update (norms over
// the LEADING axes: the weight as a (rows, cols) matrix, one norm per column; keras epsilon 1e-7).  fp32 master weights only;
// ONE workgroup per call / per column: fixed summation order, reproducible.  Off the hot path (no reference script uses them).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) l1l2_kernel(const float *__restrict__ w, float *__restrict__ g, size_t n, float l1, float l2,
                                                    float inv_grad_scale, float *__restrict__ penalty) {
    __shared__ float red[1024];
    float s1 = 0.f, s2 = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const float x = w[i];
        s1 += fabsf(x);
        s2 += x * x;
        if (g) g[i] += (l1 * (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) + 2.f * l2 * x) * inv_grad_scale;
    }
    red[threadIdx.x] = l1 * s1 + l2 * s2;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0 && penalty) *penalty += red[0];
}

// kind: 1 MaxNorm(a) | 2 NonNeg | 3 UnitNorm | 4 MinMaxNorm(a = min, b = max, rate)
__global__ void __launch_bounds__(256) weight_constraint_kernel(float *__restrict__ w, int rows, int cols, int kind, float a, float b,
                                                                float rate) {
    const int c = blockIdx.x;
    if (kind == 2) {
        for (int r = threadIdx.x; r < rows; r += 256) { const float x = w[(size_t)r * cols + c]; w[(size_t)r * cols + c] = x >= 0.f ? x : 0.f; }
        return;
    }
    __shared__ float red[256];
    float s = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) { const float x = w[(size_t)r * cols + c]; s += x * x; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const float norm = sqrtf(red[0]), eps = 1e-7f;
    float scale;
    if (kind == 1) scale = fminf(fmaxf(norm, 0.f), a) / (eps + norm);
    else if (kind == 3) scale = 1.f / (eps + norm);
    else scale = (rate * fminf(fmaxf(norm, a), b) + (1.f - rate) * norm) / (eps + norm);
    for (int r = threadIdx.x; r < rows; r += 256) w[(size_t)r * cols + c] *= scale;
}

extern "C" int dlwpcs_l1l2_regularize(const float *w, float *g, size_t n, float l1, float l2, float inv_grad_scale, float *penalty,
                                      dlwpcs_stream_t stream) {
    REQUIRE(w && (g || penalty), "l1l2_regularize: null pointer");
    REQUIRE(l1 >= 0.f && l2 >= 0.f, "l1l2_regularize: negative factor (l1=%g, l2=%g)", l1, l2);
    if (n == 0) return DLWPCS_OK;
    hipLaunchKernelGGL(l1l2_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, g, n, l1, l2, inv_grad_scale, penalty);
    return check_launch("l1l2_regularize");
}

extern "C" int dlwpcs_weight_constraint(float *w, int rows, int cols, int kind, float a, float b, float rate, dlwpcs_stream_t stream) {
    REQUIRE(w, "weight_constraint: null pointer");
    REQUIRE(rows >= 1 && cols >= 1, "weight_constraint: bad shape %d x %d", rows, cols);
    REQUIRE(kind >= DLWPCS_CONSTRAINT_MAX_NORM && kind <= DLWPCS_CONSTRAINT_MIN_MAX_NORM, "weight_constraint: unknown kind %d", kind);
    hipLaunchKernelGGL(weight_constraint_kernel, dim3(cols), dim3(256), 0, (hipStream_t)stream, w, rows, cols, kind, a, b, rate);
    return check_launch("weight_constraint");
}

extern "C" size_t dlwpcs_mse_scratch_bytes(void) { return (size_t)MSE_BLOCKS * 2 * sizeof(float); }

extern "C" int dlwpcs_mse_fwd_bwd(const void *y, const void *t, void *dy, float *loss_out, size_t n, float weight,
                                  int dtype, void *scratch, dlwpcs_stream_t stream) {
    const bool t_f32 = (dtype & DLWPCS_MSE_TARGET_F32) != 0;
    const int overwrite = (dtype & DLWPCS_MSE_OVERWRITE) ? 1 : 0;
    dtype &= ~(DLWPCS_MSE_TARGET_F32 | DLWPCS_MSE_OVERWRITE);
    REQUIRE_DTYPE(dtype, "mse_fwd_bwd");
    REQUIRE(y && t && loss_out && scratch && n > 0, "mse_fwd_bwd: bad arguments");
    size_t g = (n + 255) / 256;
    if (g > MSE_BLOCKS) g = MSE_BLOCKS;
    const float gscale = weight * 2.f / (float)n;
    const bool al = (((uintptr_t)y | (uintptr_t)t | (uintptr_t)dy) & 31) == 0;
    if (n % 8 == 0 && al) {
        const size_t n8 = n / 8;
        g = (n8 + 255) / 256;
        if (g > MSE_BLOCKS) g = MSE_BLOCKS;
        if (dtype == DLWPCS_BF16 && t_f32)
            hipLaunchKernelGGL((mse_stage1_vec_kernel<H8, F8>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                               (const H8 *)y, (const F8 *)t, (H8 *)dy, (float *)scratch, n8, gscale);
        else if (dtype == DLWPCS_BF16)
            hipLaunchKernelGGL((mse_stage1_vec_kernel<H8, H8>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                               (const H8 *)y, (const H8 *)t, (H8 *)dy, (float *)scratch, n8, gscale);
        else
            hipLaunchKernelGGL((mse_stage1_vec_kernel<F8, F8>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                               (const F8 *)y, (const F8 *)t, (F8 *)dy, (float *)scratch, n8, gscale);
    } else if (dtype == DLWPCS_BF16 && t_f32)
        hipLaunchKernelGGL((mse_stage1_kernel<bf16_t, float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t *)y, (const float *)t, (bf16_t *)dy, (float *)scratch, n, gscale);
    else if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL((mse_stage1_kernel<bf16_t, bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t *)y, (const bf16_t *)t, (bf16_t *)dy, (float *)scratch, n, gscale);
    else
        hipLaunchKernelGGL((mse_stage1_kernel<float, float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                           (const float *)y, (const float *)t, (float *)dy, (float *)scratch, n, gscale);
    hipLaunchKernelGGL(mse_stage2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)scratch, loss_out,
                       (int)g, 1.f / (float)n, weight, overwrite);
    return check_launch("mse_fwd_bwd");
}

extern "C" int dlwpcs_adam_step(float *p, const float *g, float *m, float *v, size_t n, int32_t *step_dev, float lr,
                                float beta1, float beta2, float eps, float grad_scale, dlwpcs_stream_t stream) {
    REQUIRE(p && g && m && v && step_dev, "adam_step: null pointer");
    if (n > 0)
        hipLaunchKernelGGL(adam_kernel, stream_grid(n), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_dev, lr,
                           beta1, beta2, eps, grad_scale);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    return check_launch("adam_step");
}

static int adam_fused_launch(float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev, float lr,
                             float beta1, float beta2, float eps, float grad_scale, const float *hyper_dev, int flags,
                             dlwpcs_stream_t stream) {
    REQUIRE(p && g && m && v && state_dev, "adam_step_fused: null pointer");
    REQUIRE((flags & ~DLWPCS_ADAM_ZERO_GRAD) == 0, "adam_step_fused: unknown flags %d", flags);
    if (n == 0) {
        hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev);
        return check_launch("adam_step_fused");
    }
    // few, fat workgroups: every workgroup costs one serialised ticket atomic at the end
    const bool vec = n % 4 == 0 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const size_t items = vec ? n / 4 : n;
    unsigned grid = (unsigned)((items + 255) / 256);
    if (grid > 256) grid = 256;
    const bool zero = (flags & DLWPCS_ADAM_ZERO_GRAD) != 0;
#define ADAM_LAUNCH(Z, V)                                                                                              \
    hipLaunchKernelGGL((adam_fused_kernel<Z, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, state_dev,  \
                       lr, beta1, beta2, eps, grad_scale, hyper_dev)
    if (vec) { if (zero) ADAM_LAUNCH(true, 4); else ADAM_LAUNCH(false, 4); }
    else { if (zero) ADAM_LAUNCH(true, 1); else ADAM_LAUNCH(false, 1); }
#undef ADAM_LAUNCH
    return check_launch("adam_step_fused");
}

extern "C" int dlwpcs_adam_step_fused(float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev, float lr,
                                      float beta1, float beta2, float eps, float grad_scale, int flags,
                                      dlwpcs_stream_t stream) {
    return adam_fused_launch(p, g, m, v, n, state_dev, lr, beta1, beta2, eps, grad_scale, nullptr, flags, stream);
}

extern "C" int dlwpcs_adam_step_dev(float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev,
                                    const float *hyper_dev, int flags, dlwpcs_stream_t stream) {
    REQUIRE(hyper_dev, "adam_step_dev: null hyper-parameter buffer");
    return adam_fused_launch(p, g, m, v, n, state_dev, 0.f, 0.f, 0.f, 0.f, 0.f, hyper_dev, flags, stream);
}

extern "C" int dlwpcs_batch_gather(const void *array, int64_t T, int V, int64_t S, const int32_t *samples_dev, int B,
                                   const int32_t *var_idx_dev, int nv, int n_steps, int t_off, int t_stride, void *out,
                                   int Ctot, int c_off, int c_stride, int channels_last, int dtype,
                                   dlwpcs_stream_t stream) {
    REQUIRE_DTYPE(dtype, "batch_gather");
    REQUIRE(array && samples_dev && var_idx_dev && out, "batch_gather: null pointer");
    REQUIRE(T >= 1 && V >= 1 && S >= 1 && B >= 0 && nv >= 1 && n_steps >= 1 && Ctot >= 1 && c_off >= 0 && c_stride >= 0,
            "batch_gather: bad shape T=%lld V=%d S=%lld B=%d nv=%d n_steps=%d", (long long)T, V, (long long)S, B, nv, n_steps);
    REQUIRE(c_off + (n_steps - 1) * c_stride + nv <= Ctot, "batch_gather: channel window exceeds Ctot=%d", Ctot);
    if (B == 0) return DLWPCS_OK;
    if (B > 65535) return fail(DLWPCS_E_UNSUPPORTED, "batch_gather: batch > 65535");
    hipStream_t s = (hipStream_t)stream;
    const int nch = n_steps * nv;
    if (channels_last) {
        const size_t lds = (size_t)nch * 65 * sizeof(float);
        if (lds > 64 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "batch_gather: %d gathered channels exceed the LDS tile", nch);
        // whole output rows, 16-B aligned on both sides: the 256-pixel vector kernel
        const size_t lds_rows = (size_t)nch * 257 * sizeof(float);
        const int ev = dtype == DLWPCS_BF16 ? 8 : 4;
        if (c_off == 0 && c_stride == nv && Ctot == nch && S % 4 == 0 && (4 * nch) % ev == 0 && lds_rows <= 64 * 1024 &&
            ((uintptr_t)array & 15) == 0 && ((uintptr_t)out & 15) == 0) {
            dim3 grid_r((unsigned)((S + 255) / 256), (unsigned)B);
            if (dtype == DLWPCS_BF16)
                hipLaunchKernelGGL(batch_gather_cl_rows_kernel<bf16_t>, grid_r, dim3(256), lds_rows, s, (const float *)array, (size_t)S, V,
                                   samples_dev, var_idx_dev, nv, n_steps, t_off, t_stride, (bf16_t *)out);
            else
                hipLaunchKernelGGL(batch_gather_cl_rows_kernel<float>, grid_r, dim3(256), lds_rows, s, (const float *)array, (size_t)S, V,
                                   samples_dev, var_idx_dev, nv, n_steps, t_off, t_stride, (float *)out);
            return check_launch("batch_gather");
        }
        dim3 grid((unsigned)((S + 63) / 64), (unsigned)B);
        if (dtype == DLWPCS_BF16)
            hipLaunchKernelGGL(batch_gather_cl_kernel<bf16_t>, grid, dim3(256), lds, s, (const float *)array, (size_t)S, V,
                               samples_dev, var_idx_dev, nv, n_steps, t_off, t_stride, (bf16_t *)out, Ctot, c_off, c_stride);
        else
            hipLaunchKernelGGL(batch_gather_cl_kernel<float>, grid, dim3(256), lds, s, (const float *)array, (size_t)S, V,
                               samples_dev, var_idx_dev, nv, n_steps, t_off, t_stride, (float *)out, Ctot, c_off, c_stride);
    } else {
        if (nch > 65535) return fail(DLWPCS_E_UNSUPPORTED, "batch_gather: too many channels");
        size_t gx = (S + 255) / 256;
        if (gx > 64) gx = 64;
        dim3 grid((unsigned)gx, (unsigned)nch, (unsigned)B);
        if (dtype == DLWPCS_BF16)
            hipLaunchKernelGGL(batch_gather_cf_kernel<bf16_t>, grid, dim3(256), 0, s, (const float *)array, (size_t)S, V,
                               samples_dev, var_idx_dev, nv, n_steps, t_off, t_stride, (bf16_t *)out, Ctot, c_off, c_stride);
        else
            hipLaunchKernelGGL(batch_gather_cf_kernel<float>, grid, dim3(256), 0, s, (const float *)array, (size_t)S, V,
                               samples_dev, var_idx_dev, nv, n_steps, t_off, t_stride, (float *)out, Ctot, c_off, c_stride);
    }
    return check_launch("batch_gather");
}

// ------------------------------------------------------------------------------------------------------------------
// dlwpcs_lds_oob_probe: the gather-form data gradient (conv_ws.h, EDGE) masks operands by ADDRESS -- a lane that must contribute
// nothing to an MFMA reads its operand from beyond the workgroup's LDS allocation, where the hardware returns zeros (GCN / CDNA:
// out-of-range LDS reads return 0).  This probe reads 16 B per lane at the three offsets the kernel uses and counts non-zero dwords.
// ------------------------------------------------------------------------------------------------------------------
namespace dlwpcs {
__global__ void lds_oob_probe_kernel(int32_t *nonzero) {
    extern __shared__ __attribute__((aligned(16))) uint32_t probe_smem[];
    for (int i = threadIdx.x; i < 40 * 1024; i += blockDim.x) probe_smem[i] = 0xdead0000u + i;      // 160 KB, all non-zero
    __syncthreads();
    const uint32_t offs[3] = {0x00f00000u, 1u << 23, 0x8000u << 4};
    int nz = 0;
    for (int u = 0; u < 3; ++u) {
        uint4 v;
        const uint32_t addr = offs[u] + threadIdx.x * 16 + (u == 0 ? 8160u + 32u : 0u);
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        nz += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
    }
    // (and an in-range read must still see the data)
    uint4 w;
    const uint32_t inr = threadIdx.x * 16;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(inr) : "memory");
    if (w.x != 0xdead0000u + threadIdx.x * 4) nz += 1000;
    if (nz) atomicAdd(nonzero, nz);
}
}  // namespace dlwpcs

extern "C" int dlwpcs_lds_oob_probe(int32_t *nonzero_dev, dlwpcs_stream_t stream) {
    REQUIRE(nonzero_dev, "lds_oob_probe: null output");
    hipStream_t s = (hipStream_t)stream;
    const int lds = 160 * 1024;
    hipError_t e = hipFuncSetAttribute((const void *)lds_oob_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "lds_oob_probe: hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipMemsetAsync(nonzero_dev, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "lds_oob_probe: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(lds_oob_probe_kernel, dim3(4), dim3(256), lds, s, nonzero_dev);
    return check_launch("lds_oob_probe");
}
