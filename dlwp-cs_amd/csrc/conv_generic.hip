// Generic per-face convolution (any kernel size / stride / dilation / zero 'same' padding) for the OFF-hot-path options
// of CubeSphereConv2D (DLWP/custom.py:824-842: strides, dilation_rate, padding='same').  Plain VALU direct kernels --
// correctness-first, deterministic; the hot configuration (k in {1,3}, stride 1, dilation 1) runs on conv_mfma.hip.
// T = element type of the activations (x, y, dy, dx): float, or bf16_t (DLWPCS_BF16: fp32 accumulation, one rounding on
// store, the fp32 master weights rounded to bf16 on the fly exactly as the matrix-core path packs them); parameter
// gradients are always fp32.
//
// Face 5 follows the reference literally (DLWP/custom.py:965-996): the input rows are reversed, the convolution is
// applied, and the output rows are reversed again -- expressed here as index maps, never as data movement.
#include "common.h"

namespace dlwpcs {

struct GP {
    int B, H, W, Cin, Cout, kh, kw, sh, sw, dh, dw, pad_t, pad_l, Ho, Wo, flip;
};

__device__ __forceinline__ const float *group_ptr(int f, const float *eq, const float *pol, const float *np) {
    return f < 4 ? eq : (f == 4 ? pol : (np ? np : pol));
}
template <typename T> struct GT;
template <> struct GT<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
    static __device__ __forceinline__ float wt(float w) { return w; }
};
template <> struct GT<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t *p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = f2bf(v); }
    static __device__ __forceinline__ float wt(float w) { return bf2f(f2bf(w)); }
};

template <typename T>
__global__ void __launch_bounds__(256) gconv_fwd_kernel(GP g, const T *__restrict__ x, const float *__restrict__ w_eq,
                                                        const float *__restrict__ w_pol, const float *__restrict__ w_np,
                                                        const float *__restrict__ b_eq, const float *__restrict__ b_pol,
                                                        const float *__restrict__ b_np, T *__restrict__ y, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int co = r % g.Cout; r /= g.Cout;
        const int ox = r % g.Wo; r /= g.Wo;
        const int oy = r % g.Ho; r /= g.Ho;
        const int f = r % 6;
        const size_t b = r / 6;
        const bool fl = (f == 5) && g.flip;
        const float *w = group_ptr(f, w_eq, w_pol, w_np);
        const int oyf = fl ? g.Ho - 1 - oy : oy;       // row in the flipped frame
        float acc = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            int iy = oyf * g.sh - g.pad_t + ky * g.dh;
            if (iy < 0 || iy >= g.H) continue;
            if (fl) iy = g.H - 1 - iy;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int ix = ox * g.sw - g.pad_l + kx * g.dw;
                if (ix < 0 || ix >= g.W) continue;
                const T *xp = x + (((b * 6 + f) * g.H + iy) * g.W + ix) * g.Cin;
                const float *wp = w + ((size_t)(ky * g.kw + kx) * g.Cin) * g.Cout + co;
                for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(GT<T>::ld(xp + ci), GT<T>::wt(wp[(size_t)ci * g.Cout]), acc);
            }
        }
        if (b_eq) acc += group_ptr(f, b_eq, b_pol, b_np)[co];
        GT<T>::st(y + e, acc);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gconv_bwd_data_kernel(GP g, const T *__restrict__ dy, const float *__restrict__ w_eq,
                                                             const float *__restrict__ w_pol, const float *__restrict__ w_np,
                                                             T *__restrict__ dx, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int ci = r % g.Cin; r /= g.Cin;
        const int ix = r % g.W; r /= g.W;
        const int iy = r % g.H; r /= g.H;
        const int f = r % 6;
        const size_t b = r / 6;
        const bool fl = (f == 5) && g.flip;
        const float *w = group_ptr(f, w_eq, w_pol, w_np);
        const int iyf = fl ? g.H - 1 - iy : iy;
        float acc = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int ty = iyf + g.pad_t - ky * g.dh;
            if (ty < 0 || ty % g.sh) continue;
            int oy = ty / g.sh;
            if (oy >= g.Ho) continue;
            if (fl) oy = g.Ho - 1 - oy;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int tx = ix + g.pad_l - kx * g.dw;
                if (tx < 0 || tx % g.sw) continue;
                const int ox = tx / g.sw;
                if (ox >= g.Wo) continue;
                const T *gp = dy + (((b * 6 + f) * g.Ho + oy) * g.Wo + ox) * g.Cout;
                const float *wp = w + ((size_t)(ky * g.kw + kx) * g.Cin + ci) * g.Cout;
                for (int co = 0; co < g.Cout; ++co) acc = fmaf(GT<T>::ld(gp + co), GT<T>::wt(wp[co]), acc);
            }
        }
        GT<T>::st(dx + e, acc);
    }
}

// one workgroup per weight element (group, ky, kx, ci, co) or bias element; fixed-order tree reduction
template <typename T>
__global__ void __launch_bounds__(256) gconv_bwd_weights_kernel(GP g, const T *__restrict__ x, const T *__restrict__ dy,
                                                                float *__restrict__ dw_eq, float *__restrict__ dw_pol,
                                                                float *__restrict__ dw_np, float *__restrict__ db_eq,
                                                                float *__restrict__ db_pol, float *__restrict__ db_np,
                                                                int ngroups) {
    const int nW = g.kh * g.kw * g.Cin * g.Cout;
    const int per_group = nW + g.Cout;
    const int grp = blockIdx.x / per_group;          // 0 eq, 1 pol, 2 np
    const int e = blockIdx.x % per_group;
    const bool is_bias = e >= nW;
    int f_begin, f_end;
    if (grp == 0) { f_begin = 0; f_end = 4; }
    else if (grp == 1) { f_begin = 4; f_end = (ngroups == 3) ? 5 : 6; }
    else { f_begin = 5; f_end = 6; }
    int co, ci = 0, ky = 0, kx = 0;
    if (is_bias) co = e - nW;
    else { co = e % g.Cout; ci = (e / g.Cout) % g.Cin; kx = (e / (g.Cout * g.Cin)) % g.kw; ky = e / (g.Cout * g.Cin * g.kw); }
    const size_t npos = (size_t)g.B * (f_end - f_begin) * g.Ho * g.Wo;
    float acc = 0.f;
    for (size_t pidx = threadIdx.x; pidx < npos; pidx += 256) {
        size_t r = pidx;
        const int ox = r % g.Wo; r /= g.Wo;
        const int oy = r % g.Ho; r /= g.Ho;
        const int f = f_begin + (int)(r % (f_end - f_begin));
        const size_t b = r / (f_end - f_begin);
        const float gv = GT<T>::ld(dy + (((b * 6 + f) * g.Ho + oy) * g.Wo + ox) * g.Cout + co);
        if (is_bias) { acc += gv; continue; }
        const bool fl = (f == 5) && g.flip;
        const int oyf = fl ? g.Ho - 1 - oy : oy;
        int iy = oyf * g.sh - g.pad_t + ky * g.dh;
        const int ix = ox * g.sw - g.pad_l + kx * g.dw;
        if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
        if (fl) iy = g.H - 1 - iy;
        acc = fmaf(GT<T>::ld(x + (((b * 6 + f) * g.H + iy) * g.W + ix) * g.Cin + ci), gv, acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float *dw = grp == 0 ? dw_eq : (grp == 1 ? dw_pol : dw_np);
        float *db = grp == 0 ? db_eq : (grp == 1 ? db_pol : db_np);
        if (is_bias) { if (db) db[co] = red[0]; }
        else dw[e] = red[0];
    }
}

static int check_desc(const dlwpcs_gconv_desc *d, const char *who, GP &g) {
    if (!d) return fail(DLWPCS_E_INVALID, "%s: null descriptor", who);
    if (!dtype_ok(d->dtype)) return fail(DLWPCS_E_UNSUPPORTED, "%s: dtype %d not built", who, d->dtype);
    if (d->B < 0 || d->H < 1 || d->W < 1 || d->Cin < 1 || d->Cout < 1 || d->kh < 1 || d->kw < 1 || d->sh < 1 ||
        d->sw < 1 || d->dh < 1 || d->dw < 1 || d->Ho < 1 || d->Wo < 1 || d->pad_t < 0 || d->pad_l < 0)
        return fail(DLWPCS_E_INVALID, "%s: bad descriptor", who);
    g = GP{d->B, d->H, d->W, d->Cin, d->Cout, d->kh, d->kw, d->sh, d->sw, d->dh, d->dw, d->pad_t, d->pad_l, d->Ho, d->Wo,
           d->flip_north_pole};
    return DLWPCS_OK;
}

static inline dim3 sgrid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" int dlwpcs_gconv_fwd(const dlwpcs_gconv_desc *d, const void *x, const void *w_eq, const void *w_pol,
                                const void *w_np, const void *b_eq, const void *b_pol, const void *b_np, void *y,
                                dlwpcs_stream_t stream) {
    GP g;
    int rc = check_desc(d, "gconv_fwd", g);
    if (rc) return rc;
    if (!x || !w_eq || !w_pol || !y) return fail(DLWPCS_E_INVALID, "gconv_fwd: null pointer");
    if ((b_eq == nullptr) != (b_pol == nullptr)) return fail(DLWPCS_E_INVALID, "gconv_fwd: biases must be all given or all null");
    const size_t total = (size_t)g.B * 6 * g.Ho * g.Wo * g.Cout;
    if (total == 0) return DLWPCS_OK;
    if (d->dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(gconv_fwd_kernel<bf16_t>, sgrid(total), dim3(256), 0, (hipStream_t)stream, g, (const bf16_t *)x,
                           (const float *)w_eq, (const float *)w_pol, (const float *)w_np, (const float *)b_eq,
                           (const float *)b_pol, (const float *)b_np, (bf16_t *)y, total);
    else
        hipLaunchKernelGGL(gconv_fwd_kernel<float>, sgrid(total), dim3(256), 0, (hipStream_t)stream, g, (const float *)x,
                           (const float *)w_eq, (const float *)w_pol, (const float *)w_np, (const float *)b_eq,
                           (const float *)b_pol, (const float *)b_np, (float *)y, total);
    return check_launch("gconv_fwd");
}

extern "C" int dlwpcs_gconv_bwd_data(const dlwpcs_gconv_desc *d, const void *dy, const void *w_eq, const void *w_pol,
                                     const void *w_np, void *dx, dlwpcs_stream_t stream) {
    GP g;
    int rc = check_desc(d, "gconv_bwd_data", g);
    if (rc) return rc;
    if (!dy || !w_eq || !w_pol || !dx) return fail(DLWPCS_E_INVALID, "gconv_bwd_data: null pointer");
    const size_t total = (size_t)g.B * 6 * g.H * g.W * g.Cin;
    if (total == 0) return DLWPCS_OK;
    if (d->dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(gconv_bwd_data_kernel<bf16_t>, sgrid(total), dim3(256), 0, (hipStream_t)stream, g, (const bf16_t *)dy,
                           (const float *)w_eq, (const float *)w_pol, (const float *)w_np, (bf16_t *)dx, total);
    else
        hipLaunchKernelGGL(gconv_bwd_data_kernel<float>, sgrid(total), dim3(256), 0, (hipStream_t)stream, g, (const float *)dy,
                           (const float *)w_eq, (const float *)w_pol, (const float *)w_np, (float *)dx, total);
    return check_launch("gconv_bwd_data");
}

extern "C" int dlwpcs_gconv_bwd_weights(const dlwpcs_gconv_desc *d, const void *x, const void *dy, void *dw_eq,
                                        void *dw_pol, void *dw_np, void *db_eq, void *db_pol, void *db_np,
                                        dlwpcs_stream_t stream) {
    GP g;
    int rc = check_desc(d, "gconv_bwd_weights", g);
    if (rc) return rc;
    if (!x || !dy || !dw_eq || !dw_pol) return fail(DLWPCS_E_INVALID, "gconv_bwd_weights: null pointer");
    const int ngroups = dw_np ? 3 : 2;
    const int per_group = g.kh * g.kw * g.Cin * g.Cout + g.Cout;
    if (d->dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(gconv_bwd_weights_kernel<bf16_t>, dim3((unsigned)(ngroups * per_group)), dim3(256), 0,
                           (hipStream_t)stream, g, (const bf16_t *)x, (const bf16_t *)dy, (float *)dw_eq, (float *)dw_pol,
                           (float *)dw_np, (float *)db_eq, (float *)db_pol, (float *)db_np, ngroups);
    else
        hipLaunchKernelGGL(gconv_bwd_weights_kernel<float>, dim3((unsigned)(ngroups * per_group)), dim3(256), 0,
                           (hipStream_t)stream, g, (const float *)x, (const float *)dy, (float *)dw_eq, (float *)dw_pol,
                           (float *)dw_np, (float *)db_eq, (float *)db_pol, (float *)db_np, ngroups);
    return check_launch("gconv_bwd_weights");
}
