// HBM-bound helper kernels of the DLWP-CS hot path (gfx950): stand-alone halo gather / inverse gather, activation,
// 2x2 pooling / upsampling, concat / split, layout converters, loss and Adam.  All are pure streaming kernels:
// channels_last rows are moved as float4 (16 B / lane, coalesced) whenever C % 4 == 0, grid-stride loops capped at
// ~2048 workgroups (cdna_hip_programming.md, Guideline 11/13).  No atomics anywhere -> bitwise deterministic.
#include "common.h"

namespace dlwpcs {

static inline dim3 stream_grid(size_t work_items, int block = 256) {
    size_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

template <typename V> struct VecW;
template <> struct VecW<float> { static constexpr int W = 1; };
template <> struct VecW<float4> { static constexpr int W = 4; };

__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vzero(float) { return 0.f; }
__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }

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
        V acc = base[(size_t)((f * M + yy + p) * M + xx + p) * CV];
        const bool border = (yy < p) | (yy >= N - p) | (xx < p) | (xx >= N - p);
        if (border) {
            const int4 t = *reinterpret_cast<const int4 *>(inv + (size_t)src * 4);
            if (t.x >= 0) acc = vadd(acc, base[(size_t)t.x * CV]);
            if (t.y >= 0) acc = vadd(acc, base[(size_t)t.y * CV]);
            if (t.z >= 0) acc = vadd(acc, base[(size_t)t.z * CV]);
            if (t.w >= 0) acc = vadd(acc, base[(size_t)t.w * CV]);
        }
        dx[e] = acc;
    }
}

// Adjoint of the fused conv loader: dxpad (B,6,M,M,CT) -> gradient of ONE source (channel window [choff, choff+CS))
// on the N grid, or on the N/2 grid with the 2x2 block sum of the nearest-upsample adjoint (up != 0).  p = 1.
template <typename V>
__global__ void __launch_bounds__(256) pad_bwd_src_kernel(const V *__restrict__ dxpad, V *__restrict__ dsrc,
                                                          const int32_t *__restrict__ inv, size_t total, int CSV,
                                                          int CTV, int choffV, int N, int up) {
    const int p = 1;
    const int M = N + 2 * p;
    const int No = up ? N / 2 : N;
    const int out_cells = 6 * No * No, dst_cells = 6 * M * M;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CSV);
        const size_t pix = e / CSV;
        const int cell = (int)(pix % out_cells);
        const size_t b = pix / out_cells;
        const int xo = cell % No, yo = (cell / No) % No, f = cell / (No * No);
        const V *base = dxpad + b * dst_cells * (size_t)CTV + choffV + cv;
        V acc = vzero(V());
        const int reps = up ? 2 : 1;
        for (int uy = 0; uy < reps; ++uy)
            for (int ux = 0; ux < reps; ++ux) {
                const int yy = up ? 2 * yo + uy : yo, xx = up ? 2 * xo + ux : xo;
                acc = vadd(acc, base[(size_t)((f * M + yy + p) * M + xx + p) * CTV]);
                const bool border = (yy < p) | (yy >= N - p) | (xx < p) | (xx >= N - p);
                if (border) {
                    const int src = (f * N + yy) * N + xx;
                    const int4 t = *reinterpret_cast<const int4 *>(inv + (size_t)src * 4);
                    if (t.x >= 0) acc = vadd(acc, base[(size_t)t.x * CTV]);
                    if (t.y >= 0) acc = vadd(acc, base[(size_t)t.y * CTV]);
                    if (t.z >= 0) acc = vadd(acc, base[(size_t)t.z * CTV]);
                    if (t.w >= 0) acc = vadd(acc, base[(size_t)t.w * CTV]);
                }
            }
        dsrc[e] = acc;
    }
}

// gradient of one source of a halo==0 convolution input (no padding): channel window copy, optional 2x2 sum
template <typename V>
__global__ void __launch_bounds__(256) window_src_kernel(const V *__restrict__ dxv, V *__restrict__ dsrc, size_t total,
                                                         int CSV, int CTV, int choffV, int N, int up) {
    const int No = up ? N / 2 : N;
    const int out_cells = 6 * No * No;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CSV);
        const size_t pix = e / CSV;
        const int cell = (int)(pix % out_cells);
        const size_t b = pix / out_cells;
        const int xo = cell % No, yo = (cell / No) % No, f = cell / (No * No);
        const V *base = dxv + b * (size_t)6 * N * N * CTV + choffV + cv;
        V acc = vzero(V());
        const int reps = up ? 2 : 1;
        for (int uy = 0; uy < reps; ++uy)
            for (int ux = 0; ux < reps; ++ux) {
                const int yy = up ? 2 * yo + uy : yo, xx = up ? 2 * xo + ux : xo;
                acc = vadd(acc, base[(size_t)((f * N + yy) * N + xx) * CTV]);
            }
        dsrc[e] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// activation: keras ReLU(negative_slope, max_value)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                      float alpha, float vmax) {
    const size_t n4 = n / 4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x4[i];
        v.x = act_leaky_clip(v.x, alpha, vmax); v.y = act_leaky_clip(v.y, alpha, vmax);
        v.z = act_leaky_clip(v.z, alpha, vmax); v.w = act_leaky_clip(v.w, alpha, vmax);
        y4[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = act_leaky_clip(x[i], alpha, vmax);
}

__global__ void __launch_bounds__(256) act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                      float *__restrict__ dx, size_t n, float alpha, float vmax) {
    const size_t n4 = n / 4;
    const float4 *g4 = reinterpret_cast<const float4 *>(dy);
    const float4 *y4 = reinterpret_cast<const float4 *>(y);
    float4 *o4 = reinterpret_cast<float4 *>(dx);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 g = g4[i];
        const float4 v = y4[i];
        g.x *= act_leaky_clip_grad_from_y(v.x, alpha, vmax); g.y *= act_leaky_clip_grad_from_y(v.y, alpha, vmax);
        g.z *= act_leaky_clip_grad_from_y(v.z, alpha, vmax); g.w *= act_leaky_clip_grad_from_y(v.w, alpha, vmax);
        o4[i] = g;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * act_leaky_clip_grad_from_y(y[i], alpha, vmax);
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
        y[e] = vscale(vadd(vadd(r0[0], r0[CV]), vadd(r1[0], r1[CV])), 0.25f);
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
        dx[e] = vscale(dy[((plane * No + yy / 2) * No + xx / 2) * CV + cv], 0.25f);
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
        dx[e] = vadd(vadd(r0[0], r0[CV]), vadd(r1[0], r1[CV]));
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

__global__ void __launch_bounds__(256) add_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                  float *__restrict__ y, size_t n) {
    const size_t n4 = n / 4;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        y4[i] = vadd(a4[i], b4[i]);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------------------------------
// (B, C, S) <-> (B, S, C) through a padded 32x32 LDS tile (coalesced on both sides)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                        int R, size_t Ccols) {
    // x: (batch, R, Ccols) -> y: (batch, Ccols, R)
    __shared__ float tile[32][33];
    const size_t b = blockIdx.z;
    const size_t c0 = (size_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float *xb = x + b * (size_t)R * Ccols;
    float *yb = y + b * (size_t)R * Ccols;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k;
        const size_t c = c0 + tx;
        tile[k][tx] = (r < R && c < Ccols) ? xb[(size_t)r * Ccols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const size_t c = c0 + k;
        const int r = r0 + tx;
        if (r < R && c < Ccols) yb[c * R + r] = tile[tx][k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// keras 'mse' (+ 'mae' metric) with gradient, two-stage fixed-order reduction          (Azure/train_cs.py:424-430)
// ---------------------------------------------------------------------------------------------------------------
constexpr int MSE_BLOCKS = 1024;

__global__ void __launch_bounds__(256) mse_stage1_kernel(const float *__restrict__ y, const float *__restrict__ t,
                                                         float *__restrict__ dy, float *__restrict__ partial, size_t n,
                                                         float gscale) {
    float sq = 0.f, ab = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = y[i] - t[i];
        sq += d * d;
        ab += fabsf(d);
        if (dy) dy[i] = gscale * d;
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
                                                         int nblocks, float inv_n, float weight) {
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
        loss_out[0] += (float)(s_sq[0] * inv_n) * weight;
        loss_out[1] += (float)(s_ab[0] * inv_n);
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
    const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
__global__ void step_inc_kernel(int32_t *step) { *step += 1; }

}  // namespace dlwpcs

using namespace dlwpcs;

#define REQUIRE(cond, ...) do { if (!(cond)) return fail(DLWPCS_E_INVALID, __VA_ARGS__); } while (0)
#define REQUIRE_F32(dt, who) do { if ((dt) != DLWPCS_F32) return fail(DLWPCS_E_UNSUPPORTED, who ": dtype %d not built", (int)(dt)); } while (0)

extern "C" int dlwpcs_pad_fwd(const void *x, void *y, int B, int N, int C, int p, int dtype,
                              const int32_t *table_dev, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "pad_fwd");
    REQUIRE(x && y && table_dev, "pad_fwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1 && p >= 0 && p <= N, "pad_fwd: bad shape B=%d N=%d C=%d p=%d", B, N, C, p);
    if (B == 0) return DLWPCS_OK;
    const int M = N + 2 * p;
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0) {
        const size_t total = (size_t)B * 6 * M * M * (C / 4);
        hipLaunchKernelGGL(pad_fwd_kernel<float4>, stream_grid(total), dim3(256), 0, s, (const float4 *)x, (float4 *)y,
                           table_dev, total, C / 4, 6 * N * N, 6 * M * M);
    } else {
        const size_t total = (size_t)B * 6 * M * M * C;
        hipLaunchKernelGGL(pad_fwd_kernel<float>, stream_grid(total), dim3(256), 0, s, (const float *)x, (float *)y,
                           table_dev, total, C, 6 * N * N, 6 * M * M);
    }
    return check_launch("pad_fwd");
}

extern "C" int dlwpcs_pad_bwd(const void *dy, void *dx, int B, int N, int C, int p, int dtype,
                              const int32_t *inv_table_dev, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "pad_bwd");
    REQUIRE(dy && dx && inv_table_dev, "pad_bwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1 && p >= 0 && p <= N, "pad_bwd: bad shape B=%d N=%d C=%d p=%d", B, N, C, p);
    if (B == 0) return DLWPCS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0) {
        const size_t total = (size_t)B * 6 * N * N * (C / 4);
        hipLaunchKernelGGL(pad_bwd_kernel<float4>, stream_grid(total), dim3(256), 0, s, (const float4 *)dy, (float4 *)dx,
                           inv_table_dev, total, C / 4, N, p);
    } else {
        const size_t total = (size_t)B * 6 * N * N * C;
        hipLaunchKernelGGL(pad_bwd_kernel<float>, stream_grid(total), dim3(256), 0, s, (const float *)dy, (float *)dx,
                           inv_table_dev, total, C, N, p);
    }
    return check_launch("pad_bwd");
}

namespace dlwpcs {
// used by conv_bwd_data: gradient of one virtual-input source out of the (padded or plain) virtual-input gradient
int launch_src_grad(const float *dxv, float *dsrc, const int32_t *inv, int B, int N, int CT, int choff, int CS, int up,
                    int halo, hipStream_t s) {
    const int No = up ? N / 2 : N;
    const bool vec = (CT % 4 == 0) && (choff % 4 == 0) && (CS % 4 == 0);
    if (halo) {
        if (vec) {
            const size_t total = (size_t)B * 6 * No * No * (CS / 4);
            hipLaunchKernelGGL(pad_bwd_src_kernel<float4>, stream_grid(total), dim3(256), 0, s, (const float4 *)dxv,
                               (float4 *)dsrc, inv, total, CS / 4, CT / 4, choff / 4, N, up);
        } else {
            const size_t total = (size_t)B * 6 * No * No * CS;
            hipLaunchKernelGGL(pad_bwd_src_kernel<float>, stream_grid(total), dim3(256), 0, s, dxv, dsrc, inv, total, CS,
                               CT, choff, N, up);
        }
    } else {
        if (vec) {
            const size_t total = (size_t)B * 6 * No * No * (CS / 4);
            hipLaunchKernelGGL(window_src_kernel<float4>, stream_grid(total), dim3(256), 0, s, (const float4 *)dxv,
                               (float4 *)dsrc, total, CS / 4, CT / 4, choff / 4, N, up);
        } else {
            const size_t total = (size_t)B * 6 * No * No * CS;
            hipLaunchKernelGGL(window_src_kernel<float>, stream_grid(total), dim3(256), 0, s, dxv, dsrc, total, CS, CT,
                               choff, N, up);
        }
    }
    return check_launch("src_grad");
}
}  // namespace dlwpcs

extern "C" int dlwpcs_act_fwd(const void *x, void *y, size_t n, int act, float alpha, float vmax, int dtype,
                              dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "act_fwd");
    REQUIRE(x && y, "act_fwd: null pointer");
    REQUIRE(act == DLWPCS_ACT_LEAKY_CLIP, "act_fwd: unknown activation %d", act);
    if (n == 0) return DLWPCS_OK;
    hipLaunchKernelGGL(act_fwd_kernel, stream_grid(n / 4 + 1), dim3(256), 0, (hipStream_t)stream, (const float *)x,
                       (float *)y, n, alpha, vmax);
    return check_launch("act_fwd");
}

extern "C" int dlwpcs_act_bwd(const void *dy, const void *y, void *dx, size_t n, int act, float alpha, float vmax,
                              int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "act_bwd");
    REQUIRE(dy && y && dx, "act_bwd: null pointer");
    REQUIRE(act == DLWPCS_ACT_LEAKY_CLIP, "act_bwd: unknown activation %d", act);
    if (n == 0) return DLWPCS_OK;
    hipLaunchKernelGGL(act_bwd_kernel, stream_grid(n / 4 + 1), dim3(256), 0, (hipStream_t)stream, (const float *)dy,
                       (const float *)y, (float *)dx, n, alpha, vmax);
    return check_launch("act_bwd");
}

#define POOL_LAUNCH(KERNEL, IN, OUT, TOTAL_PIX, NARG)                                                                  \
    if (C % 4 == 0) {                                                                                                  \
        const size_t total = (size_t)(TOTAL_PIX) * (C / 4);                                                            \
        hipLaunchKernelGGL(KERNEL<float4>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const float4 *)(IN), \
                           (float4 *)(OUT), total, C / 4, NARG);                                                       \
    } else {                                                                                                           \
        const size_t total = (size_t)(TOTAL_PIX) * C;                                                                  \
        hipLaunchKernelGGL(KERNEL<float>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, (const float *)(IN),   \
                           (float *)(OUT), total, C, NARG);                                                            \
    }

extern "C" int dlwpcs_avgpool2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "avgpool2_fwd");
    REQUIRE(x && y, "avgpool2_fwd: null pointer");
    REQUIRE(B >= 0 && N >= 2 && N % 2 == 0 && C >= 1, "avgpool2_fwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(avgpool2_fwd_kernel, x, y, (size_t)B * 6 * (N / 2) * (N / 2), N)
    return check_launch("avgpool2_fwd");
}
extern "C" int dlwpcs_avgpool2_bwd(const void *dy, void *dx, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "avgpool2_bwd");
    REQUIRE(dy && dx, "avgpool2_bwd: null pointer");
    REQUIRE(B >= 0 && N >= 2 && N % 2 == 0 && C >= 1, "avgpool2_bwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(avgpool2_bwd_kernel, dy, dx, (size_t)B * 6 * N * N, N)
    return check_launch("avgpool2_bwd");
}
extern "C" int dlwpcs_upsample2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "upsample2_fwd");
    REQUIRE(x && y, "upsample2_fwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1, "upsample2_fwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(upsample2_fwd_kernel, x, y, (size_t)B * 6 * (2 * N) * (2 * N), N)
    return check_launch("upsample2_fwd");
}
extern "C" int dlwpcs_upsample2_bwd(const void *dy, void *dx, int B, int N, int C, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "upsample2_bwd");
    REQUIRE(dy && dx, "upsample2_bwd: null pointer");
    REQUIRE(B >= 0 && N >= 1 && C >= 1, "upsample2_bwd: bad shape B=%d N=%d C=%d", B, N, C);
    if (B == 0) return DLWPCS_OK;
    POOL_LAUNCH(upsample2_bwd_kernel, dy, dx, (size_t)B * 6 * N * N, N)
    return check_launch("upsample2_bwd");
}

extern "C" int dlwpcs_concat2(const void *a, const void *b, void *y, size_t rows, int Ca, int Cb, int dtype,
                              dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "concat2");
    REQUIRE(a && b && y && Ca >= 1 && Cb >= 1, "concat2: bad arguments");
    if (rows == 0) return DLWPCS_OK;
    if (Ca % 4 == 0 && Cb % 4 == 0) {
        const size_t total = rows * ((Ca + Cb) / 4);
        hipLaunchKernelGGL(concat2_kernel<float4>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)a, (const float4 *)b, (float4 *)y, total, Ca / 4, Cb / 4);
    } else {
        const size_t total = rows * (Ca + Cb);
        hipLaunchKernelGGL(concat2_kernel<float>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const float *)a, (const float *)b, (float *)y, total, Ca, Cb);
    }
    return check_launch("concat2");
}

extern "C" int dlwpcs_split2(const void *y, void *a, void *b, size_t rows, int Ca, int Cb, int dtype,
                             dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "split2");
    REQUIRE(y && (a || b) && Ca >= 1 && Cb >= 1, "split2: bad arguments");
    if (rows == 0) return DLWPCS_OK;
    if (Ca % 4 == 0 && Cb % 4 == 0) {
        const size_t total = rows * ((Ca + Cb) / 4);
        hipLaunchKernelGGL(split2_kernel<float4>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)y, (float4 *)a, (float4 *)b, total, Ca / 4, Cb / 4);
    } else {
        const size_t total = rows * (Ca + Cb);
        hipLaunchKernelGGL(split2_kernel<float>, stream_grid(total), dim3(256), 0, (hipStream_t)stream,
                           (const float *)y, (float *)a, (float *)b, total, Ca, Cb);
    }
    return check_launch("split2");
}

static int launch_transpose(const void *x, void *y, int batch, size_t R, size_t Ccols, hipStream_t s, const char *who) {
    if (batch == 0 || R == 0 || Ccols == 0) return DLWPCS_OK;
    if (R > 0x7fffffffu) return fail(DLWPCS_E_INVALID, "%s: too many rows", who);
    dim3 grid((unsigned)((Ccols + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)batch);
    if (grid.y > 65535 || grid.z > 65535) return fail(DLWPCS_E_UNSUPPORTED, "%s: grid too large", who);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, (const float *)x, (float *)y, (int)R, Ccols);
    return check_launch(who);
}

extern "C" int dlwpcs_cf_to_cl(const void *x, void *y, int B, int C, size_t S, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "cf_to_cl");
    REQUIRE(x && y && B >= 0 && C >= 1, "cf_to_cl: bad arguments");
    return launch_transpose(x, y, B, (size_t)C, S, (hipStream_t)stream, "cf_to_cl");   // (B,C,S) -> (B,S,C)
}
extern "C" int dlwpcs_cl_to_cf(const void *x, void *y, int B, int C, size_t S, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "cl_to_cf");
    REQUIRE(x && y && B >= 0 && C >= 1, "cl_to_cf: bad arguments");
    // (B,S,C) -> (B,C,S): rows = S may exceed 65535*32, so put S on grid.x by swapping roles
    if (B == 0 || S == 0) return DLWPCS_OK;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((S + 31) / 32), (unsigned)B);
    if (grid.y > 65535) {
        // fall back: treat as batch of S-chunks is not possible generically; S per sample is 6*H*W < 2M in practice
        return fail(DLWPCS_E_UNSUPPORTED, "cl_to_cf: spatial size too large");
    }
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, (float *)y, (int)S,
                       (size_t)C);
    return check_launch("cl_to_cf");
}

extern "C" int dlwpcs_add(const void *a, const void *b, void *y, size_t n, int dtype, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "add");
    REQUIRE(a && b && y, "add: null pointer");
    if (n == 0) return DLWPCS_OK;
    hipLaunchKernelGGL(add_kernel, stream_grid(n / 4 + 1), dim3(256), 0, (hipStream_t)stream, (const float *)a,
                       (const float *)b, (float *)y, n);
    return check_launch("add");
}

extern "C" size_t dlwpcs_mse_scratch_bytes(void) { return (size_t)MSE_BLOCKS * 2 * sizeof(float); }

extern "C" int dlwpcs_mse_fwd_bwd(const void *y, const void *t, void *dy, float *loss_out, size_t n, float weight,
                                  int dtype, void *scratch, dlwpcs_stream_t stream) {
    REQUIRE_F32(dtype, "mse_fwd_bwd");
    REQUIRE(y && t && loss_out && scratch && n > 0, "mse_fwd_bwd: bad arguments");
    size_t g = (n + 255) / 256;
    if (g > MSE_BLOCKS) g = MSE_BLOCKS;
    const float gscale = weight * 2.f / (float)n;
    hipLaunchKernelGGL(mse_stage1_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float *)y,
                       (const float *)t, (float *)dy, (float *)scratch, n, gscale);
    hipLaunchKernelGGL(mse_stage2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)scratch, loss_out,
                       (int)g, 1.f / (float)n, weight);
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
