// Fused cubed-sphere convolution for gfx950 (MI355X): implicit-GEMM direct convolution on the matrix cores.
//
// One kernel template serves
//   * forward            y  = act( conv_valid( halo_pad(V), W_face ) + b_face )        (DLWP/custom.py:921-1002 with
//                                                                                       :1082-1308 fused into the load)
//   * data gradient      dVpad = conv_full( dy * act'(y), W_face^T )                   (same kernel, mode ZERO border)
// and a second kernel computes the weight gradient.  No im2col, nothing padded is ever materialised in HBM.
//
// GEMM view per face:  M = pixels, N = C_out, K = k*k*C_in.  Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 FLOP/clk/SIMD = the chip's 157.3 TFLOP/s fp32 peak).  Per workgroup:
//   - a band of BM <= 32*MT*WM consecutive pixels (flat row-major index inside one face of one sample) times
//     BN = 32*NT*WN output channels; wave (wm, wn) owns MT x NT accumulator tiles of 32x32 (16 VGPRs each);
//   - the input tile (band rows + k-1 halo rows, full width + k-1) is staged through LDS in chunks of KC channels,
//     channels_last, row stride KC+4 floats so that the 16-lane groups of ds_read_b128 hit distinct 16-B slots;
//   - the cube-sphere halo is resolved while staging: interior cells address their own face, border cells go through
//     the (6,N+2,N+2) gather table (L2 resident, 60 KB at N=48); nearest-upsampling (x2) and the channel concat of the
//     U-Net decoder are folded into the same address computation, so none of pad / upsample / concat costs a pass;
//   - weights are pre-packed (tiny kernel, once per call) in MFMA-B fragment order, so a lane's ds_read_b128 returns
//     the 4 consecutive K values it feeds to 4 successive MFMAs; face 5's row-reversed kernel is a packing variant.
//   - A operand: one ds_read_b128 per (tap, 8-channel group, M tile) = 4 MFMAs' worth; K order inside a group is
//     {lanes 0-31: c0..c3, lanes 32-63: c4..c7} x step j, identical on the A and B side.
#include "common.h"

namespace dlwpcs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_DIRECT = 0, MODE_HALO = 1, MODE_ZERO = 2 };

struct ConvKParams {
    const float *src0, *src1;   // virtual-input sources, channels_last
    const float *ymask;         // data-gradient mode: saved forward output, act' applied on load (or nullptr)
    const float *wpk;           // packed weights [3][NTtot][CG][TAPS][2][32][4]
    const float *bias;          // packed bias [3][NTtot*32] or nullptr
    float *out;                 // (B,6,No,No,Cout)
    const int32_t *table;       // (6, Nin+2, Nin+2) halo table (MODE_HALO, k=3)
    int B, Nin, No;             // face size of V, face size of the output
    int C0, C1, Cin, Cout;      // Cin = C0 + C1
    int CG, NTtot;              // ceil(Cin/8), ceil(Cout/32)
    int up0;                    // src0 lives on the Nin/2 grid
    int mode;
    int act;                    // epilogue activation
    float alpha, vmax;
    int pix_per_block;          // valid pixels per workgroup (<= 32*MT*WM)
    int nblk_face;              // workgroups per (sample, face)
    int W2;                     // tile width = No + KS - 1
    uint32_t magicW2, magicNo;
    int tile_rows_max;          // rows reserved in LDS
};

// ------------------------------------------------------------------------------------------------------------------
// Resolve one cell of the padded virtual input to (valid, face, vy, vx) on the Nin grid.
// ------------------------------------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ bool resolve_cell(const ConvKParams &P, int f, int iy, int ix, int &vf, int &vy, int &vx) {
    vf = f;
    if (P.mode == MODE_DIRECT) { vy = iy; vx = ix; return true; }
    if (P.mode == MODE_HALO) {
        constexpr int p = (KS - 1) / 2;
        const int N = P.Nin;
        if (iy >= p && iy < N + p && ix >= p && ix < N + p) { vy = iy - p; vx = ix - p; return true; }
        const int M = N + 2 * p;
        const int idx = P.table[(f * M + iy) * M + ix];
        vf = idx / (N * N);
        const int rem = idx - vf * N * N;
        vy = rem / N;
        vx = rem - vy * N;
        return true;
    }
    // MODE_ZERO: zero border of width KS-1 (full correlation of the data gradient)
    vy = iy - (KS - 1); vx = ix - (KS - 1);
    return (vy >= 0) & (vy < P.Nin) & (vx >= 0) & (vx < P.Nin);
}

// Stage channels [c_begin, c_begin+NCH) of tile rows [y0, y0+rows) (padded coords) into lds[pix][STRIDE].
template <int KS, int NCH, int STRIDE, bool VEC, int NTHREADS>
__device__ __forceinline__ void stage_input(const ConvKParams &P, float *lds, int b, int f, int y0, int rows, int c_begin) {
    const int tile_pix = rows * P.W2;
    const int tid = threadIdx.x;
    if (VEC) {
        constexpr int Q = NCH / 4;
        for (int item = tid; item < tile_pix * Q; item += NTHREADS) {
            const int pix = item / Q, q = item % Q;
            const int ty = __umulhi((uint32_t)pix, P.magicW2);
            const int tx = pix - ty * P.W2;
            const int c = c_begin + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int vf, vy, vx;
            if (c < P.Cin && resolve_cell<KS>(P, f, y0 + ty, tx, vf, vy, vx)) {
                if (c < P.C0) {
                    const int g = P.up0 ? (P.Nin >> 1) : P.Nin;
                    const int sy = P.up0 ? (vy >> 1) : vy, sx = P.up0 ? (vx >> 1) : vx;
                    const size_t off = ((((size_t)b * 6 + vf) * g + sy) * g + sx) * P.C0 + c;
                    v = *reinterpret_cast<const float4 *>(P.src0 + off);
                    if (P.ymask) {
                        const float4 yv = *reinterpret_cast<const float4 *>(P.ymask + off);
                        v.x *= act_leaky_clip_grad_from_y(yv.x, P.alpha, P.vmax);
                        v.y *= act_leaky_clip_grad_from_y(yv.y, P.alpha, P.vmax);
                        v.z *= act_leaky_clip_grad_from_y(yv.z, P.alpha, P.vmax);
                        v.w *= act_leaky_clip_grad_from_y(yv.w, P.alpha, P.vmax);
                    }
                } else {
                    const size_t off = ((((size_t)b * 6 + vf) * P.Nin + vy) * P.Nin + vx) * P.C1 + (c - P.C0);
                    v = *reinterpret_cast<const float4 *>(P.src1 + off);
                }
            }
            *reinterpret_cast<float4 *>(lds + pix * STRIDE + q * 4) = v;
        }
    } else {
        for (int item = tid; item < tile_pix * NCH; item += NTHREADS) {
            const int pix = item / NCH, q = item % NCH;
            const int ty = __umulhi((uint32_t)pix, P.magicW2);
            const int tx = pix - ty * P.W2;
            const int c = c_begin + q;
            float v = 0.f;
            int vf, vy, vx;
            if (c < P.Cin && resolve_cell<KS>(P, f, y0 + ty, tx, vf, vy, vx)) {
                if (c < P.C0) {
                    const int g = P.up0 ? (P.Nin >> 1) : P.Nin;
                    const int sy = P.up0 ? (vy >> 1) : vy, sx = P.up0 ? (vx >> 1) : vx;
                    const size_t off = ((((size_t)b * 6 + vf) * g + sy) * g + sx) * P.C0 + c;
                    v = P.src0[off];
                    if (P.ymask) v *= act_leaky_clip_grad_from_y(P.ymask[off], P.alpha, P.vmax);
                } else {
                    v = P.src1[((((size_t)b * 6 + vf) * P.Nin + vy) * P.Nin + vx) * P.C1 + (c - P.C0)];
                }
            }
            lds[pix * STRIDE + q] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward / data-gradient kernel
// ------------------------------------------------------------------------------------------------------------------
template <int KS, int KC, int MT, int NT, int WM, int WN, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN) conv_mfma_kernel(const ConvKParams P) {
    constexpr int TAPS = KS * KS;
    constexpr int KCP = KC + 4;
    constexpr int KCG = KC / 8;
    constexpr int NTB = NT * WN;
    constexpr int NTHREADS = 64 * WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lds_in = smem;
    float *lds_w = smem + P.tile_rows_max * P.W2 * KCP;

    const uint32_t nblk = gridDim.x;
    const uint32_t L = xcd_remap(blockIdx.x, nblk);
    const int blk = L % P.nblk_face;
    const int f = (L / P.nblk_face) % 6;
    const int b = L / (P.nblk_face * 6);
    const int nt0 = blockIdx.y * NTB;

    const int face_pix = P.No * P.No;
    const int m0 = blk * P.pix_per_block;
    const int npix = min(P.pix_per_block, face_pix - m0);
    const int y0 = __umulhi((uint32_t)m0, P.magicNo);
    const int ylast = __umulhi((uint32_t)(m0 + npix - 1), P.magicNo);
    const int rows = ylast - y0 + KS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int v = f < 4 ? 0 : (f == 4 ? 1 : 2);

    // per-lane LDS base (in floats) of the A operand for each of this wave's M tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = (wm * MT + mt) * 32 + l31;
        int base = 0;
        if (m < npix) {
            const int gm = m0 + m;
            const int oy = __umulhi((uint32_t)gm, P.magicNo);
            const int ox = gm - oy * P.No;
            base = ((oy - y0) * P.W2 + ox) * KCP;
        }
        abase[mt] = base + half * 4;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int nchunks = (P.CG + KCG - 1) / KCG;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int ncg = min(KCG, P.CG - ch * KCG);
        stage_input<KS, KC, KCP, VEC, NTHREADS>(P, lds_in, b, f, y0, rows, ch * KC);
        // weights of this chunk: for each of the block's N tiles, ncg contiguous groups of TAPS*256 floats
        {
            constexpr int GF4 = TAPS * 64;   // float4 per (ntile, cg)
            const int total = NTB * ncg * GF4;
            for (int it = tid; it < total; it += NTHREADS) {
                const int g = it / GF4, w = it % GF4;
                const int ntl = g / ncg, cgl = g % ncg;
                const int ntile = nt0 + ntl;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ntile < P.NTtot) {
                    const size_t src = ((((size_t)v * P.NTtot + ntile) * P.CG + (ch * KCG + cgl)) * GF4 + w);
                    val = reinterpret_cast<const float4 *>(P.wpk)[src];
                }
                reinterpret_cast<float4 *>(lds_w)[(ntl * KCG + cgl) * GF4 + w] = val;
            }
        }
        __syncthreads();
        for (int cgl = 0; cgl < ncg; ++cgl) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = tap / KS, dx = tap % KS;
                const int tapoff = (dy * P.W2 + dx) * KCP + cgl * 8;
                float4 a[MT], bw[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4 *>(lds_in + abase[mt] + tapoff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bw[nt] = *reinterpret_cast<const float4 *>(
                        lds_w + ((((wn * NT + nt) * KCG + cgl) * TAPS + tap) * 2 + half) * 128 + l31 * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bw[nt].x, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bw[nt].y, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bw[nt].z, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bw[nt].w, acc[mt][nt], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
    float *outp = P.out + ((size_t)b * 6 + f) * face_pix * P.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (nt0 + wn * NT + nt) * 32 + l31;
        if (co >= P.Cout) continue;
        const float bv = P.bias ? P.bias[(size_t)v * P.NTtot * 32 + co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < npix) {
                    float val = acc[mt][nt][r] + bv;
                    if (P.act == DLWPCS_ACT_LEAKY_CLIP) val = act_leaky_clip(val, P.alpha, P.vmax);
                    outp[(size_t)(m0 + m) * P.Cout + co] = val;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight packing (HWIO -> MFMA-B fragment order), 3 face variants: 0 equatorial, 1 south pole, 2 north pole.
//   transposed == 0 (forward):        B[tap=(dy,dx)][k=ci][n=co] = Wv[row(dy)][dx][ci][co]
//   transposed == 1 (data gradient):  B[tap=(ey,ex)][k=co][n=ci] = Wv[row(KS-1-ey)][KS-1-ex][ci][co]
// row(r) = KS-1-r on variant 2 when flip_north_pole (flip -> conv -> flip == row-reversed kernel), else r.
// Also packs the biases to [3][NTtot*32].
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_weights_kernel(const float *__restrict__ w_eq, const float *__restrict__ w_pol,
                                                           const float *__restrict__ w_np, float *__restrict__ out,
                                                           int KS, int Cin, int Cout, int K, int Ncol, int CG, int NTtot,
                                                           int flip, int transposed, size_t total) {
    const int TAPS = KS * KS;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int j = r % 4; r /= 4;
        const int n = r % 32; r /= 32;
        const int hf = r % 2; r /= 2;
        const int tap = r % TAPS; r /= TAPS;
        const int cg = r % CG; r /= CG;
        const int nt = r % NTtot; r /= NTtot;
        const int v = (int)r;
        const int k = cg * 8 + hf * 4 + j, col = nt * 32 + n;
        float val = 0.f;
        if (k < K && col < Ncol) {
            const float *w = v == 0 ? w_eq : (v == 1 ? w_pol : (w_np ? w_np : w_pol));
            int ty = tap / KS, tx = tap % KS;
            int ci = k, co = col;
            if (transposed) { ty = KS - 1 - ty; tx = KS - 1 - tx; ci = col; co = k; }
            if (v == 2 && flip) ty = KS - 1 - ty;
            val = w[((size_t)(ty * KS + tx) * Cin + ci) * Cout + co];
        }
        out[e] = val;
    }
}

__global__ void __launch_bounds__(256) pack_bias_kernel(const float *__restrict__ b_eq, const float *__restrict__ b_pol,
                                                        const float *__restrict__ b_np, float *__restrict__ out, int Cout,
                                                        int CoutP) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * CoutP) return;
    const int v = e / CoutP, co = e % CoutP;
    const float *bsrc = v == 0 ? b_eq : (v == 1 ? b_pol : (b_np ? b_np : b_pol));
    out[e] = co < Cout ? bsrc[co] : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel.  GEMM view per tap: D[ci][co] += sum_pixels Xpad[pixel+tap][ci] * dZ[pixel][co],
// dZ = dy * act'(y).  Workgroup = (face, pixel band, group of NB samples, 32-wide ci tile, 32-wide co tile); its four
// waves split the pixel pairs (MFMA K = 2 pixels) and each keeps all k*k taps in registers (9 x 16 VGPRs).  After the
// sample loop the four waves are summed through LDS in a fixed order and the (taps,32,32) partial is written to the
// workspace slot of (face, band, sample group); a second kernel adds the slots in a fixed order (no atomics ->
// bitwise reproducible) and applies the weight-group map (faces 0-3 -> equatorial, 4 -> polar, 5 -> polar or north
// pole, tap rows reversed when flip_north_pole).
// ------------------------------------------------------------------------------------------------------------------
struct WgradKParams {
    ConvKParams c;          // loader description of the virtual input (src0/src1/table/mode/...), ymask unused here
    const float *dy, *y;    // (B,6,No,No,Cout); y nullable
    float *partial;         // [nslots][TAPS][CinP][CoutP]
    float *bpartial;        // [nslots][CoutP] or nullptr
    int CinP, CoutP;        // multiples of 32
    int NB;                 // samples per workgroup
    int ngroups;            // ceil(B / NB)
    int mask_act;
};

template <int KS, bool VEC>
__global__ void __launch_bounds__(256, 2) wgrad_mfma_kernel(const WgradKParams W) {
    constexpr int TAPS = KS * KS;
    constexpr int XS = 32;      // X tile row stride (floats)
    const ConvKParams &P = W.c;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lds_x = smem;                                              // [tile_rows_max*W2][32]
    float *lds_dy = lds_x + P.tile_rows_max * P.W2 * XS;              // [pix_cap][32]
    const int pix_cap = (P.pix_per_block + 1) & ~1;
    int *lds_pb = reinterpret_cast<int *>(lds_dy + pix_cap * 32);     // [pix_cap] tile offsets of each output pixel
    // reduction scratch aliases the X tile after the main loop (needs 4*1024 floats = 16 KB)

    const uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = L % W.ngroups;
    const int blk = (L / W.ngroups) % P.nblk_face;
    const int f = L / (W.ngroups * P.nblk_face);
    const int cit = blockIdx.y, cot = blockIdx.z;

    const int face_pix = P.No * P.No;
    const int m0 = blk * P.pix_per_block;
    const int npix = min(P.pix_per_block, face_pix - m0);
    const int y0 = __umulhi((uint32_t)m0, P.magicNo);
    const int ylast = __umulhi((uint32_t)(m0 + npix - 1), P.magicNo);
    const int rows = ylast - y0 + KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    for (int k = tid; k < pix_cap; k += 256) {
        int base = 0;
        if (k < npix) {
            const int gm = m0 + k;
            const int oy = __umulhi((uint32_t)gm, P.magicNo);
            base = ((oy - y0) * P.W2 + (gm - oy * P.No)) * XS;
        }
        lds_pb[k] = base;
    }

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;   // bias partial: thread (co = tid&31, part = tid>>5)

    const int nsteps = pix_cap / 2;
    const int b_begin = grp * W.NB, b_end = min(P.B, b_begin + W.NB);
    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();   // previous sample's tiles fully consumed (also orders lds_pb on the first pass)
        stage_input<KS, 32, XS, VEC, 256>(P, lds_x, b, f, y0, rows, cit * 32);
        // dZ tile: [pix][32 output channels of tile cot], masked by act'(y), zero beyond npix / Cout
        {
            const float *dyb = W.dy + (((size_t)b * 6 + f) * face_pix + m0) * P.Cout;
            const float *yb = W.y ? W.y + (((size_t)b * 6 + f) * face_pix + m0) * P.Cout : nullptr;
            const bool vec_dy = (P.Cout % 4 == 0);
            if (vec_dy) {
                for (int it = tid; it < pix_cap * 8; it += 256) {
                    const int k = it >> 3, q = it & 7;
                    const int co = cot * 32 + q * 4;
                    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < npix && co < P.Cout) {
                        g = *reinterpret_cast<const float4 *>(dyb + (size_t)k * P.Cout + co);
                        if (W.mask_act) {
                            const float4 yv = *reinterpret_cast<const float4 *>(yb + (size_t)k * P.Cout + co);
                            g.x *= act_leaky_clip_grad_from_y(yv.x, P.alpha, P.vmax);
                            g.y *= act_leaky_clip_grad_from_y(yv.y, P.alpha, P.vmax);
                            g.z *= act_leaky_clip_grad_from_y(yv.z, P.alpha, P.vmax);
                            g.w *= act_leaky_clip_grad_from_y(yv.w, P.alpha, P.vmax);
                        }
                    }
                    *reinterpret_cast<float4 *>(lds_dy + k * 32 + q * 4) = g;
                }
            } else {
                for (int it = tid; it < pix_cap * 32; it += 256) {
                    const int k = it >> 5, q = it & 31;
                    const int co = cot * 32 + q;
                    float g = 0.f;
                    if (k < npix && co < P.Cout) {
                        g = dyb[(size_t)k * P.Cout + co];
                        if (W.mask_act) g *= act_leaky_clip_grad_from_y(yb[(size_t)k * P.Cout + co], P.alpha, P.vmax);
                    }
                    lds_dy[k * 32 + q] = g;
                }
            }
        }
        __syncthreads();
        if (W.bpartial && cit == 0) {
            const int co = tid & 31, part = tid >> 5;
            for (int k = part; k < npix; k += 8) bsum += lds_dy[k * 32 + co];
        }
        for (int s = wave; s < nsteps; s += 4) {
            const int k = 2 * s + half;
            const int pb = lds_pb[k];
            const float bval = lds_dy[k * 32 + l31];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = tap / KS, dx = tap % KS;
                const float aval = lds_x[pb + (dy * P.W2 + dx) * XS + l31];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bval, acc[tap], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // cross-wave reduction through LDS (fixed order w = 0..3), one tap at a time
    const int slot = (f * P.nblk_face + blk) * W.ngroups + grp;
    float *red = lds_x;
    float *pout = W.partial + (size_t)slot * TAPS * W.CinP * W.CoutP;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave * 1024 + ci * 32 + l31] = acc[tap][r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            const float sum = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
            const int ci = e >> 5, co = e & 31;
            pout[((size_t)tap * W.CinP + cit * 32 + ci) * W.CoutP + cot * 32 + co] = sum;
        }
        __syncthreads();
    }
    if (W.bpartial && cit == 0) {
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float s = 0.f;
#pragma unroll
            for (int part = 0; part < 8; ++part) s += red[part * 32 + tid];
            W.bpartial[(size_t)slot * W.CoutP + cot * 32 + tid] = s;
        }
    }
}

// Sum the per-slot partials in fixed order and route them to the weight groups.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bpartial,
                                                           float *__restrict__ dw_eq, float *__restrict__ dw_pol,
                                                           float *__restrict__ dw_np, float *__restrict__ db_eq,
                                                           float *__restrict__ db_pol, float *__restrict__ db_np,
                                                           int KS, int Cin, int Cout, int CinP, int CoutP,
                                                           int slots_per_face, int flip) {
    const int TAPS = KS * KS;
    const int nW = TAPS * Cin * Cout;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t slot_stride = (size_t)TAPS * CinP * CoutP;
    if (e < nW) {
        const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cout * Cin);
        const size_t off = ((size_t)tap * CinP + ci) * CoutP + co;
        const int ty = tap / KS, tx = tap % KS;
        const size_t off_flip = ((size_t)((KS - 1 - ty) * KS + tx) * CinP + ci) * CoutP + co;
        float s_eq = 0.f, s_4 = 0.f, s_5 = 0.f;
        for (int s = 0; s < 4 * slots_per_face; ++s) s_eq += partial[(size_t)s * slot_stride + off];
        for (int s = 4 * slots_per_face; s < 5 * slots_per_face; ++s) s_4 += partial[(size_t)s * slot_stride + off];
        // face 5 ran with the row-reversed kernel: its partial for tap row r belongs to kernel row KS-1-r
        const size_t o5 = flip ? off_flip : off;
        for (int s = 5 * slots_per_face; s < 6 * slots_per_face; ++s) s_5 += partial[(size_t)s * slot_stride + o5];
        dw_eq[e] = s_eq;
        if (dw_np) { dw_pol[e] = s_4; dw_np[e] = s_5; }
        else dw_pol[e] = s_4 + s_5;
    } else if (bpartial && e < nW + Cout) {
        const int co = e - nW;
        float s_eq = 0.f, s_4 = 0.f, s_5 = 0.f;
        for (int s = 0; s < 4 * slots_per_face; ++s) s_eq += bpartial[(size_t)s * CoutP + co];
        for (int s = 4 * slots_per_face; s < 5 * slots_per_face; ++s) s_4 += bpartial[(size_t)s * CoutP + co];
        for (int s = 5 * slots_per_face; s < 6 * slots_per_face; ++s) s_5 += bpartial[(size_t)s * CoutP + co];
        if (db_eq) db_eq[co] = s_eq;
        if (db_np) { if (db_pol) db_pol[co] = s_4; db_np[co] = s_5; }
        else if (db_pol) db_pol[co] = s_4 + s_5;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Host side: configuration choice and launches
// ------------------------------------------------------------------------------------------------------------------
int launch_src_grad(const float *dxv, float *dsrc, const int32_t *inv, int B, int N, int CT, int choff, int CS, int up,
                    int halo, hipStream_t s);

struct Work { double flops, bytes; };   // algorithmic work of one launch (for the opt-in profiler)

// rows of the face touched by `pix` consecutive flat pixels whose first pixel is a multiple of `pix`
static int tile_rows_for(int pix, int No) {
    if (pix % No == 0) return pix / No;
    int r = (pix + No - 2) / No + 1;
    return r > No ? No : r;
}

template <int KS, int KC, int MT, int NT, int WM, int WN, bool VEC>
static int launch_conv_cfg(ConvKParams P, const Work &W, hipStream_t s) {
    constexpr int BM = 32 * MT * WM, NTB = NT * WN, NTHREADS = 64 * WM * WN;
    const int face_pix = P.No * P.No;
    // band = whole rows when that does not cost extra workgroups, else a flat range of BM pixels (partial rows)
    int pix = BM < face_pix ? BM : face_pix;
    if (P.No <= BM) {
        int whole = (BM / P.No) * P.No;
        if (whole > face_pix) whole = face_pix;
        if (ceil_div(face_pix, whole) <= ceil_div(face_pix, pix)) pix = whole;
    }
    P.pix_per_block = pix;
    P.nblk_face = ceil_div(face_pix, pix);
    P.W2 = P.No + KS - 1;
    P.magicW2 = div_magic(P.W2);
    P.magicNo = div_magic(P.No);
    P.tile_rows_max = tile_rows_for(pix, P.No) + (KS - 1);
    const size_t lds = ((size_t)P.tile_rows_max * P.W2 * (KC + 4) + (size_t)NTB * (KC / 8) * KS * KS * 256) * sizeof(float);
    if (lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv: LDS tile of %zu bytes exceeds 160 KiB (N=%d)", lds, P.No);
    auto kern = conv_mfma_kernel<KS, KC, MT, NT, WM, WN, VEC>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    dim3 grid((unsigned)(P.B * 6 * P.nblk_face), (unsigned)ceil_div(P.NTtot, NTB));
    int pidx = -1;
    if (prof_enabled()) {
        char tag[128];
        snprintf(tag, sizeof(tag), "conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %s>", KS, KC, MT, NT, WM, WN, VEC ? "true" : "false");
        pidx = prof_begin(tag, W.flops, W.bytes, s);
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, s, P);
    if (pidx >= 0) prof_end(pidx, s);
    return check_launch("conv_mfma");
}

template <int KS, bool VEC>
static int launch_conv(const ConvKParams &P, const Work &W, hipStream_t s) {
    const int face_pix = P.No * P.No;
    if constexpr (KS == 1) return launch_conv_cfg<KS, 16, 3, 1, 4, 1, VEC>(P, W, s);
    else {
    if (P.NTtot == 1) return launch_conv_cfg<KS, 16, 3, 1, 4, 1, VEC>(P, W, s);
    if (P.NTtot == 2) return launch_conv_cfg<KS, 16, 3, 1, 2, 2, VEC>(P, W, s);
    if (face_pix <= 320) return launch_conv_cfg<KS, 16, 5, 1, 1, 4, VEC>(P, W, s);
    return launch_conv_cfg<KS, 16, 3, 1, 1, 4, VEC>(P, W, s);
    }
}

static int dispatch_conv(int KS, bool vec, const ConvKParams &P, const Work &W, hipStream_t s) {
    if (KS == 3) return vec ? launch_conv<3, true>(P, W, s) : launch_conv<3, false>(P, W, s);
    return vec ? launch_conv<1, true>(P, W, s) : launch_conv<1, false>(P, W, s);
}

// algorithmic work of one convolution pass: flops = 2*B*6*N^2*k^2*Cin*Cout; bytes = unpadded input and
// output touched once + the weights.
static Work conv_work(const dlwpcs_conv_desc *d) {
    const double No = d->halo ? d->N : d->N - d->ksize + 1;
    const double Cin = d->C0 + d->C1, taps = (double)d->ksize * d->ksize;
    const double n0 = d->up0 ? d->N / 2 : d->N;
    Work w;
    w.flops = 2.0 * d->B * 6 * No * No * taps * Cin * d->Cout;
    w.bytes = 4.0 * (d->B * 6.0 * (n0 * n0 * d->C0 + (double)d->N * d->N * d->C1 + No * No * d->Cout) +
                     2.0 * taps * Cin * d->Cout);
    return w;
}

struct Geometry {
    int Cin, CinP8, CG, NT_f, CoutP, NT_b, CGb, No, TAPS;
};

static int validate(const dlwpcs_conv_desc *d, const char *who) {
    if (!d) return fail(DLWPCS_E_INVALID, "%s: null descriptor", who);
    if (d->dtype != DLWPCS_F32) return fail(DLWPCS_E_UNSUPPORTED, "%s: dtype %d not built", who, d->dtype);
    if (d->ksize != 1 && d->ksize != 3) return fail(DLWPCS_E_UNSUPPORTED, "%s: kernel size %d (MFMA path serves 1 and 3)", who, d->ksize);
    if (d->B < 0 || d->N < 1 || d->C0 < 1 || d->C1 < 0 || d->Cout < 1) return fail(DLWPCS_E_INVALID, "%s: bad shape B=%d N=%d C0=%d C1=%d Cout=%d", who, d->B, d->N, d->C0, d->C1, d->Cout);
    if (d->up0 && (d->N % 2)) return fail(DLWPCS_E_INVALID, "%s: up0 needs even N", who);
    if (d->halo && d->ksize == 1) return fail(DLWPCS_E_INVALID, "%s: halo with a 1x1 kernel", who);
    if (!d->halo && d->N < d->ksize) return fail(DLWPCS_E_INVALID, "%s: N < kernel size", who);
    if (d->act != DLWPCS_ACT_NONE && d->act != DLWPCS_ACT_LEAKY_CLIP) return fail(DLWPCS_E_INVALID, "%s: unknown activation %d", who, d->act);
    if (d->N > 1024) return fail(DLWPCS_E_UNSUPPORTED, "%s: N > 1024", who);
    return DLWPCS_OK;
}

static inline int out_size(const dlwpcs_conv_desc *d) { return d->halo ? d->N : d->N - d->ksize + 1; }

// workspace layout (bytes, 256-aligned regions)
struct WsLayout {
    size_t wpk_f, bias, wpk_b, dxv, partial, bpartial, total;
    int slots_per_face, NB, ngroups, wg_pix, wg_nblk;
};

static void wgrad_tiling(const dlwpcs_conv_desc *d, int &pix, int &nblk, int &NB, int &ngroups) {
    const int No = out_size(d);
    const int face_pix = No * No;
    const int CAP = 192;
    pix = CAP;
    if (No <= CAP) pix = (CAP / No) * No;
    if (pix > face_pix) pix = face_pix;
    nblk = ceil_div(face_pix, pix);
    const int CinP = ceil_div(d->C0 + d->C1, 32) * 32, CoutP = ceil_div(d->Cout, 32) * 32;
    const int tiles = (CinP / 32) * (CoutP / 32);
    // aim for >= ~512 workgroups (2 per CU) while keeping the number of partial slots small
    NB = 1;
    while (NB < d->B && (long)6 * nblk * tiles * ceil_div(d->B, NB * 2) >= 512) NB *= 2;
    ngroups = ceil_div(d->B > 0 ? d->B : 1, NB);
}

static WsLayout ws_layout(const dlwpcs_conv_desc *d) {
    WsLayout L{};
    const int Cin = d->C0 + d->C1, TAPS = d->ksize * d->ksize;
    const int CGf = ceil_div(Cin, 8), NTf = ceil_div(d->Cout, 32);
    const int CGb = ceil_div(d->Cout, 8), NTb = ceil_div(Cin, 32);
    const int No = out_size(d);
    size_t off = 0;
    L.wpk_f = off; off += align_up((size_t)3 * NTf * CGf * TAPS * 256 * 4, 256);
    L.bias = off;  off += align_up((size_t)3 * NTf * 32 * 4, 256);
    L.wpk_b = off; off += align_up((size_t)3 * NTb * CGb * TAPS * 256 * 4, 256);
    const int Nv = d->halo ? d->N + d->ksize - 1 : d->N;      // face size of the virtual-input gradient
    L.dxv = off;   off += align_up((size_t)d->B * 6 * Nv * Nv * Cin * 4, 256);
    int pix, nblk, NB, ngroups;
    wgrad_tiling(d, pix, nblk, NB, ngroups);
    L.wg_pix = pix; L.wg_nblk = nblk; L.NB = NB; L.ngroups = ngroups;
    L.slots_per_face = nblk * ngroups;
    const int CinP = ceil_div(Cin, 32) * 32, CoutP = ceil_div(d->Cout, 32) * 32;
    // dxv and the wgrad partials are never live at the same time but are kept disjoint for simplicity of reasoning
    L.partial = off;  off += align_up((size_t)6 * L.slots_per_face * TAPS * CinP * CoutP * 4, 256);
    L.bpartial = off; off += align_up((size_t)6 * L.slots_per_face * CoutP * 4, 256);
    (void)No;
    L.total = off;
    return L;
}

static void launch_pack(const void *w_eq, const void *w_pol, const void *w_np, float *out, int KS, int Cin, int Cout,
                        int transposed, int flip, hipStream_t s) {
    const int K = transposed ? Cout : Cin, Ncol = transposed ? Cin : Cout;
    const int CG = ceil_div(K, 8), NTtot = ceil_div(Ncol, 32);
    const size_t total = (size_t)3 * NTtot * CG * KS * KS * 256;
    size_t g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float *)w_eq, (const float *)w_pol,
                       (const float *)w_np, out, KS, Cin, Cout, K, Ncol, CG, NTtot, flip, transposed, total);
}

}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" size_t dlwpcs_conv_workspace_bytes(const dlwpcs_conv_desc *d) {
    if (validate(d, "conv_workspace_bytes") != DLWPCS_OK) return 0;
    return ws_layout(d).total;
}

extern "C" int dlwpcs_conv_fwd(const dlwpcs_conv_desc *d, const void *src0, const void *src1,
                               const void *w_eq, const void *w_pol, const void *w_np,
                               const void *b_eq, const void *b_pol, const void *b_np,
                               void *y, const int32_t *table_dev,
                               void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    int rc = validate(d, "conv_fwd");
    if (rc) return rc;
    if (!src0 || !w_eq || !w_pol || !y || !workspace) return fail(DLWPCS_E_INVALID, "conv_fwd: null pointer");
    if (d->C1 > 0 && !src1) return fail(DLWPCS_E_INVALID, "conv_fwd: C1 > 0 but src1 is null");
    if (d->halo && !table_dev) return fail(DLWPCS_E_INVALID, "conv_fwd: halo requested without table");
    if ((b_eq == nullptr) != (b_pol == nullptr)) return fail(DLWPCS_E_INVALID, "conv_fwd: b_eq and b_pol must both be given or both be null");
    const WsLayout L = ws_layout(d);
    if (workspace_bytes < L.total) return fail(DLWPCS_E_WORKSPACE, "conv_fwd: workspace %zu < %zu bytes", workspace_bytes, L.total);
    if (d->B == 0) return DLWPCS_OK;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int Cin = d->C0 + d->C1;
    float *wpk = (float *)(ws + L.wpk_f), *bpk = (float *)(ws + L.bias);
    launch_pack(w_eq, w_pol, w_np, wpk, d->ksize, Cin, d->Cout, 0, d->flip_north_pole, s);
    const int NTtot = ceil_div(d->Cout, 32);
    if (b_eq) {
        const int n = 3 * NTtot * 32;
        hipLaunchKernelGGL(pack_bias_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, (const float *)b_eq,
                           (const float *)b_pol, (const float *)b_np, bpk, d->Cout, NTtot * 32);
    }
    ConvKParams P{};
    P.src0 = (const float *)src0; P.src1 = (const float *)src1; P.ymask = nullptr;
    P.wpk = wpk; P.bias = b_eq ? bpk : nullptr; P.out = (float *)y; P.table = table_dev;
    P.B = d->B; P.Nin = d->N; P.No = out_size(d);
    P.C0 = d->C0; P.C1 = d->C1; P.Cin = Cin; P.Cout = d->Cout;
    P.CG = ceil_div(Cin, 8); P.NTtot = NTtot; P.up0 = d->up0;
    P.mode = d->halo ? MODE_HALO : MODE_DIRECT;
    P.act = d->act; P.alpha = d->alpha; P.vmax = d->vmax;
    const bool vec = (d->C0 % 4 == 0) && (d->C1 % 4 == 0);
    return dispatch_conv(d->ksize, vec, P, conv_work(d), s);
}

extern "C" int dlwpcs_conv_bwd_data(const dlwpcs_conv_desc *d, const void *dy, const void *y,
                                    const void *w_eq, const void *w_pol, const void *w_np,
                                    void *dsrc0, void *dsrc1, const int32_t *inv_table_dev,
                                    void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    int rc = validate(d, "conv_bwd_data");
    if (rc) return rc;
    if (!dy || !w_eq || !w_pol || !workspace) return fail(DLWPCS_E_INVALID, "conv_bwd_data: null pointer");
    if (d->act != DLWPCS_ACT_NONE && !y) return fail(DLWPCS_E_INVALID, "conv_bwd_data: activation needs the saved output y");
    if (d->halo && !inv_table_dev) return fail(DLWPCS_E_INVALID, "conv_bwd_data: halo requested without inverse table");
    if (!dsrc0 && !dsrc1) return DLWPCS_OK;
    const WsLayout L = ws_layout(d);
    if (workspace_bytes < L.total) return fail(DLWPCS_E_WORKSPACE, "conv_bwd_data: workspace %zu < %zu bytes", workspace_bytes, L.total);
    if (d->B == 0) return DLWPCS_OK;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int Cin = d->C0 + d->C1;
    float *wpk = (float *)(ws + L.wpk_b), *dxv = (float *)(ws + L.dxv);
    launch_pack(w_eq, w_pol, w_np, wpk, d->ksize, Cin, d->Cout, 1, d->flip_north_pole, s);
    const int No = out_size(d);
    ConvKParams P{};
    P.src0 = (const float *)dy; P.src1 = nullptr; P.ymask = d->act != DLWPCS_ACT_NONE ? (const float *)y : nullptr;
    P.wpk = wpk; P.bias = nullptr; P.out = dxv; P.table = nullptr;
    P.B = d->B; P.Nin = No; P.No = No + d->ksize - 1;     // full correlation: output = input + k - 1
    P.C0 = d->Cout; P.C1 = 0; P.Cin = d->Cout; P.Cout = Cin;
    P.CG = ceil_div(d->Cout, 8); P.NTtot = ceil_div(Cin, 32); P.up0 = 0;
    P.mode = MODE_ZERO;
    P.act = DLWPCS_ACT_NONE; P.alpha = d->alpha; P.vmax = d->vmax;
    rc = dispatch_conv(d->ksize, d->Cout % 4 == 0, P, conv_work(d), s);
    if (rc) return rc;
    // dxv is the gradient of the (halo-padded, if halo) virtual input: (B,6,Nv,Nv,Cin), Nv = No + k - 1
    // halo: Nv = N + 2; plain: Nv = N.  Route to the sources (inverse halo gather, upsample adjoint, channel split).
    if (dsrc0) {
        rc = launch_src_grad(dxv, (float *)dsrc0, inv_table_dev, d->B, d->N, Cin, 0, d->C0, d->up0, d->halo, s);
        if (rc) return rc;
    }
    if (dsrc1 && d->C1 > 0) {
        rc = launch_src_grad(dxv, (float *)dsrc1, inv_table_dev, d->B, d->N, Cin, d->C0, d->C1, 0, d->halo, s);
        if (rc) return rc;
    }
    return DLWPCS_OK;
}

extern "C" int dlwpcs_conv_bwd_weights(const dlwpcs_conv_desc *d, const void *src0, const void *src1, const void *dy,
                                       const void *y, void *dw_eq, void *dw_pol, void *dw_np,
                                       void *db_eq, void *db_pol, void *db_np, const int32_t *table_dev,
                                       void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    int rc = validate(d, "conv_bwd_weights");
    if (rc) return rc;
    if (!src0 || !dy || !dw_eq || !dw_pol || !workspace) return fail(DLWPCS_E_INVALID, "conv_bwd_weights: null pointer");
    if (d->C1 > 0 && !src1) return fail(DLWPCS_E_INVALID, "conv_bwd_weights: C1 > 0 but src1 is null");
    if (d->act != DLWPCS_ACT_NONE && !y) return fail(DLWPCS_E_INVALID, "conv_bwd_weights: activation needs the saved output y");
    if (d->halo && !table_dev) return fail(DLWPCS_E_INVALID, "conv_bwd_weights: halo requested without table");
    if ((db_np != nullptr) != (dw_np != nullptr) && db_eq) return fail(DLWPCS_E_INVALID, "conv_bwd_weights: dw_np/db_np must match");
    const WsLayout L = ws_layout(d);
    if (workspace_bytes < L.total) return fail(DLWPCS_E_WORKSPACE, "conv_bwd_weights: workspace %zu < %zu bytes", workspace_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int Cin = d->C0 + d->C1, KS = d->ksize, TAPS = KS * KS;
    const int CinP = ceil_div(Cin, 32) * 32, CoutP = ceil_div(d->Cout, 32) * 32;
    if (d->B == 0) {
        (void)hipMemsetAsync(dw_eq, 0, (size_t)TAPS * Cin * d->Cout * 4, s);
        (void)hipMemsetAsync(dw_pol, 0, (size_t)TAPS * Cin * d->Cout * 4, s);
        if (dw_np) (void)hipMemsetAsync(dw_np, 0, (size_t)TAPS * Cin * d->Cout * 4, s);
        if (db_eq) (void)hipMemsetAsync(db_eq, 0, (size_t)d->Cout * 4, s);
        if (db_pol) (void)hipMemsetAsync(db_pol, 0, (size_t)d->Cout * 4, s);
        if (db_np) (void)hipMemsetAsync(db_np, 0, (size_t)d->Cout * 4, s);
        return DLWPCS_OK;
    }
    WgradKParams W{};
    ConvKParams &P = W.c;
    P.src0 = (const float *)src0; P.src1 = (const float *)src1; P.ymask = nullptr; P.table = table_dev;
    P.B = d->B; P.Nin = d->N; P.No = out_size(d);
    P.C0 = d->C0; P.C1 = d->C1; P.Cin = Cin; P.Cout = d->Cout; P.up0 = d->up0;
    P.mode = d->halo ? MODE_HALO : MODE_DIRECT;
    P.alpha = d->alpha; P.vmax = d->vmax;
    P.pix_per_block = L.wg_pix; P.nblk_face = L.wg_nblk;
    P.W2 = P.No + KS - 1; P.magicW2 = div_magic(P.W2); P.magicNo = div_magic(P.No);
    P.tile_rows_max = tile_rows_for(L.wg_pix, P.No) + (KS - 1);
    W.dy = (const float *)dy; W.y = (const float *)y;
    W.partial = (float *)(ws + L.partial);
    const bool want_bias = db_eq || db_pol || db_np;
    W.bpartial = want_bias ? (float *)(ws + L.bpartial) : nullptr;
    W.CinP = CinP; W.CoutP = CoutP; W.NB = L.NB; W.ngroups = L.ngroups;
    W.mask_act = d->act != DLWPCS_ACT_NONE;
    const int pix_cap = (L.wg_pix + 1) & ~1;
    size_t lds = ((size_t)P.tile_rows_max * P.W2 * 32 + (size_t)pix_cap * 32) * 4 + (size_t)pix_cap * 4;
    if (lds < 4 * 1024 * 4) lds = 4 * 1024 * 4;      // the 16 KB cross-wave reduction scratch aliases the tiles
    if (lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: LDS tile of %zu bytes exceeds 160 KiB", lds);
    const bool vec = (d->C0 % 4 == 0) && (d->C1 % 4 == 0);
    dim3 grid((unsigned)(6 * L.wg_nblk * L.ngroups), (unsigned)(CinP / 32), (unsigned)(CoutP / 32));
#define WG_LAUNCH(KSV, VECV)                                                                                              \
    do {                                                                                                                  \
        auto kern = wgrad_mfma_kernel<KSV, VECV>;                                                                         \
        if (lds > 64 * 1024) {                                                                                            \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));    \
        }                                                                                                                 \
        int pidx = -1;                                                                                                    \
        if (prof_enabled()) {                                                                                             \
            const Work wk = conv_work(d);                                                                                 \
            pidx = prof_begin("wgrad_mfma_kernel<" #KSV ", " #VECV ">", wk.flops, wk.bytes, s);                           \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, W);                                                             \
        if (pidx >= 0) prof_end(pidx, s);                                                                                 \
    } while (0)
    if (KS == 3) { if (vec) WG_LAUNCH(3, true); else WG_LAUNCH(3, false); }
    else { if (vec) WG_LAUNCH(1, true); else WG_LAUNCH(1, false); }
#undef WG_LAUNCH
    rc = check_launch("wgrad_mfma");
    if (rc) return rc;
    const int nout = TAPS * Cin * d->Cout + (want_bias ? d->Cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(nout, 256)), dim3(256), 0, s, W.partial, W.bpartial,
                       (float *)dw_eq, (float *)dw_pol, (float *)dw_np, (float *)db_eq, (float *)db_pol, (float *)db_np,
                       KS, Cin, d->Cout, CinP, CoutP, L.slots_per_face, d->flip_north_pole);
    return check_launch("wgrad_reduce");
}
