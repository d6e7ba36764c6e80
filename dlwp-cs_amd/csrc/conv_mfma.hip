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
#include <stdlib.h>
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
    uint32_t magicW2, magicNo, magicN, magicN2;
    int tile_rows_max;          // rows reserved in LDS
    int ntiles;                 // B * 6 * nblk_face (persistent kernel)
    long long *dbg;             // development only (-DDLWPCS_TIMELINE): s_memtime checkpoints [nblocks][64]
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

// VW consecutive channels as one register vector
template <int VW> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };

__device__ __forceinline__ void vzero(float &v) { v = 0.f; }
__device__ __forceinline__ void vzero(float2 &v) { v = make_float2(0.f, 0.f); }
__device__ __forceinline__ void vzero(float4 &v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vmask(float &v, const float &y, float a, float m) { v *= act_leaky_clip_grad_from_y(y, a, m); }
__device__ __forceinline__ void vmask(float2 &v, const float2 &y, float a, float m) {
    v.x *= act_leaky_clip_grad_from_y(y.x, a, m); v.y *= act_leaky_clip_grad_from_y(y.y, a, m);
}
__device__ __forceinline__ void vmask(float4 &v, const float4 &y, float a, float m) {
    v.x *= act_leaky_clip_grad_from_y(y.x, a, m); v.y *= act_leaky_clip_grad_from_y(y.y, a, m);
    v.z *= act_leaky_clip_grad_from_y(y.z, a, m); v.w *= act_leaky_clip_grad_from_y(y.w, a, m);
}
__device__ __forceinline__ float vsel(bool c, float v) { return c ? v : 0.f; }
__device__ __forceinline__ float2 vsel(bool c, float2 v) { return c ? v : make_float2(0.f, 0.f); }
__device__ __forceinline__ float4 vsel(bool c, float4 v) { return c ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

// Per-sample-relative element offsets of tile pixel `pix` in src0 (off0) and src1 (off1); -1 = zero cell.
// This is where the cube-sphere halo (table gather), the nearest-upsampling of src0 and the zero border of the data
// gradient are resolved.  It runs ONCE per tile and item (the result is channel-chunk invariant) so that the per-chunk
// fetch below is straight-line code: every load of a chunk is issued back to back and stays in flight.
template <int KS>
__device__ __forceinline__ void source_offsets(const ConvKParams &P, int f, int y0, int pix, bool in_tile, int &off0, int &off1) {
    const int ty = __umulhi((uint32_t)pix, P.magicW2);
    const int tx = pix - ty * P.W2;
    int vf, vy, vx;
    const bool ok = in_tile && resolve_cell<KS>(P, f, y0 + ty, tx, vf, vy, vx);
    if (!ok) { off0 = -1; off1 = -1; return; }
    const int g = P.up0 ? (P.Nin >> 1) : P.Nin;
    const int sy = P.up0 ? (vy >> 1) : vy, sx = P.up0 ? (vx >> 1) : vx;
    off0 = ((vf * g + sy) * g + sx) * P.C0;
    off1 = ((vf * P.Nin + vy) * P.Nin + vx) * P.C1;
}

// ------------------------------------------------------------------------------------------------------------------
// Forward / data-gradient kernel: software-pipelined over KC-channel chunks.
//   chunk c+1 is fetched from HBM/L2 into registers (input tile with halo + packed weights) while the matrix cores work
//   on chunk c out of LDS; the registers are written to the other LDS buffer afterwards; ONE barrier per chunk.
//   LDS per buffer: input tile rows*W2 pixels x (KC+4) floats + weights NTB*KCG*TAPS*256 floats; two buffers; sized so
//   that two workgroups fit a CU (KC = 8: 2*(24+9) KB at N = 48), i.e. two waves per SIMD feed each matrix core.
//   MASK: data-gradient mode with an activation: dz = dy * act'(y) applied while fetching.
// ------------------------------------------------------------------------------------------------------------------
template <int KS, int KC, int MT, int NT, int WM, int WN, int VW, bool MASK>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_mfma_kernel(const ConvKParams P) {
    constexpr int TAPS = KS * KS;
    constexpr int KCP = KC + 4;
    constexpr int KCG = KC / 8;
    constexpr int Q = KC / VW;                      // vectors per pixel per chunk
    constexpr int NTB = NT * WN;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int IT_IN = 3 * KC / VW;              // input vectors per thread per chunk: capacity 3*NTHREADS pixels
    constexpr int WF4 = NTB * KCG * TAPS * 64;      // float4 per weight chunk
    constexpr int IT_W = (WF4 + NTHREADS - 1) / NTHREADS;
    constexpr int GF4 = TAPS * 64;                  // float4 per (n tile, channel group)
    static_assert(NTHREADS % Q == 0, "thread -> channel-vector mapping must not depend on the item");
    typedef typename VecT<VW>::type V;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int in_floats = P.tile_rows_max * P.W2 * KCP;
    const int buf_floats = in_floats + WF4 * 4;

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
    const int nitems = rows * P.W2 * Q;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int v = f < 4 ? 0 : (f == 4 ? 1 : 2);
    const float4 *wsrc = reinterpret_cast<const float4 *>(P.wpk);

    // sample bases of the sources (src1 aliases src0 when absent so that the straight-line fetch never dereferences null)
    const int g0 = P.up0 ? (P.Nin >> 1) : P.Nin;
    const float *s0b = P.src0 + (size_t)b * 6 * g0 * g0 * P.C0;
    const float *s1b = P.C1 > 0 ? P.src1 + (size_t)b * 6 * P.Nin * P.Nin * P.C1 : s0b;
    const float *ymb = MASK ? P.ymask + (size_t)b * 6 * g0 * g0 * P.C0 : nullptr;
    const int qv = (tid % Q) * VW;                  // channel offset of this thread's vectors inside a chunk

    // per-lane LDS offset (floats) of the A operand for each of this wave's M tiles
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

    V pre_in[IT_IN];
    float4 pre_w[IT_W];
    int off0[IT_IN], off1[IT_IN];

    auto prepare = [&](int batch) {
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) {
            const int e = tid + (batch * IT_IN + i) * NTHREADS;
            source_offsets<KS>(P, f, y0, e / Q, e < nitems, off0[i], off1[i]);
        }
    };
    // straight-line: every load of the chunk is issued back to back, nothing waits until commit()
    auto fetch = [&](int ch, bool with_weights) {
        const int c = ch * KC + qv;
        const bool c_ok = c < P.Cin;
        const bool from0 = c < P.C0;
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) {
            const int o = from0 ? off0[i] : off1[i];
            const bool ok = c_ok && o >= 0;
            const float *ptr = ok ? (from0 ? s0b + (size_t)o + c : s1b + (size_t)o + (c - P.C0)) : s0b;
            V val = *reinterpret_cast<const V *>(ptr);
            if (MASK) {
                const float *yp = ok ? ymb + (size_t)o + c : ymb;
                vmask(val, *reinterpret_cast<const V *>(yp), P.alpha, P.vmax);
            }
            pre_in[i] = vsel(ok, val);
        }
        if (with_weights) {
#pragma unroll
            for (int i = 0; i < IT_W; ++i) {
                const int idx = min(tid + i * NTHREADS, WF4 - 1);
                const int g = idx / GF4, w = idx % GF4;
                const int ntl = g / KCG, cgl = g % KCG;
                const int ntile = nt0 + ntl, cg = ch * KCG + cgl;
                const bool ok = ntile < P.NTtot && cg < P.CG;
                const float4 val = wsrc[ok ? (((size_t)v * P.NTtot + ntile) * P.CG + cg) * GF4 + w : 0];
                pre_w[i] = vsel(ok, val);
            }
        }
    };
    auto commit = [&](float *buf, int batch, bool with_weights) {
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) {
            const int e = tid + (batch * IT_IN + i) * NTHREADS;
            if (e < nitems) *reinterpret_cast<V *>(buf + (e / Q) * KCP + qv) = pre_in[i];
        }
        if (with_weights) {
#pragma unroll
            for (int i = 0; i < IT_W; ++i) {
                const int idx = tid + i * NTHREADS;
                if (idx < WF4) reinterpret_cast<float4 *>(buf + in_floats)[idx] = pre_w[i];
            }
        }
    };
    auto compute = [&](const float *buf) {
        const float *lds_in = buf, *lds_w = buf + in_floats;
#pragma unroll
        for (int cgl = 0; cgl < KCG; ++cgl) {
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
    };

    const int nchunks = (P.CG + KCG - 1) / KCG;
    const int nbatch = (nitems + IT_IN * NTHREADS - 1) / (IT_IN * NTHREADS);
    if (nbatch == 1) {
        prepare(0);
        fetch(0, true);
        commit(smem, 0, true);
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            float *cur = smem + (ch & 1) * buf_floats;
            float *nxt = smem + ((ch + 1) & 1) * buf_floats;
            const bool more = ch + 1 < nchunks;
            if (more) fetch(ch + 1, true);       // global loads stay in flight while the matrix cores run
                compute(cur);
                if (more) commit(nxt, 0, true);
                __syncthreads();
            }
    } else {
        // tile wider than the register prefetch capacity (N >~ 256): stage synchronously, single buffer
        for (int ch = 0; ch < nchunks; ++ch) {
            for (int bt = 0; bt < nbatch; ++bt) { prepare(bt); fetch(ch, bt == 0); commit(smem, bt, bt == 0); }
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
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
// Persistent, software-pipelined form of the forward / data-gradient kernel (the one the hot path runs).
//
// Why: a per-tile workgroup spends ~30% of its life in a prologue (halo-table reads -> offsets -> first chunk) and an
// epilogue (bias, activation, 48 stores per lane) during which its waves issue no MFMA, and because all workgroups of a
// launch do the same work they hit those phases -- and the memory system -- in lock step across the chip.
// Here a workgroup loops over a static, strided list of tiles and treats the (tile, channel-chunk) pairs as ONE stream:
//     iteration g:  issue the global loads of chunk g+1 (possibly the first chunk of the NEXT tile)
//                   read the halo table of the next tile (at the first chunk of a tile)
//                   MFMAs of chunk g out of LDS buffer g&1
//                   write chunk g+1 from registers to LDS buffer (g+1)&1; table entries -> offsets of the next tile
//                   (last chunk of a tile) bias + activation + stores of the finished tile, accumulators reset
//                   one barrier
// so after the first tile nothing the matrix cores need is ever waited for: loads have a whole chunk of MFMAs (~7k
// cycles) to land, stores drain behind the next tile's MFMAs.  Needs >= 2 chunks per tile (C_in > KC).
// ------------------------------------------------------------------------------------------------------------------
#ifdef DLWPCS_TIMELINE
#define TL_MARK() do { if (tlp && tli < 64) tlp[tli++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TL_MARK() do { } while (0)
#endif

template <int KS, int KC, int MT, int NT, int WM, int WN, int VW, bool MASK>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_mfma_pp_kernel(const ConvKParams P) {
    constexpr int TAPS = KS * KS;
    constexpr int KCP = KC + 4;
    constexpr int KCG = KC / 8;
    constexpr int Q = KC / VW;
    constexpr int NTB = NT * WN;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int IT_IN = 3 * KC / VW;
    constexpr int WF4 = NTB * KCG * TAPS * 64;
    constexpr int IT_W = (WF4 + NTHREADS - 1) / NTHREADS;
    constexpr int GF4 = TAPS * 64;
    static_assert(NTHREADS % Q == 0, "thread -> channel-vector mapping must not depend on the item");
    typedef typename VecT<VW>::type V;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int in_floats = P.tile_rows_max * P.W2 * KCP;
    const int buf_floats = in_floats + WF4 * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int nt0 = blockIdx.y * NTB;
    const int face_pix = P.No * P.No;
    const int g0 = P.up0 ? (P.Nin >> 1) : P.Nin;
    const int qv = (tid % Q) * VW;
    const float4 *wsrc = reinterpret_cast<const float4 *>(P.wpk);
    const int G = gridDim.x;
#ifdef DLWPCS_TIMELINE
    int tli = 0;
    long long *tlp = (P.dbg && tid == 0 && blockIdx.y == 0) ? P.dbg + (size_t)blockIdx.x * 64 : nullptr;
#endif

    struct Geo { int b, f, v, m0, npix, y0, nitems; };
    auto geo_of = [&](int t) {
        Geo gq;
        const uint32_t L = xcd_remap((uint32_t)t, (uint32_t)P.ntiles);
        const int blk = L % P.nblk_face;
        gq.f = (L / P.nblk_face) % 6;
        gq.b = L / (P.nblk_face * 6);
        gq.v = gq.f < 4 ? 0 : (gq.f == 4 ? 1 : 2);
        gq.m0 = blk * P.pix_per_block;
        gq.npix = min(P.pix_per_block, face_pix - gq.m0);
        gq.y0 = __umulhi((uint32_t)gq.m0, P.magicNo);
        const int ylast = __umulhi((uint32_t)(gq.m0 + gq.npix - 1), P.magicNo);
        gq.nitems = (ylast - gq.y0 + KS) * P.W2 * Q;
        return gq;
    };

    // bias of the three face variants, read once
    float bias_v[3][NT];
#pragma unroll
    for (int vv = 0; vv < 3; ++vv)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = (nt0 + wn * NT + nt) * 32 + l31;
            bias_v[vv][nt] = (P.bias && co < P.Cout) ? P.bias[(size_t)vv * P.NTtot * 32 + co] : 0.f;
        }

    f32x16 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    zero_acc();

    int abase[MT];
    auto set_abase = [&](const Geo &gq) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = (wm * MT + mt) * 32 + l31;
            int base = 0;
            if (m < gq.npix) {
                const int gm = gq.m0 + m;
                const int oy = __umulhi((uint32_t)gm, P.magicNo);
                const int ox = gm - oy * P.No;
                base = ((oy - gq.y0) * P.W2 + ox) * KCP;
            }
            abase[mt] = base + half * 4;
        }
    };

    V pre_in[IT_IN];
    float4 pre_w[IT_W];
    int sidx_c[IT_IN];      // flat source index (face*Nin + row)*Nin + col of each tile pixel, current tile; -1 = zero
    int sidx_n[IT_IN];      // same for the next tile (filled by the halo-table loads in flight)

    // read the halo table (or form the identity / zero-border index) for every tile pixel of a tile
    auto issue_tables = [&](const Geo &gq, int (&sidx)[IT_IN]) {
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) {
            const int e = tid + i * NTHREADS;
            const int ec = min(e, gq.nitems - 1);
            const int pix = ec / Q;
            const int ty = __umulhi((uint32_t)pix, P.magicW2);
            const int tx = pix - ty * P.W2;
            const int iy = gq.y0 + ty;
            int val;
            if (P.mode == MODE_HALO) {
                const int M = P.Nin + KS - 1;
                val = P.table[(gq.f * M + iy) * M + tx];
            } else if (P.mode == MODE_DIRECT) {
                val = (gq.f * P.Nin + iy) * P.Nin + tx;
            } else {    // MODE_ZERO: zero border of width KS-1
                const int vy = iy - (KS - 1), vx = tx - (KS - 1);
                const bool ok = (vy >= 0) & (vy < P.Nin) & (vx >= 0) & (vx < P.Nin);
                val = ok ? (gq.f * P.Nin + vy) * P.Nin + vx : -1;
            }
            sidx[i] = e < gq.nitems ? val : -1;
        }
    };
    // straight-line: all loads of the chunk are issued back to back; offsets derived from the flat source index
    auto fetch = [&](const Geo &gq, int ch, const int (&sidx)[IT_IN]) {
        const float *s0b = P.src0 + (size_t)gq.b * 6 * g0 * g0 * P.C0;
        const float *s1b = P.C1 > 0 ? P.src1 + (size_t)gq.b * 6 * P.Nin * P.Nin * P.C1 : s0b;
        const float *ymb = MASK ? P.ymask + (size_t)gq.b * 6 * g0 * g0 * P.C0 : nullptr;
        const int c = ch * KC + qv;
        const bool c_ok = c < P.Cin;
        const bool from0 = c < P.C0;
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) {
            const int idx = sidx[i];
            const bool ok = c_ok && idx >= 0;
            const int ii = ok ? idx : 0;
            int o;
            if (from0) {
                if (P.up0) {
                    const int vf = __umulhi((uint32_t)ii, P.magicN2);
                    const int rem = ii - vf * P.Nin * P.Nin;
                    const int vy = __umulhi((uint32_t)rem, P.magicN);
                    const int vx = rem - vy * P.Nin;
                    o = ((vf * g0 + (vy >> 1)) * g0 + (vx >> 1)) * P.C0 + c;
                } else {
                    o = ii * P.C0 + c;
                }
            } else {
                o = ii * P.C1 + (c - P.C0);
            }
            const float *ptr = (from0 ? s0b : s1b) + (ok ? (size_t)o : 0);
            V val = *reinterpret_cast<const V *>(ptr);
            if (MASK) vmask(val, *reinterpret_cast<const V *>(ymb + (ok ? (size_t)o : 0)), P.alpha, P.vmax);
            pre_in[i] = vsel(ok, val);
        }
#pragma unroll
        for (int i = 0; i < IT_W; ++i) {
            const int idx = min(tid + i * NTHREADS, WF4 - 1);
            const int g = idx / GF4, w = idx % GF4;
            const int ntl = g / KCG, cgl = g % KCG;
            const int ntile = nt0 + ntl, cg = ch * KCG + cgl;
            const bool ok = ntile < P.NTtot && cg < P.CG;
            const float4 val = wsrc[ok ? (((size_t)gq.v * P.NTtot + ntile) * P.CG + cg) * GF4 + w : 0];
            pre_w[i] = vsel(ok, val);
        }
    };
    auto commit = [&](float *buf, const Geo &gq) {
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) {
            const int e = tid + i * NTHREADS;
            if (e < gq.nitems) *reinterpret_cast<V *>(buf + (e / Q) * KCP + qv) = pre_in[i];
        }
#pragma unroll
        for (int i = 0; i < IT_W; ++i) {
            const int idx = tid + i * NTHREADS;
            if (idx < WF4) reinterpret_cast<float4 *>(buf + in_floats)[idx] = pre_w[i];
        }
    };
    auto compute = [&](const float *buf) {
        const float *lds_in = buf, *lds_w = buf + in_floats;
#pragma unroll
        for (int cgl = 0; cgl < KCG; ++cgl) {
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
    };
    // bias + activation + stores of a finished tile (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
    // Wide path (C_out % 4 == 0): each half M tile (16 pixels x 32 channels) is transposed through a wave-private
    // 16 x 36-float LDS patch so that every lane stores 16 B and 8 lanes cover one 128-B line: 4 dwordx4 stores per
    // M tile instead of 16 dword stores (the store ISSUE rate, not bandwidth, is what the epilogue costs).
    float *stage = smem + 2 * buf_floats + wave * (16 * 36);
    auto epilogue = [&](const Geo &gq) {
        float *outp = P.out + ((size_t)gq.b * 6 + gq.f) * face_pix * P.Cout;
        const bool wide = (P.Cout & 3) == 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cot = (nt0 + wn * NT + nt) * 32;
            const int co = cot + l31;
            const float bv = gq.v == 0 ? bias_v[0][nt] : (gq.v == 1 ? bias_v[1][nt] : bias_v[2][nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (wide) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const int r = h * 8 + rr;
                            float val = acc[mt][nt][r] + bv;
                            if (P.act == DLWPCS_ACT_LEAKY_CLIP) val = act_leaky_clip(val, P.alpha, P.vmax);
                            stage[((rr & 3) + 8 * (rr >> 2) + 4 * half) * 36 + l31] = val;
                        }
                        // wave-private patch: a wave executes in lock step, LDS ops complete in order -> no barrier
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int row = (lane >> 3) + 8 * j;
                            const int quad = lane & 7;
                            const float4 v4 = *reinterpret_cast<const float4 *>(stage + row * 36 + quad * 4);
                            const int m = (wm * MT + mt) * 32 + h * 16 + row;
                            const int c4 = cot + quad * 4;
                            if (m < gq.npix && c4 < P.Cout)
                                *reinterpret_cast<float4 *>(outp + (size_t)(gq.m0 + m) * P.Cout + c4) = v4;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                } else if (co < P.Cout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (m < gq.npix) {
                            float val = acc[mt][nt][r] + bv;
                            if (P.act == DLWPCS_ACT_LEAKY_CLIP) val = act_leaky_clip(val, P.alpha, P.vmax);
                            outp[(size_t)(gq.m0 + m) * P.Cout + co] = val;
                        }
                    }
                }
            }
        }
    };

    const int nchunks = (P.CG + KCG - 1) / KCG;          // >= 2 (launcher guarantees)
    int t = blockIdx.x;
    if (t >= P.ntiles) return;
    TL_MARK();
    Geo cur = geo_of(t);
    issue_tables(cur, sidx_c);
    set_abase(cur);
    fetch(cur, 0, sidx_c);
    commit(smem, cur);
    __syncthreads();
    TL_MARK();
    int gpar = 0;
    // One tile per trip; the chunk loop is peeled into first / middle / last so that every register array (prefetch
    // registers, source indices, accumulators) is defined unconditionally on every path (no phi copies -> no spills).
    // On the final tile the "next tile" is the tile itself: its table reads and first chunk are fetched again and
    // never used, which keeps the code straight-line.
    while (true) {
        const bool have_next = t + G < P.ntiles;
        const Geo nxt = geo_of(have_next ? t + G : t);
        // ---- first chunk: also start reading the next tile's halo-table entries
        {
            float *bcur = smem + gpar * buf_floats, *bnxt = smem + (gpar ^ 1) * buf_floats;
            fetch(cur, 1, sidx_c);
            issue_tables(nxt, sidx_n);
            TL_MARK();
            compute(bcur);
            TL_MARK();
            commit(bnxt, cur);
            TL_MARK();
            __syncthreads();
            gpar ^= 1;
        }
        // ---- middle chunks
        for (int c = 1; c < nchunks - 1; ++c) {
            float *bcur = smem + gpar * buf_floats, *bnxt = smem + (gpar ^ 1) * buf_floats;
            fetch(cur, c + 1, sidx_c);
            compute(bcur);
            commit(bnxt, cur);
            __syncthreads();
            gpar ^= 1;
        }
        // ---- last chunk: the stream moves on to the next tile; finish this one
        {
            float *bcur = smem + gpar * buf_floats, *bnxt = smem + (gpar ^ 1) * buf_floats;
            fetch(nxt, 0, sidx_n);
            TL_MARK();
            compute(bcur);
            TL_MARK();
            commit(bnxt, nxt);
            epilogue(cur);
            zero_acc();
            TL_MARK();
            __syncthreads();
            gpar ^= 1;
        }
        if (!have_next) break;
        t += G;
        cur = nxt;
#pragma unroll
        for (int i = 0; i < IT_IN; ++i) sidx_c[i] = sidx_n[i];
        set_abase(cur);
    }
    TL_MARK();
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
    ConvKParams c;          // description of the virtual input (src0/src1/table/mode/...), ymask unused here
    const float *dy, *y;    // (B,6,No,No,Cout); y nullable
    float *partial;         // [nworkers][TAPS][CinP][CoutP]
    float *bpartial;        // [nworkers][CoutP] or nullptr
    int CinP, CoutP;        // multiples of 32
    int n_eq, n_4, n_5;     // workers per face class (equatorial faces 0-3 / face 4 / face 5); grid.x = their sum
    int pipelined;          // two LDS buffers + register prefetch (else synchronous staging, one buffer)
    uint32_t magicN, magicN2;
};

// Weight-gradient kernel, persistent form.  GEMM view per tap: D[ci][co] += sum_pixels Xpad[pixel+tap][ci] * dZ[pixel][co],
// dZ = dy * act'(y).  The accumulators D (k*k taps x 32 x 32, 9 x 16 VGPRs per lane) do not depend on WHICH pixels are
// summed, so a worker (workgroup of 8 waves, one per CU) owns one (ci tile, co tile) pair and streams through a STATIC,
// strided list of work items (sample, face, band of <= 192 pixels) of its face class; per item the 8 waves split the
// pixel pairs (MFMA K = 2 pixels).  Two-deep software pipeline across items: while item t is on the matrix cores, the
// X / dZ tiles of item t+1 are in flight to registers and the halo-table entries of item t+2 are being read.
// At the end the 8 waves are summed through LDS in a fixed order and ONE partial per worker is written; a second kernel
// adds the workers' partials in a fixed order per face class (no atomics -> bitwise reproducible) and applies the
// weight-group map (class 0 -> equatorial kernel, 1 -> polar, 2 -> polar or north pole, tap rows reversed when flipping).
template <int KS, int VW, bool MASK>
__global__ void __launch_bounds__(512, 2) wgrad_mfma_kernel(const WgradKParams W) {
    constexpr int TAPS = KS * KS;
    constexpr int XS = 32;                  // X tile row stride (floats) = the 32 input channels of this ci tile
    constexpr int QX = 32 / VW;             // vectors per X pixel
    constexpr int IT_X = 28 / VW;           // X vectors per thread per item: capacity 448 tile pixels
    constexpr int IT_DY = 3;                // dZ float4 per thread per item: capacity 192 pixels (x 8 float4)
    constexpr int NT_ = 512;
    typedef typename VecT<VW>::type V;
    const ConvKParams &P = W.c;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pix_cap = (P.pix_per_block + 1) & ~1;
    const int x_floats = P.tile_rows_max * P.W2 * XS;
    const int buf_floats = x_floats + pix_cap * 32 + pix_cap;      // X tile, dZ tile, per-pixel X offsets
    // the 32 KB cross-wave reduction scratch aliases the buffers after the main loop

    const int worker = blockIdx.x;
    const int cit = blockIdx.y, cot = blockIdx.z;
    int j, nj, nfaces, fbase;
    if (worker < W.n_eq) { j = worker; nj = W.n_eq; nfaces = 4; fbase = 0; }
    else if (worker < W.n_eq + W.n_4) { j = worker - W.n_eq; nj = W.n_4; nfaces = 1; fbase = 4; }
    else { j = worker - W.n_eq - W.n_4; nj = W.n_5; nfaces = 1; fbase = 5; }
    const int nbands = P.nblk_face;
    const int total_items = P.B * nfaces * nbands;
    const int n_my = j < total_items ? (total_items - j + nj - 1) / nj : 0;

    const int face_pix = P.No * P.No;
    const bool vec_dy = (P.Cout % 4 == 0);
    const int nitems_dy = vec_dy ? pix_cap * 8 : pix_cap * 32;
    const int M = P.Nin + KS - 1;           // padded face size (MODE_HALO)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int cx = cit * 32 + (tid % QX) * VW;          // this thread's input channel(s): fixed for the whole kernel
    const bool cx_ok = cx < P.Cin;
    const bool from0 = cx < P.C0;
    const int g0 = P.up0 ? (P.Nin >> 1) : P.Nin;
    const int csub = from0 ? cx : cx - P.C0;
    const int cstride = from0 ? P.C0 : P.C1;

    struct Item { int b, f, m0, npix, y0, rows; };
    auto item_of = [&](int k) {
        Item it;
        const int t = j + k * nj;
        const int band = t % nbands;
        const int r = t / nbands;
        it.f = fbase + r % nfaces;
        it.b = r / nfaces;
        it.m0 = band * P.pix_per_block;
        it.npix = min(P.pix_per_block, face_pix - it.m0);
        it.y0 = __umulhi((uint32_t)it.m0, P.magicNo);
        const int ylast = __umulhi((uint32_t)(it.m0 + it.npix - 1), P.magicNo);
        it.rows = ylast - it.y0 + KS;
        return it;
    };

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;   // bias partial: thread (co = tid&31, part = tid>>5)

    V pre_x[IT_X];
    float4 pre_dy[IT_DY];
    int tbl[IT_X];      // stage A: raw halo-table entries of item k+2 (or tile coordinates when there is no halo)
    int off[IT_X];      // stage B: element offsets (per sample) of item k+1, -1 = nothing to load

    // stage A: read the halo table for the tile pixels of an item (flat source index on the Nin grid)
    auto issue_table = [&](const Item &it, int batch) {
        const int nitems_x = it.rows * P.W2 * QX;
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int e = min(tid + (batch * IT_X + i) * NT_, nitems_x - 1);
            const int pix = e / QX;
            const int ty = __umulhi((uint32_t)pix, P.magicW2);
            const int tx = pix - ty * P.W2;
            if (P.mode == MODE_HALO) tbl[i] = P.table[(it.f * M + it.y0 + ty) * M + tx];
            else tbl[i] = (it.f * P.Nin + it.y0 + ty) * P.Nin + tx;          // MODE_DIRECT: identity
        }
    };
    // stage B: table entry -> element offset into the source this thread reads (upsampling folded in)
    auto make_offsets = [&](const Item &it, int batch) {
        const int nitems_x = it.rows * P.W2 * QX;
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int e = tid + (batch * IT_X + i) * NT_;
            const int idx = tbl[i];
            const int vf = __umulhi((uint32_t)idx, W.magicN2);
            const int rem = idx - vf * P.Nin * P.Nin;
            const int vy = __umulhi((uint32_t)rem, W.magicN);
            const int vx = rem - vy * P.Nin;
            int o;
            if (from0) {
                const int sy = P.up0 ? (vy >> 1) : vy, sx = P.up0 ? (vx >> 1) : vx;
                o = ((vf * g0 + sy) * g0 + sx) * P.C0;
            } else {
                o = idx * P.C1;
            }
            off[i] = (e < nitems_x && cx_ok) ? o + csub : -1;
        }
        (void)cstride;
    };
    // stage C: straight-line loads of the X tile (through the offsets) and of the dZ tile
    auto issue_data = [&](const Item &it, int batch) {
        const float *sb = from0 ? P.src0 + (size_t)it.b * 6 * g0 * g0 * P.C0
                                : P.src1 + (size_t)it.b * 6 * P.Nin * P.Nin * P.C1;
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const bool ok = off[i] >= 0;
            const V val = *reinterpret_cast<const V *>(ok ? sb + (size_t)off[i] : P.src0);
            pre_x[i] = vsel(ok, val);
        }
        const size_t rowbase = (((size_t)it.b * 6 + it.f) * face_pix + it.m0) * P.Cout;
        const float *dyb = W.dy + rowbase;
        const float *yb = MASK ? W.y + rowbase : nullptr;
#pragma unroll
        for (int i = 0; i < IT_DY; ++i) {
            const int e = tid + (batch * IT_DY + i) * NT_;
            if (vec_dy) {
                const int k = e >> 3, co = cot * 32 + (e & 7) * 4;
                const bool ok = e < nitems_dy && k < it.npix && co < P.Cout;
                const size_t o = ok ? (size_t)k * P.Cout + co : 0;
                float4 g = *reinterpret_cast<const float4 *>(dyb + o);
                if (MASK) vmask(g, *reinterpret_cast<const float4 *>(yb + o), P.alpha, P.vmax);
                pre_dy[i] = vsel(ok, g);
            } else {
                // scalar dZ path (Cout % 4 != 0): 4 consecutive scalars per slot
                float gs[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e4 = e * 4 + u;
                    const int k = e4 >> 5, co = cot * 32 + (e4 & 31);
                    const bool ok = e4 < nitems_dy && k < it.npix && co < P.Cout;
                    const size_t o = ok ? (size_t)k * P.Cout + co : 0;
                    float g = dyb[o];
                    if (MASK) vmask(g, yb[o], P.alpha, P.vmax);
                    gs[u] = ok ? g : 0.f;
                }
                pre_dy[i] = make_float4(gs[0], gs[1], gs[2], gs[3]);
            }
        }
    };
    auto commit = [&](float *buf, const Item &it, int batch) {
        const int nitems_x = it.rows * P.W2 * QX;
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int e = tid + (batch * IT_X + i) * NT_;
            if (e < nitems_x) *reinterpret_cast<V *>(buf + (e / QX) * XS + (tid % QX) * VW) = pre_x[i];
        }
#pragma unroll
        for (int i = 0; i < IT_DY; ++i) {
            const int e = tid + (batch * IT_DY + i) * NT_;
            // both dZ paths hold 4 consecutive floats of the [pix][32] tile per slot
            if (e * 4 < pix_cap * 32) *reinterpret_cast<float4 *>(buf + x_floats + e * 4) = pre_dy[i];
        }
        if (batch == 0) {
            int *pb = reinterpret_cast<int *>(buf + x_floats + pix_cap * 32);
            for (int k = tid; k < pix_cap; k += NT_) {
                int base = 0;
                if (k < it.npix) {
                    const int gm = it.m0 + k;
                    const int oy = __umulhi((uint32_t)gm, P.magicNo);
                    base = ((oy - it.y0) * P.W2 + (gm - oy * P.No)) * XS;
                }
                pb[k] = base;
            }
        }
    };
    const int nsteps = pix_cap / 2;
    auto compute = [&](const float *buf, const Item &it) {
        const float *lds_x = buf, *lds_dy = buf + x_floats;
        const int *lds_pb = reinterpret_cast<const int *>(buf + x_floats + pix_cap * 32);
        if (W.bpartial && cit == 0) {
            const int co = tid & 31, part = tid >> 5;
            for (int k = part; k < it.npix; k += 16) bsum += lds_dy[k * 32 + co];
        }
        for (int s = wave; s < nsteps; s += 8) {
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
    };

    if (W.pipelined) {
        if (n_my > 0) {
            const Item i0 = item_of(0);
            issue_table(i0, 0);
            make_offsets(i0, 0);
            issue_data(i0, 0);
            if (n_my > 1) issue_table(item_of(1), 0);
            commit(smem, i0, 0);
            if (n_my > 1) make_offsets(item_of(1), 0);
        }
        __syncthreads();
        for (int k = 0; k < n_my; ++k) {
            float *cur = smem + (k & 1) * buf_floats;
            float *nxt = smem + ((k + 1) & 1) * buf_floats;
            const Item it = item_of(k);
            if (k + 1 < n_my) issue_data(item_of(k + 1), 0);       // uses off[] of item k+1
            if (k + 2 < n_my) issue_table(item_of(k + 2), 0);      // table entries of item k+2 in flight
            compute(cur, it);
            if (k + 1 < n_my) commit(nxt, item_of(k + 1), 0);
            if (k + 2 < n_my) make_offsets(item_of(k + 2), 0);
            __syncthreads();
        }
    } else {
        for (int k = 0; k < n_my; ++k) {
            const Item it = item_of(k);
            const int nbx = (it.rows * P.W2 * QX + IT_X * NT_ - 1) / (IT_X * NT_);
            const int nbd = (nitems_dy + IT_DY * NT_ * (vec_dy ? 1 : 4) - 1) / (IT_DY * NT_ * (vec_dy ? 1 : 4));
            const int nb = max(nbx, nbd);
            __syncthreads();
            for (int bt = 0; bt < nb; ++bt) { issue_table(it, bt); make_offsets(it, bt); issue_data(it, bt); commit(smem, it, bt); }
            __syncthreads();
            compute(smem, it);
        }
        __syncthreads();
    }

    // cross-wave reduction through LDS (fixed order w = 0..7), one tap at a time
    float *red = smem;
    float *pout = W.partial + (size_t)worker * TAPS * W.CinP * W.CoutP;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave * 1024 + ci * 32 + l31] = acc[tap][r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * NT_;
            const float sum = ((red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e])) +
                              ((red[4096 + e] + red[5120 + e]) + (red[6144 + e] + red[7168 + e]));
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
            for (int part = 0; part < 16; ++part) s += red[part * 32 + tid];
            W.bpartial[(size_t)worker * W.CoutP + cot * 32 + tid] = s;
        }
    }
}

// Sum the per-slot partials in a fixed order and route them to the weight groups.  Workgroup = 16 outputs x 16 slot
// phases: thread (o, ph) adds slots ph, ph+16, ... (fixed order), the 16 phases are then combined through LDS in a fixed
// tree -> bitwise reproducible, and enough workgroups (outputs/16) to fill the chip.  accumulate != 0: add to the
// destination instead of overwriting it (shared layers / direct accumulation into the flat gradient buffer).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bpartial,
                                                           float *__restrict__ dw_eq, float *__restrict__ dw_pol,
                                                           float *__restrict__ dw_np, float *__restrict__ db_eq,
                                                           float *__restrict__ db_pol, float *__restrict__ db_np,
                                                           int KS, int Cin, int Cout, int CinP, int CoutP,
                                                           int n_eq, int n_4, int n_5, int flip, int accumulate) {
    const int TAPS = KS * KS;
    const int nW = TAPS * Cin * Cout;
    const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + o;
    const size_t slot_stride = (size_t)TAPS * CinP * CoutP;
    float s_eq = 0.f, s_4 = 0.f, s_5 = 0.f;
    const bool is_w = e < nW, is_b = (!is_w) && bpartial && e < nW + Cout;
    const int e4 = n_eq, e5 = n_eq + n_4, e6 = n_eq + n_4 + n_5;
    if (is_w) {
        const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cout * Cin);
        const size_t off = ((size_t)tap * CinP + ci) * CoutP + co;
        const int ty = tap / KS, tx = tap % KS;
        // face 5 ran with the row-reversed kernel: its partial for tap row r belongs to kernel row KS-1-r
        const size_t o5 = flip ? ((size_t)((KS - 1 - ty) * KS + tx) * CinP + ci) * CoutP + co : off;
        for (int s = ph; s < e4; s += 16) s_eq += partial[(size_t)s * slot_stride + off];
        for (int s = e4 + ph; s < e5; s += 16) s_4 += partial[(size_t)s * slot_stride + off];
        for (int s = e5 + ph; s < e6; s += 16) s_5 += partial[(size_t)s * slot_stride + o5];
    } else if (is_b) {
        const int co = e - nW;
        for (int s = ph; s < e4; s += 16) s_eq += bpartial[(size_t)s * CoutP + co];
        for (int s = e4 + ph; s < e5; s += 16) s_4 += bpartial[(size_t)s * CoutP + co];
        for (int s = e5 + ph; s < e6; s += 16) s_5 += bpartial[(size_t)s * CoutP + co];
    }
    __shared__ float red[3][256];
    red[0][threadIdx.x] = s_eq; red[1][threadIdx.x] = s_4; red[2][threadIdx.x] = s_5;
    __syncthreads();
    for (int st = 8; st > 0; st >>= 1) {
        if (ph < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st * 16];
            red[1][threadIdx.x] += red[1][threadIdx.x + st * 16];
            red[2][threadIdx.x] += red[2][threadIdx.x + st * 16];
        }
        __syncthreads();
    }
    if (ph != 0) return;
    s_eq = red[0][o]; s_4 = red[1][o]; s_5 = red[2][o];
    if (is_w) {
        if (accumulate) {
            dw_eq[e] += s_eq;
            if (dw_np) { dw_pol[e] += s_4; dw_np[e] += s_5; } else dw_pol[e] += s_4 + s_5;
        } else {
            dw_eq[e] = s_eq;
            if (dw_np) { dw_pol[e] = s_4; dw_np[e] = s_5; } else dw_pol[e] = s_4 + s_5;
        }
    } else if (is_b) {
        const int co = e - nW;
        if (accumulate) {
            if (db_eq) db_eq[co] += s_eq;
            if (db_np) { if (db_pol) db_pol[co] += s_4; db_np[co] += s_5; } else if (db_pol) db_pol[co] += s_4 + s_5;
        } else {
            if (db_eq) db_eq[co] = s_eq;
            if (db_np) { if (db_pol) db_pol[co] = s_4; db_np[co] = s_5; } else if (db_pol) db_pol[co] = s_4 + s_5;
        }
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

template <int KS, int KC, int MT, int NT, int WM, int WN, int VW, bool MASK>
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
    const size_t buf = ((size_t)P.tile_rows_max * P.W2 * (KC + 4) + (size_t)NTB * (KC / 8) * KS * KS * 256) * sizeof(float);
    const size_t cap_items = (size_t)(3 * KC / VW) * NTHREADS;
    const bool pipelined = (size_t)P.tile_rows_max * P.W2 * (KC / VW) <= cap_items;
    size_t lds = pipelined ? 2 * buf : buf;
    if (lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv: LDS tile of %zu bytes exceeds 160 KiB (N=%d)", lds, P.No);
    auto kern = conv_mfma_kernel<KS, KC, MT, NT, WM, WN, VW, MASK>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    P.ntiles = P.B * 6 * P.nblk_face;
    P.magicN = div_magic(P.Nin);
    P.magicN2 = div_magic(P.Nin * P.Nin);
    P.dbg = nullptr;
#ifdef DLWPCS_TIMELINE
    { const char *e = getenv("DLWPCS_DBG_PTR"); P.dbg = e ? (long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
    const int nchunks = ceil_div(P.CG, KC / 8);
    const bool persistent = pipelined && nchunks >= 2 && (long)P.Nin * P.Nin * 6 < (1l << 24);
    int pidx = -1;
    if (persistent) {
        auto kpp = conv_mfma_pp_kernel<KS, KC, MT, NT, WM, WN, VW, MASK>;
        lds += (size_t)(WM * WN) * 16 * 36 * sizeof(float);      // wave-private epilogue transpose patches
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kpp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        // one resident set of workgroups: 256 CUs x as many workgroups as the LDS footprint admits (<= 2)
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu > 2) per_cu = 2;
        if (per_cu < 1) per_cu = 1;
        int gx = 256 * per_cu;
        if (gx > P.ntiles) gx = P.ntiles;
        dim3 grid((unsigned)gx, (unsigned)ceil_div(P.NTtot, NTB));
        if (prof_enabled()) {
            char tag[128];
            snprintf(tag, sizeof(tag), "conv_mfma_pp_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", KS, KC, MT, NT, WM, WN, VW, MASK ? "true" : "false");
            pidx = prof_begin(tag, W.flops, W.bytes, s);
        }
        hipLaunchKernelGGL(kpp, grid, dim3(NTHREADS), lds, s, P);
    } else {
        dim3 grid((unsigned)P.ntiles, (unsigned)ceil_div(P.NTtot, NTB));
        if (prof_enabled()) {
            char tag[128];
            snprintf(tag, sizeof(tag), "conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", KS, KC, MT, NT, WM, WN, VW, MASK ? "true" : "false");
            pidx = prof_begin(tag, W.flops, W.bytes, s);
        }
        hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, s, P);
    }
    if (pidx >= 0) prof_end(pidx, s);
    return check_launch("conv_mfma");
}

template <int KS, int VW, bool MASK>
static int launch_conv(const ConvKParams &P, const Work &W, hipStream_t s) {
    const int face_pix = P.No * P.No;
    if constexpr (KS == 1) return launch_conv_cfg<KS, 8, 3, 1, 4, 1, VW, MASK>(P, W, s);
    else if constexpr (VW != 4) return launch_conv_cfg<KS, 8, 3, 1, 4, 1, VW, MASK>(P, W, s);   // odd channel counts: one generic tiling
    else {
        if (P.NTtot == 1) return launch_conv_cfg<KS, 8, 3, 1, 4, 1, VW, MASK>(P, W, s);
        if (P.NTtot == 2) return launch_conv_cfg<KS, 8, 3, 1, 2, 2, VW, MASK>(P, W, s);
        if (face_pix <= 320) return launch_conv_cfg<KS, 8, 5, 1, 1, 4, VW, MASK>(P, W, s);
        return launch_conv_cfg<KS, 8, 3, 1, 1, 4, VW, MASK>(P, W, s);
    }
}

template <int KS, bool MASK>
static int dispatch_vw(int vw, const ConvKParams &P, const Work &W, hipStream_t s) {
    if (vw == 4) return launch_conv<KS, 4, MASK>(P, W, s);
    if (vw == 2) return launch_conv<KS, 2, MASK>(P, W, s);
    return launch_conv<KS, 1, MASK>(P, W, s);
}

static int dispatch_conv(int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s) {
    const bool mask = P.ymask != nullptr;
    if (KS == 3) return mask ? dispatch_vw<3, true>(vw, P, W, s) : dispatch_vw<3, false>(vw, P, W, s);
    return mask ? dispatch_vw<1, true>(vw, P, W, s) : dispatch_vw<1, false>(vw, P, W, s);
}

static inline int vec_width(int c0, int c1) {
    if (c0 % 4 == 0 && c1 % 4 == 0) return 4;
    if (c0 % 2 == 0 && c1 % 2 == 0) return 2;
    return 1;
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
    int n_eq, n_4, n_5, wg_pix, wg_nblk;
};

// persistent weight-gradient launch geometry: pixels per work item, items (bands) per face, workers per face class
static void wgrad_tiling(const dlwpcs_conv_desc *d, int &pix, int &nblk, int &n_eq, int &n_4, int &n_5) {
    const int No = out_size(d);
    const int face_pix = No * No;
    const int CAP = 192;     // pixels per work item: 2 LDS buffers of (X tile + dZ tile) in one CU's 160 KB
    pix = CAP;
    if (No <= CAP) pix = (CAP / No) * No;
    if (pix > face_pix) pix = face_pix;
    nblk = ceil_div(face_pix, pix);
    const int CinP = ceil_div(d->C0 + d->C1, 32) * 32, CoutP = ceil_div(d->Cout, 32) * 32;
    const int pairs = (CinP / 32) * (CoutP / 32);
    int wpp = 256 / pairs;               // one worker per CU (256 CUs) spread over the (ci, co) tile pairs
    if (wpp < 3) wpp = 3;
    const long items_per_face = (long)(d->B > 0 ? d->B : 1) * nblk;
    if (wpp > 6 * items_per_face) wpp = (int)(6 * items_per_face);
    if (wpp < 3) wpp = 3;
    n_4 = (wpp + 3) / 6; if (n_4 < 1) n_4 = 1;
    n_5 = n_4;
    n_eq = wpp - n_4 - n_5; if (n_eq < 1) n_eq = 1;
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
    int pix, nblk;
    wgrad_tiling(d, pix, nblk, L.n_eq, L.n_4, L.n_5);
    L.wg_pix = pix; L.wg_nblk = nblk;
    const int nworkers = L.n_eq + L.n_4 + L.n_5;
    const int CinP = ceil_div(Cin, 32) * 32, CoutP = ceil_div(d->Cout, 32) * 32;
    // dxv and the wgrad partials are never live at the same time but are kept disjoint for simplicity of reasoning
    L.partial = off;  off += align_up((size_t)nworkers * TAPS * CinP * CoutP * 4, 256);
    L.bpartial = off; off += align_up((size_t)nworkers * CoutP * 4, 256);
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
    return dispatch_conv(d->ksize, vec_width(d->C0, d->C1), P, conv_work(d), s);
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
    rc = dispatch_conv(d->ksize, vec_width(d->Cout, 0), P, conv_work(d), s);
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
    if (d->B == 0 && (d->flags & DLWPCS_CONV_ACCUMULATE_WGRAD)) return DLWPCS_OK;
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
    W.CinP = CinP; W.CoutP = CoutP;
    W.n_eq = L.n_eq; W.n_4 = L.n_4; W.n_5 = L.n_5;
    W.magicN = div_magic(P.Nin); W.magicN2 = div_magic(P.Nin * P.Nin);
    if (P.C1 == 0) P.src1 = P.src0;
    const bool mask = d->act != DLWPCS_ACT_NONE;
    const int pix_cap = (L.wg_pix + 1) & ~1;
    const int vw = vec_width(d->C0, d->C1);
    const size_t bufb = ((size_t)P.tile_rows_max * P.W2 * 32 + (size_t)pix_cap * 32 + pix_cap) * 4;
    const bool fits_regs = (size_t)P.tile_rows_max * P.W2 * (32 / vw) <= (size_t)(28 / vw) * 512 &&
                           (size_t)pix_cap * 8 <= (size_t)3 * 512;
    W.pipelined = (fits_regs && 2 * bufb <= 160 * 1024) ? 1 : 0;
    size_t lds = W.pipelined ? 2 * bufb : bufb;
    if (lds < 8 * 1024 * 4) lds = 8 * 1024 * 4;       // the 32 KB cross-wave reduction scratch aliases the buffers
    if (lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: LDS tile of %zu bytes exceeds 160 KiB", lds);
    dim3 grid((unsigned)(L.n_eq + L.n_4 + L.n_5), (unsigned)(CinP / 32), (unsigned)(CoutP / 32));
#define WG_LAUNCH(KSV, VWV, MASKV)                                                                                        \
    do {                                                                                                                  \
        auto kern = wgrad_mfma_kernel<KSV, VWV, MASKV>;                                                                   \
        if (lds > 64 * 1024) {                                                                                            \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));    \
        }                                                                                                                 \
        int pidx = -1;                                                                                                    \
        if (prof_enabled()) {                                                                                             \
            const Work wk = conv_work(d);                                                                                 \
            pidx = prof_begin("wgrad_mfma_kernel<" #KSV ", " #VWV ", " #MASKV ">", wk.flops, wk.bytes, s);                \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, W);                                                             \
        if (pidx >= 0) prof_end(pidx, s);                                                                                 \
    } while (0)
#define WG_VW(KSV, MASKV)                                                                                                 \
    do { if (vw == 4) WG_LAUNCH(KSV, 4, MASKV); else if (vw == 2) WG_LAUNCH(KSV, 2, MASKV); else WG_LAUNCH(KSV, 1, MASKV); } while (0)
    if (KS == 3) { if (mask) WG_VW(3, true); else WG_VW(3, false); }
    else { if (mask) WG_VW(1, true); else WG_VW(1, false); }
#undef WG_VW
#undef WG_LAUNCH
    rc = check_launch("wgrad_mfma");
    if (rc) return rc;
    const int nout = TAPS * Cin * d->Cout + (want_bias ? d->Cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(nout, 16)), dim3(256), 0, s, W.partial, W.bpartial,
                       (float *)dw_eq, (float *)dw_pol, (float *)dw_np, (float *)db_eq, (float *)db_pol, (float *)db_np,
                       KS, Cin, d->Cout, CinP, CoutP, L.n_eq, L.n_4, L.n_5, d->flip_north_pole,
                       (d->flags & DLWPCS_CONV_ACCUMULATE_WGRAD) ? 1 : 0);
    return check_launch("wgrad_reduce");
}
