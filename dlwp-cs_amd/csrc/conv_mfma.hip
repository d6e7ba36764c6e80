// Fused cubed-sphere convolution for gfx950 (MI355X): implicit-GEMM direct convolution on the matrix cores.
//
// One kernel template (conv_mfma_ws_kernel) serves
//   * forward            y  = act( conv_valid( halo_pad(V), W_face ) + b_face )        (DLWP/custom.py:921-1002 with
//                                                                                       :1082-1308 fused into the load)
//   * data gradient      dVpad = conv_full( dz, W_face^T ), dz = dy * act'(y)          (same kernel, mode ZERO border)
// and two more compute the weight gradient (wgrad_mfma_kernel: fp32 MFMA; wgrad_bf16_kernel: bf16 MFMA with LDS transpose
// reads) followed by a fixed-order reduction.  No im2col, nothing padded is ever materialised in HBM.
//
// Element type T of the activations: float (v_mfma_f32_32x32x2_f32, exact fp32, 157.3 TFLOP/s) or bf16
// (v_mfma_f32_32x32x16_bf16, ~2.5 PFLOP/s, fp32 accumulate; parameters stay fp32 and are rounded while packing).
// GEMM view per face:  M = pixels, N = C_out, K = k*k*C_in.  Per workgroup (one per CU, persistent, 4 consumer + 4 producer
// waves):
//   - a band of BM <= 32*MT*WM consecutive pixels (flat row-major index inside one face of one sample) times
//     BN = 32*NT*WN output channels; wave (wm, wn) owns MT x NT accumulator tiles of 32x32 (16 VGPRs each);
//   - the input tile (band rows + k-1 halo rows, full width + k-1) is staged through LDS in 64-B channel chunks (16 fp32 /
//     32 bf16 channels), channels_last, pixel stride 80 B so that the 16-lane groups of ds_read_b128 hit distinct slots;
//   - the cube-sphere halo is resolved while staging: border cells go through the (6,N+2,N+2) gather table (L2 resident,
//     60 KB at N=48); nearest-upsampling (x2) and the channel concat of the U-Net decoder are folded into the same address
//     computation, so none of pad / upsample / concat costs a pass;
//   - weights arrive pre-packed in MFMA fragment order (dlwpcs_pack_batch: every layer of a model in one launch per pass,
//     or per call into the workspace): a lane's ds_read_b128 returns the K values of 4 fp32 MFMAs / 1 bf16 MFMA; face 5's
//     row-reversed kernel is a packing variant;
//   - the MFMA runs as D[co][pixel] (weights = A operand), the epilogue (bias, activation, rounding) leaves through a
//     wave-private LDS patch as whole-line stores; in data-gradient mode interior cells go straight to the sources.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "mfma_common.h"
#include "conv_ws.h"
#include "conv_launch.h"

namespace dlwpcs {

// ------------------------------------------------------------------------------------------------------------------
// Weight packing (HWIO -> MFMA-B fragment order), 3 face variants: 0 equatorial, 1 south pole, 2 north pole.
//   transposed == 0 (forward):        B[tap=(dy,dx)][k=ci][n=co] = Wv[row(dy)][dx][ci][co]
//   transposed == 1 (data gradient):  B[tap=(ey,ex)][k=co][n=ci] = Wv[row(KS-1-ey)][KS-1-ex][ci][co]
// row(r) = KS-1-r on variant 2 when flip_north_pole (flip -> conv -> flip == row-reversed kernel), else r.
// Also packs the biases to [3][NTtot*32].
// ------------------------------------------------------------------------------------------------------------------
// T = MFMA operand type: float -> 4 values per 16-B lane entry (K group of 8), bf16_t -> 8 values (K group of 16; the
// fp32 master weights are rounded to bf16 here, once per call).
template <typename T>
__device__ __forceinline__ void pack_weights_range(const float *__restrict__ w_eq, const float *__restrict__ w_pol,
                                                   const float *__restrict__ w_np, T *__restrict__ out,
                                                   int KS, int Cin, int Cout, int K, int Ncol, int CG, int NTtot,
                                                   int flip, int transposed, size_t total, size_t first, size_t stride) {
    // one 16-B lane entry (J consecutive K values of one column) per thread and iteration: the index arithmetic happens once
    // per entry and the store is one 16-B write (element by element -- seven divisions and a 2-byte store per value -- the
    // per-step packing of a whole model was ~11 us of integer arithmetic)
    constexpr int J = 16 / (int)sizeof(T);
    const int TAPS = KS * KS;
    const size_t entries = total / J;
    for (size_t e = first; e < entries; e += stride) {
        unsigned r = (unsigned)e;             // totals are a few million at most: 32-bit divisions (64-bit ones cost ~10x)
        const int n = r % 32; r /= 32;
        const int hf = r % 2; r /= 2;
        const int tap = r % TAPS; r /= TAPS;
        const int cg = r % CG; r /= CG;
        const int nt = r % NTtot; r /= NTtot;
        const int v = (int)r;
        const int k0 = (cg * 2 + hf) * J, col = nt * 32 + n;
        const float *w = v == 0 ? w_eq : (v == 1 ? w_pol : (w_np ? w_np : w_pol));
        int ty = tap / KS, tx = tap % KS;
        if (transposed) { ty = KS - 1 - ty; tx = KS - 1 - tx; }
        if (v == 2 && flip) ty = KS - 1 - ty;
        const size_t tbase = (size_t)(ty * KS + tx) * Cin;
        float val[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int k = k0 + j;
            // forward: (ci, co) = (k, col); data gradient: (ci, co) = (col, k)
            const size_t idx = transposed ? (tbase + col) * Cout + k : (tbase + k) * Cout + col;
            val[j] = (k < K && col < Ncol) ? w[idx] : 0.f;
        }
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(out + e * J) = make_float4(val[0], val[1], val[2], val[3]);
        } else {
            *reinterpret_cast<uint4 *>(out + e * J) = make_uint4(f2bf2(val[0], val[1]), f2bf2(val[2], val[3]),
                                                                f2bf2(val[4], val[5]), f2bf2(val[6], val[7]));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) pack_weights_kernel(const float *__restrict__ w_eq, const float *__restrict__ w_pol,
                                                           const float *__restrict__ w_np, T *__restrict__ out,
                                                           int KS, int Cin, int Cout, int K, int Ncol, int CG, int NTtot,
                                                           int flip, int transposed, size_t total) {
    pack_weights_range<T>(w_eq, w_pol, w_np, out, KS, Cin, Cout, K, Ncol, CG, NTtot, flip, transposed, total,
                          (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// Every layer of a model in ONE launch (dlwpcs_pack_batch): block (x, item) packs item `blockIdx.y`'s forward weights,
// data-gradient weights and biases with a grid-stride loop over x.
__global__ void __launch_bounds__(256) pack_batch_kernel(const dlwpcs_pack_item *__restrict__ items) {
    const dlwpcs_pack_item it = items[blockIdx.y];
    const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const int KS = it.ksize, TAPS = KS * KS;
    const bool bf = it.dtype == DLWPCS_BF16;
    const int cgw = bf ? 16 : 8, per_lane = bf ? 8 : 4;
    const float *w_eq = (const float *)it.w_eq, *w_pol = (const float *)it.w_pol, *w_np = (const float *)it.w_np;
    if (it.wpk_fwd) {
        const int CG = (it.Cin + cgw - 1) / cgw, NTtot = (it.Cout + 31) / 32;
        const size_t total = (size_t)3 * NTtot * CG * TAPS * 64 * per_lane;
        if (bf) pack_weights_range<bf16_t>(w_eq, w_pol, w_np, (bf16_t *)it.wpk_fwd, KS, it.Cin, it.Cout, it.Cin, it.Cout, CG, NTtot,
                                           it.flip_north_pole, 0, total, first, stride);
        else pack_weights_range<float>(w_eq, w_pol, w_np, (float *)it.wpk_fwd, KS, it.Cin, it.Cout, it.Cin, it.Cout, CG, NTtot,
                                       it.flip_north_pole, 0, total, first, stride);
    }
    if (it.wpk_bwd) {
        const int CG = (it.Cout + cgw - 1) / cgw, NTtot = (it.Cin + 31) / 32;
        const size_t total = (size_t)3 * NTtot * CG * TAPS * 64 * per_lane;
        if (bf) pack_weights_range<bf16_t>(w_eq, w_pol, w_np, (bf16_t *)it.wpk_bwd, KS, it.Cin, it.Cout, it.Cout, it.Cin, CG, NTtot,
                                           it.flip_north_pole, 1, total, first, stride);
        else pack_weights_range<float>(w_eq, w_pol, w_np, (float *)it.wpk_bwd, KS, it.Cin, it.Cout, it.Cout, it.Cin, CG, NTtot,
                                       it.flip_north_pole, 1, total, first, stride);
    }
    if (it.bias_pk && it.b_eq) {
        const int CoutP = ((it.Cout + 31) / 32) * 32;
        const float *b_eq = (const float *)it.b_eq, *b_pol = (const float *)it.b_pol, *b_np = (const float *)it.b_np;
        for (size_t e = first; e < (size_t)3 * CoutP; e += stride) {
            const int v = (int)(e / CoutP), co = (int)(e % CoutP);
            const float *bsrc = v == 0 ? b_eq : (v == 1 ? b_pol : (b_np ? b_np : b_pol));
            ((float *)it.bias_pk)[e] = co < it.Cout ? bsrc[co] : 0.f;
        }
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
// taps summed across the consumer waves per LDS round of the weight-gradient epilogue (scratch = WG_TG x 4 waves x 4 KB)
constexpr int WG_TG = 3;
constexpr int WG_SCRATCH_FLOATS = WG_TG * 4096;

struct WgradKParams {
    ConvKParams c;          // description of the virtual input (src0/src1/table/mode/...), ymask unused here
    const void *dy, *y;     // (B,6,No,No,Cout), element type T; y nullable
    float *partial;         // [nworkers][TAPS][CinP][CoutP]
    float *bpartial;        // [nworkers][CoutP] or nullptr
    int CinP, CoutP;        // multiples of 32
    int n_eq, n_4, n_5;     // workers per face class (equatorial faces 0-3 / face 4 / face 5); grid.x = their sum
    uint32_t magicB, magicNb;   // exact-division magics of the batch size and of the bands per face (wgrad_bf16_kernel)
    void *dz_out;               // wgrad_bf16_kernel with MASK: ci-tile-0 workers also store dz (same shape as dy), or nullptr
};

// Weight-gradient kernel: persistent + wave-specialised.  GEMM view per tap:
//     D[ci][co] += sum_pixels Xpad[pixel + tap][ci] * dZ[pixel][co],      dZ = dy * act'(y).
// The accumulators D (k*k taps x 32 x 32, 9 x 16 VGPRs per lane) do not depend on WHICH pixels are summed, so a worker
// (one workgroup per CU: 4 consumer + 4 producer waves) owns one (ci tile, co tile) pair and streams through a STATIC,
// strided list of work items (sample, face, band of <= 192 pixels) of its face class:
//     producers:  fill(0); B; fill(1); B; ...     X tile (band + halo rows, 32 channels) and dZ tile (band, 32 channels)
//     consumers:           B; mma(0);  B; ...     the 4 consumer waves split the pixel pairs (MFMA K = 2 pixels)
// Producer code is straight-line per item (see conv_mfma_ws_kernel for why); the halo-table entries of item t+1 are read
// together with the data of item t.  Consumers double-buffer their operand registers (10 LDS reads of step s+1 are issued
// before the 9 MFMAs of step s).  At the end the 4 consumer waves are summed through LDS in a fixed order and ONE partial
// per worker is written; wgrad_reduce_kernel adds the workers' partials in a fixed order per face class (no atomics ->
// bitwise reproducible) and applies the weight-group map (class 0 -> equatorial kernel, 1 -> polar, 2 -> polar or north
// pole, tap rows reversed when flipping).  Bias gradients: the producers of ci tile 0 sum dZ as it passes through their
// registers.
//
// T = element type of X and dZ in HBM.  bf16 inputs are widened to fp32 by the producers on their way into LDS, the
// consumers are the same exact-fp32 MFMA loop for both (wgrad_bf16_kernel below is the bf16-MFMA version for the
// hot-path shapes; this kernel is its general fallback).
// fp32 <- raw register vectors
__device__ __forceinline__ void st_f32(float *d, float v) { *d = v; }
__device__ __forceinline__ void st_f32(float *d, float2 v) { *reinterpret_cast<float2 *>(d) = v; }
__device__ __forceinline__ void st_f32(float *d, float4 v) { *reinterpret_cast<float4 *>(d) = v; }
__device__ __forceinline__ void st_f32(float *d, uint16_t v) { *d = bf2f(v); }
__device__ __forceinline__ void st_f32(float *d, uint32_t v) { *reinterpret_cast<float2 *>(d) = make_float2(bf_lo(v), bf_hi(v)); }
__device__ __forceinline__ void st_f32(float *d, uint2 v) { *reinterpret_cast<float4 *>(d) = make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y)); }
__device__ __forceinline__ void st_f32(float *d, uint4 v) {
    reinterpret_cast<float4 *>(d)[0] = make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
    reinterpret_cast<float4 *>(d)[1] = make_float4(bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w));
}
__device__ __forceinline__ float4 to_f4(float4 v) { return v; }
__device__ __forceinline__ float4 to_f4(uint2 v) { return make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y)); }
__device__ __forceinline__ float4 pack4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
__device__ __forceinline__ uint2 pack4(bf16_t a, bf16_t b, bf16_t c, bf16_t d) {
    return make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)d << 16));
}

template <typename T, int KS, int VW, bool MASK>
__global__ void __launch_bounds__(512) wgrad_mfma_kernel(const WgradKParams W) {
    constexpr int TAPS = KS * KS;
    constexpr int XS = 32;                  // X tile row stride (floats) = the 32 input channels of this ci tile
    constexpr int QX = 32 / VW;             // vectors per X pixel
    constexpr int NCT = 256;                // consumer threads == producer threads
    constexpr int IT_X = 56 / VW;           // X vectors per producer thread per item: capacity 448 tile pixels
    constexpr int IT_DY = 6;                // dZ quads per producer thread per item: capacity 192 pixels (x 8 quads)
    typedef typename VecT<T, VW>::type V;
    typedef typename VecT<T, 4>::type DV;   // 4 consecutive output channels as loaded from HBM
    const ConvKParams &P = W.c;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int pix_cap = (P.pix_per_block + 1) & ~1;
    const int x_floats = P.tile_rows_max * P.W2 * XS;
    const int buf_floats = x_floats + pix_cap * 32;                // X tile, dZ tile
    // the 16 KB cross-wave reduction scratch aliases buffer 0 after the main loop

    const int worker = blockIdx.x;
    const int cit = blockIdx.y, cot = blockIdx.z;
    int j, nj, nfaces, fbase;
    if (worker < W.n_eq) { j = worker; nj = W.n_eq; nfaces = 4; fbase = 0; }
    else if (worker < W.n_eq + W.n_4) { j = worker - W.n_eq; nj = W.n_4; nfaces = 1; fbase = 4; }
    else { j = worker - W.n_eq - W.n_4; nj = W.n_5; nfaces = 1; fbase = 5; }
    const int nbands = P.nblk_face;
    // Items in (face, band)-major, SAMPLE-minor order, one contiguous range per worker (see wgrad_bf16_kernel): gather
    // offsets are rebuilt only at the few (face, band) changes, per item only a scalar sample base moves.
    const int total_items = P.B * nfaces * nbands;
    const int t_first = (int)(((long)total_items * j) / nj), t_last = (int)(((long)total_items * (j + 1)) / nj);
    const int n_my = t_last - t_first;
    const int face_pix = P.No * P.No;
    const int tid = threadIdx.x;

    struct Item { int b, f, combo, m0, npix, y0, nitems; };
    auto item_of = [&](int k) {
        Item it;
        const int t = t_first + max(min(k, n_my - 1), 0);
        it.combo = W.magicB ? __umulhi((uint32_t)t, W.magicB) : t;                    // t / B   (magic 0 <=> divisor 1)
        it.b = t - it.combo * P.B;
        const int fl = W.magicNb ? __umulhi((uint32_t)it.combo, W.magicNb) : it.combo;   // combo / nbands
        const int band = it.combo - fl * nbands;
        it.f = fbase + fl;
        it.m0 = band * P.pix_per_block;
        it.npix = min(P.pix_per_block, face_pix - it.m0);
        it.y0 = __umulhi((uint32_t)it.m0, P.magicNo);
        const int ylast = __umulhi((uint32_t)(it.m0 + it.npix - 1), P.magicNo);
        it.nitems = (ylast - it.y0 + KS) * P.W2 * QX;
        return it;
    };

    if (tid >= NCT) {
        // =========================================== producers ===========================================
        const int ptid = tid - NCT;
        if (P.tune & TUNE_WG_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(2);
        const int cx = cit * 32 + (ptid % QX) * VW;         // this thread's input channel(s): fixed for the whole kernel
        const bool cx_ok = cx < P.Cin;
        const bool from0 = cx < P.C0;
        const int g0 = P.up0 ? (P.Nin >> 1) : P.Nin;
        const int cs = from0 ? cx : cx - P.C0;
        const int cstride = from0 ? P.C0 : P.C1;
        const bool up = from0 && P.up0;
        const int M = P.Nin + KS - 1;
        const bool vec_dy = (P.Cout % 4 == 0);      // block-uniform
        const bool want_bias = W.bpartial != nullptr && cit == 0;
        float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);     // dZ column sums of this thread's (pixel subset, 4 channels)
        // per-slot gather offsets (elements relative to the sample's base; -1 = zero cell), valid for one (face, band)
        int xoff[IT_X];
        int cur_combo = -1;
        auto rebuild = [&](const Item &it) {
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
                const int e = min(ptid + i * NCT, it.nitems - 1);
                const int pix = e / QX;
                const int ty = __umulhi((uint32_t)pix, P.magicW2);
                const int tx = pix - ty * P.W2;
                const int iy = it.y0 + ty;
                int ii;
                if (P.mode == MODE_HALO) ii = P.table[(it.f * M + iy) * M + tx];
                else ii = (it.f * P.Nin + iy) * P.Nin + tx;
                const int r = __umulhi((uint32_t)ii, P.magicN);      // row face*Nin + y of the Nin grid -> row r/2 of Nin/2
                const int pix_up = (r >> 1) * g0 + ((ii - r * P.Nin) >> 1);
                const int spix = up ? pix_up : ii;
                xoff[i] = (cx_ok && ptid + i * NCT < it.nitems) ? spix * cstride + cs : -1;
            }
            cur_combo = it.combo;
        };
        const size_t sample_elems = from0 ? (size_t)6 * g0 * g0 * P.C0 : (size_t)6 * P.Nin * P.Nin * P.C1;
        const T *src_base = reinterpret_cast<const T *>(from0 ? P.src0 : P.src1);
        Item cur = item_of(0);
        for (int k = 0; k < n_my; ++k) {
            float *buf = smem + (k & 1) * buf_floats;
            const Item nxt = item_of(k + 1);
            if (cur.combo != cur_combo) rebuild(cur);              // uniform, a few times per worker
            const T *sb = src_base + (size_t)cur.b * sample_elems;
            // ---- X tile: every load in flight at once.  (VW = 1, odd channel counts: 56 scalar loads per item -- with their
            // offsets and the dZ registers that spilled ~170 dwords; there the tile goes through the registers in XG groups and
            // dZ in DG halves, each written to LDS before the next is fetched.  The vector instantiations are one group.)
            constexpr int XG = VW == 1 ? 4 : (VW == 2 && MASK ? 2 : 1), XN = IT_X / XG;
            constexpr int DG = VW == 1 ? 2 : 1, DN = IT_DY / DG;
            static_assert(IT_X % XG == 0 && IT_DY % DG == 0, "register groups must divide the slot counts");
            V xv[IT_X];
            bool xok[IT_X];
            auto x_store = [&](int i0) {
#pragma unroll
                for (int u = 0; u < XN; ++u) {
                    const int i = i0 + u;
                    const int e = ptid + i * NCT;
                    if (e < cur.nitems) st_f32(buf + (e / QX) * XS + (ptid % QX) * VW, vsel(xok[i], xv[i]));
                }
            };
#pragma unroll
            for (int g = 0; g < XG; ++g) {
#pragma unroll
                for (int u = 0; u < XN; ++u) {
                    const int i = g * XN + u;
                    const int o = xoff[i];
                    xv[i] = *reinterpret_cast<const V *>(sb + (uint32_t)max(o, 0));
                    xok[i] = o >= 0;
                }
                if constexpr (XG > 1) { x_store(g * XN); asm volatile("" ::: "memory"); }
            }
            // ---- dZ tile [pix][32 output channels of tile cot] = dy * act'(y), zero beyond npix / Cout
            const size_t rowbase = (((size_t)cur.b * 6 + cur.f) * face_pix + cur.m0) * P.Cout;
            const T *dyb = reinterpret_cast<const T *>(W.dy) + rowbase;
            const T *yb = MASK ? reinterpret_cast<const T *>(W.y) + rowbase : nullptr;
            float4 dv[IT_DY], yv[MASK ? IT_DY : 1];
            DV dvr[IT_DY], yvr[MASK ? IT_DY : 1];
            bool dok[IT_DY];
            auto dz_store = [&](int i0) {
#pragma unroll
                for (int u = 0; u < DN; ++u) {
                    const int i = i0 + u;
                    const int e = ptid + i * NCT;
                    if (e * 4 < pix_cap * 32) *reinterpret_cast<float4 *>(buf + x_floats + e * 4) = dv[i];
                    if (want_bias) { bsum.x += dv[i].x; bsum.y += dv[i].y; bsum.z += dv[i].z; bsum.w += dv[i].w; }
                }
            };
#pragma unroll
            for (int h = 0; h < DG; ++h) {
                if (vec_dy) {
#pragma unroll
                    for (int u = 0; u < DN; ++u) {
                        const int i = h * DN + u;
                        const int e = ptid + i * NCT;
                        const int kk = e >> 3, co = cot * 32 + (e & 7) * 4;
                        const bool ok = kk < cur.npix && co < P.Cout;
                        const size_t o = ok ? (size_t)kk * P.Cout + co : 0;
                        dvr[i] = *reinterpret_cast<const DV *>(dyb + o);
                        if (MASK) yvr[i] = *reinterpret_cast<const DV *>(yb + o);
                        dok[i] = ok;
                    }
                } else {
                    // C_out % 4 != 0 (e.g. the 14-channel head): scalar gather, 4 consecutive tile floats per slot
#pragma unroll
                    for (int u = 0; u < DN; ++u) {
                        const int i = h * DN + u;
                        T gs[4], ys[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int e4 = (ptid + i * NCT) * 4 + q;
                            const int kk = e4 >> 5, co = cot * 32 + (e4 & 31);
                            const bool ok = kk < cur.npix && co < P.Cout;
                            const size_t o = ok ? (size_t)kk * P.Cout + co : 0;
                            gs[q] = ok ? dyb[o] : (T)0;
                            ys[q] = MASK ? yb[o] : (T)0;
                        }
                        dvr[i] = pack4(gs[0], gs[1], gs[2], gs[3]);
                        if (MASK) yvr[i] = pack4(ys[0], ys[1], ys[2], ys[3]);
                        dok[i] = true;
                    }
                }
                // dZ = dy * act'(y), applied only after EVERY load of the item (of the half) has been issued
#pragma unroll
                for (int u = 0; u < DN; ++u) {
                    const int i = h * DN + u;
                    dv[i] = to_f4(dvr[i]);
                    if (MASK) {
                        yv[i] = to_f4(yvr[i]); vmask(dv[i], yv[i], P.alpha, P.vmax);
                        if constexpr (sizeof(T) == 2) {
                            // bf16 mode: dz is a bf16 tensor everywhere else (hand-over, pre-masked gradients, the bf16 kernels):
                            // round the product here too, so that every path multiplies the same numbers
                            dv[i].x = bf2f(f2bf(dv[i].x)); dv[i].y = bf2f(f2bf(dv[i].y));
                            dv[i].z = bf2f(f2bf(dv[i].z)); dv[i].w = bf2f(f2bf(dv[i].w));
                        }
                    }
                    dv[i] = vsel(dok[i], dv[i]);
                }
                // DLWPCS_CONV_REUSE_DZ (fp32 tensors, C_out % 4 == 0): the workers of ci tile 0 see every dZ element exactly
                // once -> they hand dz to the data-gradient kernel that follows
                if constexpr (MASK && sizeof(T) == 4) {
                    if (W.dz_out != nullptr && cit == 0 && vec_dy) {
                        float *dzb = reinterpret_cast<float *>(W.dz_out) + rowbase;
#pragma unroll
                        for (int u = 0; u < DN; ++u) {
                            const int i = h * DN + u;
                            const int e = ptid + i * NCT;
                            const int kk = e >> 3, co = cot * 32 + (e & 7) * 4;
                            if (kk < cur.npix && co < P.Cout) *reinterpret_cast<float4 *>(dzb + (size_t)kk * P.Cout + co) = dv[i];
                        }
                    }
                }
                if constexpr (DG > 1) { dz_store(h * DN); asm volatile("" ::: "memory"); }
            }
            // ---- registers -> LDS
            if constexpr (XG == 1) x_store(0);
            if constexpr (DG == 1) dz_store(0);
            __syncthreads();            // B_k: item k is in LDS
            cur = nxt;
        }
        // ---- bias partial: thread (q = ptid & 7, 32 pixel phases) holds sums of channels 4q..4q+3 -> fixed-order tree
        __syncthreads();                // consumers are done with the buffers (matches the consumers' final barrier)
        if (want_bias) {
            float *red = smem + WG_SCRATCH_FLOATS;   // behind the consumers' reduction scratch
            red[ptid * 4 + 0] = bsum.x; red[ptid * 4 + 1] = bsum.y; red[ptid * 4 + 2] = bsum.z; red[ptid * 4 + 3] = bsum.w;
        }
        __syncthreads();
        if (want_bias && ptid < 32) {
            // channel c = ptid: quad q = c >> 2, component c & 3; sum over the 32 threads with (t & 7) == q
            float sum = 0.f;
            const float *red = smem + WG_SCRATCH_FLOATS;
#pragma unroll
            for (int ph = 0; ph < 32; ++ph) sum += red[((ph * 8 + (ptid >> 2)) * 4) + (ptid & 3)];
            W.bpartial[(size_t)worker * W.CoutP + cot * 32 + ptid] = sum;
        }
        // the consumers' reduction loop below executes 2 barriers per round: keep the barrier counts of both halves equal
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += (TAPS >= WG_TG ? WG_TG : 1)) { __syncthreads(); __syncthreads(); }
        return;
    }

    // ============================================= consumers =============================================
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nsteps = pix_cap / 2;             // pixel pairs per item
    const int S = (((nsteps + 3) / 4) + 1) & ~1; // steps per consumer wave, rounded up to even (extra steps add zero)
    for (int k = 0; k < n_my; ++k) {
        __syncthreads();                        // B_k
        const float *lds_x = smem + (k & 1) * buf_floats, *lds_dy = lds_x + x_floats;
        const Item it = item_of(k);
        // operands of step si: pixel kk = 2*s + half, s = wave + 4*si.  Branch-free: out-of-range steps read a clamped
        // (valid) address and multiply by a zero dZ, so the loop body is straight-line and the two register sets can be
        // software-pipelined (10 LDS reads of step si+1 before the 9 MFMAs of step si).
        auto frag = [&](int si, float (&a)[TAPS], float &bq) {
            const int s = wave + 4 * si;
            const int k2 = 2 * s + half;
            const int kk = min(k2, pix_cap - 1);
            const int gm = it.m0 + min(kk, it.npix - 1);
            const int oy = __umulhi((uint32_t)gm, P.magicNo);
            const int pb = ((oy - it.y0) * P.W2 + (gm - oy * P.No)) * XS + l31;
            const float bv = lds_dy[kk * 32 + l31];
            bq = (s < nsteps) ? bv : 0.f;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) a[tap] = lds_x[pb + ((tap / KS) * P.W2 + (tap % KS)) * XS];
        };
        float fa[2][TAPS], fb[2];
        frag(0, fa[0], fb[0]);
        for (int si = 0; si < S; si += 2) {
            frag(si + 1, fa[1], fb[1]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][tap], fb[0], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);
            frag(si + 2, fa[0], fb[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][tap], fb[1], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);
        }
    }
    __syncthreads();                            // all consumers finished reading the last buffer
    __syncthreads();                            // (producers stage their bias sums between these two)

    // cross-wave reduction through LDS (fixed order w = 0..3), WG_TG taps per round (2 barriers per round, not per tap)
    float *red = smem;
    float *pout = W.partial + (size_t)worker * TAPS * W.CinP * W.CoutP;
    const int ctid = tid;
    constexpr int TG = TAPS >= WG_TG ? WG_TG : 1;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TG) {
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                red[(tt * 4 + wave) * 1024 + ci * 32 + l31] = acc[t0 + tt][r];
            }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = ctid + i * NCT;
                const float *rt = red + tt * 4096;
                const float sum = (rt[e] + rt[1024 + e]) + (rt[2048 + e] + rt[3072 + e]);
                const int ci = e >> 5, co = e & 31;
                pout[((size_t)(t0 + tt) * W.CinP + cit * 32 + ci) * W.CoutP + cot * 32 + co] = sum;
            }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores (hot-path shapes of the bf16 mode: every channel count a multiple of 8).
//
// Same worker structure as wgrad_mfma_kernel (persistent, 4 consumer + 4 producer waves, one partial per worker, static
// strided item lists per face class), but the contraction runs on v_mfma_f32_32x32x16_bf16 with K = 16 PIXELS per
// instruction.  Both operands are stored channels-contiguous ([pixel][32 channels] bf16, 64 B per pixel, as they arrive
// from HBM) while the instruction wants, per lane, 8 consecutive K values of ONE channel: the transposition is done by
// the LDS itself with ds_read_b64_tr_b16 -- the 16 lanes of a group each pass the address of 4 contiguous channels of one
// of 4 pixels and receive one channel of those 4 pixels (probed on the MI355X with tools/probe_tr.hip: lane n gets
// column n, element j = row j).  The K -> pixel mapping is free as long as X and dZ agree, so pixels are taken in flat
// band order and every lane computes its own two pixel addresses; the 64-B pixel stride makes each 32-lane access group
// (4 pixels x 32 channels = 256 B) hit all 64 banks exactly once.
// dZ = dy * act'(y) is rounded to bf16 by the producers (it is a bf16 tensor in this mode); bias gradients are summed
// from those rounded values in fp32.  Items are <= 384 pixels (16-pixel K slabs; rows beyond npix are zero-filled).
// ------------------------------------------------------------------------------------------------------------------
// XV = channels per X load (8 = 16-B vectors; 2 = 4-B vectors for channel counts that are only even, e.g. the 14-channel
// network input), QX = X vectors staged per tile pixel (QX * XV channels of the 32-wide ci tile; the rest of the LDS row
// is never written and only feeds accumulator rows >= C_in, which the reduction ignores).
// CT = ci tiles (of 32 channels) per worker.  CT = 2 (C_in a multiple of 64): the worker stages 64 input channels per pixel
// (two [pixel][32 ch] planes) against ONE dZ tile and its consumer waves split as (ci tile, slab parity) instead of four
// slab phases -- each dZ element is loaded, masked and written to LDS once per 64 input channels instead of once per 32,
// which is what the producers, the bottleneck of this kernel, spend most of their time on.  Items are then <= 192 pixels.
// DV = channels per dZ load: 8 (16-B vectors), or 2 for output-channel counts that are only even and <= 16 (the 14-channel
// 1x1 head): 8 four-byte vectors per pixel, columns 16..31 of the dZ tile are never written and only feed accumulator
// columns >= C_out.
template <int KS, bool MASK, int XV, int QX, int CT, int DV>
__global__ void __launch_bounds__(512) wgrad_bf16_kernel(const WgradKParams W) {
    constexpr int QD = DV == 8 ? 4 : 8;     // dZ vectors staged per pixel
    typedef typename VecT<bf16_t, DV>::type DVec;
    constexpr int TAPS = KS * KS;
    constexpr int PB = 64;                  // LDS bytes per pixel and plane: 32 channels bf16 (X planes and dZ tile alike)
    constexpr int NCT = 256;                // consumer threads == producer threads
    constexpr int QXT = QX * CT;            // X vectors staged per tile pixel
    // X vectors per producer thread per item: capacity IT_X * 256 / QXT tile pixels (512 / 320 / 512|256)
    constexpr int IT_X = XV == 8 ? (CT == 2 ? 10 : 8) : 16;
    typedef typename VecT<bf16_t, XV>::type XVec;
    constexpr int IT_DY = (CT == 2 ? 3 : 6) * (QD / 4);   // dZ vectors per producer thread per item: capacity 192 / 384 pixels
    constexpr int NPH = 4 / CT;             // consumer waves sharing the slabs of one ci tile
    static_assert(CT == 1 || (XV == 8 && QX == 4), "two ci tiles per worker need full 16-B X vectors");
    // cross-wave reduction of the epilogue: tap t is summed by the wave of phase t % NPH; every other wave of the ci tile
    // parks its accumulators of that tap in one of its RSLOTS 4-KB LDS slots
    constexpr int RSLOTS = TAPS - TAPS / NPH;
    constexpr int RED_FLOATS = 4 * RSLOTS * 1024;
    const ConvKParams &P = W.c;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pix_cap = (P.pix_per_block + 15) & ~15;
    const int plane_bytes = P.tile_rows_max * P.W2 * PB;
    const int x_bytes = CT * plane_bytes;
    const int buf_bytes = x_bytes + pix_cap * PB;
    // the 16 KB cross-wave reduction scratch (+ 8 KB bias staging) aliases the buffers after the main loop

    const int worker = blockIdx.x;
    const int cit = blockIdx.y, cot = blockIdx.z;
    int j, nj, nfaces, fbase;
    if (worker < W.n_eq) { j = worker; nj = W.n_eq; nfaces = 4; fbase = 0; }
    else if (worker < W.n_eq + W.n_4) { j = worker - W.n_eq; nj = W.n_4; nfaces = 1; fbase = 4; }
    else { j = worker - W.n_eq - W.n_4; nj = W.n_5; nfaces = 1; fbase = 5; }
    const int nbands = P.nblk_face;
    // Items in (face, band)-major, SAMPLE-minor order; worker j of its class owns one contiguous range.  Consecutive items
    // of a worker are then the same tile position in consecutive samples: every gather offset, validity flag and LDS
    // address stays the same and only a scalar sample base moves (offsets are rebuilt at the <= 2 combo changes per worker).
    const int total_items = P.B * nfaces * nbands;
    const int t_first = (int)(((long)total_items * j) / nj), t_last = (int)(((long)total_items * (j + 1)) / nj);
    const int n_my = t_last - t_first;
    const int face_pix = P.No * P.No;
    const int tid = threadIdx.x;

    struct Item { int b, f, combo, m0, npix, y0, nitems; };
    auto item_of = [&](int k) {
        Item it;
        const int t = t_first + max(min(k, n_my - 1), 0);
        // (div_magic(1) does not fit 32 bits: a divisor of 1 is passed as magic 0)
        it.combo = W.magicB ? __umulhi((uint32_t)t, W.magicB) : t;                    // t / B
        it.b = t - it.combo * P.B;
        const int fl = W.magicNb ? __umulhi((uint32_t)it.combo, W.magicNb) : it.combo;   // combo / nbands
        const int band = it.combo - fl * nbands;
        it.f = fbase + fl;
        it.m0 = band * P.pix_per_block;
        it.npix = min(P.pix_per_block, face_pix - it.m0);
        it.y0 = __umulhi((uint32_t)it.m0, P.magicNo);
        const int ylast = __umulhi((uint32_t)(it.m0 + it.npix - 1), P.magicNo);
        it.nitems = (ylast - it.y0 + KS) * P.W2 * QXT;      // QXT channel vectors per tile pixel
        return it;
    };

    if (tid >= NCT) {
        // =========================================== producers ===========================================
        const int ptid = tid - NCT;
        if (P.tune & TUNE_WG_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(2);
        const int qx = ptid % QD;                           // this thread's dZ channel group: fixed for the whole kernel
        const int cx = cit * 32 * CT + (ptid % QXT) * XV;   // this thread's X channels: fixed as well (256 % QXT == 0)
        const bool cx_ok = cx < P.Cin;
        const bool from0 = cx < P.C0;
        const int g0 = P.up0 ? (P.Nin >> 1) : P.Nin;
        const int cs = from0 ? cx : cx - P.C0;
        const int cstride = from0 ? P.C0 : P.C1;
        const bool up = from0 && P.up0;
        const int M = P.Nin + KS - 1;
        const int co = cot * 32 + qx * DV;
        const bool co_ok = co < P.Cout;
        const bool want_bias = W.bpartial != nullptr && cit == 0;
        float bsum[DV];
#pragma unroll
        for (int u = 0; u < DV; ++u) bsum[u] = 0.f;
        // Per-slot gather offsets (elements, relative to the sample's base pointer; -1 = zero cell), valid for one
        // (face, band) combination: halo-table lookup + upsample decode happen here, once per combo, not once per item.
        int xoff[IT_X];
        int cur_combo = -1;
        auto rebuild = [&](const Item &it) {
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
                const int e = min(ptid + i * NCT, it.nitems - 1);
                const int pix = e / QXT;
                const int ty = __umulhi((uint32_t)pix, P.magicW2);
                const int tx = pix - ty * P.W2;
                const int iy = it.y0 + ty;
                int ii;                                              // flat cell on the Nin grid INSIDE the face plane set
                if (P.mode == MODE_HALO) ii = P.table[(it.f * M + iy) * M + tx];
                else ii = (it.f * P.Nin + iy) * P.Nin + tx;
                const int r = __umulhi((uint32_t)ii, P.magicN);      // row face*Nin + y of the Nin grid -> row r/2 of Nin/2
                const int pix_up = (r >> 1) * g0 + ((ii - r * P.Nin) >> 1);
                const int spix = up ? pix_up : ii;
                xoff[i] = (cx_ok && ptid + i * NCT < it.nitems) ? spix * cstride + cs : -1;
            }
            cur_combo = it.combo;
        };
        // dZ slot offsets never change: slot i is pixel kk = (ptid + i*NCT) / QD of the band, channels co..co+DV-1
        int doff[IT_DY];
#pragma unroll
        for (int i = 0; i < IT_DY; ++i) doff[i] = ((ptid + i * NCT) / QD) * P.Cout + co;
        const size_t sample_elems = from0 ? (size_t)6 * g0 * g0 * P.C0 : (size_t)6 * P.Nin * P.Nin * P.C1;
        const bf16_t *src_base = reinterpret_cast<const bf16_t *>(from0 ? P.src0 : P.src1);
        // Two register sets, prefetch distance 2: the loads of item k+1 are issued BEFORE the loads of item k are waited for,
        // so a full item's worth of loads is always in flight and the HBM/L2 latency never shows (with one set the
        // producers spent more than half of every item waiting; the consumers, 16x faster than in the fp32 kernel, idled).
        struct Stage {
            XVec xv[IT_X];
            DVec dv[IT_DY], yv[MASK ? IT_DY : 1];
            bool xok[IT_X], dok[IT_DY];
        };
        auto issue = [&](const Item &it, Stage &st) {
            if (it.combo != cur_combo) rebuild(it);                 // uniform, <= 2-3 times per worker
            const bf16_t *sb = src_base + (size_t)it.b * sample_elems;
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
                const int o = xoff[i];
                st.xv[i] = *reinterpret_cast<const XVec *>(sb + (uint32_t)max(o, 0));
                st.xok[i] = o >= 0;
            }
            // dZ tile [pix][32 output channels of tile cot] = dy * act'(y), zero beyond npix / Cout.  (Issuing these BEFORE the
            // X gather -- they need no table -- measured 0.7 % slower on the training step.)
            const size_t rowbase = (((size_t)it.b * 6 + it.f) * face_pix + it.m0) * P.Cout;
            const bf16_t *dyb = reinterpret_cast<const bf16_t *>(W.dy) + rowbase;
            const bf16_t *yb = MASK ? reinterpret_cast<const bf16_t *>(W.y) + rowbase : nullptr;
            const int dlim = it.npix * P.Cout;                      // slot valid <=> its pixel < npix
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                const bool ok = co_ok && doff[i] < dlim;
                const uint32_t o = ok ? (uint32_t)doff[i] : 0u;
                st.dv[i] = *reinterpret_cast<const DVec *>(dyb + o);
                if (MASK) st.yv[i] = *reinterpret_cast<const DVec *>(yb + o);
                st.dok[i] = ok;
            }
        };
        auto commit = [&](const Item &it, int k, Stage &st) {
            char *buf = smem + (k & 1) * buf_bytes;
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                if (MASK) vmask(st.dv[i], st.yv[i], P.alpha, P.vmax);
                st.dv[i] = vsel(st.dok[i], st.dv[i]);
            }
            // DLWPCS_CONV_REUSE_DZ: the workers of ci tile 0 see every dZ element exactly once -> they hand dz to the
            // data-gradient kernel that follows (which then needs neither y nor the act' arithmetic)
            if (MASK && W.dz_out != nullptr && cit == 0) {
                bf16_t *dzb = reinterpret_cast<bf16_t *>(W.dz_out) + (((size_t)it.b * 6 + it.f) * face_pix + it.m0) * P.Cout;
                const rsrc_t dzr = make_rsrc(dzb, (uint32_t)(it.npix * P.Cout * 2));
#pragma unroll
                for (int i = 0; i < IT_DY; ++i) bstv(st.dv[i], dzr, st.dok[i] ? (uint32_t)doff[i] * 2 : ST_SKIP);
            }
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
                const int e = ptid + i * NCT;
                if (e < it.nitems)
                    *reinterpret_cast<XVec *>(buf + ((ptid % QXT) / QX) * plane_bytes + (size_t)(e / QXT) * PB +
                                              ((ptid % QXT) % QX) * (XV * 2)) = vsel(st.xok[i], st.xv[i]);
            }
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                const int e = ptid + i * NCT;
                if (e < pix_cap * QD)
                    *reinterpret_cast<DVec *>(buf + x_bytes + (size_t)(e / QD) * PB + (e % QD) * (DV * 2)) = st.dv[i];
                if (want_bias) {
                    if constexpr (DV == 8) {
                        bsum[0] += bf_lo(st.dv[i].x); bsum[1] += bf_hi(st.dv[i].x); bsum[2] += bf_lo(st.dv[i].y); bsum[3] += bf_hi(st.dv[i].y);
                        bsum[4] += bf_lo(st.dv[i].z); bsum[5] += bf_hi(st.dv[i].z); bsum[6] += bf_lo(st.dv[i].w); bsum[7] += bf_hi(st.dv[i].w);
                    } else {
                        bsum[0] += bf_lo(st.dv[i]); bsum[1] += bf_hi(st.dv[i]);
                    }
                }
            }
            // B_k: item k is in LDS.  A RAW barrier behind an explicit LDS wait: __syncthreads() makes hipcc drain vmcnt(0)
            // first, i.e. wait for the loads of item k+1 that were issued a moment ago -- the whole HBM latency (~4 k cycles,
            // measured with the s_memtime marks) would be paid at every barrier, with the consumers idling behind it.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (n_my > 0) {
            Stage A, B;
            Item i0 = item_of(0), i1 = item_of(1);          // item_of clamps: prefetches past the end re-read the last item
            issue(i0, A);
            for (int k = 0; k < n_my; k += 2) {
                // invariant: A holds item k (in flight)
                const Item i2 = item_of(k + 2);
                issue(i1, B);
                commit(i0, k, A);
                if (k + 1 >= n_my) break;
                const Item i3 = item_of(k + 3);
                issue(i2, A);
                commit(i1, k + 1, B);
                i0 = i2; i1 = i3;
            }
        }
        // ---- bias partial: thread (q = ptid % QD, 256/QD pixel phases) holds sums of channels DV*q.. -> fixed-order sum
        __syncthreads();                // consumers are done with the buffers (matches the consumers' final barrier)
        float *red = reinterpret_cast<float *>(smem) + RED_FLOATS;   // behind the consumers' reduction slots
        if (want_bias) {
#pragma unroll
            for (int u = 0; u < DV; ++u) red[ptid * DV + u] = bsum[u];
        }
        __syncthreads();                // (the consumers parked their accumulators between the two barriers)
        if (want_bias && ptid < 32) {
            float sum = 0.f;
            if (ptid < QD * DV) {
#pragma unroll
                for (int ph = 0; ph < NCT / QD; ++ph) sum += red[(ph * QD + ptid / DV) * DV + (ptid % DV)];
            }
            W.bpartial[(size_t)worker * W.CoutP + cot * 32 + ptid] = sum;
        }
        return;
    }

    // ============================================= consumers =============================================
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int li = lane & 15;
    const int choff = ((((lane >> 4) & 1) * 16) + (li & 3) * 4) * 2;   // byte offset of this lane's 4 contiguous channels
    const int prow = li >> 2;                                            // which of the 4 pixels of a transpose block
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int tapoff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) tapoff[t] = ((t / KS) * P.W2 + (t % KS)) * PB;

    const int nslab = pix_cap / 16;             // K slabs (16 pixels) per item
    const int S = (((nslab + NPH - 1) / NPH) + 1) & ~1;   // slabs per consumer wave, rounded up to even (extras add zero)
    const int ct = wave % CT, ph = wave / CT;   // this wave's ci tile and slab phase
    for (int k = 0; k < n_my; ++k) {
        __syncthreads();                        // B_k
        const char *lds_x0 = smem + (k & 1) * buf_bytes, *lds_dy = lds_x0 + x_bytes, *lds_x = lds_x0 + ct * plane_bytes;
        const Item it = item_of(k);
        // operands of slab si of this wave: K index kk = 8*half + 4*jj + prow (jj = 0, 1) <-> flat pixel 16*s + kk.
        // Branch-free: out-of-range slabs / pixels read a clamped (valid) X address and a zero dZ.
        auto frag = [&](int si, uint4 (&a)[TAPS], uint4 &bq) {
            const int s = ph + NPH * si;
            const int sc = min(s, nslab - 1);
            const bool live = s < nslab;
            int xaddr[2], daddr[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int m = 16 * sc + 8 * half + 4 * jj + prow;
                const int gm = it.m0 + min(m, it.npix - 1);
                const int oy = __umulhi((uint32_t)gm, P.magicNo);
                xaddr[jj] = ((oy - it.y0) * P.W2 + (gm - oy * P.No)) * PB + choff;
                daddr[jj] = m * PB + choff;
            }
            const uint2 b0 = lds_tr16(lds_dy + daddr[0]), b1 = lds_tr16(lds_dy + daddr[1]);
            bq = vsel(live, make_uint4(b0.x, b0.y, b1.x, b1.y));
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const uint2 a0 = lds_tr16(lds_x + xaddr[0] + tapoff[tap]), a1 = lds_tr16(lds_x + xaddr[1] + tapoff[tap]);
                a[tap] = make_uint4(a0.x, a0.y, a1.x, a1.y);
            }
        };
        uint4 fa[2][TAPS], fb[2];
        frag(0, fa[0], fb[0]);
        for (int si = 0; si < S; si += 2) {
            frag(si + 1, fa[1], fb[1]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) frag_mma<bf16_t>(acc[tap], fa[0][tap], fb[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * TAPS + 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);
            frag(si + 2, fa[0], fb[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) frag_mma<bf16_t>(acc[tap], fa[1][tap], fb[1]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * TAPS + 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);
        }
    }
    __syncthreads();                            // all consumers finished reading the last buffer
    // Cross-wave reduction, ONE LDS round for all taps: the NPH waves of a ci tile hold K-split partial sums of the same
    // (taps, 32, 32) block.  Tap t belongs to the wave of phase t % NPH; the others park their 16 accumulator registers of
    // that tap in LDS ([slot][r / 4][lane][4]: one ds_write_b128 per register quad), the owner adds them in phase order
    // (fixed -> bitwise reproducible) and writes the rows straight from its registers: in the C/D layout register r of lanes
    // 0-31 / 32-63 is row ci = (r & 3) + 8 (r >> 2) + 4 half of 32 consecutive output channels = two whole 128-B lines per
    // store instruction.  (The previous version went through LDS in three rounds of 4-B accesses: 10 k cycles per worker.)
    float4 *red4 = reinterpret_cast<float4 *>(smem);
    const rsrc_t pr = make_rsrc(W.partial + (size_t)worker * TAPS * W.CinP * W.CoutP, (uint32_t)(TAPS * W.CinP * W.CoutP * 4));
    auto slot_of = [](int t, int p) { int c = 0; for (int u = 0; u < t; ++u) c += (u % NPH != p) ? 1 : 0; return c; };
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (ph != t % NPH) {
            float4 *dst = red4 + (size_t)((ct * NPH + ph) * RSLOTS + slot_of(t, ph)) * 256 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[q * 64] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        }
    }
    __syncthreads();                            // (the producers staged their bias sums meanwhile)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (ph == t % NPH) {
            const int o = t % NPH;
            f32x16 v[NPH];
#pragma unroll
            for (int p = 0; p < NPH; ++p) {
                if (p == o) { v[p] = acc[t]; continue; }
                const float4 *src = red4 + (size_t)((ct * NPH + p) * RSLOTS + slot_of(t, p)) * 256 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 f = src[q * 64];
                    v[p][4 * q] = f.x; v[p][4 * q + 1] = f.y; v[p][4 * q + 2] = f.z; v[p][4 * q + 3] = f.w;
                }
            }
            const uint32_t row0 = (uint32_t)(((t * W.CinP + (cit * CT + ct) * 32 + 4 * half) * W.CoutP + cot * 32 + l31) * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sum;
                if (NPH == 4) sum = (v[0][r] + v[1][r]) + (v[2][r] + v[3][r]);
                else if (NPH == 2) sum = v[0][r] + v[1][r];
                else sum = v[0][r];
                bst32(__float_as_uint(sum), pr, row0 + (uint32_t)(((r & 3) + 8 * (r >> 2)) * W.CoutP * 4));
            }
        }
    }
}

// Sum the per-slot partials in a fixed order and route them to the weight groups.  Workgroup = 16 output groups x 16 slot
// phases: thread (o, ph) adds slots ph, ph+16, ... (fixed order), the 16 phases are then combined through LDS in a fixed
// tree -> bitwise reproducible, and enough workgroups to fill the chip.  VEC = outputs per thread: 4 consecutive output
// channels as one 16-B load per slot when C_out % 4 == 0 (the 16 threads of a phase then read 256 contiguous bytes per
// slot instead of 64), else 1.  accumulate != 0: add to the destination instead of overwriting it (shared layers / direct
// accumulation into the flat gradient buffer).
// A workgroup reduces RED_OG output groups (VEC floats each) x 256 / RED_OG slot phases.  Measured on the 152 MB of partials of
// a training step (MI355X): 16 groups x 16 phases 43-46 us, 32 x 8 51 us, 64 x 4 52-53 us, 128 x 2 51 us -- longer contiguous
// reads per slot do not pay, the slot-phase parallelism does.
#ifndef DLWPCS_RED_OG
#define DLWPCS_RED_OG 16
#endif
constexpr int RED_OG = DLWPCS_RED_OG;
template <int VEC>
__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ partial, const float *__restrict__ bpartial,
                                                  float *__restrict__ dw_eq, float *__restrict__ dw_pol,
                                                  float *__restrict__ dw_np, float *__restrict__ db_eq,
                                                  float *__restrict__ db_pol, float *__restrict__ db_np,
                                                  int KS, int Cin, int Cout, int CinP, int CoutP,
                                                  int n_eq, int n_4, int n_5, int flip, int accumulate, int block) {
    typedef float VT __attribute__((ext_vector_type(VEC)));
    constexpr int OG = RED_OG, NPH = 256 / RED_OG;     // output groups per workgroup x slot phases
    const int TAPS = KS * KS;
    const int nW = TAPS * Cin * Cout;
    const int o = threadIdx.x % OG, ph = threadIdx.x / OG;
    const int e = (block * OG + o) * VEC;              // first of this thread's VEC consecutive outputs
    const size_t slot_stride = (size_t)TAPS * CinP * CoutP;
    VT s_eq = 0.f, s_4 = 0.f, s_5 = 0.f;
    const bool is_w = e < nW, is_b = (!is_w) && bpartial && e < nW + Cout;
    const int e4 = n_eq, e5 = n_eq + n_4, e6 = n_eq + n_4 + n_5;
    if (is_w) {
        const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cout * Cin);
        const size_t off = ((size_t)tap * CinP + ci) * CoutP + co;
        const int ty = tap / KS, tx = tap % KS;
        // face 5 ran with the row-reversed kernel: its partial for tap row r belongs to kernel row KS-1-r
        const size_t o5 = flip ? ((size_t)((KS - 1 - ty) * KS + tx) * CinP + ci) * CoutP + co : off;
#pragma unroll 8
        for (int s = ph; s < e4; s += NPH) s_eq += *reinterpret_cast<const VT *>(partial + (size_t)s * slot_stride + off);
#pragma unroll 4
        for (int s = e4 + ph; s < e5; s += NPH) s_4 += *reinterpret_cast<const VT *>(partial + (size_t)s * slot_stride + off);
#pragma unroll 4
        for (int s = e5 + ph; s < e6; s += NPH) s_5 += *reinterpret_cast<const VT *>(partial + (size_t)s * slot_stride + o5);
    } else if (is_b) {
        const int co = e - nW;
#pragma unroll 8
        for (int s = ph; s < e4; s += NPH) s_eq += *reinterpret_cast<const VT *>(bpartial + (size_t)s * CoutP + co);
#pragma unroll 4
        for (int s = e4 + ph; s < e5; s += NPH) s_4 += *reinterpret_cast<const VT *>(bpartial + (size_t)s * CoutP + co);
#pragma unroll 4
        for (int s = e5 + ph; s < e6; s += NPH) s_5 += *reinterpret_cast<const VT *>(bpartial + (size_t)s * CoutP + co);
    }
    __shared__ VT red[3][256];
    red[0][threadIdx.x] = s_eq; red[1][threadIdx.x] = s_4; red[2][threadIdx.x] = s_5;
    __syncthreads();
    for (int st = NPH / 2; st > 0; st >>= 1) {
        if (ph < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st * OG];
            red[1][threadIdx.x] += red[1][threadIdx.x + st * OG];
            red[2][threadIdx.x] += red[2][threadIdx.x + st * OG];
        }
        __syncthreads();
    }
    if (ph != 0) return;
    s_eq = red[0][o]; s_4 = red[1][o]; s_5 = red[2][o];
    auto put = [&](float *dst, int idx, VT v) {
        VT *p = reinterpret_cast<VT *>(dst + idx);
        *p = accumulate ? *p + v : v;
    };
    if (is_w) {
        put(dw_eq, e, s_eq);
        if (dw_np) { put(dw_pol, e, s_4); put(dw_np, e, s_5); } else put(dw_pol, e, s_4 + s_5);
    } else if (is_b) {
        const int co = e - nW;
        if (db_eq) put(db_eq, co, s_eq);
        if (db_np) { if (db_pol) put(db_pol, co, s_4); put(db_np, co, s_5); } else if (db_pol) put(db_pol, co, s_4 + s_5);
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bpartial,
                                                           float *__restrict__ dw_eq, float *__restrict__ dw_pol,
                                                           float *__restrict__ dw_np, float *__restrict__ db_eq,
                                                           float *__restrict__ db_pol, float *__restrict__ db_np,
                                                           int KS, int Cin, int Cout, int CinP, int CoutP,
                                                           int n_eq, int n_4, int n_5, int flip, int accumulate) {
    wgrad_reduce_body<VEC>(partial, bpartial, dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np, KS, Cin, Cout, CinP, CoutP, n_eq,
                           n_4, n_5, flip, accumulate, (int)blockIdx.x);
}

// All layers of a backward pass in one launch (DLWPCS_CONV_DEFER_REDUCE): a workgroup finds its item by walking the
// (short) item list — workgroup-uniform scalar loads — and runs the same body, so the summation order and therefore
// the result bits are those of the per-layer launches.
constexpr int REDUCE_BATCH_MAX = 32;
struct ReduceStarts { int first[REDUCE_BATCH_MAX + 1]; };      // first[i] = first workgroup of item i, by value (kernarg)
__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(const dlwpcs_reduce_item *__restrict__ items, int n_items,
                                                                 ReduceStarts st) {
    const int b = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < REDUCE_BATCH_MAX; ++k) i += (k < n_items && b >= st.first[k]) ? 1 : 0;     // registers only
    const dlwpcs_reduce_item it = items[i];
    const int blk = b - st.first[i];
    if (blk >= it.nblocks) return;
    if (it.vec == 4)
        wgrad_reduce_body<4>(it.partial, it.bpartial, it.dw_eq, it.dw_pol, it.dw_np, it.db_eq, it.db_pol, it.db_np, it.ksize,
                             it.Cin, it.Cout, it.CinP, it.CoutP, it.n_eq, it.n_4, it.n_5, it.flip_north_pole, it.accumulate, blk);
    else
        wgrad_reduce_body<1>(it.partial, it.bpartial, it.dw_eq, it.dw_pol, it.dw_np, it.db_eq, it.db_pol, it.db_np, it.ksize,
                             it.Cin, it.Cout, it.CinP, it.CoutP, it.n_eq, it.n_4, it.n_5, it.flip_north_pole, it.accumulate, blk);
}

// VEC = 4 needs every vector to stay inside one row of C_out values and 16-B aligned destinations
static bool wgrad_reduce_vec(const void *dw_eq, const void *dw_pol, const void *dw_np, const void *db_eq, const void *db_pol,
                             const void *db_np, int Cout) {
    auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    return Cout % 4 == 0 && al16(dw_eq) && al16(dw_pol) && al16(dw_np) && al16(db_eq) && al16(db_pol) && al16(db_np);
}
static int wgrad_reduce_blocks(int KS, int Cin, int Cout, bool want_bias, bool vec) {
    const int nout = KS * KS * Cin * Cout + (want_bias ? Cout : 0);
    return vec ? ceil_div(nout / 4, RED_OG) : ceil_div(nout, RED_OG);
}

static void launch_wgrad_reduce(hipStream_t s, const float *partial, const float *bpartial, void *dw_eq, void *dw_pol,
                                void *dw_np, void *db_eq, void *db_pol, void *db_np, int KS, int Cin, int Cout, int CinP,
                                int CoutP, int n_eq, int n_4, int n_5, int flip, int accumulate, bool want_bias) {
    const bool vec = wgrad_reduce_vec(dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np, Cout);
    const int nblocks = wgrad_reduce_blocks(KS, Cin, Cout, want_bias, vec);
    if (vec)
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(nblocks), dim3(256), 0, s, partial, bpartial,
                           (float *)dw_eq, (float *)dw_pol, (float *)dw_np, (float *)db_eq, (float *)db_pol, (float *)db_np,
                           KS, Cin, Cout, CinP, CoutP, n_eq, n_4, n_5, flip, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(nblocks), dim3(256), 0, s, partial, bpartial,
                           (float *)dw_eq, (float *)dw_pol, (float *)dw_np, (float *)db_eq, (float *)db_pol, (float *)db_np,
                           KS, Cin, Cout, CinP, CoutP, n_eq, n_4, n_5, flip, accumulate);
}

// ------------------------------------------------------------------------------------------------------------------
// Host side: configuration choice and launches
// ------------------------------------------------------------------------------------------------------------------
int launch_src_grad(const void *dxv, void *dsrc, const int32_t *inv, int B, int N, int CT, int choff, int CS, int up,
                    int halo, int dtype, hipStream_t s, const void *msrc, float m_alpha, float m_vmax, int *masked);
int launch_src_pair(const void *dxv, void *dsrc0, void *dsrc1, const int32_t *inv, int B, int N, int C0, int C1, int up0,
                    int dtype, hipStream_t s, const void *m0, const void *m1, float m_alpha, float m_vmax);
int launch_ring_fix(const void *dxv, void *dsrc, const int32_t *inv, int B, int N, int CT, int choff, int CS, int dtype,
                    hipStream_t s, const void *msrc, float m_alpha, float m_vmax);
int launch_mask_inplace(void *dx, const void *m, size_t n, float alpha, float vmax, int dtype, hipStream_t s);


// ------------------------------------------------------------------------------------------------------------------
// Pointwise head (k = 1, no halo, 32 -> 8..32 channels, bf16): the output layer of the U-Net (32 -> 14) moves 20 MB and
// does 5 % of the FLOPs; through the tiled conv kernel it costs as much as a 3x3 layer.  Here one wave handles 16 pixels
// per v_mfma_f32_16x16x32_bf16 (K = all 32 input channels): lane (n = lane & 15, q = lane >> 4) loads 16 B = channels
// 8q..8q+7 of pixel n, so ONE load instruction of the wave covers 16 complete 64-B pixel rows; the weights of the three
// face classes stay in registers (read from the dlwpcs_pack_batch fragment buffers), the bias rides in as the C operand.
// D layout: lane holds output channels 4q..4q+3 of pixel n -> one 8-B store per lane, 16 x 2*Cout contiguous bytes/wave.
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PwParams {
    const bf16_t *in;        // forward: x (pix, 32); data gradient: dy (pix, Cout)
    const bf16_t *wpk;       // forward: wpk_fwd; data gradient: wpk_bwd
    const float *bias;       // bias_pk [3][32] or null (forward only)
    bf16_t *out;             // forward: y (pix, Cout); data gradient: dx (pix, 32)
    long ngroups;            // groups of 16 consecutive pixels
    int groups_per_face;     // N * N / 16
    int Cout;
    float alpha, vmax;
    // data gradient, pre-masked gradients (round 6): dx *= act'(mask; m_alpha, m_vmax), mask (pix, 32) = the layer's input, the output of
    // an activated layer -- inside the launch instead of a masking pass over dx behind it (14.7 us per head of the production model)
    const bf16_t *mask;
    float m_alpha, m_vmax;
};

constexpr int PW_U = 4;      // 16-pixel groups in flight per wave

// Each wave owns a contiguous range of groups (32-bit indices throughout; the face class of a group is tracked
// incrementally in scalar registers: no divisions in the loop, the weight fragment is re-selected only when it changes).
struct PwRange { int g, end, rem, face; };
__device__ __forceinline__ PwRange pw_range(const PwParams &P) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = (int)gridDim.x * 4;
    const int ng = (int)P.ngroups;
    const int per = (ng + nwaves - 1) / nwaves;
    PwRange r;
    r.g = wave * per < ng ? wave * per : ng;
    r.end = r.g + per < ng ? r.g + per : ng;
    const unsigned f = (unsigned)r.g / (unsigned)P.groups_per_face;
    r.rem = r.g - (int)f * P.groups_per_face;
    r.face = (int)(f % 6u);
    return r;
}
__device__ __forceinline__ void pw_next(PwRange &r, int gpf) {
    ++r.g;
    if (++r.rem == gpf) { r.rem = 0; r.face = r.face == 5 ? 0 : r.face + 1; }
}

template <bool ACT, int MT>      // MT = 16-channel output tiles: 1 (C_out <= 16) or 2 (C_out <= 32)
__global__ void __launch_bounds__(256) pw_fwd_kernel(PwParams P) {
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    // packed forward weights: [variant][k-group of 8 (4 of them)][32 columns][8]; column = output channel.  A wave's range
    // rarely spans two face classes: the fragments and the bias quads are (re)loaded only when the class changes (every
    // wave pre-loading all three classes — thousands of waves on the same few cache lines — tripled the kernel time).
    const bf16_t *wlane = P.wpk + (q * 32 + n) * 8;
    const float *blane = P.bias ? P.bias + q * 4 : nullptr;
    const int Cout = P.Cout, gpf = P.groups_per_face;
    PwRange r = pw_range(P);
    const bf16_t *src = P.in + (unsigned)(n * 32 + q * 8);
    // a group's 16 output rows are 32 * Cout contiguous bytes, but a lane's 4 channels sit at a 4-B aligned offset inside
    // them: direct 8-B stores are split and merge badly.  The wave transposes through a private LDS patch and lanes
    // 0 .. 2*Cout-1 write aligned 16-B pieces (no fence: one wave's LDS accesses execute in order).
    __shared__ __attribute__((aligned(16))) uint32_t pw_stage[4][128 * MT];
    uint32_t *stage = pw_stage[threadIdx.x >> 6];
    const bool wr16 = lane < 2 * Cout;
    bf16_t *dst = P.out + (unsigned)(lane * 8);
    int vcur = -1;
    uint4 a[MT];
    f32x4 b[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { a[t] = make_uint4(0, 0, 0, 0); b[t] = (f32x4)0.f; }
    while (r.g < r.end) {
        uint4 xv[PW_U];
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            const int g = r.g + u < r.end ? r.g + u : r.end - 1;
            xv[u] = *reinterpret_cast<const uint4 *>(src + (unsigned)g * 512u);
        }
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            if (r.g < r.end) {
                const int v = r.face < 4 ? 0 : r.face - 3;
                if (v != vcur) {
                    vcur = v;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        a[t] = *reinterpret_cast<const uint4 *>(wlane + v * 1024 + t * 128);
                        if (blane) b[t] = *reinterpret_cast<const f32x4 *>(blane + v * 32 + t * 16);
                    }
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, xv[u]),
                                                                      b[t], 0, 0, 0);
                    if (ACT) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const float x = d[k] < 0.f ? d[k] * P.alpha : d[k]; d[k] = x > P.vmax ? P.vmax : x; }   // (NaN stays NaN)
                    }
                    const int co0 = t * 16 + q * 4, sidx = (n * Cout + co0) >> 1;
                    if (co0 + 4 <= Cout) { stage[sidx] = f2bf2(d[0], d[1]); stage[sidx + 1] = f2bf2(d[2], d[3]); }
                    else if (co0 + 2 <= Cout) stage[sidx] = f2bf2(d[0], d[1]);
                }
                __builtin_amdgcn_wave_barrier();
                if (wr16) *reinterpret_cast<uint4 *>(dst + (unsigned)r.g * (unsigned)(16 * Cout)) = reinterpret_cast<const uint4 *>(stage)[lane];
                __builtin_amdgcn_wave_barrier();
                pw_next(r, gpf);
            }
        }
    }
}

// dx (pix, 32) = dy (pix, Cout) . W^T: K = Cout zero-padded to 32 (lane groups past Cout carry zeros), two 16-row M tiles
template <bool MASKED>
__global__ void __launch_bounds__(256) pw_dgrad_kernel(PwParams P) {
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    // packed data-gradient weights: [variant][k-group of 8 (2 per 16 output channels, zero-padded past Cout)][32 columns][8];
    // column = ci.  Fragments of a face class are loaded when the class changes.
    const int Cout = P.Cout, gpf = P.groups_per_face;
    const int kgroups = ((Cout + 15) / 16) * 2;              // k-groups present in the packed buffer
    const bool kvalid = q < kgroups && q * 8 < Cout;
    const bf16_t *wlane = P.wpk + (q * 32 + n) * 8;
    PwRange r = pw_range(P);
    const bf16_t *src = P.in + (unsigned)(n * Cout + q * 8);
    bf16_t *dst = P.out + (unsigned)(n * 32 + q * 4);
    // lane group q holds channels 8q .. min(8q + 8, Cout) - 1: up to four dwords
    const int nch = Cout - q * 8;
    const bool full = nch >= 8, l1 = !full && nch > 0, l2 = !full && nch > 2, l3 = !full && nch > 4, l4 = !full && nch > 6;
    int vcur = -1;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    while (r.g < r.end) {
        uint4 bv[PW_U];
        uint2 xm[MASKED ? PW_U : 1][2];
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            const int g = r.g + u < r.end ? r.g + u : r.end - 1;
            const bf16_t *sp = src + (unsigned)g * (unsigned)(16 * Cout);
            bv[u] = make_uint4(0, 0, 0, 0);
            if (full) bv[u] = *reinterpret_cast<const uint4_a4 *>(sp);
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(sp);
            if (l1) bv[u].x = s32[0];
            if (l2) bv[u].y = s32[1];
            if (l3) bv[u].z = s32[2];
            if (l4) bv[u].w = s32[3];
            if constexpr (MASKED) {
                // the mask operand at the channels this lane's dx quads cover (4q .. and 16 + 4q .. of pixel n)
                const bf16_t *xp = P.mask + (unsigned)g * 512u + (unsigned)(n * 32 + q * 4);
                xm[u][0] = *reinterpret_cast<const uint2 *>(xp);
                xm[u][1] = *reinterpret_cast<const uint2 *>(xp + 16);
            }
        }
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            if (r.g < r.end) {
                const int v = r.face < 4 ? 0 : r.face - 3;
                if (v != vcur) {
                    vcur = v;
                    if (kvalid) {
                        a0 = *reinterpret_cast<const uint4 *>(wlane + v * kgroups * 256);
                        a1 = *reinterpret_cast<const uint4 *>(wlane + v * kgroups * 256 + 128);
                    }
                }
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bv[u]),
                                                                         (f32x4)0.f, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bv[u]),
                                                                         (f32x4)0.f, 0, 0, 0);
                bf16_t *o_ptr = dst + (unsigned)r.g * 512u;
                uint2 o;
                o.x = f2bf2(d0[0], d0[1]); o.y = f2bf2(d0[2], d0[3]);
                // (the mask multiplies the ROUNDED gradient, like the masking pass it replaces: same bits)
                if constexpr (MASKED) { o.x = bmask2(o.x, xm[u][0].x, P.m_alpha, P.m_vmax); o.y = bmask2(o.y, xm[u][0].y, P.m_alpha, P.m_vmax); }
                *reinterpret_cast<uint2 *>(o_ptr) = o;
                o.x = f2bf2(d1[0], d1[1]); o.y = f2bf2(d1[2], d1[3]);
                if constexpr (MASKED) { o.x = bmask2(o.x, xm[u][1].x, P.m_alpha, P.m_vmax); o.y = bmask2(o.y, xm[u][1].y, P.m_alpha, P.m_vmax); }
                *reinterpret_cast<uint2 *>(o_ptr + 16) = o;
                pw_next(r, gpf);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused training tail of the network (bf16): pointwise head + 'mse' loss (+ 'mae') + its gradient + the head's data
// gradient in ONE pass over the 32-channel input -- y = W x + b is rounded to bf16 exactly as pw_fwd_kernel stores it, the
// loss terms and dy = gscale (y - t) are formed in fp32 against the fp32 target (mse_stage1_kernel's arithmetic), dy is
// rounded to bf16 and goes through the wave's LDS patch twice: as aligned 16-B pieces to HBM (the head's weight gradient
// reads it) and, re-read as this lane's K fragment, into the two MFMAs of pw_dgrad_kernel.  The prediction itself is never
// written.  Replaces pw_fwd + mse_stage1 + pw_dgrad (3 launches, ~30 us) of the unfused step; same dy / dx bits.
// Per-workgroup partial sums -> head_stage2_kernel (fixed order).
// ------------------------------------------------------------------------------------------------------------------
struct HeadParams {
    PwParams f;              // in = x, wpk = wpk_fwd, bias, out = dy (pix, Cout)
    const bf16_t *wpk_bwd;
    const float *target;     // (pix, Cout) fp32
    bf16_t *dx;              // (pix, 32)
    float *partial;          // [gridDim.x][2]
    float gscale;            // weight * 2 / n
    int mask_dx;             // pre-masked gradients: dx *= act'(x; m_alpha, m_vmax), x being the output of an activated layer
    float m_alpha, m_vmax;
};

template <int MT>
__global__ void __launch_bounds__(256) pw_head_train_kernel(HeadParams H) {
    const PwParams &P = H.f;
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const bf16_t *wlane = P.wpk + (q * 32 + n) * 8;
    const float *blane = P.bias ? P.bias + q * 4 : nullptr;
    const int Cout = P.Cout, gpf = P.groups_per_face;
    const int kgroups = ((Cout + 15) / 16) * 2;
    const bool kvalid = q < kgroups && q * 8 < Cout;
    const bf16_t *wblane = H.wpk_bwd + (q * 32 + n) * 8;
    PwRange r = pw_range(P);
    const bf16_t *src = P.in + (unsigned)(n * 32 + q * 8);
    __shared__ __attribute__((aligned(16))) uint32_t pw_stage[4][128 * MT + 4];
    uint32_t *stage = pw_stage[threadIdx.x >> 6];
    if (lane < 4) stage[128 * MT + lane] = 0u;              // the K fragment of the last pixel may read 4 B past the rows
    const bool wr16 = lane < 2 * Cout;
    bf16_t *dst = P.out + (unsigned)(lane * 8);
    bf16_t *dxl = H.dx + (unsigned)(n * 32 + q * 4);
    // this lane's K fragment of dy: channels 8q .. 8q+7 of pixel n = bytes n*2*Cout + 16q .. +15 of the patch (4-B aligned)
    const int nch = Cout - q * 8;
    const uint32_t *frag = stage + ((n * Cout + q * 8) >> 1);
    const bool k1 = nch > 0, k2 = nch > 2, k3 = nch > 4, k4 = nch > 6;
    int vcur = -1;
    uint4 a[MT], a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    f32x4 b[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { a[t] = make_uint4(0, 0, 0, 0); b[t] = (f32x4)0.f; }
    float sq = 0.f, ab = 0.f;
    while (r.g < r.end) {
        uint4 xv[PW_U];
        uint2 xm[PW_U][2];
        float2 tv[PW_U][MT][2];
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            const int g = r.g + u < r.end ? r.g + u : r.end - 1;
            xv[u] = *reinterpret_cast<const uint4 *>(src + (unsigned)g * 512u);
            if (H.mask_dx) {
                // x at the channels this lane's dx quads cover (4q.. and 16 + 4q.. of pixel n): same lines as xv, L1 hits
                const bf16_t *xp = P.in + (unsigned)g * 512u + (unsigned)(n * 32 + q * 4);
                xm[u][0] = *reinterpret_cast<const uint2 *>(xp);
                xm[u][1] = *reinterpret_cast<const uint2 *>(xp + 16);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int co0 = t * 16 + q * 4;
                const float *tp = H.target + ((size_t)g * 16 + n) * Cout + co0;
                tv[u][t][0] = co0 + 2 <= Cout ? *reinterpret_cast<const float2 *>(tp) : make_float2(0.f, 0.f);
                tv[u][t][1] = co0 + 4 <= Cout ? *reinterpret_cast<const float2 *>(tp + 2) : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            if (r.g < r.end) {
                const int v = r.face < 4 ? 0 : r.face - 3;
                if (v != vcur) {
                    vcur = v;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        a[t] = *reinterpret_cast<const uint4 *>(wlane + v * 1024 + t * 128);
                        if (blane) b[t] = *reinterpret_cast<const f32x4 *>(blane + v * 32 + t * 16);
                    }
                    if (kvalid) {
                        a0 = *reinterpret_cast<const uint4 *>(wblane + v * kgroups * 256);
                        a1 = *reinterpret_cast<const uint4 *>(wblane + v * kgroups * 256 + 128);
                    }
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, xv[u]),
                                                                            b[t], 0, 0, 0);
                    const int co0 = t * 16 + q * 4, sidx = (n * Cout + co0) >> 1;
                    // the prediction as the unfused path stores it (bf16), loss terms and gradient in fp32
                    const uint32_t y01 = f2bf2(d[0], d[1]), y23 = f2bf2(d[2], d[3]);
                    const float e0 = bf_lo(y01) - tv[u][t][0].x, e1 = bf_hi(y01) - tv[u][t][0].y;
                    const float e2 = bf_lo(y23) - tv[u][t][1].x, e3 = bf_hi(y23) - tv[u][t][1].y;
                    if (co0 + 2 <= Cout) {
                        sq += e0 * e0; ab += fabsf(e0); sq += e1 * e1; ab += fabsf(e1);
                        stage[sidx] = f2bf2(H.gscale * e0, H.gscale * e1);
                    }
                    if (co0 + 4 <= Cout) {
                        sq += e2 * e2; ab += fabsf(e2); sq += e3 * e3; ab += fabsf(e3);
                        stage[sidx + 1] = f2bf2(H.gscale * e2, H.gscale * e3);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (wr16) *reinterpret_cast<uint4 *>(dst + (unsigned)r.g * (unsigned)(16 * Cout)) = reinterpret_cast<const uint4 *>(stage)[lane];
                uint4 bv = make_uint4(0, 0, 0, 0);
                if (k1) bv.x = frag[0];
                if (k2) bv.y = frag[1];
                if (k3) bv.z = frag[2];
                if (k4) bv.w = frag[3];
                __builtin_amdgcn_wave_barrier();
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bv),
                                                                         (f32x4)0.f, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bv),
                                                                         (f32x4)0.f, 0, 0, 0);
                bf16_t *o_ptr = dxl + (unsigned)r.g * 512u;
                uint2 o;
                o.x = f2bf2(d0[0], d0[1]); o.y = f2bf2(d0[2], d0[3]);
                // (the mask multiplies the ROUNDED gradient, like the unfused sequence data gradient -> elementwise mask: same bits)
                if (H.mask_dx) { o.x = bmask2(o.x, xm[u][0].x, H.m_alpha, H.m_vmax); o.y = bmask2(o.y, xm[u][0].y, H.m_alpha, H.m_vmax); }
                *reinterpret_cast<uint2 *>(o_ptr) = o;
                o.x = f2bf2(d1[0], d1[1]); o.y = f2bf2(d1[2], d1[3]);
                if (H.mask_dx) { o.x = bmask2(o.x, xm[u][1].x, H.m_alpha, H.m_vmax); o.y = bmask2(o.y, xm[u][1].y, H.m_alpha, H.m_vmax); }
                *reinterpret_cast<uint2 *>(o_ptr + 16) = o;
                pw_next(r, gpf);
            }
        }
    }
    // workgroup partial sums in a fixed order: lanes -> waves -> workgroup
    __shared__ float s_sq[256], s_ab[256];
    s_sq[threadIdx.x] = sq; s_ab[threadIdx.x] = ab;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { s_sq[threadIdx.x] += s_sq[threadIdx.x + st]; s_ab[threadIdx.x] += s_ab[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { H.partial[2 * blockIdx.x] = s_sq[0]; H.partial[2 * blockIdx.x + 1] = s_ab[0]; }
}

__global__ void __launch_bounds__(256) head_stage2_kernel(const float *__restrict__ partial, float *__restrict__ loss_out,
                                                          int nblocks, float inv_n, float weight, int overwrite) {
    loss_stage2_body(partial, loss_out, nblocks, inv_n, weight, overwrite);
}

static bool pw_applies(const dlwpcs_conv_desc *d) {
    return d->dtype == DLWPCS_BF16 && d->ksize == 1 && !d->halo && !d->up0 && d->C1 == 0 && d->C0 == 32 && d->Cout % 2 == 0 &&
           d->Cout >= 8 && d->Cout <= 32 && ((long)d->N * d->N) % 16 == 0 && d->B > 0 &&
           (long)d->B * 6 * d->N * d->N * 32 < (1l << 31);        // 32-bit element offsets
}
static unsigned pw_grid(long ngroups) {
    // latency-bound streaming: as many waves in flight as the chip holds (8 per SIMD), >= PW_U groups per wave
    long blocks = (ngroups + 4 * PW_U - 1) / (4 * PW_U);
    return (unsigned)(blocks > 2048 ? 2048 : blocks);
}

static inline int cin_logical(const dlwpcs_conv_desc *d);

// the gather-form plan of the data-gradient launch in flight on this thread (conv_launch.h)
thread_local ConvEdgeArgs g_edge_args{};

struct WgradBf16Name { static const char *str() { return "wgrad_bf16_kernel"; } };
struct WgradMfmaName { static const char *str() { return "wgrad_mfma_kernel"; } };
struct PwHeadName { static const char *str() { return "pw_head_train_kernel"; } };
struct PwFwdName { static const char *str() { return "pw_fwd_kernel"; } };
static const int g_plain_tags[] = {prof_register_tag("pw_dgrad_kernel<false>"), prof_register_tag("pw_dgrad_kernel<true>"),
                                   prof_register_tag("wgrad_reduce_batch_kernel")};

static int dispatch_conv(int dtype, int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s) {
    if (KS == 3 && P.mode == MODE_HALO && P.edge) {
        // data gradient in gather form (bf16, 16-B channel vectors; conv_bwd_data_impl checked)
        if (dtype == DLWPCS_BF16 && vw == 8) return dispatch_conv_edge(P, W, s);
        return fail(DLWPCS_E_UNSUPPORTED, "conv: gather-form data gradient serves bf16 with 16-B channel vectors");
    }
    return dtype == DLWPCS_BF16 ? dispatch_conv_bf16(KS, vw, P, W, s) : dispatch_conv_f32(KS, vw, P, W, s);
}

// widest channel vector (elements) both sources allow: 16 B at most
static inline int vec_width(int c0, int c1, int dtype) {
    if (dtype == DLWPCS_BF16 && c0 % 8 == 0 && c1 % 8 == 0) return 8;
    if (c0 % 4 == 0 && c1 % 4 == 0) return 4;
    if (c0 % 2 == 0 && c1 % 2 == 0) return 2;
    return 1;
}
static inline int cgw_of(int dtype) { return dtype == DLWPCS_BF16 ? 16 : 8; }

// algorithmic work of one convolution pass: flops = 2*B*6*N^2*k^2*Cin*Cout; bytes = unpadded input and
// output touched once + the weights.
static Work conv_work(const dlwpcs_conv_desc *d) {
    const double No = d->halo ? d->N : d->N - d->ksize + 1;
    const double Cin = cin_logical(d), taps = (double)d->ksize * d->ksize;
    const double n0 = d->up0 ? d->N / 2 : d->N;
    Work w;
    w.flops = 2.0 * d->B * 6 * No * No * taps * Cin * d->Cout;
    w.bytes = (double)dtype_size(d->dtype) * d->B * 6.0 * (n0 * n0 * d->C0 + (double)d->N * d->N * d->C1 + No * No * d->Cout) +
              4.0 * 2.0 * taps * Cin * d->Cout;
    return w;
}

struct Geometry {
    int Cin, CinP8, CG, NT_f, CoutP, NT_b, CGb, No, TAPS;
};

static int validate(const dlwpcs_conv_desc *d, const char *who) {
    if (!d) return fail(DLWPCS_E_INVALID, "%s: null descriptor", who);
    if (!dtype_ok(d->dtype)) return fail(DLWPCS_E_UNSUPPORTED, "%s: dtype %d not built", who, d->dtype);
    if (d->ksize != 1 && d->ksize != 3) return fail(DLWPCS_E_UNSUPPORTED, "%s: kernel size %d (MFMA path serves 1 and 3)", who, d->ksize);
    if (d->B < 0 || d->N < 1 || d->C0 < 1 || d->C1 < 0 || d->Cout < 1) return fail(DLWPCS_E_INVALID, "%s: bad shape B=%d N=%d C0=%d C1=%d Cout=%d", who, d->B, d->N, d->C0, d->C1, d->Cout);
    if (d->up0 && (d->N % 2)) return fail(DLWPCS_E_INVALID, "%s: up0 needs even N", who);
    if (d->halo && d->ksize == 1) return fail(DLWPCS_E_INVALID, "%s: halo with a 1x1 kernel", who);
    if (!d->halo && d->N < d->ksize) return fail(DLWPCS_E_INVALID, "%s: N < kernel size", who);
    if (d->act != DLWPCS_ACT_NONE && d->act != DLWPCS_ACT_LEAKY_CLIP) return fail(DLWPCS_E_INVALID, "%s: unknown activation %d", who, d->act);
    // keras ReLU: negative_slope >= 0, max_value >= 0 (the backward kernels read act' off the saved OUTPUT: y < 0 <=> x < 0)
    if (d->act == DLWPCS_ACT_LEAKY_CLIP && (!(d->alpha >= 0.f) || !(d->vmax >= 0.f)))
        return fail(DLWPCS_E_INVALID, "%s: activation needs negative_slope >= 0 and max_value >= 0, got %g / %g", who, d->alpha, d->vmax);
    if (d->N > 1024) return fail(DLWPCS_E_UNSUPPORTED, "%s: N > 1024", who);
    if (d->c0_valid != 0) {
        if (d->c0_valid < 1 || d->c0_valid > d->C0) return fail(DLWPCS_E_INVALID, "%s: c0_valid %d outside [1, C0 = %d]", who, d->c0_valid, d->C0);
        if (d->C1 != 0) return fail(DLWPCS_E_INVALID, "%s: c0_valid needs a single source (C1 = 0)", who);
        if (ceil_div(d->c0_valid, 32) != ceil_div(d->C0, 32))
            return fail(DLWPCS_E_INVALID, "%s: c0_valid %d and C0 %d must share their last 32-channel tile", who, d->c0_valid, d->C0);
    }
    return DLWPCS_OK;
}

static inline int out_size(const dlwpcs_conv_desc *d) { return d->halo ? d->N : d->N - d->ksize + 1; }
// Logical input channels = rows of the HWIO kernels.  c0_valid > 0: src0 is stored with C0 channels per pixel of which only
// the first c0_valid are real (the rest are zero padding up to the next vector width, e.g. 7 variables in an 8-channel
// layout); the kernels keep c0_valid input rows, the packed fragments carry zeros for the padding (pack_weights_range zero-
// fills k >= K) and the weight-gradient reduction emits the real rows only.
static inline int cin_logical(const dlwpcs_conv_desc *d) { return (d->c0_valid > 0 ? d->c0_valid : d->C0) + d->C1; }

// workspace layout (bytes, 256-aligned regions)
struct WsLayout {
    size_t wpk_f, bias, wpk_b, dxv, dz, partial, bpartial, total;
    int n_eq, n_4, n_5, wg_pix, wg_nblk;
};

// persistent weight-gradient launch geometry: pixels per work item, items (bands) per face, workers per face class
// bf16 matrix-core weight gradient (wgrad_bf16_kernel): bf16 tensors whose channel counts are all multiples of 8
static bool wgrad_bf16_eligible(const dlwpcs_conv_desc *d) {
    // dZ vectors: 8 channels, or (1x1 kernels only, e.g. the 14-channel head) 2 channels with C_out <= 16
    const bool dz_ok = d->Cout % 8 == 0 || (d->ksize == 1 && d->Cout % 2 == 0 && d->Cout <= 16);
    if (d->Cout % 8 != 0 && (d->C0 % 8 != 0 || d->C1 % 8 != 0)) return false;      // 4-B vectors on one side only
    return d->dtype == DLWPCS_BF16 && d->C0 % 2 == 0 && d->C1 % 2 == 0 && dz_ok;
}
// X staging of wgrad_bf16_kernel: channels per load, vectors per tile pixel, tile-pixel capacity of the producers
static void wgrad_bf16_xcfg(const dlwpcs_conv_desc *d, int &xv, int &qx, int &cap_px, int &ct) {
    ct = 1;
    if (d->C0 % 8 == 0 && d->C1 % 8 == 0) {
        xv = 8; qx = 4; cap_px = 512;
        if ((d->C0 + d->C1) % 64 == 0 && d->Cout % 8 == 0) { ct = 2; cap_px = 320; }   // two ci tiles per worker, items <= 192 px
    }
    else if (d->C0 + d->C1 <= 16) { xv = 2; qx = 8; cap_px = 512; }
    else { xv = 2; qx = 16; cap_px = 256; }
}

// DLWPCS_CONV_REUSE_DZ hand-over happens iff the flag is set, there is an activation, and conv_bwd_weights takes the
// wgrad_bf16_kernel path (evaluated identically by conv_bwd_weights, which writes dz, and conv_bwd_data, which reads it)
static bool dz_handover(const dlwpcs_conv_desc *d);

static void wgrad_tiling(const dlwpcs_conv_desc *d, int &pix, int &nblk, int &n_eq, int &n_4, int &n_5) {
    const int No = out_size(d);
    const int face_pix = No * No;
    // pixels per work item: 2 LDS buffers of (X tile + dZ tile) in one CU's 160 KB (bf16 tiles are half the bytes)
    int CAP = 192;
    int ci_tile = 32;                    // input channels per worker
    if (wgrad_bf16_eligible(d)) {
        int xv, qx, cap_px, ct;
        wgrad_bf16_xcfg(d, xv, qx, cap_px, ct);
        CAP = ct == 2 ? 192 : 384;
        ci_tile = 32 * ct;
        // the X tile (item rows + k-1 halo rows, full padded width) must fit the producers' register capacity
        while (CAP > 96 && (long)(tile_rows_for(No <= CAP ? (CAP / No) * No : CAP, No) + d->ksize - 1) * (No + d->ksize - 1) > cap_px)
            CAP -= 96;
    }
    pix = CAP;
    if (No <= CAP) pix = (CAP / No) * No;
    if (pix > face_pix) pix = face_pix;
    nblk = ceil_div(face_pix, pix);
    const int CinP = ceil_div(d->C0 + d->C1, 32) * 32, CoutP = ceil_div(d->Cout, 32) * 32;
    const int pairs = ceil_div(CinP, ci_tile) * (CoutP / 32);
    int wpp = 256 / pairs;               // one worker per CU (256 CUs) spread over the (ci, co) tile pairs
    if (wpp < 3) wpp = 3;
    const long items_per_face = (long)(d->B > 0 ? d->B : 1) * nblk;
    if (wpp > 6 * items_per_face) wpp = (int)(6 * items_per_face);
    if (wpp < 3) wpp = 3;
    n_4 = (wpp + 3) / 6; if (n_4 < 1) n_4 = 1;
    n_5 = n_4;
    n_eq = wpp - n_4 - n_5; if (n_eq < 1) n_eq = 1;
}

static WsLayout ws_layout(const dlwpcs_conv_desc *d);
// wgrad_bf16_kernel applies: eligible dtype / channel counts and the item's X tile fits the producers' registers
static bool wgrad_bf16_fits(const dlwpcs_conv_desc *d, const struct WsLayout &L);

static WsLayout ws_layout(const dlwpcs_conv_desc *d) {
    WsLayout L{};
    const int Cin = d->C0 + d->C1, TAPS = d->ksize * d->ksize;
    const int cgw = cgw_of(d->dtype);
    const int CGf = ceil_div(Cin, cgw), NTf = ceil_div(d->Cout, 32);
    const int CGb = ceil_div(d->Cout, cgw), NTb = ceil_div(Cin, 32);
    const int No = out_size(d);
    size_t off = 0;
    L.wpk_f = off; off += align_up((size_t)3 * NTf * CGf * TAPS * 256 * 4, 256);
    L.bias = off;  off += align_up((size_t)3 * NTf * 32 * 4, 256);
    L.wpk_b = off; off += align_up((size_t)3 * NTb * CGb * TAPS * 256 * 4, 256);
    const int Nv = d->halo ? d->N + d->ksize - 1 : d->N;      // face size of the virtual-input gradient
    L.dxv = off;   off += align_up((size_t)d->B * 6 * Nv * Nv * Cin * dtype_size(d->dtype), 256);
    L.dz = off;    off += align_up((size_t)d->B * 6 * No * No * d->Cout * dtype_size(d->dtype), 256);   // REUSE_DZ hand-over
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

static bool wgrad_bf16_fits(const dlwpcs_conv_desc *d, const WsLayout &L) {
    if (!wgrad_bf16_eligible(d)) return false;
    int xv, qx, cap_px, ct;
    wgrad_bf16_xcfg(d, xv, qx, cap_px, ct);
    const int No = out_size(d), KS = d->ksize;
    const long tile_px = (long)(tile_rows_for(L.wg_pix, No) + KS - 1) * (No + KS - 1);
    return tile_px <= cap_px && L.wg_pix <= 384;
}
static bool dz_handover(const dlwpcs_conv_desc *d) {
    if (!(d->flags & DLWPCS_CONV_REUSE_DZ) || d->act == DLWPCS_ACT_NONE || d->B == 0) return false;
    if (d->dtype == DLWPCS_F32) return d->Cout % 4 == 0;        // wgrad_mfma_kernel<float>, vector dZ path
    return wgrad_bf16_fits(d, ws_layout(d));
}

static void launch_pack(const void *w_eq, const void *w_pol, const void *w_np, void *out, int KS, int Cin, int Cout,
                        int transposed, int flip, int dtype, hipStream_t s) {
    const int K = transposed ? Cout : Cin, Ncol = transposed ? Cin : Cout;
    const int CG = ceil_div(K, cgw_of(dtype)), NTtot = ceil_div(Ncol, 32);
    const size_t total = (size_t)3 * NTtot * CG * KS * KS * 64 * (16 / dtype_size(dtype));
    size_t g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    if (dtype == DLWPCS_BF16)
        hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, s, (const float *)w_eq,
                           (const float *)w_pol, (const float *)w_np, (bf16_t *)out, KS, Cin, Cout, K, Ncol, CG, NTtot, flip,
                           transposed, total);
    else
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3((unsigned)g), dim3(256), 0, s, (const float *)w_eq,
                           (const float *)w_pol, (const float *)w_np, (float *)out, KS, Cin, Cout, K, Ncol, CG, NTtot, flip,
                           transposed, total);
}

}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" size_t dlwpcs_conv_workspace_bytes(const dlwpcs_conv_desc *d) {
    if (validate(d, "conv_workspace_bytes") != DLWPCS_OK) return 0;
    return ws_layout(d).total;
}

extern "C" size_t dlwpcs_conv_packed_bytes(const dlwpcs_conv_desc *d, int which) {
    if (validate(d, "conv_packed_bytes") != DLWPCS_OK) return 0;
    const WsLayout L = ws_layout(d);
    if (which == DLWPCS_PACK_FWD) return L.bias - L.wpk_f;
    if (which == DLWPCS_PACK_BIAS) return L.wpk_b - L.bias;
    if (which == DLWPCS_PACK_BWD) return L.dxv - L.wpk_b;
    fail(DLWPCS_E_INVALID, "conv_packed_bytes: which = %d", which);
    return 0;
}

extern "C" int dlwpcs_pack_batch(const dlwpcs_pack_item *items_dev, int n_items, dlwpcs_stream_t stream) {
    if (n_items < 0 || (n_items > 0 && !items_dev)) return fail(DLWPCS_E_INVALID, "pack_batch: bad arguments");
    if (n_items == 0) return DLWPCS_OK;
    if (n_items > 65535) return fail(DLWPCS_E_UNSUPPORTED, "pack_batch: more than 65535 items");
    // 96 x n_items workgroups (the largest U-Net layer packs ~0.9 M values = ~110 k 16-B entries per direction, 4-5 per thread;
    // small layers' surplus workgroups exit at once).  Measured on the unet2 step, 32 / 64 / 96 / 128 / 192 / 448 per item:
    // 11.5 / 8.9 / 8.5 / 9.7 / 8.9 / 9.7 us.
    const int gx = 96;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)gx, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    return check_launch("pack_batch");
}

extern "C" int dlwpcs_head_mse_tail(const dlwpcs_conv_desc *d, float weight, int overwrite, void *scratch, float *loss_out,
                                    dlwpcs_loss_tail *tail) {
    if (!d || !scratch || !loss_out || !tail) return fail(DLWPCS_E_INVALID, "head_mse_tail: null pointer");
    if (!pw_applies(d)) return fail(DLWPCS_E_UNSUPPORTED, "head_mse_tail: not a layer dlwpcs_head_mse_step serves");
    const double n = (double)d->B * 6 * d->N * d->N * d->Cout;
    tail->partial = (const float *)scratch; tail->loss_out = loss_out;
    tail->nblocks = (int)pw_grid((long)d->B * 6 * d->N * d->N / 16);
    tail->inv_n = (float)(1.0 / n); tail->weight = weight; tail->overwrite = overwrite & 1;
    return DLWPCS_OK;
}
extern "C" int dlwpcs_loss_tail_run(const dlwpcs_loss_tail *tail, dlwpcs_stream_t stream) {
    if (!tail || !tail->partial || !tail->loss_out || tail->nblocks < 1) return fail(DLWPCS_E_INVALID, "loss_tail_run: bad tail");
    hipLaunchKernelGGL(head_stage2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tail->partial, tail->loss_out, tail->nblocks,
                       tail->inv_n, tail->weight, tail->overwrite);
    return check_launch("loss_tail_run");
}

extern "C" size_t dlwpcs_head_mse_scratch_bytes(void) { return (size_t)2048 * 2 * sizeof(float); }

static int head_mse_impl(const dlwpcs_conv_desc *d, const void *x, const void *wpk_fwd, const void *bias_pk,
                         const void *wpk_bwd, const float *target, float weight, void *dy, void *dx,
                         float *loss_out, int overwrite, void *scratch, int mask_dx, float m_alpha, float m_vmax,
                         dlwpcs_stream_t stream);

extern "C" int dlwpcs_head_mse_step(const dlwpcs_conv_desc *d, const void *x, const void *wpk_fwd, const void *bias_pk,
                                    const void *wpk_bwd, const float *target, float weight, void *dy, void *dx,
                                    float *loss_out, int overwrite, void *scratch, dlwpcs_stream_t stream) {
    return head_mse_impl(d, x, wpk_fwd, bias_pk, wpk_bwd, target, weight, dy, dx, loss_out, overwrite, scratch, 0, 0.f, 0.f, stream);
}

extern "C" int dlwpcs_head_mse_step_masked(const dlwpcs_conv_desc *d, const void *x, const void *wpk_fwd, const void *bias_pk,
                                           const void *wpk_bwd, const float *target, float weight, void *dy, void *dx,
                                           float *loss_out, int overwrite, void *scratch, float m_alpha, float m_vmax,
                                           dlwpcs_stream_t stream) {
    if (!(m_alpha >= 0.f) || !(m_vmax >= 0.f))
        return fail(DLWPCS_E_INVALID, "head_mse_step_masked: activation needs negative_slope >= 0 and max_value >= 0");
    return head_mse_impl(d, x, wpk_fwd, bias_pk, wpk_bwd, target, weight, dy, dx, loss_out, overwrite, scratch, 1, m_alpha, m_vmax, stream);
}

static int head_mse_impl(const dlwpcs_conv_desc *d, const void *x, const void *wpk_fwd, const void *bias_pk,
                         const void *wpk_bwd, const float *target, float weight, void *dy, void *dx,
                         float *loss_out, int overwrite, void *scratch, int mask_dx, float m_alpha, float m_vmax,
                         dlwpcs_stream_t stream) {
    int rc = validate(d, "head_mse_step");
    if (rc) return rc;
    if (!x || !wpk_fwd || !wpk_bwd || !target || !dy || !dx || !loss_out || !scratch)
        return fail(DLWPCS_E_INVALID, "head_mse_step: null pointer");
    if (!pw_applies(d) || d->act != DLWPCS_ACT_NONE || d->c0_valid != 0)
        return fail(DLWPCS_E_UNSUPPORTED, "head_mse_step: serves the bf16 pointwise head (k = 1, 32 input channels, even C_out in "
                                         "8..32, no activation)");
    hipStream_t s = (hipStream_t)stream;
    HeadParams H{};
    H.f.in = (const bf16_t *)x; H.f.wpk = (const bf16_t *)wpk_fwd; H.f.bias = (const float *)bias_pk; H.f.out = (bf16_t *)dy;
    H.f.ngroups = (long)d->B * 6 * d->N * d->N / 16; H.f.groups_per_face = d->N * d->N / 16; H.f.Cout = d->Cout;
    H.wpk_bwd = (const bf16_t *)wpk_bwd; H.target = target; H.dx = (bf16_t *)dx; H.partial = (float *)scratch;
    const double n = (double)d->B * 6 * d->N * d->N * d->Cout;
    H.gscale = (float)(weight * 2.0 / n);
    H.mask_dx = mask_dx; H.m_alpha = m_alpha; H.m_vmax = m_vmax;
    const unsigned grid = pw_grid(H.f.ngroups);
    int pidx = -1;
    if (prof_enabled()) {
        Work wk = conv_work(d);
        wk.flops *= 2.0;                                                        // forward + data gradient
        wk.bytes = (double)d->B * 6 * d->N * d->N * (2.0 * 32 * 2 + d->Cout * (4.0 + 2.0));   // x, dx, target, dy
        pidx = prof_begin(d->Cout <= 16 ? KTag<PwHeadName, void, 1>::tag() : KTag<PwHeadName, void, 2>::tag(), wk.flops, wk.bytes, s);
    }
    if (d->Cout <= 16) hipLaunchKernelGGL((pw_head_train_kernel<1>), dim3(grid), dim3(256), 0, s, H);
    else hipLaunchKernelGGL((pw_head_train_kernel<2>), dim3(grid), dim3(256), 0, s, H);
    if (pidx >= 0) prof_end(pidx, s);
    // DLWPCS_HEAD_DEFER_STAGE2: the caller finishes the loss (dlwpcs_head_mse_tail + dlwpcs_loss_tail_run, or inside the
    // weight-gradient reduction: dlwpcs_wgrad_batch_adam_tail)
    if (!(overwrite & DLWPCS_HEAD_DEFER_STAGE2))
        hipLaunchKernelGGL(head_stage2_kernel, dim3(1), dim3(256), 0, s, (const float *)scratch, loss_out, (int)grid,
                           (float)(1.0 / n), weight, overwrite & 1);
    return check_launch("head_mse_step");
}

// the pointwise output layer behind the convolution, folded into its epilogue where the tiling allows (dlwpcs_conv_fwd_head)
struct ConvHeadArgs { const void *wpk; const float *bias; void *out; int *done; };
static int conv_fwd_impl(const dlwpcs_conv_desc *d, const void *src0, const void *src1,
                         const void *w_eq, const void *w_pol, const void *w_np,
                         const void *b_eq, const void *b_pol, const void *b_np,
                         void *y, const int32_t *table_dev,
                         void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream, void *y_pooled, int *pool_done,
                         const ConvHeadArgs *head = nullptr);

extern "C" int dlwpcs_conv_fwd(const dlwpcs_conv_desc *d, const void *src0, const void *src1,
                               const void *w_eq, const void *w_pol, const void *w_np,
                               const void *b_eq, const void *b_pol, const void *b_np,
                               void *y, const int32_t *table_dev,
                               void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    return conv_fwd_impl(d, src0, src1, w_eq, w_pol, w_np, b_eq, b_pol, b_np, y, table_dev, workspace, workspace_bytes, stream,
                         nullptr, nullptr);
}

extern "C" int dlwpcs_avgpool2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream);

extern "C" int dlwpcs_conv_fwd_pool(const dlwpcs_conv_desc *d, const void *src0, const void *src1,
                                    const void *w_eq, const void *w_pol, const void *w_np,
                                    const void *b_eq, const void *b_pol, const void *b_np,
                                    void *y, void *y_pooled, const int32_t *table_dev,
                                    void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    if (!y_pooled) return fail(DLWPCS_E_INVALID, "conv_fwd_pool: null pointer");
    if (!d || !d->halo || d->N % 2) return fail(DLWPCS_E_INVALID, "conv_fwd_pool: needs a halo convolution on an even face size");
    int done = 0;
    int rc = conv_fwd_impl(d, src0, src1, w_eq, w_pol, w_np, b_eq, b_pol, b_np, y, table_dev, workspace, workspace_bytes, stream,
                           y_pooled, &done);
    if (rc || done || d->B == 0) return rc;
    // the tiling of this shape cannot pool in the epilogue: the pooling launch of its own, same result
    return dlwpcs_avgpool2_fwd(y, y_pooled, d->B, d->N, d->Cout, d->dtype, stream);
}

// Convolution + the pointwise output layer behind it (inference: the U-Net's last 3x3 layer and its 1x1 head,
// Azure/train_cs.py:300-305), see include/dlwpcs.h.
extern "C" int dlwpcs_conv_fwd_head(const dlwpcs_conv_desc *d, const void *src0, const void *src1, const void *wpk_fwd,
                                    const void *bias_pk, const dlwpcs_conv_desc *dh, const void *head_wpk_fwd,
                                    const void *head_bias_pk, void *y, void *y_head, const int32_t *table_dev,
                                    void *workspace, size_t workspace_bytes, int *fused, dlwpcs_stream_t stream) {
    if (fused) *fused = 0;
    int rc = validate(d, "conv_fwd_head");
    if (rc) return rc;
    rc = validate(dh, "conv_fwd_head (head)");
    if (rc) return rc;
    if (!wpk_fwd || !head_wpk_fwd || !y || !y_head) return fail(DLWPCS_E_INVALID, "conv_fwd_head: null pointer");
    if (!(d->flags & DLWPCS_CONV_PREPACKED) || !(dh->flags & DLWPCS_CONV_PREPACKED))
        return fail(DLWPCS_E_INVALID, "conv_fwd_head: both layers take dlwpcs_pack_batch operands (DLWPCS_CONV_PREPACKED)");
    const int No = out_size(d);
    if (dh->B != d->B || dh->N != No || dh->C0 != d->Cout || dh->dtype != d->dtype)
        return fail(DLWPCS_E_INVALID, "conv_fwd_head: the head (B=%d N=%d C_in=%d) does not consume the layer's output (B=%d N=%d C=%d)",
                    dh->B, dh->N, dh->C0, d->B, No, d->Cout);
    if (d->flags & DLWPCS_CONV_OUT_PADDED) return fail(DLWPCS_E_INVALID, "conv_fwd_head: DLWPCS_CONV_OUT_PADDED belongs to the head's descriptor");
    if (d->B == 0) return DLWPCS_OK;
    // folded: bf16 pointwise head without activation whose stored rows are 32 channels (C_out = 32, or padded to 32)
    const int head_rows = (dh->flags & DLWPCS_CONV_OUT_PADDED) ? (dh->Cout + 7) / 8 * 8 : dh->Cout;
    const bool can = pw_applies(dh) && dh->act == DLWPCS_ACT_NONE && dh->c0_valid == 0 && head_rows == 32 && d->Cout == 32 &&
                     d->dtype == DLWPCS_BF16 && d->ksize == 3;
    int done = 0;
    ConvHeadArgs H{head_wpk_fwd, (const float *)head_bias_pk, y_head, &done};
    rc = conv_fwd_impl(d, src0, src1, wpk_fwd, nullptr, nullptr, bias_pk, nullptr, nullptr, y, table_dev, workspace, workspace_bytes,
                       stream, nullptr, nullptr, can ? &H : nullptr);
    if (rc) return rc;
    if (done) { if (fused) *fused = 1; return DLWPCS_OK; }
    // the tiling of this shape cannot fold the head: its own launch on y, same contract as dlwpcs_conv_fwd
    return conv_fwd_impl(dh, y, nullptr, head_wpk_fwd, nullptr, nullptr, head_bias_pk, nullptr, nullptr, y_head, nullptr, workspace,
                         workspace_bytes, stream, nullptr, nullptr);
}

static int conv_fwd_impl(const dlwpcs_conv_desc *d, const void *src0, const void *src1,
                         const void *w_eq, const void *w_pol, const void *w_np,
                         const void *b_eq, const void *b_pol, const void *b_np,
                         void *y, const int32_t *table_dev,
                         void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream, void *y_pooled, int *pool_done,
                         const ConvHeadArgs *head) {
    int rc = validate(d, "conv_fwd");
    if (rc) return rc;
    if (!src0 || !w_eq || (!w_pol && !(d->flags & DLWPCS_CONV_PREPACKED)) || !y || !workspace)
        return fail(DLWPCS_E_INVALID, "conv_fwd: null pointer");
    if (d->C1 > 0 && !src1) return fail(DLWPCS_E_INVALID, "conv_fwd: C1 > 0 but src1 is null");
    if (d->halo && !table_dev) return fail(DLWPCS_E_INVALID, "conv_fwd: halo requested without table");
    const bool prepacked = (d->flags & DLWPCS_CONV_PREPACKED) != 0;
    if (!prepacked && (b_eq == nullptr) != (b_pol == nullptr)) return fail(DLWPCS_E_INVALID, "conv_fwd: b_eq and b_pol must both be given or both be null");
    const WsLayout L = ws_layout(d);
    if (workspace_bytes < L.total) return fail(DLWPCS_E_WORKSPACE, "conv_fwd: workspace %zu < %zu bytes", workspace_bytes, L.total);
    if (d->B == 0) return DLWPCS_OK;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int Cin = d->C0 + d->C1;
    const void *wpk = prepacked ? w_eq : ws + L.wpk_f;      // PREPACKED: w_eq / b_eq are dlwpcs_pack_batch outputs
    const float *bpk = prepacked ? (const float *)b_eq : (const float *)(ws + L.bias);
    if (!prepacked) launch_pack(w_eq, w_pol, w_np, ws + L.wpk_f, d->ksize, cin_logical(d), d->Cout, 0, d->flip_north_pole, d->dtype, s);
    const int NTtot = ceil_div(d->Cout, 32);
    if (b_eq && !prepacked) {
        const int n = 3 * NTtot * 32;
        hipLaunchKernelGGL(pack_bias_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, (const float *)b_eq,
                           (const float *)b_pol, (const float *)b_np, (float *)(ws + L.bias), d->Cout, NTtot * 32);
    }
    ConvKParams P{};
    P.src0 = src0; P.src1 = src1; P.ymask = nullptr;
    P.wpk = wpk; P.bias = b_eq ? bpk : nullptr; P.out = y; P.table = table_dev;
    P.B = d->B; P.Nin = d->N; P.No = out_size(d);
    P.C0 = d->C0; P.C1 = d->C1; P.Cin = Cin; P.Cout = d->Cout;
    P.CG = ceil_div(Cin, cgw_of(d->dtype)); P.NTtot = NTtot; P.up0 = d->up0;
    P.mode = d->halo ? MODE_HALO : MODE_DIRECT;
    P.act = d->act; P.alpha = d->alpha; P.vmax = d->vmax;
    P.pool_out = y_pooled; P.pool_done = pool_done;
    if (head) { P.head_w = head->wpk; P.head_b = head->bias; P.head_out = head->out; P.head_done = head->done; }
    if ((d->flags & DLWPCS_CONV_OUT_PADDED) && !pw_applies(d))
        return fail(DLWPCS_E_UNSUPPORTED, "conv_fwd: DLWPCS_CONV_OUT_PADDED serves the bf16 pointwise output layer only");
    if (pw_applies(d)) {
        PwParams Q{};
        Q.in = (const bf16_t *)src0; Q.wpk = (const bf16_t *)wpk; Q.bias = b_eq ? bpk : nullptr; Q.out = (bf16_t *)y;
        Q.ngroups = (long)d->B * 6 * d->N * d->N / 16; Q.groups_per_face = d->N * d->N / 16; Q.Cout = d->Cout;
        // padded rows: the packed weights and biases of the channels beyond C_out are zero, so the kernel simply writes
        // ceil8(C_out) channels per pixel (the head has no activation; act(0) = 0 anyway)
        if (d->flags & DLWPCS_CONV_OUT_PADDED) Q.Cout = (d->Cout + 7) / 8 * 8;
        Q.alpha = d->alpha; Q.vmax = d->vmax;
        int pidx = -1;
        const bool actv = d->act != DLWPCS_ACT_NONE;
        if (prof_enabled()) {
            const Work wk = conv_work(d);
            const char *tag = Q.Cout <= 16 ? (actv ? KTag<PwFwdName, void, true, 1>::tag() : KTag<PwFwdName, void, false, 1>::tag())
                                           : (actv ? KTag<PwFwdName, void, true, 2>::tag() : KTag<PwFwdName, void, false, 2>::tag());
            pidx = prof_begin(tag, wk.flops, wk.bytes, s);
        }
        const dim3 pgrid(pw_grid(Q.ngroups));
        if (Q.Cout <= 16) {
            if (actv) hipLaunchKernelGGL((pw_fwd_kernel<true, 1>), pgrid, dim3(256), 0, s, Q);
            else hipLaunchKernelGGL((pw_fwd_kernel<false, 1>), pgrid, dim3(256), 0, s, Q);
        } else {
            if (actv) hipLaunchKernelGGL((pw_fwd_kernel<true, 2>), pgrid, dim3(256), 0, s, Q);
            else hipLaunchKernelGGL((pw_fwd_kernel<false, 2>), pgrid, dim3(256), 0, s, Q);
        }
        if (pidx >= 0) prof_end(pidx, s);
        return check_launch("pw_fwd");
    }
    // the network's input layer (14 or 26 channels): 16-B vectors with a shifted tail instead of 4-B loads
    if (d->dtype == DLWPCS_BF16 && d->ksize == 3 && d->halo && d->C1 == 0 && d->C0 >= 8 && d->C0 % 2 == 0 && d->C0 % 8 != 0 &&
        NTtot <= 2) {
        // (one 16-channel operand group for <= 16 channels: half the MFMAs of a 32-channel chunk)
        return dispatch_conv_tail8(d->C0 <= 16 ? 16 : 32, NTtot, P, conv_work(d), s);
    }
    return dispatch_conv(d->dtype, d->ksize, vec_width(d->C0, d->C1, d->dtype), P, conv_work(d), s);
}

// dz_given: `dy` is already dz = dy * act'(y) (pre-masked gradient convention): no mask on load, y unused.
// m0 / m1 (nullable): the sources themselves; dsrc0 / dsrc1 come out multiplied by act'(m; m_alpha, m_vmax).  The multiply is
// fused wherever a kernel of this call writes the final value (direct-store epilogue + ring fix-up, inverse gather); what is
// left (in-place 1x1 / 'valid' gradients, odd vector widths) gets one elementwise launch.
static int conv_bwd_data_impl(const dlwpcs_conv_desc *d, const void *dy, const void *y, bool dz_given,
                              const void *w_eq, const void *w_pol, const void *w_np,
                              void *dsrc0, void *dsrc1, const void *m0, const void *m1, float m_alpha, float m_vmax,
                              const int32_t *inv_table_dev, void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream,
                              const char *who, int *ring_query = nullptr) {
    int rc = validate(d, who);
    if (rc) return rc;
    const bool prepacked = (d->flags & DLWPCS_CONV_PREPACKED) != 0;
    if (!dy || !w_eq || (!w_pol && !prepacked) || !workspace) return fail(DLWPCS_E_INVALID, "%s: null pointer", who);
    const bool has_act = d->act != DLWPCS_ACT_NONE && !dz_given;
    if (has_act && !y) return fail(DLWPCS_E_INVALID, "%s: activation needs the saved output y", who);
    if (d->halo && !inv_table_dev) return fail(DLWPCS_E_INVALID, "%s: halo requested without inverse table", who);
    if ((m0 || m1) && (!(m_alpha >= 0.f) || !(m_vmax >= 0.f)))
        return fail(DLWPCS_E_INVALID, "%s: mask activation needs negative_slope >= 0 and max_value >= 0, got %g / %g", who, m_alpha, m_vmax);
    if (!dsrc0 && !dsrc1) return DLWPCS_OK;
    if (!dsrc0) m0 = nullptr;
    if (!dsrc1 || d->C1 == 0) m1 = nullptr;
    const WsLayout L = ws_layout(d);
    if (workspace_bytes < L.total) return fail(DLWPCS_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, L.total);
    if (d->B == 0) return DLWPCS_OK;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int Cin = d->C0 + d->C1;
    const void *wpk = prepacked ? w_eq : ws + L.wpk_b;      // PREPACKED: w_eq is the wpk_bwd output of dlwpcs_pack_batch
    void *dxv = ws + L.dxv;
    if (!prepacked) launch_pack(w_eq, w_pol, w_np, ws + L.wpk_b, d->ksize, cin_logical(d), d->Cout, 1, d->flip_north_pole, d->dtype, s);
    const int No = out_size(d);
    const int n0 = d->up0 ? d->N / 2 : d->N;
    const size_t n_src0 = (size_t)d->B * 6 * n0 * n0 * d->C0, n_src1 = (size_t)d->B * 6 * d->N * d->N * d->C1;
    // what is still to be multiplied by act'(m) when the kernels of this call are done
    auto finish_masks = [&](bool todo0, bool todo1) -> int {
        int r = DLWPCS_OK;
        if (todo0 && m0) r = launch_mask_inplace(dsrc0, m0, n_src0, m_alpha, m_vmax, d->dtype, s);
        if (!r && todo1 && m1) r = launch_mask_inplace(dsrc1, m1, n_src1, m_alpha, m_vmax, d->dtype, s);
        return r;
    };
    if (pw_applies(d) && !has_act && dsrc0) {
        PwParams Q{};
        Q.in = (const bf16_t *)dy; Q.wpk = (const bf16_t *)wpk; Q.bias = nullptr; Q.out = (bf16_t *)dsrc0;
        Q.ngroups = (long)d->B * 6 * d->N * d->N / 16; Q.groups_per_face = d->N * d->N / 16; Q.Cout = d->Cout;
        int pidx = -1;
        // pre-masked gradients: act'(m0) inside the launch (the masking pass behind it cost 14.7 us per head of the production model)
        const bool in_kernel_mask = m0 != nullptr;
        if (prof_enabled()) {
            const Work wk = conv_work(d);
            pidx = prof_begin(in_kernel_mask ? "pw_dgrad_kernel<true>" : "pw_dgrad_kernel<false>", wk.flops, wk.bytes, s);
        }
        Q.mask = (const bf16_t *)m0; Q.m_alpha = m_alpha; Q.m_vmax = m_vmax;
        if (in_kernel_mask) hipLaunchKernelGGL(pw_dgrad_kernel<true>, dim3(pw_grid(Q.ngroups)), dim3(256), 0, s, Q);
        else hipLaunchKernelGGL(pw_dgrad_kernel<false>, dim3(pw_grid(Q.ngroups)), dim3(256), 0, s, Q);
        if (pidx >= 0) prof_end(pidx, s);
        rc = check_launch("pw_dgrad");
        return rc ? rc : finish_masks(!in_kernel_mask, false);
    }
    ConvKParams P{};
    // DLWPCS_CONV_REUSE_DZ: the weight-gradient call that ran just before left dz = dy * act'(y) in the workspace
    const bool dz_ready = !dz_given && dz_handover(d);
    P.src0 = dz_ready ? (const void *)(ws + L.dz) : dy; P.src1 = nullptr;
    P.ymask = (has_act && !dz_ready) ? y : nullptr;
    P.wpk = wpk; P.bias = nullptr; P.out = dxv; P.table = nullptr;
    P.B = d->B; P.Nin = No; P.No = No + d->ksize - 1;     // full correlation: output = input + k - 1
    P.C0 = d->Cout; P.C1 = 0; P.Cin = d->Cout; P.Cout = Cin;
    P.CG = ceil_div(d->Cout, cgw_of(d->dtype)); P.NTtot = ceil_div(Cin, 32); P.up0 = 0;
    P.mode = MODE_ZERO;
    P.act = DLWPCS_ACT_NONE; P.alpha = d->alpha; P.vmax = d->vmax;
    // direct mode: interior cells of the padded gradient go straight to the (non-upsampled) sources' gradient tensors, only
    // the halo ring is materialised in dxv and a border fix-up replaces the full inverse-gather pass
    int direct_done = 0, mask_done = 0;
    const bool can_direct = d->halo && d->ksize == 3;
    P.d0 = (can_direct && dsrc0 && !d->up0) ? dsrc0 : nullptr;
    P.d1 = (can_direct && dsrc1 && d->C1 > 0) ? dsrc1 : nullptr;
    P.dsplit = d->C0;
    P.direct_done = &direct_done;
    P.m0 = P.d0 ? m0 : nullptr; P.m1 = P.d1 ? m1 : nullptr;
    P.m_alpha = m_alpha; P.m_vmax = m_vmax; P.m_thr1 = bf16_mask_threshold(m_vmax);
    P.mask_done = &mask_done;
    // ---- gather form (DLWPCS_CONV_DGRAD_GATHER; conv_ws.h EDGE): the gradient on the N x N grid, every cell complete when it is
    // stored.  Sources that are not upsampled are written directly (masked where asked), an upsampled source 0 goes through the
    // workspace and ONE 2 x 2 block-sum launch; no halo ring, no fix-up, no inverse gather.
    if ((d->flags & DLWPCS_CONV_DGRAD_GATHER) && d->halo && d->ksize == 3 && d->dtype == DLWPCS_BF16 && !P.ymask && d->N >= 8 &&
        vec_width(d->Cout, 0, d->dtype) == 8 && d->C0 % 8 == 0 && d->C1 % 8 == 0) {
        ConvKParams G = P;
        const int M = d->N + 2;
        const int32_t *plan = inv_table_dev + (size_t)6 * d->N * d->N * 4 + DGG_HEADER;
        G.table = plan; G.edge = 1;
        G.mode = MODE_HALO; G.Nin = d->N; G.No = d->N;
        ConvEdgeArgs EA{};
        EA.src = plan + (size_t)6 * M * M;
        int32_t wids[36];
        int grc = dgrad_gather_wids(d->N, wids);
        for (int k = 0; k < 36; ++k) EA.wids[k] = (int8_t)wids[k];
        g_edge_args = EA;
        G.d0 = (dsrc0 && !d->up0) ? dsrc0 : nullptr;
        G.d1 = (dsrc1 && d->C1 > 0) ? dsrc1 : nullptr;
        G.m0 = G.d0 ? m0 : nullptr; G.m1 = G.d1 ? m1 : nullptr;
        // an upsampled source 0: its gradient = the 2 x 2 block sums of its channels' gradient, written by the kernel's epilogue as
        // a second output (masked there too) where the tiling allows (g_usum) -- else through the workspace + one window-sum launch
        int g_usum = 0;
        G.pool_out = (dsrc0 && d->up0) ? dsrc0 : nullptr; G.pool_mask = G.pool_out ? m0 : nullptr; G.pool_done = &g_usum;
        int g_direct = 0, g_mask = 0;
        G.direct_done = &g_direct; G.mask_done = &g_mask;
        G.dry_run = 1;
        const bool need_direct = G.d0 || G.d1;
        if (!grc) grc = dispatch_conv(d->dtype, d->ksize, 8, G, conv_work(d), s);
        if (!grc && (g_direct || !need_direct)) {
            if (ring_query) { *ring_query = 0; return DLWPCS_OK; }      // nothing is deferred: there is no ring
            G.dry_run = 0;
            grc = dispatch_conv(d->dtype, d->ksize, 8, G, conv_work(d), s);
            if (grc) return grc;
            bool todo0 = m0 != nullptr && !(G.d0 && (g_mask & 1)) && !g_usum, todo1 = m1 != nullptr && !(G.d1 && (g_mask & 2));
            if (dsrc0 && !G.d0 && !g_usum) {
                int masked = 0;
                grc = launch_src_grad(dxv, dsrc0, nullptr, d->B, d->N, Cin, 0, d->C0, d->up0, 0, d->dtype, s, m0, m_alpha, m_vmax, &masked);
                if (grc) return grc;
                if (masked) todo0 = false;
            }
            if (dsrc1 && d->C1 > 0 && !G.d1) {
                int masked = 0;
                grc = launch_src_grad(dxv, dsrc1, nullptr, d->B, d->N, Cin, d->C0, d->C1, 0, 0, d->dtype, s, m1, m_alpha, m_vmax, &masked);
                if (grc) return grc;
                if (masked) todo1 = false;
            }
            return finish_masks(todo0, todo1);
        }
        // (a tiling without the line-store epilogue: the padded-grid path below)
    }
    // no halo, no upsample, one source: the virtual input IS the source -> write its gradient in place (no routing pass)
    const bool whole = !d->halo && !d->up0 && d->C1 == 0 && dsrc0;
    if (whole) P.out = dsrc0;
    P.dry_run = ring_query ? 1 : 0;
    rc = dispatch_conv(d->dtype, d->ksize, vec_width(d->Cout, 0, d->dtype), P, conv_work(d), s);
    if (ring_query) { *ring_query = (!rc && direct_done && P.d0 && !m0) ? 1 : 0; return rc; }
    if (rc) return rc;
    if (whole) return finish_masks(true, false);
    // dxv is the gradient of the (halo-padded, if halo) virtual input: (B,6,Nv,Nv,Cin), Nv = No + k - 1
    // halo: Nv = N + 2; plain: Nv = N.  Route to the sources (inverse halo gather, upsample adjoint, channel split).
    // A source the epilogue wrote directly: its ring fix-up masks iff the epilogue did (mask_done); otherwise both stay
    // unmasked and the elementwise pass finishes.
    bool todo0 = m0 != nullptr, todo1 = m1 != nullptr;
    const bool direct0 = direct_done && P.d0, direct1 = direct_done && P.d1;
    const void *rm0 = (direct0 && !(mask_done & 1)) ? nullptr : m0;      // mask handed to the kernel that finishes source 0
    const void *rm1 = (direct1 && !(mask_done & 2)) ? nullptr : m1;
    if (dsrc0 && dsrc1 && d->C1 > 0 && d->halo && direct_done && !P.d0 && P.d1) {     // decoder layer: both in one launch
        rc = launch_src_pair(dxv, dsrc0, dsrc1, inv_table_dev, d->B, d->N, d->C0, d->C1, d->up0, d->dtype, s, m0, rm1, m_alpha, m_vmax);
        if (rc) return rc;
        return finish_masks(false, todo1 && !rm1);
    }
    if (dsrc0) {
        if (direct0 && !m0 && (d->flags & DLWPCS_CONV_DEFER_RING0)) {
            // the caller folds this fix-up into its next pass over dsrc0 (dlwpcs_avgpool2_bwd_ring): the ring stays in dxv
        } else if (direct0) {
            rc = launch_ring_fix(dxv, dsrc0, inv_table_dev, d->B, d->N, Cin, 0, d->C0, d->dtype, s, rm0, m_alpha, m_vmax);
            if (rm0) todo0 = false;
        } else {
            int masked = 0;
            rc = launch_src_grad(dxv, dsrc0, inv_table_dev, d->B, d->N, Cin, 0, d->C0, d->up0, d->halo, d->dtype, s, m0, m_alpha, m_vmax, &masked);
            if (masked) todo0 = false;
        }
        if (rc) return rc;
    }
    if (dsrc1 && d->C1 > 0) {
        if (direct1) {
            rc = launch_ring_fix(dxv, dsrc1, inv_table_dev, d->B, d->N, Cin, d->C0, d->C1, d->dtype, s, rm1, m_alpha, m_vmax);
            if (rm1) todo1 = false;
        } else {
            int masked = 0;
            rc = launch_src_grad(dxv, dsrc1, inv_table_dev, d->B, d->N, Cin, d->C0, d->C1, 0, d->halo, d->dtype, s, m1, m_alpha, m_vmax, &masked);
            if (masked) todo1 = false;
        }
        if (rc) return rc;
    }
    return finish_masks(todo0, todo1);
}

extern "C" int dlwpcs_conv_bwd_data(const dlwpcs_conv_desc *d, const void *dy, const void *y,
                                    const void *w_eq, const void *w_pol, const void *w_np,
                                    void *dsrc0, void *dsrc1, const int32_t *inv_table_dev,
                                    void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    return conv_bwd_data_impl(d, dy, y, false, w_eq, w_pol, w_np, dsrc0, dsrc1, nullptr, nullptr, 0.f, 0.f, inv_table_dev, workspace,
                              workspace_bytes, stream, "conv_bwd_data");
}

extern "C" int dlwpcs_conv_bwd_data_masked(const dlwpcs_conv_desc *d, const void *dz,
                                           const void *w_eq, const void *w_pol, const void *w_np,
                                           void *dsrc0, void *dsrc1, const void *m0, const void *m1, float m_alpha, float m_vmax,
                                           const int32_t *inv_table_dev,
                                           void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    return conv_bwd_data_impl(d, dz, nullptr, true, w_eq, w_pol, w_np, dsrc0, dsrc1, m0, m1, m_alpha, m_vmax, inv_table_dev, workspace,
                              workspace_bytes, stream, "conv_bwd_data_masked");
}

extern "C" int dlwpcs_conv_ring_info(const dlwpcs_conv_desc *d, size_t *dxv_offset, int *channels) {
    if (validate(d, "conv_ring_info") != DLWPCS_OK || !dxv_offset || !channels) return 0;
    if (!d->halo || d->ksize != 3 || d->up0 || d->B < 1) return 0;
    const WsLayout L = ws_layout(d);
    // the configuration choice of the data-gradient launch itself, without launching (placeholder non-null pointers)
    int q = 0;
    char dummy = 0;
    dlwpcs_conv_desc dd = *d;
    dd.act = DLWPCS_ACT_NONE;
    dd.flags |= DLWPCS_CONV_PREPACKED;
    const int rc = conv_bwd_data_impl(&dd, &dummy, nullptr, true, &dummy, nullptr, nullptr, &dummy, d->C1 > 0 ? &dummy : nullptr, nullptr,
                                      nullptr, 0.f, 0.f, (const int32_t *)&dummy, &dummy, L.total, nullptr, "conv_ring_info", &q);
    if (rc || !q) return 0;
    *dxv_offset = L.dxv;
    *channels = d->C0 + d->C1;
    return 1;
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
        const size_t wbytes = (size_t)TAPS * cin_logical(d) * d->Cout * 4;
        (void)hipMemsetAsync(dw_eq, 0, wbytes, s);
        (void)hipMemsetAsync(dw_pol, 0, wbytes, s);
        if (dw_np) (void)hipMemsetAsync(dw_np, 0, wbytes, s);
        if (db_eq) (void)hipMemsetAsync(db_eq, 0, (size_t)d->Cout * 4, s);
        if (db_pol) (void)hipMemsetAsync(db_pol, 0, (size_t)d->Cout * 4, s);
        if (db_np) (void)hipMemsetAsync(db_np, 0, (size_t)d->Cout * 4, s);
        return DLWPCS_OK;
    }
    WgradKParams W{};
    ConvKParams &P = W.c;
    P.src0 = src0; P.src1 = src1; P.ymask = nullptr; P.table = table_dev;
    P.B = d->B; P.Nin = d->N; P.No = out_size(d);
    P.C0 = d->C0; P.C1 = d->C1; P.Cin = Cin; P.Cout = d->Cout; P.up0 = d->up0;
    P.mode = d->halo ? MODE_HALO : MODE_DIRECT;
    P.alpha = d->alpha; P.vmax = d->vmax;
    P.pix_per_block = L.wg_pix; P.nblk_face = L.wg_nblk;
    P.W2 = P.No + KS - 1; P.magicW2 = div_magic(P.W2); P.magicNo = div_magic(P.No);
    P.tile_rows_max = tile_rows_for(L.wg_pix, P.No) + (KS - 1);
    W.dy = dy; W.y = y;
    W.dz_out = dz_handover(d) ? ws + L.dz : nullptr;
    W.partial = (float *)(ws + L.partial);
    const bool want_bias = db_eq || db_pol || db_np;
    W.bpartial = want_bias ? (float *)(ws + L.bpartial) : nullptr;
    W.CinP = CinP; W.CoutP = CoutP;
    W.n_eq = L.n_eq; W.n_4 = L.n_4; W.n_5 = L.n_5;
    P.magicN = div_magic(P.Nin);
    P.tune = tune_bits();
    W.magicB = P.B > 1 ? div_magic(P.B) : 0; W.magicNb = P.nblk_face > 1 ? div_magic(P.nblk_face) : 0;
    if (P.C1 == 0) P.src1 = P.src0;
    const bool mask = d->act != DLWPCS_ACT_NONE;
    dim3 grid((unsigned)(L.n_eq + L.n_4 + L.n_5), (unsigned)(CinP / 32), (unsigned)(CoutP / 32));
    int xv = 0, qx = 0, cap_px = 0, ct = 1;
    if (wgrad_bf16_eligible(d)) wgrad_bf16_xcfg(d, xv, qx, cap_px, ct);
    if (wgrad_bf16_fits(d, L)) {
        const int pcap = (L.wg_pix + 15) & ~15;
        size_t lds = 2 * ((size_t)ct * P.tile_rows_max * P.W2 * 64 + (size_t)pcap * 64);
        grid.y = (unsigned)(CinP / (32 * ct));
        {   // the epilogue's reduction slots + bias staging alias the buffers: 4 * (TAPS - TAPS / (4 / ct)) slots of 4 KB + 8 KB
            const int nph = 4 / ct, rslots = TAPS - TAPS / nph;
            const size_t need = ((size_t)4 * rslots * 1024 + 2048) * 4;
            if (lds < need) lds = need;
        }
        if (lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: LDS tile of %zu bytes exceeds 160 KiB", lds);
        if ((long)P.Nin * P.Nin >= (1l << 16))
            return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: face size %d too large for the 16-bit index arithmetic", P.Nin);
#define WGB_LAUNCH(KSV, MASKV, XVV, QXV, CTV, DVV)                                                                        \
    do {                                                                                                                  \
        auto kern = wgrad_bf16_kernel<KSV, MASKV, XVV, QXV, CTV, DVV>;                                                    \
        if (lds > 64 * 1024) {                                                                                            \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));    \
        }                                                                                                                 \
        int pidx = -1;                                                                                                    \
        if (prof_enabled()) {                                                                                             \
            const Work wk = conv_work(d);                                                                                 \
            pidx = prof_begin(KTag<WgradBf16Name, void, KSV, MASKV, XVV, QXV, CTV, DVV>::tag(), wk.flops, wk.bytes, s);   \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, W);                                                             \
        if (pidx >= 0) prof_end(pidx, s);                                                                                 \
    } while (0)
#define WGB_X(KSV, MASKV)                                                                                                 \
    do {                                                                                                                  \
        if (xv == 8 && ct == 2) WGB_LAUNCH(KSV, MASKV, 8, 4, 2, 8);                                                       \
        else if (xv == 8) WGB_LAUNCH(KSV, MASKV, 8, 4, 1, 8);                                                             \
        else if (qx == 8) WGB_LAUNCH(KSV, MASKV, 2, 8, 1, 8);                                                             \
        else WGB_LAUNCH(KSV, MASKV, 2, 16, 1, 8);                                                                         \
    } while (0)
        if (d->Cout % 8 != 0) {
            // 1x1 kernel with an even C_out <= 16: 4-B dZ vectors; X must take the plain 16-B path
            if (xv != 8) return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: even-only C_in and C_out together");
            if (mask) WGB_LAUNCH(1, true, 8, 4, 1, 2); else WGB_LAUNCH(1, false, 8, 4, 1, 2);
        }
        else if (KS == 3) { if (mask) WGB_X(3, true); else WGB_X(3, false); }
        else { if (mask) WGB_X(1, true); else WGB_X(1, false); }
#undef WGB_X
#undef WGB_LAUNCH
        rc = check_launch("wgrad_bf16");
        if (rc || (d->flags & DLWPCS_CONV_DEFER_REDUCE)) return rc;
        launch_wgrad_reduce(s, W.partial, W.bpartial, dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np, KS, cin_logical(d), d->Cout, CinP, CoutP,
                            L.n_eq, L.n_4, L.n_5, d->flip_north_pole, (d->flags & DLWPCS_CONV_ACCUMULATE_WGRAD) ? 1 : 0, want_bias);
        return check_launch("wgrad_reduce");
    }
    const int pix_cap = (L.wg_pix + 1) & ~1;
    const int vw = vec_width(d->C0, d->C1, d->dtype);
    const bool bf = d->dtype == DLWPCS_BF16;
    const size_t bufb = ((size_t)P.tile_rows_max * P.W2 * 32 + (size_t)pix_cap * 32) * 4;
    if ((size_t)P.tile_rows_max * P.W2 > 448 || pix_cap > 192)
        return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: tile of %d x %d pixels exceeds the producers' register capacity", P.tile_rows_max, P.W2);
    if ((long)P.Nin * P.Nin >= (1l << 16))
        return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: face size %d too large for the 16-bit index arithmetic", P.Nin);
    size_t lds = 2 * bufb;
    if (lds < (WG_SCRATCH_FLOATS + 1024) * 4) lds = (WG_SCRATCH_FLOATS + 1024) * 4;   // reduction scratch + bias staging alias the buffers
    if (lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv_bwd_weights: LDS tile of %zu bytes exceeds 160 KiB", lds);
#define WG_LAUNCH(TV, TS, KSV, VWV, MASKV)                                                                                \
    do {                                                                                                                  \
        auto kern = wgrad_mfma_kernel<TV, KSV, VWV, MASKV>;                                                               \
        if (lds > 64 * 1024) {                                                                                            \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));    \
        }                                                                                                                 \
        int pidx = -1;                                                                                                    \
        if (prof_enabled()) {                                                                                             \
            const Work wk = conv_work(d);                                                                                 \
            pidx = prof_begin(KTag<WgradMfmaName, TV, KSV, VWV, MASKV>::tag(), wk.flops, wk.bytes, s);                    \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, W);                                                             \
        if (pidx >= 0) prof_end(pidx, s);                                                                                 \
    } while (0)
#define WG_VW(KSV, MASKV)                                                                                                 \
    do {                                                                                                                  \
        if (bf) {                                                                                                         \
            if (vw == 8) WG_LAUNCH(bf16_t, "unsigned short", KSV, 8, MASKV);                                              \
            else if (vw >= 2) WG_LAUNCH(bf16_t, "unsigned short", KSV, 2, MASKV);                                         \
            else WG_LAUNCH(bf16_t, "unsigned short", KSV, 1, MASKV);                                                      \
        } else {                                                                                                          \
            if (vw == 4) WG_LAUNCH(float, "float", KSV, 4, MASKV);                                                        \
            else if (vw == 2) WG_LAUNCH(float, "float", KSV, 2, MASKV);                                                   \
            else WG_LAUNCH(float, "float", KSV, 1, MASKV);                                                                \
        }                                                                                                                 \
    } while (0)
    if (KS == 3) { if (mask) WG_VW(3, true); else WG_VW(3, false); }
    else { if (mask) WG_VW(1, true); else WG_VW(1, false); }
#undef WG_VW
#undef WG_LAUNCH
    rc = check_launch("wgrad_mfma");
    if (rc || (d->flags & DLWPCS_CONV_DEFER_REDUCE)) return rc;
    launch_wgrad_reduce(s, W.partial, W.bpartial, dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np, KS, cin_logical(d), d->Cout, CinP, CoutP,
                        L.n_eq, L.n_4, L.n_5, d->flip_north_pole, (d->flags & DLWPCS_CONV_ACCUMULATE_WGRAD) ? 1 : 0, want_bias);
    return check_launch("wgrad_reduce");
}

extern "C" int dlwpcs_conv_wgrad_reduce_item(const dlwpcs_conv_desc *d, void *dw_eq, void *dw_pol, void *dw_np, void *db_eq,
                                             void *db_pol, void *db_np, void *workspace, size_t workspace_bytes,
                                             dlwpcs_reduce_item *item) {
    int rc = validate(d, "conv_wgrad_reduce_item");
    if (rc) return rc;
    if (!dw_eq || !dw_pol || !workspace || !item) return fail(DLWPCS_E_INVALID, "conv_wgrad_reduce_item: null pointer");
    if ((db_np != nullptr) != (dw_np != nullptr) && db_eq) return fail(DLWPCS_E_INVALID, "conv_wgrad_reduce_item: dw_np/db_np must match");
    const WsLayout L = ws_layout(d);
    if (workspace_bytes < L.total)
        return fail(DLWPCS_E_WORKSPACE, "conv_wgrad_reduce_item: workspace %zu < %zu bytes", workspace_bytes, L.total);
    char *ws = (char *)workspace;
    const int Cin = d->C0 + d->C1;
    const bool want_bias = db_eq || db_pol || db_np;
    const bool vec = wgrad_reduce_vec(dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np, d->Cout);
    dlwpcs_reduce_item it{};
    it.partial = (const float *)(ws + L.partial);
    it.bpartial = want_bias ? (const float *)(ws + L.bpartial) : nullptr;
    it.dw_eq = (float *)dw_eq; it.dw_pol = (float *)dw_pol; it.dw_np = (float *)dw_np;
    it.db_eq = (float *)db_eq; it.db_pol = (float *)db_pol; it.db_np = (float *)db_np;
    it.ksize = d->ksize; it.Cin = cin_logical(d); it.Cout = d->Cout;
    it.CinP = ceil_div(Cin, 32) * 32; it.CoutP = ceil_div(d->Cout, 32) * 32;
    it.n_eq = L.n_eq; it.n_4 = L.n_4; it.n_5 = L.n_5;
    it.flip_north_pole = d->flip_north_pole;
    it.accumulate = (d->flags & DLWPCS_CONV_ACCUMULATE_WGRAD) ? 1 : 0;
    it.vec = vec ? 4 : 1;
    it.nblocks = d->B == 0 ? 0 : wgrad_reduce_blocks(d->ksize, cin_logical(d), d->Cout, want_bias, vec);   // B == 0: nothing was produced
    *item = it;
    return DLWPCS_OK;
}

extern "C" int dlwpcs_wgrad_reduce_batch(const dlwpcs_reduce_item *items_dev, const dlwpcs_reduce_item *items_host,
                                         int n_items, dlwpcs_stream_t stream) {
    if (n_items < 0) return fail(DLWPCS_E_INVALID, "wgrad_reduce_batch: n_items %d", n_items);
    if (n_items == 0) return DLWPCS_OK;
    if (!items_dev || !items_host) return fail(DLWPCS_E_INVALID, "wgrad_reduce_batch: null item table");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n_items; base += REDUCE_BATCH_MAX) {
        const int n = n_items - base < REDUCE_BATCH_MAX ? n_items - base : REDUCE_BATCH_MAX;
        ReduceStarts st{};
        int total = 0;
        double bytes = 0.0;          // algorithmic traffic: every partial read once, every gradient written once
        for (int k = 0; k < n; ++k) {
            const dlwpcs_reduce_item &it = items_host[base + k];
            if (it.nblocks < 0) return fail(DLWPCS_E_INVALID, "wgrad_reduce_batch: item %d has nblocks < 0", base + k);
            st.first[k] = total;
            total += it.nblocks;
            if (it.nblocks > 0)
                bytes += 4.0 * ((double)(it.n_eq + it.n_4 + it.n_5) * it.ksize * it.ksize * it.CinP * it.CoutP +
                                2.0 * it.ksize * it.ksize * it.Cin * it.Cout);
        }
        for (int k = n; k <= REDUCE_BATCH_MAX; ++k) st.first[k] = total;
        if (total == 0) continue;
        int pidx = -1;
        if (prof_enabled()) pidx = prof_begin("wgrad_reduce_batch_kernel", 0.0, bytes, s);
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)total), dim3(256), 0, s, items_dev + base, n, st);
        if (pidx >= 0) prof_end(pidx, s);
    }
    return check_launch("wgrad_reduce_batch");
}
