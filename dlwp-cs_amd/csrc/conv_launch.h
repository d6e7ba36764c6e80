// Launch side of the forward / data-gradient kernel (conv_ws.h): tiling choice per layer shape, LDS budget, the launch itself.
// Templates only -- every translation unit that includes this header instantiates the kernels it dispatches to: the instantiations
// are split over conv_inst_f32.hip / conv_inst_bf16.hip / conv_inst_edge.hip so that they compile in parallel (one file: 3.5 min).
#pragma once
#include "conv_ws.h"

namespace dlwpcs {

struct Work { double flops, bytes; };   // algorithmic work of one launch (for the opt-in profiler)

// pixels per tile for a workgroup that can hold BM pixels: whole rows when that does not cost extra tiles, else a flat
// range of BM pixels (partial rows)
static int tile_pixels(int BM, int No) {
    const int face_pix = No * No;
    int pix = BM < face_pix ? BM : face_pix;
    if (No <= BM) {
        int whole = (BM / No) * No;
        if (whole > face_pix) whole = face_pix;
        if (ceil_div(face_pix, whole) <= ceil_div(face_pix, pix)) pix = whole;
    }
    return pix;
}

// ILV (conv_ws.h): cost of the most expensive tile list when `gx` workers share the 6 * B * ceil(face_pix / pix) tiles of a launch, a
// tile of mn M tiles costing 2 x ceil(mn / WM) + 1.  split = false: the plain contiguous split (some worker has ceil(ntiles / gx) full
// tiles).  split = true: the M tiles dealt round-robin and, where the last tile of a face is cheaper than the others, the best cost
// split (*gb workers take all the short tiles + *fb full ones; 0 / 0: the plain split is as good).
static long ilv_cost(int pix, int face_pix, int B, int gx, int WM, int *gb, int *fb, bool split, int c2 = 2) {
    auto cost = [&](int npix) { const int mn = (npix + 31) / 32; return 4 * ceil_div(mn, WM) + c2; };       // (in half units)
    const int nbl = ceil_div(face_pix, pix), cF = cost(pix), cL = cost(face_pix - (nbl - 1) * pix);
    const long ntiles = 6l * B * nbl;
    const long plain = (long)ceil_div((int)ntiles, gx) * cF;       // (some workgroup has that many tiles, full ones in general)
    if (gb) *gb = 0;
    if (fb) *fb = 0;
    if (!split || cL >= cF || nbl < 2) return plain;
    const long F = 6l * (nbl - 1) * B, S = 6l * B;
    long best = plain;
    for (int GB = 1; GB < gx; ++GB) {
        const int GA = gx - GB;
        // FB full tiles for the B group: as many as keep a B list (ceil(FB / GB) full + ceil(S / GB) short) under the A lists
        const long sB = ceil_div((int)S, GB) * (long)cL;
        for (long mB = 0; mB <= ceil_div((int)F, gx) + 1; ++mB) {
            const long FB = mB * GB < F ? mB * GB : F, FA = F - FB;
            const long cA = (long)ceil_div((int)FA, GA) * cF, cB = mB * cF + sB;
            const long mx = cA > cB ? cA : cB;
            if (mx < best) { best = mx; if (gb) *gb = GB; if (fb) *fb = (int)FB; }
        }
    }
    return best;
}

// the gather-form plan of the data-gradient launch in flight on this thread (conv_bwd_data_impl sets it around its dispatch_conv;
// every other launch passes the empty record: the kernels that read it are the EDGE instantiations only)
extern thread_local ConvEdgeArgs g_edge_args;

struct ConvWsName { static const char *str() { return "conv_mfma_ws_kernel"; } };

template <typename T, int KS, int KC, int MT, int NT, int WM, int WN, int VW, int MODE, bool MASK, bool TAIL8 = false, bool MOUT = false,
          bool EDGE = false>
static int launch_conv_cfg(ConvKParams P, const Work &W, hipStream_t s) {
    constexpr int ES = sizeof(T), CGW = 32 / ES;
    constexpr int BM = 32 * MT * WM, NTB = NT * WN, NTHREADS = 64 * WM * WN;
    const int face_pix = P.No * P.No;
    // column blocks (conv_ws.h): forward pass with the halo gather on wide faces -- two strips when that doubles the rows of a band
    P.ncol = 1;
    if (MODE == MODE_HALO && !EDGE && KS == 3 && (tune_bits() & TUNE_CONV_STRIPS) && P.No >= 64 && P.No % 2 == 0 && P.Cout % (16 / ES) == 0) {
        const int wt = P.No / 2, rows2 = BM / wt, rows1 = BM / P.No;
        // whole rows of the strip, bands that tile it evenly, and at least twice the rows of a full-width band
        if (BM % wt == 0 && rows2 <= P.No && P.No % rows2 == 0 && rows1 >= 1 && rows2 >= 2 * rows1) P.ncol = 2;
    }
    P.Wt = P.No / P.ncol;
    const int strip_pix = P.No * P.Wt;
    int pix = P.ncol > 1 ? (BM / P.Wt) * P.Wt : tile_pixels(BM, P.No);
    // fp32 forward pass (MFMA-bound: a tile costs its M tiles): tiles of k x WM M tiles, k = 1 .. MT, dealt to the waves round-robin
    // (ILV) with the tile list cut by cost -- whichever setting has the cheapest longest list; the plain tiling unless one beats it
    P.ilv_fwd = 0;
    int fwd_gb = 0, fwd_fb = 0;
    const int gy_ = ceil_div(P.NTtot, NTB);
    const int gx_ = 256 / gy_ < 1 ? 1 : 256 / gy_;
    if (MODE == MODE_HALO && !EDGE && KS == 3 && MT == 3 && sizeof(T) == 4 && (tune_bits() & TUNE_CONV_ILV) && (tune_bits() & TUNE_CONV_ILV_FWD) &&
        P.ncol == 1 && P.pool_out == nullptr && gx_ > 1) {
        constexpr int c2 = 2;       // fixed cost of a tile in half M-tile rounds (swept 0 .. 4 on the fp32 step and encoder6: flat within 0.4 %)
        long best = ilv_cost(pix, face_pix, P.B, gx_, WM, nullptr, nullptr, false, c2);
        for (int k = MT; k >= 1; --k) {
            const int cand = 32 * WM * k;
            if (cand >= face_pix) continue;
            int gb = 0, fb = 0;
            const long c = ilv_cost(cand, face_pix, P.B, gx_, WM, &gb, &fb, true, c2);
            if (c < best) { best = c; pix = cand; P.ilv_fwd = 1; fwd_gb = gb; fwd_fb = fb; }
        }
    }
    P.pix_per_block = pix;
    P.nblk_face = ceil_div(strip_pix, pix);
    P.W2 = P.Wt + KS - 1;
    P.magicW2 = div_magic(P.W2);
    P.magicWt = div_magic(P.Wt);
    P.magicNcol = P.ncol > 1 ? div_magic(P.ncol) : 0;
    P.magicNo = div_magic(P.No);
    P.magicN = div_magic(P.Nin);
    P.magicB = P.B > 1 ? div_magic(P.B) : 0;
    P.magicNblk = P.nblk_face > 1 ? div_magic(P.nblk_face) : 0;
    P.tile_rows_max = tile_rows_for(pix, P.Wt) + (KS - 1);
    P.ntiles = P.B * 6 * P.ncol * P.nblk_face;
    P.split_gb = P.split_fb = 0;
    P.tune = tune_bits();
    const size_t in_b = (size_t)P.tile_rows_max * P.W2 * (KC * ES + 16), w_b = (size_t)NTB * (KC / CGW) * (KS * KS + (EDGE ? 3 : 0)) * 1024;
    // pooled second output: every consumer wave must own whole PAIRS of tile rows (its 32 * MT pixels and the tile a multiple of
    // two rows, all tiles full), whole 32-channel output tiles, and LDS room for one patch per M tile -- else the caller pools
    // with a launch of its own (pool_done stays 0)
    // (... or, faces whose row is exactly one wave's 32 * MT pixels -- N = 96 -- with four consumer waves on a 4-row tile: the
    // waves take half-rows of two rows each instead, P.colsplit)
    const bool rowpairs = (32 * MT) % (2 * P.Wt) == 0 && pix % (2 * P.Wt) == 0;
    const bool halfrows = !rowpairs && 32 * MT == P.Wt && WM == 4 && WN == 1 && pix == 4 * P.Wt && P.Wt % 4 == 0;
    bool pool = MODE != MODE_ZERO && P.pool_out != nullptr && P.No % 2 == 0 && P.Wt % 2 == 0 && (rowpairs || halfrows) &&
                strip_pix % pix == 0 && P.Cout % 32 == 0 && P.Cout % (16 / ES) == 0;
    // (gather-form data gradient: the "pooled" output is the 2 x 2 sum of an upsampled source's channels -- whole n tiles of it)
    if (EDGE) pool = pool && !MOUT && ES == 2 && P.dsplit > 0 && P.dsplit % 32 == 0;
    const size_t buf = in_b + w_b;
    size_t patch_b = (size_t)(WM * WN) * 32 * (32 * ES + 16);
    if (pool && 2 * buf + patch_b * MT <= 160 * 1024) patch_b *= MT; else pool = false;
    size_t lds = 2 * buf + patch_b;                                              // + wave-private epilogue patches
    P.patches = 1;
    P.wstat = 0;
    const int nchunks = ceil_div(P.CG, KC / CGW);
    if (nchunks > 2 && nchunks <= 4 && (tune_bits() & TUNE_CONV_WSTAT) && 2 * in_b + nchunks * w_b + patch_b <= 160 * 1024) {
        P.wstat = nchunks;                                                       // one resident weight area per chunk
        lds = 2 * in_b + nchunks * w_b + patch_b;
    } else if (lds > 160 * 1024) { lds = 2 * buf; P.patches = 0; pool = false; } // large faces: direct quad stores instead
    if (!pool) P.pool_out = nullptr;
    // pointwise output layer folded into the epilogue: bf16, the layer's 32 output channels = ONE n tile of ONE wave column, the
    // line-store epilogue (its rows are the head's rows: 32 channels), no pooled second output
    {
        const bool head = P.head_w != nullptr && P.head_out != nullptr && ES == 2 && NT == 1 && WN == 1 && !EDGE && !MOUT && !MASK &&
                          MODE != MODE_ZERO && P.Cout == 32 && P.patches && !pool;
        if (head) P.out = P.head_out; else P.head_w = nullptr;
        if (P.head_done) *P.head_done = head ? 1 : 0;
    }
    P.colsplit = (pool && halfrows) ? 1 : 0;
    if (P.pool_done) *P.pool_done = pool ? 1 : 0;
    if ((MODE == MODE_ZERO || EDGE) && KS == 3 && P.patches && P.Cout % (16 / ES) == 0 && P.dsplit % (16 / ES) == 0) {
        if (P.direct_done) *P.direct_done = (P.d0 || P.d1) ? 1 : 0;             // the line-store epilogue honours d0 / d1
    } else {
        P.d0 = P.d1 = nullptr;
    }
    if (EDGE) {
        // The producers stage ONE weight-id triple per tile behind the nine taps (the top row's, or the bottom row's when the tile
        // holds the face's last row; the polar faces' two triples are equal): a tile must not hold both edge rows of a face.  Faces
        // that fit into one tile (N <= 16) keep the padded-grid path: served in gather form -- both triples staged, or two half-face
        // tiles -- they measured 29-46 us against 23-25 (EXPERIMENTS.md).
        if (P.nblk_face < 2)
            return fail(DLWPCS_E_UNSUPPORTED, "conv: gather-form data gradient: a tile holds both edge rows of a face (N=%d)", P.No);
    }
    if (lds > 160 * 1024)
        return fail(DLWPCS_E_UNSUPPORTED, "conv: LDS tile of %zu bytes exceeds 160 KiB (face size %d)", lds, P.No);
    if (P.ncol > 1 && !P.patches)
        return fail(DLWPCS_E_UNSUPPORTED, "conv: internal: column strips need the line-store epilogue (face size %d)", P.No);
    if (P.tile_rows_max > 32)
        return fail(DLWPCS_E_UNSUPPORTED, "conv: %d tile rows exceed the producers' 5-bit row field", P.tile_rows_max);
    if ((size_t)P.tile_rows_max * P.W2 > (size_t)3 * NTHREADS)
        return fail(DLWPCS_E_UNSUPPORTED, "conv: tile of %d x %d pixels exceeds the producers' register capacity", P.tile_rows_max, P.W2);
    if ((long)6 * face_pix * P.Cout * ES >= (1l << 31))
        return fail(DLWPCS_E_UNSUPPORTED, "conv: one sample of the output (%ld bytes) exceeds the 32-bit store offsets",
                    (long)6 * face_pix * P.Cout * ES);
    if ((long)P.Nin * P.Nin * 6 >= (1l << 16) * 6 && (long)P.Nin * P.Nin >= (1l << 16))
        return fail(DLWPCS_E_UNSUPPORTED, "conv: face size %d too large for the 16-bit index arithmetic", P.Nin);
    auto kern = conv_mfma_ws_kernel<T, KS, KC, MT, NT, WM, WN, VW, MODE, MASK, TAIL8, MOUT, EDGE>;
    if (!MOUT || !(P.d0 || P.d1)) {
        // the epilogue masks only what it stores directly: whoever routes the rest (ring fix-up, inverse gather) applies the rest
        if (!(P.d0)) P.m0 = nullptr;
        if (!(P.d1)) P.m1 = nullptr;
        if (!MOUT) P.m0 = P.m1 = nullptr;
    }
    if (P.mask_done) *P.mask_done = (MOUT && (P.m0 || P.m1)) ? ((P.m0 ? 1 : 0) | (P.m1 ? 2 : 0)) : 0;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int gy = ceil_div(P.NTtot, NTB);
    int gx = 256 / gy;                              // one workgroup (consumers + producers) per CU, all groups resident at once
    if (gx < 1) gx = 1;
    if (gx > P.ntiles) gx = P.ntiles;
    dim3 grid((unsigned)gx, (unsigned)gy);
    // ILV cost split (conv_ws.h): data gradient, MT = 3, the last band of a face shorter than the others.  Tile cost = 2 x rounds of
    // M tiles per consumer wave + 1.  Choose how many workers (GB) take the short tiles + FB full ones so that the most expensive
    // list is as cheap as possible; keep the plain split unless that beats it.
    if (MODE == MODE_ZERO && MT == 3 && sizeof(T) == 4 && (P.tune & TUNE_CONV_ILV) && P.nblk_face > 1 && gx > 1)
        (void)ilv_cost(pix, face_pix, P.B, gx, WM, &P.split_gb, &P.split_fb, true);
    if (P.ilv_fwd) { P.split_gb = fwd_gb; P.split_fb = fwd_fb; }
    if (P.dry_run) return DLWPCS_OK;
    int pidx = -1;
    // (the tag carries every template argument, as rocprofv3 / nm -C print the instantiation: bench.py joins its PMC records on this
    // name; tests/test_abi.py checks every registered tag against the library's own kernel symbols)
    const char *tag = KTag<ConvWsName, T, KS, KC, MT, NT, WM, WN, VW, MODE, MASK, TAIL8, MOUT, EDGE>::tag();
    if (prof_enabled()) {
        pidx = prof_begin(tag, W.flops, W.bytes, s);
    }
    hipLaunchKernelGGL(kern, grid, dim3(2 * NTHREADS), lds, s, P, g_edge_args);
    if (pidx >= 0) prof_end(pidx, s);
    return check_launch("conv_mfma");
}

template <typename T, int KS, int VW, int MODE, bool MASK, bool MOUT = false, bool EDGE = false>
static int launch_conv(const ConvKParams &P, const Work &W, hipStream_t s) {
    const int face_pix = P.No * P.No;
    constexpr int VWF = 16 / (int)sizeof(T);        // full 16-B vectors
    constexpr int K2 = 64 / (int)sizeof(T);         // channels in a 64-B chunk row (16 fp32 / 32 bf16)
    constexpr int K1 = K2 / 2;
    if constexpr (KS == 1) return launch_conv_cfg<T, KS, K1, 3, 1, 4, 1, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
    else if constexpr (VW != VWF) return launch_conv_cfg<T, KS, K1, 3, 1, 4, 1, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);   // odd channel counts
    else {
        // Measured (MI355X, batch 32): the smaller tiles (MT = 2 / MT = 1) that would even out the tile count per CU
        // lose more to halo re-staging (the producers become the bottleneck) than they gain -> fixed MT = 3 tilings.
        if (P.NTtot == 1) return launch_conv_cfg<T, KS, K2, 3, 1, 4, 1, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
        // data gradient with 64 output channels: the padded grid (N + 2 columns) leaves a 192-pixel tile 3 rows at N = 48
        // (5 fetched per 3 computed, 150 of 192 pixels used); two 32-channel groups with 384-pixel tiles get 7 rows (9 per 7,
        // 350 of 384) and read the smaller operand (dz) twice.  bf16 step -0.7 % (fp32 -0.3 %); the same split for the forward pass
        // measured +-0.
        if (P.NTtot == 2 && (MODE == MODE_ZERO || EDGE) && (tune_bits() & TUNE_CONV_SPLIT2_BWD))
            return launch_conv_cfg<T, KS, K2, 3, 1, 4, 1, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
        // bf16, 64 output channels from 65-128 input channels (3-4 chunks): 32 output channels per workgroup, whose 4 x 18 KB of
        // fragments fit as resident areas beside two 384-pixel input buffers (64 per workgroup would need 4 x 37 KB)
        // (faces of more than 320 pixels: at N = 12 a 384-pixel tile is 37 % full and the layer came out 6 us slower; the
        // 128 -> 64 forward at N = 24: 33.3 -> 28.5 us)
        if (P.NTtot == 2 && sizeof(T) == 2 && P.CG > 2 * (K2 / (32 / (int)sizeof(T))) && P.CG <= 4 * (K2 / (32 / (int)sizeof(T))) &&
            face_pix > 320 && (tune_bits() & TUNE_CONV_WSTAT))
            return launch_conv_cfg<T, KS, K2, 3, 1, 4, 1, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
        if (P.NTtot == 2) return launch_conv_cfg<T, KS, K2, 3, 1, 2, 2, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
        // more than 64 output channels: 64 per workgroup (two 32-channel chunks whose weight fragments STAY in the two LDS
        // buffers) and the 256 workgroups split over the output-channel groups, instead of 128 channels per workgroup in four
        // 16-channel chunks whose 37 KB of fragments had to be re-fetched every chunk (the 128-channel data gradient at N = 24:
        // 43.9 us, producers weight-fetch-bound; whole bf16 step -1.9 %, fp32 -0.9 %)
        // (not the forward pass on faces of <= 320 pixels: 64 -> 128 at N = 12 measured 13.4 us with the 160-pixel tiling, 14.9 split)
        if ((tune_bits() & TUNE_CONV_SPLIT_N) && (face_pix > 320 || MODE == MODE_ZERO))
            return launch_conv_cfg<T, KS, K2, 3, 1, 2, 2, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
        if (face_pix <= 320) {
            // (gather form: one wave over the whole face would hold both edge rows -- dispatch_conv_edge takes the 4-wave M split)
            if constexpr (EDGE) return fail(DLWPCS_E_UNSUPPORTED, "conv: gather-form data gradient: one-wave tiling of a small face");
            else return launch_conv_cfg<T, KS, K1, 5, 1, 1, 4, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
        }
        return launch_conv_cfg<T, KS, K1, 3, 1, 1, 4, VW, MODE, MASK, false, MOUT, EDGE>(P, W, s);
    }
}

template <typename T, int KS, int MODE, bool MASK>
static int dispatch_vw(int vw, const ConvKParams &P, const Work &W, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        if (vw == 4) return launch_conv<T, KS, 4, MODE, MASK>(P, W, s);
    } else {
        if (vw == 8) return launch_conv<T, KS, 8, MODE, MASK>(P, W, s);
    }
    if (vw >= 2) return launch_conv<T, KS, 2, MODE, MASK>(P, W, s);
    return launch_conv<T, KS, 1, MODE, MASK>(P, W, s);
}

// the entry points of the instantiation units
int dispatch_conv_f32(int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s);      // conv_inst_f32.hip
int dispatch_conv_bf16(int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s);     // conv_inst_bf16.hip (not the gather form)
int dispatch_conv_tail8(int kc, int NTtot, const ConvKParams &P, const Work &W, hipStream_t s); // conv_inst_bf16.hip: TAIL8 forward
int dispatch_conv_edge(const ConvKParams &P, const Work &W, hipStream_t s);                     // conv_inst_edge.hip: gather-form data gradient

}  // namespace dlwpcs
