// Forward / data-gradient convolution of libdlwpcs (gfx950): the persistent, wave-specialised kernel (body: conv_ws_body, launched
// per layer by conv_launch.h through conv_mfma_ws_kernel).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "mfma_common.h"

namespace dlwpcs {

enum { MODE_DIRECT = 0, MODE_HALO = 1, MODE_ZERO = 2 };

struct ConvKParams {
    const void *src0, *src1;    // virtual-input sources, channels_last, element type T
    const void *ymask;          // data-gradient mode: saved forward output (T), act' applied on load (or nullptr)
    const void *wpk;            // packed weights [3][NTtot][CG][TAPS][2][32][16 B]: 4 fp32 / 8 bf16 per lane
    const float *bias;          // packed bias [3][NTtot*32] fp32 or nullptr
    void *out;                  // (B,6,No,No,Cout), element type T
    const int32_t *table;       // (6, Nin+2, Nin+2) halo table (MODE_HALO, k=3)
    int edge;                   // EDGE instantiations (data gradient in gather form, see conv_ws_body and ConvEdgeArgs): requested
    int B, Nin, No;             // face size of V, face size of the output
    int C0, C1, Cin, Cout;      // Cin = C0 + C1
    int CG, NTtot;              // ceil(Cin/CGW) (CGW = 8 fp32 / 16 bf16 channels per MFMA operand group), ceil(Cout/32)
    int up0;                    // src0 lives on the Nin/2 grid
    int mode;
    int act;                    // epilogue activation
    float alpha, vmax;
    int pix_per_block;          // valid pixels per workgroup (<= 32*MT*WM)
    int nblk_face;              // workgroups per (sample, face)
    int W2;                     // tile width = No + KS - 1
    uint32_t magicW2, magicNo, magicN;
    uint32_t magicB, magicNblk; // exact-division magics of B and nblk_face (0 when the divisor is 1)
    int patches;                // LDS holds the wave-private epilogue patches (0: no room -> direct quad stores)
    int wstat;                  // > 0: one resident LDS weight area per channel chunk (= the chunk count), see the kernel
    // Data-gradient direct mode (MODE_ZERO, k = 3, halo): output channels [0, dsplit) belong to source 0, the rest to source
    // 1; where d0 / d1 is non-null the INTERIOR cells of the padded gradient go straight to that source's gradient tensor
    // (B,6,No-2,No-2,channels of the source) and only the halo ring is written to `out`
    void *d0, *d1;
    int dsplit;
    // Pre-masked gradients (data gradient, direct mode, bf16, MOUT instantiations): where m0 / m1 is non-null the cells that go
    // straight to d0 / d1 are multiplied by act'(m) first -- m0 / m1 are the SOURCES themselves (outputs of the activated layers
    // that produced them, same shape as d0 / d1), so that the gradient arrives at those layers as dz = dy * act'(y) already
    const void *m0, *m1;
    float m_alpha, m_vmax;
    uint32_t m_thr1;            // bf16_mask_threshold(m_vmax)
    int *direct_done;           // HOST pointer: set to 1 by launch_conv_cfg when the kernel it launched honours d0 / d1
    int *mask_done;             // HOST pointer: bit 0 / 1 set when the launched kernel masks what it stores to d0 / d1
    int dry_run;                // host only: choose the configuration, report direct_done / mask_done, launch nothing
    // Forward pass with the 2x2 average pooling of the output as a SECOND output (Azure/train_cs.py:282,287: AveragePooling3D
    // behind the block's last convolution): (B,6,No/2,No/2,Cout), written by the epilogue out of the LDS patches (launch_conv_cfg
    // checks the tiling: every consumer wave owns whole pairs of rows).  pool_done: HOST pointer, set to 1 when the launched
    // kernel does it.
    void *pool_out;
    int *pool_done;
    // EDGE (data gradient in gather form) with an UPSAMPLED source 0 (UpSampling3D in front of the convolution, Azure/train_cs.py:292,
    // 298): the gradient of that source is the 2 x 2 block SUM of the gradient of its channels on the fine grid.  The same second-
    // output machinery writes it -- pool_out = the source's gradient (B,6,No/2,No/2,dsplit), scale 1 instead of 1/4, times
    // act'(pool_mask) where pool_mask (the source itself) is given -- and the n tiles of that source store nothing on the fine grid.
    const void *pool_mask;
    int colsplit;               // pooled output, faces whose row is exactly one wave's 32 * MT pixels (N = 96): see launch_conv_cfg
    // Forward pass with the POINTWISE OUTPUT LAYER behind it folded into the epilogue (inference: the last 3x3 convolution of the
    // U-Net + the 1x1 head, Azure/train_cs.py:300-305): head_w / head_b are the head's dlwpcs_pack_batch operands (forward
    // fragments [3][4 k groups of 8][32 columns][8] bf16, bias [3][32] fp32, zero beyond its C_out); the tile's activated
    // 32-channel result, rounded to bf16 exactly as it would have been stored, is the B operand of two more MFMAs straight out of
    // the accumulators and what is stored (to head_out, rows of 32 channels) is the head's result.  bf16, 32 output channels, one
    // n tile per wave.  head_done: HOST pointer, set to 1 when the launched kernel does it (else `out` gets the layer's own output).
    const void *head_w;
    const float *head_b;
    void *head_out;
    int *head_done;
    int tune;                   // scheduling tunables (tune_bits(): DLWPCS_TUNE, default set below)
    int tile_rows_max;          // rows reserved in LDS
    int ntiles;                 // B * 6 * nblk_face (persistent kernel)
    int split_gb, split_fb;     // ILV cost split (launch_conv_cfg): the LAST split_gb workers take all the short tiles (the last band
                                // of every face) and split_fb of the full ones, the others the remaining full tiles; 0: plain split
    // COLUMN BLOCKS (round 6; forward pass on wide faces): a face of No x No cells is cut into ncol strips of Wt = No / ncol columns and
    // a tile is a band of rows of ONE strip -- at N = 96 a 384-pixel tile is 8 rows x 48 columns (fetches 10 x 50 cells: 1.30 x) instead
    // of 4 rows x 96 (6 x 98: 1.53 x, and the counters say the L2 absorbs none of it).  Everything inside a tile (band geometry, LDS
    // addresses, the waves' pixel ownership, W2 = Wt + KS - 1) is that of a face of width Wt; the strip's first column x0 enters where
    // global addresses are formed (halo-table lookup, store / pooled-store offsets).  ncol = 1, Wt = No: the layout of rounds 1-5.
    int Wt, ncol;
    uint32_t magicWt, magicNcol;
    int ilv_fwd;                // fp32 forward pass: M tiles dealt round-robin + cost split (ILV in the kernel), chosen by launch_conv_cfg
};

// EDGE instantiations: `P.table` is the forward halo table inside a dlwpcs_dgrad_gather_plan buffer, `src` its border-cell records
// [6][4 No - 4][8] (six window positions, the wrong-tap mask, 0), `wids` the two weight-id triples per face [6][2][3] BY VALUE (scalar
// loads out of the kernel arguments: no memory round trip in front of the fragment loads that need them).  A kernel argument of
// its own.
struct ConvEdgeArgs { const int32_t *src; int8_t wids[36]; };

// Scheduling tunables, bit set.  Defaults are the measured-best values; DLWPCS_TUNE=<int> overrides them for A/B runs.
//   1: weight-gradient kernels: producer waves run at s_setprio 2 (they are the second-dispatched, i.e. arbitration-losing,
//      half of the workgroup and the consumers wait for them at every barrier)
//      (default since round 3: fp32 step -1.9 %, encoder6 -1.6 % on top of bit 2; the bf16 step runs the batched kernel instead)
//   2: forward / data-gradient kernel: the same for its producer waves (default since round 3: with the consumers' waits for
//      store acknowledgements gone the producers' issue slots matter again: fp32 step -1.1 %, encoder6 -1.2 %, rollout -0.7 %,
//      bf16 step -0.2 %)
//   4: forward / data-gradient kernel: weight fragments stay in LDS across tiles (see `wres` in the producer)
//  16: forward / data-gradient kernel, more than 64 output channels: 64 per workgroup and the workgroups split over the
//      output-channel groups (see launch_conv), instead of 128 per workgroup in 16-channel chunks
//  32: data gradient with 64 output channels (= the layer's input channels): two groups of 32 with the 384-pixel tiling
//  64: 3-4 channel chunks per tile: one resident LDS weight area per chunk (P.wstat), with the tiling that makes them fit
// 256: forward / data-gradient kernel: late start of the workgroups with the shorter tile list, bits 12..17 = how late (see the
//      kernel; default 3)
// 512: data-gradient kernel: M tiles of a tile dealt to the consumer waves round-robin, short tiles skip the M tiles they do not
//      have, tile list split by cost (see ILV in the kernel)
//1024: forward kernel on faces of >= 64 cells: tiles are bands of a column strip (ConvKParams::ncol)
//2048: fp32 forward kernel: tiles of fewer M tiles, dealt round-robin, tile list split by cost where that shortens the longest list (ConvKParams::ilv_fwd)
enum { TUNE_WG_PRODUCER_PRIO = 1, TUNE_CONV_PRODUCER_PRIO = 2, TUNE_CONV_WEIGHTS_STAY = 4, TUNE_CONV_SPLIT_N = 16,
       TUNE_CONV_SPLIT2_BWD = 32, TUNE_CONV_WSTAT = 64, TUNE_CONV_STAGGER = 256, TUNE_CONV_ILV = 512, TUNE_CONV_STRIPS = 1024,
       TUNE_CONV_ILV_FWD = 2048 };
static int tune_bits() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLWPCS_TUNE"); v = e ? atoi(e) : (TUNE_WG_PRODUCER_PRIO | TUNE_CONV_PRODUCER_PRIO | TUNE_CONV_WEIGHTS_STAY | TUNE_CONV_SPLIT_N | TUNE_CONV_SPLIT2_BWD | TUNE_CONV_WSTAT |
                               TUNE_CONV_STAGGER | TUNE_CONV_ILV | TUNE_CONV_STRIPS | TUNE_CONV_ILV_FWD | (3 << 12));
                 }
    return v;
}

// ------------------------------------------------------------------------------------------------------------------
// Forward / data-gradient kernel: persistent, wave-specialised.
//
// A workgroup = NCW consumer waves (one per SIMD: ds_read + MFMA only) + NCW producer waves (global -> LDS copies); one
// workgroup per CU.  The (tile, channel chunk) pairs of the workgroup's static tile list form one stream; LDS holds
// two chunk buffers:
//     producers:  fill(0); B; fill(1); B; fill(2); B; ...          (B = workgroup barrier, one per chunk)
//     consumers:           B; mma(0);  B; mma(1);  B; mma(2); ...
// so chunk g+1 is being fetched while the matrix cores run chunk g, and a consumer's instruction stream between two
// barriers is nothing but LDS fragment reads (double-buffered in registers) and MFMAs.
//
// Producer code is STRAIGHT-LINE per chunk: every address is computed with selects, every load of the chunk (weights,
// input tile, and -- at a tile's first chunk -- the next tile's halo-table entries) is issued back to back and waited
// for once.  This matters: with any branch between two loads hipcc waits vmcnt(0) per load (measured: 12 serialised L2
// round trips, ~17k cycles per chunk, the consumers idle at every barrier).  MODE is a template parameter for that reason.
//   MODE_HALO  : cube-sphere halo resolved through the (6,N+2,N+2) table          (forward, fused padding)
//   MODE_DIRECT: input consumed as is ('valid' on an already padded tensor, 1x1)   (forward)
//   MODE_ZERO  : zero border of width k-1 = full correlation                        (data gradient)
//   MASK       : dz = dy * act'(y) applied while fetching                           (data gradient through an activation)
// ------------------------------------------------------------------------------------------------------------------
// T = element type of the activations in HBM / LDS (float, or bf16_t with bf16 MFMA); all LDS geometry is in BYTES and
// identical for both: a pixel row holds KC channels (64 B at KC = 16 fp32 / 32 bf16) + 16 B pad.
// TAIL8 (bf16, VW = 8, one source): the source's channel count is even and >= 8 but not a multiple of 8 (14 = 7 variables
// x 2 steps, 26 = 13 x 2).  Pixel rows are then only 4-B aligned; the vector that would run past the last channel is loaded
// as the pixel's LAST 8 channels (in bounds) and shifted into place, instead of falling back to 4-B loads (7 / 13 per pixel).
// Who sets the tile period (s_memtime marks, 32 -> 32 channels at N = 48, cycles per tile): the consumers used to, with 2.2 k
// of MFMA phase + 3.1 k of epilogue + 0.8 k of set-up against ~4 k for the producers.  Two attempts to run the epilogue
// BESIDE the next tile's MFMAs (a second team of consumer waves: three waves per SIMD = 168 VGPRs, spilled; the epilogue cut
// into slices between the MFMAs of the same wave: hipcc hoists the slices' arithmetic into clumps, and the extra VALU work in
// the MFMA phase slows the co-resident producer wave) both measured slower and are gone.  What worked was making the
// epilogue itself cheap -- it is VALU-issue bound, sharing its SIMD with a producer wave: accumulators start at the bias,
// the activation is max + min instead of compare + select, store addresses are per-(face, band) constants -- 1.75 k + 0.44 k cycles now -- and then
// the producers' address arithmetic per load was cut to one multiply-add (see `sup`): the two sides are now balanced within
// ~10 % (4.4 k consumer vs ~4.9 k producer cycles per tile).
// conv_ws_body: the work of worker `lw` of the launch's `G` workers (they share the tile list); `by` = which group of N tiles.  Every
// one of the workgroup's 2 * 64 * WM * WN threads calls it and RETURNS from it.
// EDGE (round 5 form): the DATA GRADIENT IN GATHER FORM.  MODE_HALO on the N x N grid with the flipped operand pack: the main loop is
// the plain correlation of the HALO-PADDED dz (the forward's own gather).  Taps that stay inside the face, and taps that cross an
// equatorial-equatorial edge (same kernel, same orientation), are terms of the adjoint of DLWP/custom.py:1198-1308 as they are.  A
// tap that crosses any other edge is WRONG: the adjoint wants the same halo cell's dz row times a tap of the NEIGHBOUR's kernel
// (rotated into this face's frame) -- a WEIGHT SUBSTITUTION, done inside the matrix phase by addressing alone (see the consumers).
// Every cell is complete when it is stored: no halo ring, no fix-up launch, one rounding.  Stores go to d0 / d1 like the direct
// mode's interior cells (all cells are interior here), masks (MOUT) included.
template <typename T, int KS, int KC, int MT, int NT, int WM, int WN, int VW, int MODE, bool MASK, bool TAIL8 = false, bool MOUT = false,
          bool EDGE = false>
__device__ __forceinline__ void conv_ws_body(const ConvKParams &P, char *smem, const uint32_t lw, const int G, const int by,
                                             const ConvEdgeArgs &E = ConvEdgeArgs{}) {
    static_assert(!TAIL8 || (VW == 8 && sizeof(T) == 2 && !MASK), "TAIL8: bf16 16-B vectors, forward only");
    static_assert(!MOUT || ((MODE == MODE_ZERO || EDGE) && KS == 3 && sizeof(T) == 2 && !MASK), "MOUT: bf16 data gradient, direct mode");
    static_assert(!EDGE || (MODE == MODE_HALO && KS == 3 && !MASK && !TAIL8 && VW * sizeof(T) == 16), "EDGE: gather-form data gradient");
    constexpr int ES = sizeof(T);
    constexpr int CGW = 32 / ES;                    // channels per MFMA operand group (two 16-B half fragments)
    constexpr int TAPS = KS * KS;
    constexpr int RB = KC * ES + 16;                // LDS bytes per tile pixel
    constexpr int KCG = KC / CGW;
    constexpr int Q = KC / VW;
    constexpr int NTB = NT * WN;
    constexpr int NCT = 64 * WM * WN;               // consumer threads == producer threads
    constexpr int GF4 = TAPS * 64;                  // 16-B entries per (variant, n tile, operand group) of the packed operands
    // LDS weight area of a chunk: [n tile][operand group][tap][64 lanes] 16-B entries -- and, EDGE, behind it the three substitute
    // fragments of the gather form (the weight-id triple of the tile's edge row), [slot][n tile][operand group][64]
    constexpr int WF4M = NTB * KCG * TAPS * 64;     // 16-B entries of the nine taps
    constexpr int XF4 = NTB * KCG * 64;             // ... of one substitute slot
    constexpr int WF4 = EDGE ? WF4M + 3 * XF4 : WF4M;
    constexpr int ITS = 3 * KC / VW;                // input vectors per producer thread per chunk (3*NCT pixels)
    constexpr int ITW = (WF4 + NCT - 1) / NCT;
    static_assert(NCT % Q == 0, "thread -> channel-vector mapping must not depend on the item");
    static_assert(KC % CGW == 0 && KC % VW == 0, "chunk must hold whole operand groups and whole vectors");
    typedef typename VecT<T, VW>::type V;
    const int in_bytes = P.tile_rows_max * P.W2 * RB;
    // LDS: two chunk buffers [input tile | weight fragments], then the epilogue patches.  With 3-4 chunks per tile (P.wstat) the
    // fragments get one RESIDENT area per chunk behind two input-only buffers instead: two buffers alternate between two
    // different chunks' fragments and would re-fetch 18-37 KB every chunk (the 128 -> 64 forward layers ran weight-fetch-bound).
    const int in_step = P.wstat ? in_bytes : in_bytes + WF4 * 16;     // distance between the two input buffers
    const int w_base = P.wstat ? 2 * in_bytes : in_bytes;             // first weight area
    const int w_step = P.wstat ? WF4 * 16 : in_bytes + WF4 * 16;      // distance between weight areas (per chunk / per buffer)
    const int patch_base = P.wstat ? 2 * in_bytes + P.wstat * WF4 * 16 : 2 * (in_bytes + WF4 * 16);

    const int tid = threadIdx.x;
    const bool is_producer = tid >= NCT;
    const int nt0 = by * NTB;
    const int face_pix = P.No * P.No;
    const int strip_pix = P.No * P.Wt;              // cells of one column strip (= face_pix without column blocks)
    const int g0 = P.up0 ? (P.Nin >> 1) : P.Nin;
    const int nchunks = (P.CG + KCG - 1) / KCG;

    // Tiles in (face, band)-major, SAMPLE-minor order; every workgroup owns one contiguous range (the ranges themselves
    // are laid out XCD-aware: neighbouring ranges on the same XCD's L2).  Consecutive tiles of a workgroup are then the same
    // tile position in consecutive samples: gather offsets, validity flags and LDS addresses stay put, only a scalar sample
    // base moves; they are rebuilt at the few (face, band) changes.
    // ILV (data gradient, MT = 3; round 4).  The data gradient is computed on the padded grid: (N + 2)^2 = 2500 pixels per face at
    // N = 48 are 6 tiles of 384 pixels and one of 196 -- and with a wave owning 96 CONSECUTIVE pixels that short tile costs what a
    // full one does (wave 0 has its three M tiles either way), the 42 tiles per sample run as 6 rounds per CU where their pixels
    // are worth 4.9.  With ILV the M tiles of a tile are dealt to the consumer waves round-robin (M tile j -> wave j % WM), a wave
    // runs the MFMA loop and the epilogue instantiated for the number of M tiles it actually has (1, 2 or 3), and the tile list
    // is cut by COST (2 x rounds of M tiles + 1 per tile) instead of by count.  Which wave computes a pixel changes, what is
    // computed for it does not: same bits (tests/test_gpu_parity.py::test_data_gradient_tile_interleave_is_bitwise_neutral).
    // fp32 only: there a tile's time is its MFMA work and the cost split pays (unet2 fp32 step 3.278 -> 3.189 ms, same box); the bf16
    // kernels are bandwidth-bound in steady state -- their launch time did not move with the split (25.2 -> 25.0 us) and the step lost
    // the staggered start's 2-3 us -- so they keep the plain map.
    // (round 6: the fp32 FORWARD pass too, where launch_conv_cfg found tiles of fewer M tiles + the cost split worthwhile -- P.ilv_fwd:
    // 576 tiles of 192 pixels at N = 24 are 2.25 per workgroup = three rounds of three M tiles per wave where tiles of 128 pixels, two
    // M tiles per wave, come out at seven; never with the pooled second output, whose waves own whole pairs of rows)
    constexpr bool ILV_OK = (MODE == MODE_ZERO || (MODE == MODE_HALO && !EDGE && KS == 3)) && MT == 3 && sizeof(T) == 4;
    const bool ilv = ILV_OK && (P.tune & TUNE_CONV_ILV) != 0 && (MODE == MODE_ZERO || P.ilv_fwd != 0);
    // The tile list of this worker: n_my tiles, tile_of(q) = the q-th.  Plain split: one contiguous range of the (face, band)-major,
    // sample-minor list.  Cost split (P.split_gb > 0; the host found it worthwhile): per face the list is (nbl - 1) * B FULL tiles and
    // then B SHORT ones (the last band).  Workers [0, GA) share the full tiles [0, F - split_fb) of the "full list" evenly, workers
    // [GA, G) the remaining split_fb full tiles and ALL short ones -- at N = 48, batch 32: 192 workers x 5 full tiles (cost 35) and
    // 64 workers x (3 full + 3 short) (cost 36) where the plain split has workgroups with 6 full tiles (42).
    const int nbl = P.nblk_face;
    const bool csplit = ilv && P.split_gb > 0;
    // (EDGE, measured: every tile of a polar face runs the substitute steps in all of its waves, a tile of an equatorial face in at
    // most one, and a worker's 4.5 contiguous tiles are all of one kind -- but neither an equatorial + a polar run per worker (a second
    // (face, band) costs a halo-table round trip and a weight reload in the producers' path: +4 us per launch) nor ranges cut by cost
    // (a polar tile = 20 .. 28 sixteenths: +-0 .. +3 us, the longer equatorial lists lose what the polar ones gain) beat the plain split)
    int t_first = 0, t_last = 0;            // plain split
    int f0 = 0, f1 = 0, s0 = 0, s1 = 0;     // cost split: ranges in the full list / in the short list
    if (csplit) {
        const int GB = P.split_gb, GA = G - GB;
        const int Ftot = 6 * (nbl - 1) * P.B, Stot = 6 * P.B, FA = Ftot - P.split_fb;
        if ((int)lw < GA) {
            f0 = (int)(((long)FA * lw) / GA); f1 = (int)(((long)FA * (lw + 1)) / GA);
        } else {
            const int u = (int)lw - GA;
            f0 = FA + (int)(((long)P.split_fb * u) / GB); f1 = FA + (int)(((long)P.split_fb * (u + 1)) / GB);
            s0 = (int)(((long)Stot * u) / GB); s1 = (int)(((long)Stot * (u + 1)) / GB);
        }
    } else {
        t_first = (int)(((long)P.ntiles * lw) / G);
        t_last = (int)(((long)P.ntiles * (lw + 1)) / G);
    }
    const int n_my = csplit ? (f1 - f0) + (s1 - s0) : t_last - t_first;
    auto tile_of = [&](int q) __attribute__((always_inline)) {
        if (!csplit) return t_first + q;
        const int nf = f1 - f0, perF = (nbl - 1) * P.B;
        if (q < nf) { const int i = f0 + q, f = i / perF; return f * nbl * P.B + (i - f * perF); }
        const int i = s0 + (q - nf), f = i / P.B;
        return f * nbl * P.B + perF + (i - f * P.B);
    };
    // TUNE_CONV_STAGGER: the workgroups whose tile list is one shorter than the longest (4 against 5 tiles at N = 48: half of
    // them) start ~1.3 us late -- (tune >> 12) & 63 sleeps of 1024 cycles.  They have a tile's worth of slack, and the chip's 256
    // workgroups no longer hit memory and the matrix cores in lockstep at the start of the kernel (the first tile of a workgroup
    // costs twice a later one).  Measured on the bf16 training step, 0 / 1 / 2 / 3 / 4 / 6 / 8 sleeps: 0.6842 / 0.6795 / 0.6781 /
    // 0.6768 / 0.6775 / 0.6786 / 0.6797 ms; 16 sleeps +24 us.
    // (not with the cost split: there every workgroup's list is as expensive as the next one's, whatever its length)
    if ((P.tune & TUNE_CONV_STAGGER) && !csplit && n_my * G < P.ntiles) {
#pragma unroll 1
        for (int i = 0; i < ((P.tune >> 12) & 63); ++i) __builtin_amdgcn_s_sleep(16);
    }
    struct Geo { int b, f, v, combo, m0, npix, y0, nitems, fcls, x0; };
    auto geo_of = [&](int t) __attribute__((always_inline)) {
        Geo gq;
        gq.combo = P.magicB ? __umulhi((uint32_t)t, P.magicB) : t;                      // t / B   (magic 0 <=> divisor 1)
        gq.b = t - gq.combo * P.B;
        const int fs = P.magicNblk ? __umulhi((uint32_t)gq.combo, P.magicNblk) : gq.combo;      // combo / nblk_face = face * ncol + strip
        const int blk = gq.combo - fs * P.nblk_face;
        gq.f = P.magicNcol ? __umulhi((uint32_t)fs, P.magicNcol) : fs;                  // (magic 0 <=> one strip)
        gq.x0 = (fs - gq.f * P.ncol) * P.Wt;
        gq.v = gq.f < 4 ? 0 : (gq.f == 4 ? 1 : 2);
        gq.m0 = blk * P.pix_per_block;                                                  // (flat in the strip: rows of Wt cells)
        gq.npix = min(P.pix_per_block, strip_pix - gq.m0);
        gq.y0 = __umulhi((uint32_t)gq.m0, P.magicWt);
        const int ylast = __umulhi((uint32_t)(gq.m0 + gq.npix - 1), P.magicWt);
        gq.nitems = (ylast - gq.y0 + KS) * P.W2 * Q;
        gq.fcls = 2 * gq.f + (ylast == P.No - 1 ? 1 : 0);     // EDGE: which weight-id triple (face; the tile holds the face's last row)
        return gq;
    };

    if (is_producer) {
        // =========================================== producers ===========================================
        const int ptid = tid - NCT;
        const int qv = (ptid % Q) * VW;
        const uint4 *wsrc = reinterpret_cast<const uint4 *>(P.wpk);
        if (P.tune & TUNE_CONV_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(2);
        // Tile-invariant slot constants: slot i of this thread is tile pixel (ty, tx) in EVERY tile (only the band's first
        // row y0, the face and the sample change), so the divisions happen once per kernel.  Packed per slot:
        //   bits 4:0 = ty, bit 5 = column valid (MODE_ZERO border), bits 31:6 = OFF + offset of (ty, tx) from the band's
        //   first row on the source grid (halo table row stride M / source row stride Nin).
        constexpr int PADZ = (MODE == MODE_ZERO) ? KS - 1 : 0;
        const int rstride = (MODE == MODE_HALO) ? P.Nin + KS - 1 : P.Nin;
        const int OFF = PADZ * (rstride + 1);
        const int nitems_cap = P.tile_rows_max * P.W2 * Q;
        int slot_c[ITS];
#pragma unroll
        for (int i = 0; i < ITS; ++i) {
            const int e = min(ptid + i * NCT, nitems_cap - 1);
            const int pix = e / Q;
            const int ty = __umulhi((uint32_t)pix, P.magicW2);
            const int tx = pix - ty * P.W2;
            const int vx = tx - PADZ;
            const int xok = (vx >= 0) & (vx < P.Nin);
            slot_c[i] = (((ty - PADZ) * rstride + vx + OFF) << 6) | (xok << 5) | ty;
        }
        // flat source index on the Nin grid (-1 = zero cell) of every item slot of a tile, all loads in flight at once;
        // full-vector kernels (SUP) also keep (as an offset) the pixel index on the nearest-upsampled source's own grid (row r = face*Nin + y
        // of the Nin grid -> row r/2 = face*g0 + y/2 of the Nin/2 grid, Nin = 2*g0 even; column x -> x/2), so that issuing a
        // chunk's loads costs one multiply-add per vector
        constexpr bool SUP = VW * ES == 16 && !MASK;     // (the act' mask's second load stream leaves no registers)
        int sidx[ITS];
        int sup[SUP ? ITS : 1];
        auto upmap = [&](int ii) __attribute__((always_inline)) {
            const int r = __umulhi((uint32_t)ii, P.magicN);
            return (r >> 1) * g0 + ((ii - r * P.Nin) >> 1);
        };
        auto lookup = [&](const Geo &gq) __attribute__((always_inline)) {
            const int base = (gq.f * rstride + gq.y0) * rstride + gq.x0 - OFF;
#pragma unroll
            for (int i = 0; i < ITS; ++i) {
                const int sc = slot_c[i];
                const int a = base + (sc >> 6);
                const bool live = ptid + i * NCT < gq.nitems;
                int v0;
                // (a slot beyond THIS band's rows would index past the band -- for the last band of face 5 past the end of
                // the table: dead slots read entry 0)
                if (MODE == MODE_HALO) v0 = P.table[live ? a : 0];
                else if (MODE == MODE_DIRECT) v0 = a;
                else {
                    const int vy = gq.y0 + (sc & 31) - PADZ;
                    v0 = ((vy >= 0) & (vy < P.Nin) & ((sc >> 5) & 1)) ? a : -1;
                }
                sidx[i] = live ? v0 : -1;
            }
            if constexpr (SUP) {
                // kept as the DIFFERENCE to sidx: `up ? sup[i] : sidx[i]` becomes a select of two stack addresses in LLVM and
                // sends both arrays (and the kernel arguments with them) to scratch memory
#pragma unroll
                for (int i = 0; i < ITS; ++i) sup[i] = sidx[i] >= 0 ? upmap(sidx[i]) - sidx[i] : 0;
            }
        };
        int g = 0;
        // Weight-stationary LDS: the weight fragments of (face variant, chunk) are the same for every tile, and consecutive
        // tiles of a workgroup are the same (face, band) in consecutive samples.  wres[b] = what buffer b's weight area holds;
        // a chunk whose fragments are already there skips their loads and LDS writes.  With one chunk per tile both buffers
        // converge after two tiles, with an even chunk count chunk ch always lands in buffer ch & 1; other counts simply
        // never match.  (Measured need: at 64 -> 64 channels the fragments were 74 of the 107 KB a tile pulled through the
        // CU's load path, which is what bounds these kernels -- ~10 B/clk/CU -- not the matrix cores.)
        int wres[4] = {-1, -1, -1, -1};     // by buffer (g & 1), or by chunk with resident areas (P.wstat)

        // per-thread constants of a chunk: source, channel offset inside it, TAIL8 shift
        auto chunk_src = [&](const Geo &gc, int ch, const T *&sb, int &cstride, int &cs_ld, int &sh, bool &c_ok, bool &up)
            __attribute__((always_inline)) {
            const T *s0b = reinterpret_cast<const T *>(P.src0) + (size_t)gc.b * 6 * g0 * g0 * P.C0;
            const T *s1b = P.C1 > 0 ? reinterpret_cast<const T *>(P.src1) + (size_t)gc.b * 6 * P.Nin * P.Nin * P.C1 : s0b;
            const int c = ch * KC + qv;
            c_ok = c < P.Cin;
            const bool from0 = c < P.C0;
            sb = from0 ? s0b : s1b;
            const int cs = from0 ? c : c - P.C0;              // channel inside the chosen source
            cstride = from0 ? P.C0 : P.C1;
            up = from0 && P.up0;
            cs_ld = cs; sh = 0;
            if constexpr (TAIL8) {
                if (c_ok && cs + 8 > cstride) { sh = (cs + 8 - cstride) >> 1; cs_ld = cstride - 8; }
            }
        };
        // weight fragments of (face variant v, chunk ch) -> registers -> the weight area of LDS buffer b
        auto load_w = [&](int v, int fcls, int ch, uint4 (&wv)[ITW]) __attribute__((always_inline)) {
            // EDGE: substitute slot sl = fragment (variant, tap) = weight id sl of the tile's triple (E.wids [face][top | bottom row][3], -1: none)
            int wid[ITW];
            if constexpr (EDGE) {
#pragma unroll
                for (int u = 0; u < ITW; ++u) {
                    const int sl = max(min(ptid + u * NCT, WF4 - 1) - WF4M, 0) / XF4;
                    wid[u] = (int)E.wids[fcls * 3 + sl];
                }
            }
#pragma unroll
            for (int u = 0; u < ITW; ++u) {
                const int idx = min(ptid + u * NCT, WF4 - 1);
                const bool sub = EDGE && idx >= WF4M;
                const int gg = sub ? ((idx - WF4M) % XF4) / 64 : idx / GF4, w = sub ? (idx & 63) : idx % GF4;
                const int ntl = gg / KCG, cgl = gg % KCG;
                const int ntile = nt0 + ntl, cg = ch * KCG + cgl;
                bool ok = ntile < P.NTtot && cg < P.CG;
                int vv = v, ww = w;
                if constexpr (EDGE) {
                    const int id = max(wid[u], 0);
                    ok = ok && (!sub || wid[u] >= 0);
                    vv = sub ? id / 9 : v;
                    ww = sub ? (id % 9) * 64 + w : w;
                }
                wv[u] = vsel(ok, wsrc[ok ? (((size_t)vv * P.NTtot + ntile) * P.CG + cg) * GF4 + ww : 0]);
            }
        };
        auto store_w = [&](int area, const uint4 (&wv)[ITW]) __attribute__((always_inline)) {
            char *wa = smem + w_base + area * w_step;
#pragma unroll
            for (int u = 0; u < ITW; ++u) {
                const int idx = ptid + u * NCT;
                if (idx < WF4) reinterpret_cast<uint4 *>(wa)[idx] = wv[u];
            }
        };
        // issue(): (rarely) the weight fragments -> LDS buffer g & 1, then every load of the chunk's input tile, back to back
        auto issue = [&](const Geo &gc, int ch, V (&val)[ITS], V (&ymv)[MASK ? ITS : 1], uint32_t &okm) __attribute__((always_inline)) {
            // (EDGE: the area also holds the substitute fragments of the tile's triple -- keyed by face and edge-row class)
            const int wkey = (EDGE ? gc.fcls : gc.v) * 1024 + ch;
            const int area = P.wstat ? ch : (g & 1);
            const bool need_w = !(P.tune & TUNE_CONV_WEIGHTS_STAY) || wres[area] != wkey;
            wres[area] = wkey;
            if (need_w) {
                // uniform and rare (weights stay): the fragments are fetched and written in a block of their own, all loads in
                // flight at once (fetching them four at a time cost 2-3 serial L2 round trips on each workgroup's first tiles:
                // +5 % on the whole training step)
                uint4 wv[ITW];
                load_w(gc.v, gc.fcls, ch, wv);
                store_w(area, wv);
            }
            const T *sb; int cstride, cs_ld, sh; bool c_ok, up;
            chunk_src(gc, ch, sb, cstride, cs_ld, sh, c_ok, up);
            const T *ymb = MASK ? reinterpret_cast<const T *>(P.ymask) + (size_t)gc.b * 6 * g0 * g0 * P.C0 : nullptr;
            okm = 0;
#pragma unroll
            for (int i = 0; i < ITS; ++i) {
                int idx = sidx[i];
                if constexpr (SUP) idx += up ? sup[i] : 0;
                const bool ok = c_ok && idx >= 0;
                int pix = ok ? idx : 0;
                if constexpr (!SUP) pix = up ? upmap(pix) : pix;
                const size_t oo = ok ? (size_t)pix * cstride + cs_ld : 0;
                if constexpr (TAIL8) val[i] = *reinterpret_cast<const uint4_a4 *>(sb + oo);
                else val[i] = *reinterpret_cast<const V *>(sb + oo);
                if (MASK) ymv[i] = *reinterpret_cast<const V *>(ymb + oo);
                okm |= (uint32_t)ok << i;
            }
        };
        // commit(): the loaded vectors -> LDS buffer g & 1, barrier B_g
        auto commit = [&](const Geo &gc, int ch, V (&val)[ITS], V (&ymv)[MASK ? ITS : 1], uint32_t okm) __attribute__((always_inline)) {
            char *buf = smem + (g & 1) * in_step;
            if constexpr (TAIL8) {
                const T *sb; int cstride, cs_ld, sh; bool c_ok, up;
                chunk_src(gc, ch, sb, cstride, cs_ld, sh, c_ok, up);
                if (sh) {
#pragma unroll
                    for (int i = 0; i < ITS; ++i) val[i] = vshl_dwords(val[i], sh);
                }
            }
            // act' mask only after EVERY load has been issued (a use right behind its load makes hipcc wait per load)
            if (MASK) {
#pragma unroll
                for (int i = 0; i < ITS; ++i) vmask(val[i], ymv[i], P.alpha, P.vmax);
            }
#pragma unroll
            for (int i = 0; i < ITS; ++i) {
                const int e = ptid + i * NCT;
                if (e < gc.nitems) *reinterpret_cast<V *>(buf + (e / Q) * RB + qv * ES) = vsel(((okm >> i) & 1u) != 0, val[i]);
            }
            __syncthreads();            // B_g: chunk g is in LDS
            ++g;
        };

        // (Measured negative: issuing the loads of chunk g+1 BEFORE chunk g is written to LDS -- two register sets, one chunk
        // ahead -- hides the 1.2-1.9 k cycles a producer waits for its loads, but its LDS writes then land in the consumers'
        // MFMA phase, whose fragment reads slow down by more than was gained: 2.2 k -> 3.0 k cycles per tile at 32 -> 32
        // channels, 0.936 -> 1.02 ms per training step.)
        int cur_combo = -1;
        V val[ITS], ymv[MASK ? ITS : 1];
        uint32_t okm = 0;
        // (Measured negative: requesting the first two chunks' weight fragments before the first tile's halo-table lookup and
        // its input vectors before the weight stores -- two dependent round trips instead of three at the start of the kernel
        // -- made the training step 3 % SLOWER; the first chunk's fragments alone before the lookup: 1 % slower.  The loads of a
        // wave return in order: whatever is requested ahead of the table entries delays them.)
        for (int q = 0; q < n_my; ++q) {
            const Geo gq = geo_of(tile_of(q));
            if (gq.combo != cur_combo) { lookup(gq); cur_combo = gq.combo; }        // uniform; a few times per workgroup
            for (int ch = 0; ch < nchunks; ++ch) {
                issue(gq, ch, val, ymv, okm);
                commit(gq, ch, val, ymv, okm);
            }
        }
        if (P.tune & TUNE_CONV_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(0);
        return;
    }

    // ============================================= consumers =============================================
    // (wave index as a SCALAR: values derived from it -- n tile, per-wave buffer descriptors -- are then uniform for the compiler too;
    // a descriptor it cannot prove uniform costs a waterfall loop around every store)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    // NOTE: no s_setprio(1) here: a prioritised wave waiting for the busy matrix pipe still wins its SIMD's issue
    // arbitration and starves the co-resident producer wave's address arithmetic.

    // pixel `loc` (0 .. 32 * MT - 1) of this wave -> its index in the tile (row-major, rows of No pixels).  Normally the wave owns
    // 32 * MT consecutive tile pixels.  P.colsplit (pooled second output on faces whose ROW is 32 * MT pixels, tile = 4 rows, 4
    // consumer waves): wave wm owns the half-rows [ (wm & 1) * No/2, + No/2 ) of tile rows 2 * (wm >> 1) and + 1 -- whole
    // 2 x 2 pooling blocks again.
    auto tile_pix = [&](int loc) __attribute__((always_inline)) {
        const int lin = ilv ? (((loc >> 5) * WM + wm) * 32 + (loc & 31)) : wm * MT * 32 + loc;
        const int hN = P.Wt >> 1;
        const int r = loc >= hN ? 1 : 0;
        const int cs = (2 * (wm >> 1) + r) * P.Wt + (wm & 1) * hN + (loc - r * hN);
        return P.colsplit ? cs : lin;
    };
    f32x16 acc[MT][NT];
    constexpr int LPP = 32 * ES / 16;           // epilogue: lanes per pixel on the way out (16 B each): 8 fp32 / 4 bf16
    constexpr int PPP = 64 / LPP;               // pixels per store pass
    constexpr int NPS = 32 / PPP;               // store passes per M tile
    constexpr bool DIRECT = (MODE == MODE_ZERO || EDGE) && KS == 3;   // data gradient: interior cells go straight to the sources
    constexpr int RING = EDGE ? 0 : 1;          // width of the ring around the sources' cells on the output grid (gather form: none)
    int g = 0;
    int abase[MT];
    int cur_combo = -1, cur_v = -1;
    // EDGE: per (face, band).  A lane owns pixel l31 of each of the wave's MT M tiles.  For a border cell the plan record
    // (dlwpcs_dgrad_gather_plan) says which taps of its 3 x 3 window are WRONG (they cross an edge behind which the neighbour's
    // kernel, rotated into this face's frame, applies) and where the cell's true terms read instead: term k of the first / second
    // slot set = the dz row of the halo cell at window position slot[k] / slot[3 + k] times weight id k of the tile's triple.
    // Both are pure ADDRESSING for the matrix phase: a lane whose tap is wrong reads its pixel operand from an LDS address beyond
    // the workgroup's allocation -- the hardware returns zeros there (dlwpcs_lds_oob_probe,
    // tests/test_gpu_dgrad_gather.py::test_lds_reads_beyond_the_allocation_return_zero) -- so the own-kernel MFMA of that tap adds
    // nothing to the lane's pixel column, and three (six in waves that hold a cube corner) more steps per operand group run the
    // substitute fragments (tap slots TAPS + k of the LDS weight area: the producers stage the tile's triple beside the nine taps)
    // against per-lane operand addresses (out of range = no such term).  No cancellation, no scratch accumulator, no per-level
    // loop, no global load in the consumers' path: the corrections are steps of the same software pipeline as the nine taps.
    //   Per lane and M tile, packed (the matrix phase has few registers to spare): es[mt][j >> 1], 16 bits each = the LDS
    //   offset / 16 of term j = set * 3 + k inside the chunk buffer, bit 15 = no such term (<< 4 it lands beyond the allocation);
    //   ewp = the wrong-tap masks, 8 bits per M tile (the centre tap is never wrong: bits 0-3 = taps 0-3, 4-7 = taps 5-8).  Uniform:
    //   e_any (some lane of the wave has a correction in this (face, band): the wave runs mma_chunk_edge), e_two (some lane has a
    //   second-slot term: cube corners).
    uint32_t es[EDGE ? MT : 1][3];
    uint32_t ewp[EDGE ? (MT + 3) / 4 : 1];
    uint32_t e_any = 0, e_two = 0;
    float4 bq[NT][4];
    // pointwise output layer folded into the epilogue (P.head_w): the head's two A fragments (K = 16 channels each) and its bias quads
    // in the D layout, per face variant like the layer's own bias quads
    constexpr bool HEAD_OK = ES == 2 && NT == 1 && WN == 1 && !EDGE && !MOUT && !MASK && MODE != MODE_ZERO;
    uint4 hfrag[HEAD_OK ? 2 : 1];
    float4 hbq[HEAD_OK ? 4 : 1];
    // store pass (nt, mt, ps) of this lane: byte offset of its 16 B inside ONE sample of the destination (ST_SKIP: nothing to
    // store) and, data gradient in direct mode, which destination (2 bits each: 0 = out, 1 = d0, 2 = d1).  Like the LDS
    // addresses they depend on the (face, band) only, not on the sample.
    // (SOFF: kept in registers when there are at most 12 passes; the MT = 5 tilings recompute them per store)
    constexpr bool SOFF = NT * MT * NPS <= 12;
    uint32_t soff[SOFF ? NT : 1][SOFF ? MT : 1][SOFF ? NPS : 1];      // BYTE offsets, ST_SKIP = nothing to store
    uint32_t ssel = 0;
    auto store_off = [&](const Geo &gq, int nt, int mt, int ps, uint32_t &sel) __attribute__((always_inline)) {
        const int px = ps * PPP + lane / LPP, q = lane % LPP;
        const int mm = tile_pix(mt * 32 + px);
        const int c = (nt0 + wn * NT + nt) * 32 + q * (16 / ES);
        const int gm = gq.m0 + mm;
        const int sy = __umulhi((uint32_t)gm, P.magicWt), sx = gm - sy * P.Wt;      // row / column inside the strip
        int off = (gq.f * face_pix + sy * P.No + gq.x0 + sx) * P.Cout + c;
        sel = 0;
        if constexpr (DIRECT) {
            // direct mode: an interior cell of the padded gradient IS cell (oy-1, ox-1) of the source  (never with column blocks)
            const int oy = sy, ox = sx;
            const int Ns = P.No - 2 * RING;
            const bool in0 = c < P.dsplit;
            const bool have = (in0 ? P.d0 : P.d1) != nullptr;
            const int cs = in0 ? c : c - P.dsplit, CS = in0 ? P.dsplit : P.Cout - P.dsplit;
            const bool interior = ((uint32_t)(oy - RING) < (uint32_t)Ns) & ((uint32_t)(ox - RING) < (uint32_t)Ns);
            if (interior && have) {
                off = ((gq.f * Ns + (oy - RING)) * Ns + (ox - RING)) * CS + cs;
                sel = in0 ? 1u : 2u;
            }
        }
        if constexpr (EDGE) { if (P.pool_out != nullptr && !MOUT && c < P.dsplit) return ST_SKIP; }     // (that source gets the 2 x 2 sums only)
        return (mm < gq.npix && c < P.Cout) ? (uint32_t)off * ES : ST_SKIP;
    };

    constexpr int PROW = 32 * ES + 16;          // patch row: one pixel's 32 channels + pad
    // pooling as a second output: one patch per M tile of the wave (all of them are read back once the wave's rows are complete)
    // (MOUT instantiations of the gather form have no registers to spare for it: an upsampled source beside directly masked ones
    // takes the workspace + window-sum launch)
    const bool pooling = MODE != MODE_ZERO && !(EDGE && MOUT) && P.pool_out != nullptr;
    const int pool_cs = EDGE ? P.dsplit : P.Cout;                       // channels of the pooled output
    // EDGE: does n tile nt of this wave belong to the upsampled source (pooled sum only) or to the other one (fine stores only)?
    auto pools_nt = [&](int nt) { return !EDGE || (nt0 + wn * NT + nt) * 32 < P.dsplit; };
    char *const patch0 = smem + patch_base + wave * (32 * PROW) * (pooling ? MT : 1);
    const int patch_step = pooling ? 32 * PROW : 0;
    // ---- pooled second output.  The wave's MT * 32 pixels are whole pairs of tile rows: pooled pixel pp of the wave is the
    // mean of local pixels i00 = 2 * (pp / (No/2)) * No + 2 * (pp % (No/2)), i00 + 1, i00 + No, i00 + No + 1, read from the
    // wave's MT patches (bf16 / fp32 values exactly as stored to `out`), summed like avgpool2_fwd_kernel: (a + b) + (c + d),
    // x 0.25, rounded once -- the same bits as the separate launch.
    constexpr int PITEMS = MT * 8 * LPP;            // 16-B vectors of the wave's pooled pixels
    constexpr int PNP = (PITEMS + 63) / 64;         // passes
    uint32_t plds0[PNP], plds1[PNP], pgo[NT][PNP];  // LDS offsets of i00 / i00 + No, byte offset in one sample of pool_out
    auto pool_setup = [&](const Geo &gq) {
        const int hN = P.No >> 1;
        const int Wl = P.colsplit ? (P.Wt >> 1) : P.Wt, hW = Wl >> 1;   // the wave's pixels as rows of Wl (local, row-major)
#pragma unroll
        for (int ps = 0; ps < PNP; ++ps) {
            const int item = ps * 64 + lane;
            const int pp = min(item / LPP, MT * 8 - 1), q = item % LPP;
            const int prow = pp / hW, pcol = pp - prow * hW;
            const int i00 = 2 * prow * Wl + 2 * pcol, i10 = i00 + Wl;
            plds0[ps] = (uint32_t)((i00 >> 5) * (32 * PROW) + (i00 & 31) * PROW + q * 16);
            plds1[ps] = (uint32_t)((i10 >> 5) * (32 * PROW) + (i10 & 31) * PROW + q * 16);
            const int gm = gq.m0 + tile_pix(i00);
            const int oy = __umulhi((uint32_t)gm, P.magicWt), ox = gq.x0 + (gm - oy * P.Wt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = (nt0 + wn * NT + nt) * 32 + q * (16 / ES);
                const bool ok = item / LPP < MT * 8 && tile_pix(i00) < gq.npix && c < pool_cs;
                pgo[nt][ps] = ok ? (uint32_t)((((gq.f * hN + (oy >> 1)) * hN + (ox >> 1)) * pool_cs + c) * ES) : ST_SKIP;
            }
        }
    };
    // ---- per-tile set-up: LDS addresses and store offsets (rebuilt at (face, band) changes), bias quads (reloaded at face
    // variant changes), accumulators = bias
    auto setup = [&](const Geo &gq) {
        if (gq.combo != cur_combo) {            // uniform; the same for every sample of a combo
            cur_combo = gq.combo;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = tile_pix(mt * 32 + l31);
                int base = 0;
                if (m < gq.npix) {
                    const int gm = gq.m0 + m;
                    const int oy = __umulhi((uint32_t)gm, P.magicWt);
                    const int ox = gm - oy * P.Wt;
                    base = ((oy - gq.y0) * P.W2 + ox) * RB;
                }
                abase[mt] = base + half * 16;
            }
            if constexpr (EDGE) {
                // ONE memory round trip per (face, band): the border-cell records.  Consumer waves do this while the producers' first
                // loads are in flight.
                // (equatorial faces: only the bands that hold the first or the last row have corrections)
                const int ylast = __umulhi((uint32_t)(gq.m0 + gq.npix - 1), P.magicNo);
                const bool maybe = gq.f >= 4 || gq.y0 == 0 || ylast == P.No - 1;
                bool any = false, two = false;
#pragma unroll
                for (int w = 0; w < (MT + 3) / 4; ++w) ewp[w] = 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = tile_pix(mt * 32 + l31);
                    const int gm = gq.m0 + min(m, gq.npix - 1);
                    const int oy = __umulhi((uint32_t)gm, P.magicNo), ox = gm - oy * P.No;
                    const bool border = maybe && m < gq.npix && ((oy == 0) | (oy == P.No - 1) | (ox == 0) | (ox == P.No - 1));
                    const int ord = oy == 0 ? ox : (oy == P.No - 1 ? P.No + ox : 2 * P.No + 2 * (oy - 1) + (ox ? 1 : 0));
                    int4 e0 = make_int4(-1, -1, -1, -1), e1 = make_int4(-1, -1, 0, 0);
                    if (maybe) {
                        const int4 *esrc = reinterpret_cast<const int4 *>(E.src + ((size_t)gq.f * (4 * P.No - 4) + (border ? ord : 0)) * 8);
                        e0 = esrc[0]; e1 = esrc[1];
                    }
                    // window position a * 3 + b -> the halo cell's operand in the chunk buffer (abase = the window's first cell)
                    auto fld = [&](int pos) {
                        const int pa = (max(pos, 0) * 11) >> 5, pb = max(pos, 0) - 3 * pa;
                        return (border && pos >= 0) ? (uint32_t)(abase[mt] + (pa * P.W2 + pb) * RB) >> 4 : 0x8000u;
                    };
                    es[mt][0] = fld(e0.x) | fld(e0.y) << 16;
                    es[mt][1] = fld(e0.z) | fld(e0.w) << 16;
                    es[mt][2] = fld(e1.x) | fld(e1.y) << 16;
                    const uint32_t wrong = border ? (uint32_t)e1.z : 0u;
                    ewp[mt / 4] |= ((wrong & 15u) | ((wrong >> 5) << 4)) << (8 * (mt % 4));
                    any |= border && (wrong != 0 || e0.x >= 0 || e0.y >= 0 || e0.z >= 0 || e0.w >= 0 || e1.x >= 0 || e1.y >= 0);
                    two |= border && (e0.w >= 0 || e1.x >= 0 || e1.y >= 0);
                }
                e_any = __builtin_amdgcn_ballot_w64(any) != 0 ? 1u : 0u;
                e_two = __builtin_amdgcn_ballot_w64(two) != 0 ? 1u : 0u;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(es[mt][0]), "+v"(es[mt][1]), "+v"(es[mt][2]));
#pragma unroll
                for (int w = 0; w < (MT + 3) / 4; ++w) asm volatile("" : "+v"(ewp[w]));
            }
            if constexpr (MODE != MODE_ZERO) {
                if (pooling) pool_setup(gq);
            }
            if constexpr (SOFF) {
                ssel = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int ps = 0; ps < NPS; ++ps) {
                            uint32_t sel;
                            soff[nt][mt][ps] = store_off(gq, nt, mt, ps, sel);
                            ssel |= sel << (2 * ((nt * MT + mt) * NPS + ps));
                        }
            }
        }
        // the bias quads of the face variant (the packed bias vector is zero-padded to NTtot*32 floats, so every quad is
        // readable; an N tile beyond C_out -- NTtot not a multiple of the workgroup's N tiles -- re-reads the last tile's)
        if (!EDGE && gq.v != cur_v) {
            cur_v = gq.v;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int jq = 0; jq < 4; ++jq)
                    bq[nt][jq] = P.bias ? *reinterpret_cast<const float4 *>(P.bias + (size_t)gq.v * P.NTtot * 32 +
                                                                             min(nt0 + wn * NT + nt, P.NTtot - 1) * 32 + 8 * jq + 4 * half)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            // The quads are waited for HERE, inside the (rare) branch: the empty asm redefines the registers, so nothing is
            // pending on them at the join.  Otherwise the accumulator set-up below -- every tile -- sits behind a
            // s_waitcnt vmcnt(0) (a load MAY be in flight; with the previous tile's stores in flight too the counter cannot be
            // split), i.e. behind the acknowledgement of the previous epilogue's stores: ~1000 of a 5750-cycle tile period
            // (s_memtime marks, 32 -> 32 at N = 48).
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int jq = 0; jq < 4; ++jq)
                    asm volatile("" : "+v"(bq[nt][jq].x), "+v"(bq[nt][jq].y), "+v"(bq[nt][jq].z), "+v"(bq[nt][jq].w));
            if constexpr (HEAD_OK) {
                if (P.head_w != nullptr) {      // (uniform)
                    // MFMA step s of the head contracts the channels this lane's accumulator quads 2s and 2s + 1 hold: 16 s + 4 half +
                    // (0..3) and 16 s + 8 + 4 half + (0..3) -- the half-entries [4 half, 4 half + 4) of k groups 2s and 2s + 1 of
                    // output channel l31 in the packed forward fragments
                    const uint2 *hw = reinterpret_cast<const uint2 *>(P.head_w) + (size_t)gq.v * 256 + l31 * 2 + half;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const uint2 lo = hw[(2 * st) * 64], hi = hw[(2 * st + 1) * 64];
                        hfrag[st] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq)
                        hbq[jq] = P.head_b ? *reinterpret_cast<const float4 *>(P.head_b + (size_t)gq.v * 32 + 8 * jq + 4 * half)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int st = 0; st < 2; ++st) asm volatile("" : "+v"(hfrag[st].x), "+v"(hfrag[st].y), "+v"(hfrag[st].z), "+v"(hfrag[st].w));
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) asm volatile("" : "+v"(hbq[jq].x), "+v"(hbq[jq].y), "+v"(hbq[jq].z), "+v"(hbq[jq].w));
                }
            }
        }
        // the accumulators start at the bias (row = output channel in the MFMA's D[co][pixel] layout): no add in the epilogue
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    if constexpr (EDGE) {       // (a data gradient has no bias: no quads to keep)
                        acc[mt][nt][4 * jq] = 0.f; acc[mt][nt][4 * jq + 1] = 0.f; acc[mt][nt][4 * jq + 2] = 0.f; acc[mt][nt][4 * jq + 3] = 0.f;
                        continue;
                    }
                    acc[mt][nt][4 * jq] = bq[nt][jq].x; acc[mt][nt][4 * jq + 1] = bq[nt][jq].y;
                    acc[mt][nt][4 * jq + 2] = bq[nt][jq].z; acc[mt][nt][4 * jq + 3] = bq[nt][jq].w;
                }
    };

    // ---- tile epilogue: bias + activation + stores.  The MFMA ran as D[co][pixel] (weights as the A operand), so in
    // the C/D layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) a lane owns ONE pixel and, per r>>2, FOUR
    // CONSECUTIVE output channels.  Storing those quads directly costs one L2 write request per lane and quad (8 per
    // 64-B line; measured: ~1 request/clk/CU, 3 k of a 5 k-cycle epilogue), so each M tile goes through a wave-private
    // LDS patch [32 pixels][32 channels + 16 B pad] instead: 4 quad writes per lane in, 16 B per lane out with the
    // lanes of a pixel contiguous -> every store instruction writes whole lines.  No fence: the LDS executes one wave's
    // instructions in order; wave_barrier only pins the compiler's schedule (a release fence here waits vmcnt(0),
    // i.e. for the previous stores to land -- that was the cost of the first LDS epilogue).
    //
    // Per (n tile, m tile) pair: 4 x "one quad -> patch" then NPS x "one store pass out of the patch" (epi_slice, called
    // with compile-time-constant i from a fully unrolled loop).
    constexpr int SPP = 4 + NPS;                // slices per (n tile, m tile) pair
    constexpr int NSLICE = NT * MT * SPP;
    // no activation == ReLU(alpha = 1, max = +inf): the epilogue applies the activation unconditionally and stays
    // straight-line (a branch per quad chops it into 5-instruction blocks whose dependent chains cannot interleave).
    // FAST (0 <= alpha <= 1, max >= 0, i.e. every activation of the reference's models and "none"): the activation is
    // min(max(x, alpha*x), max) -- 2.5 VALU instructions per value (v_max_f32, v_min_f32, half a packed multiply) instead of 4.5
    // (canonicalise, min, compare, select, half a multiply); the consumers' epilogue is VALU-issue bound.  (NOT med3(x, alpha*x, max): that is alpha*x,
    // not max, once alpha*x itself exceeds max.)  NaN: keras' ReLU (Azure/train_cs.py:199) propagates it.  v_max_f32(NaN, alpha * NaN)
    // is NaN (both operands), but v_min_f32 is IEEE minNum and would turn it into `max`; gfx950's v_minimum3_f32 is the IEEE-754-2019
    // minimum (NaN if any operand is), same issue cost: a NaN pre-activation leaves the layer as NaN.
    const float e_alpha = P.act == DLWPCS_ACT_LEAKY_CLIP ? P.alpha : 1.f;
    const float e_vmax = P.act == DLWPCS_ACT_LEAKY_CLIP ? P.vmax : __builtin_inff();
    const bool fast_act = e_alpha >= 0.f && e_alpha <= 1.f && e_vmax >= 0.f;
    auto quad = [&](auto fast_tag, const f32x16 &a, int jq) {
        float4 v4 = make_float4(a[4 * jq], a[4 * jq + 1], a[4 * jq + 2], a[4 * jq + 3]);
        if constexpr (MODE == MODE_ZERO) return v4;         // data gradient: never an activation (2.5 instructions per value)
        if constexpr (decltype(fast_tag)::value == 2) return v4;    // no activation: identity (a NaN stays a NaN)
        if constexpr (decltype(fast_tag)::value == 1) {
            // v_max_f32 / v_min_f32 as (pure) asm: fmaxf / fminf -- and v_med3_f32 with an infinite operand, which LLVM folds
            // back into them -- put a canonicalising `v_max x, x` in front of every value that comes out of an accumulator
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            auto lc = [&](float x, float ax) {
                float t, y;
                asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(ax));
                asm("v_minimum3_f32 %0, %1, %2, %2" : "=v"(y) : "v"(t), "v"(e_vmax));
                return y;
            };
            const f32x2 a01 = f32x2{v4.x, v4.y} * e_alpha, a23 = f32x2{v4.z, v4.w} * e_alpha;     // v_pk_mul_f32
            v4.x = lc(v4.x, a01.x); v4.y = lc(v4.y, a01.y); v4.z = lc(v4.z, a23.x); v4.w = lc(v4.w, a23.y);
        } else {
            v4.x = act_leaky_clip(v4.x, e_alpha, e_vmax); v4.y = act_leaky_clip(v4.y, e_alpha, e_vmax);
            v4.z = act_leaky_clip(v4.z, e_alpha, e_vmax); v4.w = act_leaky_clip(v4.w, e_alpha, e_vmax);
        }
        return v4;
    };
    // pointwise output layer folded in: acc[mt][0] <- W_head * bf16(act(acc[mt][0])) + b_head, in place (the epilogue then stores it
    // without an activation).  The activated values are rounded to bf16 as the stand-alone layer would have stored them.
    const bool headed = HEAD_OK && P.head_w != nullptr;
    auto head_apply = [&](auto fast_tag) {
        if constexpr (HEAD_OK) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                uint4 bf[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const float4 lo = quad(fast_tag, acc[mt][0], 2 * st), hi = quad(fast_tag, acc[mt][0], 2 * st + 1);
                    bf[st] = make_uint4(f2bf2(lo.x, lo.y), f2bf2(lo.z, lo.w), f2bf2(hi.x, hi.y), f2bf2(hi.z, hi.w));
                }
                f32x16 a2;
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    a2[4 * jq] = hbq[jq].x; a2[4 * jq + 1] = hbq[jq].y; a2[4 * jq + 2] = hbq[jq].z; a2[4 * jq + 3] = hbq[jq].w;
                }
                frag_mma<T>(a2, hfrag[0], bf[0]);
                frag_mma<T>(a2, hfrag[1], bf[1]);
                acc[mt][0] = a2;
            }
        }
    };
    // per-sample buffer descriptors of the destinations (uniform): write-through stores, see common.h
    auto out_of = [&](const Geo &gq) {
        return make_rsrc(reinterpret_cast<T *>(P.out) + (size_t)gq.b * 6 * face_pix * P.Cout, (uint32_t)(6 * face_pix * P.Cout * ES));
    };
    auto d0_of = [&](const Geo &gq) {
        const int spix = 6 * (P.No - 2 * RING) * (P.No - 2 * RING);
        return make_rsrc(P.d0 ? reinterpret_cast<T *>(P.d0) + (size_t)gq.b * spix * P.dsplit : nullptr, (uint32_t)(spix * P.dsplit * ES));
    };
    auto d1_of = [&](const Geo &gq) {
        const int spix = 6 * (P.No - 2 * RING) * (P.No - 2 * RING), c1 = P.Cout - P.dsplit;
        return make_rsrc(P.d1 ? reinterpret_cast<T *>(P.d1) + (size_t)gq.b * spix * c1 : nullptr, (uint32_t)(spix * c1 * ES));
    };
    // MOUT: the sources' own values at the cells this lane will store to d0 / d1, requested at the start of the tile (they land
    // during its matrix phase) with the store offsets of the (face, band)
    static_assert(!MOUT || SOFF, "MOUT needs the per-band store offsets in registers");
    uint4 ymq[MOUT ? NT : 1][MOUT ? MT : 1][MOUT ? NPS : 1];
    auto mask_load = [&](const Geo &gq) {
        if constexpr (MOUT) {
            const int spix = 6 * (P.No - 2 * RING) * (P.No - 2 * RING), c1 = P.Cout - P.dsplit;
            const rsrc_t r0 = make_rsrc(P.m0 ? reinterpret_cast<const T *>(P.m0) + (size_t)gq.b * spix * P.dsplit : nullptr,
                                        (uint32_t)(spix * P.dsplit * ES));
            const rsrc_t r1 = make_rsrc(P.m1 ? reinterpret_cast<const T *>(P.m1) + (size_t)gq.b * spix * c1 : nullptr,
                                        (uint32_t)(spix * c1 * ES));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int ps = 0; ps < NPS; ++ps) {
                        uint32_t boff = soff[nt][mt][ps];
                        const uint32_t sel = (ssel >> (2 * ((nt * MT + mt) * NPS + ps))) & 3u;
#ifdef DLWPCS_ABL_MASK8     // (side builds only: what would an 8x smaller mask tensor be worth?  wrong numbers, right traffic)
                        if (boff != ST_SKIP) boff = (boff >> 3) & ~15u;
#endif
                        u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r0, sel == 1 ? boff : ST_SKIP, 0, 0);
                        if (P.m1 != nullptr) {      // (uniform; a skip connection that is masked directly -- not in the U-Nets)
                            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r1, sel == 2 ? boff : ST_SKIP, 0, 0);
                            a = a | b;
                        }
                        ymq[nt][mt][ps] = make_uint4(a.x, a.y, a.z, a.w);
                    }
        }
    };
    uint4 ypm[EDGE && !MOUT ? NT : 1][EDGE && !MOUT ? PNP : 1];      // EDGE: act' operands of the pooled sums (the upsampled source's own values)
    auto pool_mask_load = [&](const Geo &gq) {
        if constexpr (EDGE && !MOUT) {
            if (pooling && P.pool_mask != nullptr) {
                const int ppix = 6 * (P.No >> 1) * (P.No >> 1);
                const rsrc_t rm = make_rsrc(reinterpret_cast<const T *>(P.pool_mask) + (size_t)gq.b * ppix * pool_cs, (uint32_t)(ppix * pool_cs * ES));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int ps = 0; ps < PNP; ++ps) {
                        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rm, pgo[nt][ps], 0, 0);
                        ypm[nt][ps] = make_uint4(a.x, a.y, a.z, a.w);
                    }
            }
        }
    };
    const float pool_scale = EDGE ? 1.f : 0.25f;
    auto pool_pass = [&](int nt, rsrc_t d_pool) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PNP; ++ps) {
            const uint4 a = *reinterpret_cast<const uint4 *>(patch0 + plds0[ps]);
            const uint4 b = *reinterpret_cast<const uint4 *>(patch0 + plds0[ps] + PROW);
            const uint4 c = *reinterpret_cast<const uint4 *>(patch0 + plds1[ps]);
            const uint4 d = *reinterpret_cast<const uint4 *>(patch0 + plds1[ps] + PROW);
            uint4 o;
            if constexpr (ES == 4) {
                auto avg = [&](uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
                    return __float_as_uint(((__uint_as_float(x) + __uint_as_float(y)) + (__uint_as_float(z) + __uint_as_float(w))) * pool_scale);
                };
                o = make_uint4(avg(a.x, b.x, c.x, d.x), avg(a.y, b.y, c.y, d.y), avg(a.z, b.z, c.z, d.z), avg(a.w, b.w, c.w, d.w));
            } else {
                auto avg2 = [&](uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
                    const float lo = ((bf_lo(x) + bf_lo(y)) + (bf_lo(z) + bf_lo(w))) * pool_scale;
                    const float hi = ((bf_hi(x) + bf_hi(y)) + (bf_hi(z) + bf_hi(w))) * pool_scale;
                    return f2bf2(lo, hi);
                };
                o = make_uint4(avg2(a.x, b.x, c.x, d.x), avg2(a.y, b.y, c.y, d.y), avg2(a.z, b.z, c.z, d.z), avg2(a.w, b.w, c.w, d.w));
            }
            if constexpr (EDGE && !MOUT && ES == 2) {
                if (P.pool_mask != nullptr) vmask_pk(o, ypm[nt][ps], P.m_alpha, P.m_thr1);      // (uniform)
            }
            bst128(o, d_pool, pgo[nt][ps]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto epi_slice = [&](auto fast_tag, auto ud_tag, int i, const Geo &gq, rsrc_t d_out, rsrc_t d_0, rsrc_t d_1, const rsrc_t (&dsel)[NT],
                         uint32_t on_u, const auto &A) {
        const int pr = i / SPP, k = i % SPP;
        const int nt = pr / MT, mt = pr % MT;
        char *const patch = patch0 + mt * patch_step;
        if (k < 4) {
            const float4 v4 = quad(fast_tag, A[mt][nt], k);
            char *pp = patch + l31 * PROW + (8 * k + 4 * half) * ES;
            if constexpr (ES == 4) *reinterpret_cast<float4 *>(pp) = v4;
            else *reinterpret_cast<uint2 *>(pp) = make_uint2(f2bf2(v4.x, v4.y), f2bf2(v4.z, v4.w));
            if (k == 3) __builtin_amdgcn_wave_barrier();
        } else {
            const int ps = k - 4;
            const int px = ps * PPP + lane / LPP, q = lane % LPP;
            uint4 v = *reinterpret_cast<const uint4 *>(patch + px * PROW + q * 16);
            uint32_t boff, sel;
            if constexpr (SOFF) { boff = soff[nt][mt][ps]; sel = (ssel >> (2 * ((nt * MT + mt) * NPS + ps))) & 3u; }
            else boff = store_off(gq, nt, mt, ps, sel);
            if constexpr (MOUT) {
                // (measured: the whole masking -- these ~11 instructions per bf16 pair and the six 16-B loads per tile -- costs
                // the training step ~14 us over four layers; the loads alone nothing.  Branch-free on purpose: a uniform branch
                // around this block made the step 25 us SLOWER.)
                uint4 vm = v;
                vmask_pk(vm, ymq[nt][mt][ps], P.m_alpha, P.m_thr1);
                // (uniform destinations: which n tiles are masked was decided per tile, as a scalar)
                const bool on = decltype(ud_tag)::value ? ((on_u >> nt) & 1u) != 0 : ((sel == 1 && P.m0 != nullptr) || (sel == 2 && P.m1 != nullptr));
                // component by component: `v = on ? vm : v` on the uint4 becomes a select of two STACK ADDRESSES in LLVM -- both
                // values went to scratch memory and came back through a scratch load behind s_waitcnt vmcnt(0), which also
                // waited for the stores of the previous slice (measured: 43.8 us against 26.1 us unmasked, 32 -> 32 at N = 48)
                v.x = on ? vm.x : v.x; v.y = on ? vm.y : v.y; v.z = on ? vm.z : v.z; v.w = on ? vm.w : v.w;
            }
            if constexpr (decltype(ud_tag)::value) {
                // gather form, whole 32-channel n tiles per source: the destination is the same for every lane of the pass
                bst128(v, dsel[nt], boff);
            } else if constexpr (DIRECT) {
                // three possible destinations: one store each, the lanes of the other two skip
                bst128(v, d_out, sel == 0 ? boff : ST_SKIP);
                bst128(v, d_0, sel == 1 ? boff : ST_SKIP);
                bst128(v, d_1, sel == 2 ? boff : ST_SKIP);
            } else {
                bst128(v, d_out, boff);
            }
            if (ps == NPS - 1) __builtin_amdgcn_wave_barrier();
        }
    };
    auto pool_of = [&](const Geo &gq) {
        const int ppix = 6 * (P.No >> 1) * (P.No >> 1);
        return make_rsrc(pooling ? reinterpret_cast<T *>(P.pool_out) + (size_t)gq.b * ppix * pool_cs : nullptr,
                         (uint32_t)(ppix * pool_cs * ES));
    };
    auto epilogue_lines = [&](const Geo &gq, const auto &A, auto mta_tag) {
        constexpr int MTA = decltype(mta_tag)::value;       // M tiles this wave has in this tile (ILV), else MT
        if constexpr (MOUT) {
            // Every mask value is waited for HERE, before the first store of the epilogue: with loads and stores both in flight
            // hipcc cannot count (gfx9 has one vmcnt for both and they complete out of order), so each later use of a mask
            // register became s_waitcnt vmcnt(0) -- one store round trip per slice, six per tile (measured: 43.8 us against
            // 26.1 us unmasked on the 32 -> 32 layer at N = 48).  The empty asm redefines the registers: nothing pending on them.
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int ps = 0; ps < NPS; ++ps)
                        asm volatile("" : "+v"(ymq[nt][mt][ps].x), "+v"(ymq[nt][mt][ps].y), "+v"(ymq[nt][mt][ps].z), "+v"(ymq[nt][mt][ps].w));
        }
        if constexpr (EDGE && !MOUT) {      // (the pooled sums' mask operands: waited for once, like the MOUT values above)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int ps = 0; ps < PNP; ++ps)
                    asm volatile("" : "+v"(ypm[nt][ps].x), "+v"(ypm[nt][ps].y), "+v"(ypm[nt][ps].z), "+v"(ypm[nt][ps].w));
        }
        const rsrc_t d_out = out_of(gq);
        const rsrc_t d_0 = DIRECT ? d0_of(gq) : d_out, d_1 = DIRECT ? d1_of(gq) : d_out;
        const rsrc_t d_pool = pool_of(gq);
        // (the pooled output is written when the slices of an n tile are through: its patches are complete then)
        // EDGE: every cell goes straight to a source's gradient (or, an upsampled source, to the workspace); when the sources' channel
        // counts are multiples of 32 an n tile belongs to ONE of them and a store pass is one store instruction instead of three
        rsrc_t dsel[NT];
        uint32_t on_u = 0;
        bool ud = false;
        if constexpr (EDGE) {
            ud = (P.dsplit & 31) == 0;
            // (base pointer and size are chosen as scalars and the descriptor built from them: a select of two descriptors becomes
            // control flow around every store)
            const int spix = 6 * P.No * P.No, c1 = P.Cout - P.dsplit;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool in0 = (nt0 + wn * NT + nt) * 32 < P.dsplit;
                const bool have = (in0 ? P.d0 : P.d1) != nullptr;
                const int cs = have ? (in0 ? P.dsplit : c1) : P.Cout;
                T *const base = have ? (in0 ? reinterpret_cast<T *>(P.d0) : reinterpret_cast<T *>(P.d1)) : reinterpret_cast<T *>(P.out);
                dsel[nt] = make_rsrc(base + (size_t)gq.b * spix * cs, (uint32_t)(spix * cs * ES));
                if (have && (in0 ? P.m0 : P.m1) != nullptr) on_u |= 1u << nt;
            }
        }
        auto run = [&](auto tag, auto ud_tag) {
#pragma unroll
            for (int i = 0; i < NSLICE; ++i) {
                // (EDGE, an n tile of the upsampled source: its patches are filled, nothing is stored on the fine grid)
                const bool sum_only = EDGE && i % SPP >= 4 && pooling && pools_nt((i / SPP) / MT);
                if ((i / SPP) % MT < MTA && !sum_only) epi_slice(tag, ud_tag, i, gq, d_out, d_0, d_1, dsel, on_u, A);   // (compile-time: the loop is unrolled)
                if constexpr (MODE != MODE_ZERO) {
                    if ((i + 1) % (MT * SPP) == 0 && pooling && pools_nt(i / (MT * SPP))) pool_pass(i / (MT * SPP), d_pool);
                }
            }
        };
        if constexpr (EDGE) {                                                // (a data gradient never has an activation)
            if (ud) run(std::integral_constant<int, 2>{}, std::true_type{});
            else run(std::integral_constant<int, 2>{}, std::false_type{});
        }
        else if constexpr (DIRECT) run(std::integral_constant<int, 2>{}, std::false_type{});
        else if (P.act != DLWPCS_ACT_LEAKY_CLIP || headed) run(std::integral_constant<int, 2>{}, std::false_type{});
        else if (fast_act) run(std::integral_constant<int, 1>{}, std::false_type{});
        else run(std::integral_constant<int, 0>{}, std::false_type{});
    };
    // fallback (odd channel counts, or no LDS room for the patches): quads / scalars straight from the accumulators
    auto epilogue_plain = [&](const Geo &gq) {
        T *outp = reinterpret_cast<T *>(P.out) + ((size_t)gq.b * 6 + gq.f) * face_pix * P.Cout;      // (never with column blocks)
        const bool wide = (P.Cout & 3) == 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cot = (nt0 + wn * NT + nt) * 32;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = tile_pix(mt * 32 + l31);
                T *dst = outp + (size_t)(gq.m0 + m) * P.Cout + cot + 4 * half;
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    const int co = cot + 8 * jq + 4 * half;
                    const float4 v4 = quad(std::integral_constant<int, 0>{}, acc[mt][nt], jq);
                    if (m >= gq.npix) continue;
                    if (wide) {
                        if (co < P.Cout) {
                            if constexpr (ES == 4) *reinterpret_cast<float4 *>(dst + 8 * jq) = v4;
                            else *reinterpret_cast<uint2 *>(dst + 8 * jq) = make_uint2(f2bf2(v4.x, v4.y), f2bf2(v4.z, v4.w));
                        }
                    } else {
                        const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (co + u < P.Cout) {
                                if constexpr (ES == 4) dst[8 * jq + u] = vs[u];
                                else dst[8 * jq + u] = f2bf(vs[u]);
                            }
                    }
                }
            }
        }
    };

    // ---- one chunk of MFMAs out of LDS buffer g & 1 (between two barriers: fragment reads + MFMAs only)
    // Explicit two-register-set pipeline over the (channel group, tap) steps: the fragments of step s+1 are read from LDS
    // BEFORE the MFMAs of step s are issued (left alone, the compiler reuses one register set and stalls on lgkmcnt after
    // every step).
    constexpr int NSTEP = KCG * TAPS;
    auto mma_chunk = [&](int ch, auto mta_tag) {
        constexpr int MTA = decltype(mta_tag)::value;       // M tiles this wave has in this tile (ILV), else MT
        const char *lds_in = smem + (g & 1) * in_step, *lds_w = smem + w_base + (P.wstat ? ch : (g & 1)) * w_step;
        uint4 fa[2][MT], fb[2][NT];
        auto load_frag = [&](int step, uint4 (&a)[MT], uint4 (&bw)[NT]) {
            const int cgl = step / TAPS, tap = step % TAPS;
            const int dy = tap / KS, dx = tap % KS;
            const int tapoff = (dy * P.W2 + dx) * RB + cgl * 32;
#pragma unroll
            for (int mt = 0; mt < MTA; ++mt) a[mt] = *reinterpret_cast<const uint4 *>(lds_in + abase[mt] + tapoff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bw[nt] = *reinterpret_cast<const uint4 *>(
                    lds_w + ((((wn * NT + nt) * KCG + cgl) * TAPS + tap) * 2 + half) * 512 + l31 * 16);
        };
        load_frag(0, fa[0], fb[0]);
        // (the groups of all sched_group_barriers of the unrolled block form ONE pipeline, filled in program order: without a group
        // of its own for step 0's reads the first DS group takes THEM, the first MFMA group step 0's MFMAs, and so on -- reads and
        // MFMAs of the same step back to back, one fragment set, every step waiting for its own reads (what rounds 1-4 shipped))
        __builtin_amdgcn_sched_group_barrier(0x100, MTA + NT, 0);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            const int cur = step & 1;
            if (step + 1 < NSTEP) load_frag(step + 1, fa[cur ^ 1], fb[cur ^ 1]);
#pragma unroll
            for (int mt = 0; mt < MTA; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) frag_mma<T>(acc[mt][nt], fb[cur][nt], fa[cur][mt]);   // D[co][pixel]
            // (round 3: spreading the reads behind the individual MFMAs -- (MFMA, 2 reads), (MFMA, 1), (MFMA, 1) -- as in the batched
            // weight-gradient kernel measured +-0 here: three MFMAs already hide four reads.  Keeping the 18 weight fragments of a
            // one-chunk layer in 72 VGPRs (3 reads per 3 MFMAs instead of 4; a template variant of its own): the 32 -> 32 data
            // gradient at N = 48 alone 28.5 -> 27.8 us, the whole training step +6 us -- dropped.)
            __builtin_amdgcn_sched_group_barrier(0x100, MTA + NT, 0);                         // DS reads of step s+1 first
            __builtin_amdgcn_sched_group_barrier(0x008, MmaPerFrag<T>::N * MTA * NT, 0);      // then the MFMAs of step s
        }
    };

    // ---- EDGE: the chunk of a wave that holds border cells with corrections (e_any).  The nine tap steps with the wrong taps'
    // lanes reading zeros, then -- same register pipeline -- the substitute steps: (slot set, weight id k, operand group) with the
    // fragment of tap slot TAPS + k as the A operand and per-lane operand addresses.
    auto mma_chunk_edge = [&](const Geo &gq, int ch, auto two_tag) {
      if constexpr (EDGE) {
        constexpr int NSETS = decltype(two_tag)::value ? 2 : 1;
        constexpr int NX = NSETS * 3 * KCG;
        const char *lds_in = smem + (g & 1) * in_step, *lds_w = smem + w_base + (P.wstat ? ch : (g & 1)) * w_step;
        const char *lds_x = lds_w + WF4M * 16;                          // the substitute fragments
        // (the packed records are redefined per chunk -- empty asm, in place -- so that their decoding stays INSIDE the chunk: hoisted
        // out of the tile loop, the 24 masked bases and 18 operand addresses are 40 more live registers, i.e. spills -- and a spill
        // reloaded in the epilogue is a scratch load behind s_waitcnt vmcnt(0), i.e. behind the acknowledgement of every store
        // issued so far: measured 39 -> 65 us on the 32 -> 32 layer at N = 48)
        uint32_t (&rs)[MT][3] = es;
        uint32_t (&rw)[(MT + 3) / 4] = ewp;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(rs[mt][0]), "+v"(rs[mt][1]), "+v"(rs[mt][2]));
#pragma unroll
        for (int w = 0; w < (MT + 3) / 4; ++w) asm volatile("" : "+v"(rw[w]));
        uint4 fa[2][MT], fb[2][NT];
        auto load_frag = [&](int step, uint4 (&a)[MT], uint4 (&bw)[NT]) {
            if (step < NSTEP) {
                const int cgl = step / TAPS, tap = step % TAPS;
                const int dy = tap / KS, dx = tap % KS;
                const int tapoff = (dy * P.W2 + dx) * RB + cgl * 32;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // a wrong tap's lanes read 8 MB up: zeros (the window's centre is the cell itself: never wrong)
                    int base = abase[mt];
                    if (tap != TAPS / 2) base += (int)((rw[mt / 4] >> (8 * (mt % 4) + (tap < TAPS / 2 ? tap : tap - 1))) & 1u) << 23;
                    a[mt] = *reinterpret_cast<const uint4 *>(lds_in + base + tapoff);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bw[nt] = *reinterpret_cast<const uint4 *>(
                        lds_w + ((((wn * NT + nt) * KCG + cgl) * TAPS + tap) * 2 + half) * 512 + l31 * 16);
            } else {
                const int x = step - NSTEP, j = x / KCG, cgl = x % KCG;       // j = set * 3 + k
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint32_t f16 = (j & 1) ? rs[mt][j >> 1] >> 16 : rs[mt][j >> 1] & 0xffffu;
                    a[mt] = *reinterpret_cast<const uint4 *>(lds_in + (int)(f16 << 4) + cgl * 32);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)         // the substitute fragment: tap slot TAPS + k of the weight area
                    bw[nt] = *reinterpret_cast<const uint4 *>(
                        lds_x + (((j % 3) * NTB * KCG + (wn * NT + nt) * KCG + cgl) * 2 + half) * 512 + l31 * 16);
            }
        };
        load_frag(0, fa[0], fb[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);       // (step 0's reads: a group of their own, see mma_chunk)
#pragma unroll
        for (int step = 0; step < NSTEP + NX; ++step) {
            const int cur = step & 1;
            if (step + 1 < NSTEP + NX) load_frag(step + 1, fa[cur ^ 1], fb[cur ^ 1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) frag_mma<T>(acc[mt][nt], fb[cur][nt], fa[cur][mt]);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MmaPerFrag<T>::N * MT * NT, 0);
        }
      }
    };

    const bool lines = P.patches && (P.Cout % (16 / ES)) == 0;
    for (int q = 0; q < n_my; ++q) {
        const Geo gq = geo_of(tile_of(q));
        setup(gq);
        mask_load(gq);
        pool_mask_load(gq);
        // (ILV: how many of its M tiles this wave has in this tile: M tile j of the tile's ceil(npix / 32) belongs to wave j % WM)
        const int my_mt = ilv ? max(0, min(MT, (((gq.npix + 31) >> 5) - wm + WM - 1) / WM)) : MT;
        for (int ch = 0; ch < nchunks; ++ch, ++g) {
            // B_g: chunk g has been written by the producers.  A RAW barrier behind an explicit LDS wait: __syncthreads() makes
            // hipcc drain vmcnt(0) first, i.e. wait for the previous tile's epilogue stores to be acknowledged and -- MOUT -- for
            // the mask values requested a moment ago (measured on the 32 -> 32 data gradient at N = 48: 43.8 us masked against
            // 26.1 plain, nearly all of it this wait).  The consumers only owe the producers their LDS reads.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (ILV_OK) {
                if (my_mt >= MT) mma_chunk(ch, std::integral_constant<int, MT>{});
                else if (my_mt == 2) mma_chunk(ch, std::integral_constant<int, 2>{});
                else if (my_mt == 1) mma_chunk(ch, std::integral_constant<int, 1>{});
            } else if constexpr (EDGE) {
                // (uniform per (face, band) and wave; most waves of the equatorial faces have no correction at all)
                if (!e_any) mma_chunk(ch, std::integral_constant<int, MT>{});
                else if (!e_two) mma_chunk_edge(gq, ch, std::false_type{});
                else mma_chunk_edge(gq, ch, std::true_type{});
            } else {
                mma_chunk(ch, std::integral_constant<int, MT>{});
            }
        }
        if constexpr (HEAD_OK) {
            if (headed) {       // (uniform; launch_conv_cfg grants it with the line-store epilogue only)
                if (fast_act) head_apply(std::integral_constant<int, 1>{});
                else head_apply(std::integral_constant<int, 0>{});
            }
        }
        if (lines) {
            if constexpr (ILV_OK) {
                if (my_mt >= MT) epilogue_lines(gq, acc, std::integral_constant<int, MT>{});
                else if (my_mt == 2) epilogue_lines(gq, acc, std::integral_constant<int, 2>{});
                else if (my_mt == 1) epilogue_lines(gq, acc, std::integral_constant<int, 1>{});
            } else {
                epilogue_lines(gq, acc, std::integral_constant<int, MT>{});
            }
        } else {
            epilogue_plain(gq);
        }
    }
}

// the per-layer launch: one workgroup per CU, workers laid out XCD-aware over the tile list
template <typename T, int KS, int KC, int MT, int NT, int WM, int WN, int VW, int MODE, bool MASK, bool TAIL8 = false, bool MOUT = false,
          bool EDGE = false>
__global__ void __launch_bounds__(2 * 64 * WM * WN) conv_mfma_ws_kernel(const ConvKParams P, const ConvEdgeArgs E) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_ws_body<T, KS, KC, MT, NT, WM, WN, VW, MODE, MASK, TAIL8, MOUT, EDGE>(P, smem, xcd_remap(blockIdx.x, gridDim.x), (int)gridDim.x,
                                                                             (int)blockIdx.y, E);
}

}  // namespace dlwpcs
