// Batched weight gradients of a whole backward pass (gfx950, bf16 mode): ONE persistent launch for every layer.
//
// Per layer the weight gradient is  dW[tap][ci][co] = sum over (sample, face of the weight group, pixel)
//     Xpad[pixel + tap][ci] * dZ[pixel][co]                                   (DLWP/custom.py:921-1002, transposed)
// and it depends on nothing but the layer's saved input and the gradient dZ w.r.t. its pre-activation output: it is off
// the critical path of the backward pass.  Launched layer by layer (conv_mfma.hip, wgrad_bf16_kernel) each of the 11
// launches of the DLWP-CS U-Net splits ~4.5 work items per CU, i.e. spends more time filling and draining than streaming
// (measured on MI355X, batch 32: 339 us for the 11 launches of which 146 us do not scale with the batch), and leaves 256
// full-size fp32 partial sums per layer behind (153 MB per step for a 2.7 MB gradient).  Here all layers form ONE work list:
//   * a SEGMENT = (layer, ci-tile group, co-tile group, face class, contiguous range of work items); the host plan
//     (dlwpcs_wgrad_batch_plan) cuts the list into one chain of segments per CU with equal estimated cost, so a layer gets
//     CUs in proportion to its work (a dozen instead of 256 per layer: ~20x fewer partial sums) and every CU streams ~45
//     items instead of 4.5;
//   * the kernel walks its chain: per segment the main loop of wgrad_bf16_kernel (4 producer waves: global -> registers ->
//     LDS with the halo gather / upsample / concat resolved in the address; 4 consumer waves: ds_read_b64_tr_b16 + bf16 MFMA,
//     K = 16 pixels), then one partial sum per segment;
//   * a worker covers up to 64 input x 64 output channels (CT x NT tiles of 32 x 32, one consumer wave each): X and dZ are
//     then read once per layer where the per-layer kernel re-read X per 32-wide output tile;
//   * dZ arrives PRE-MASKED (the producer of the gradient applied act'(y)): no second load stream, no mask arithmetic;
//   * wb_reduce_kernel adds the few partial sums per (layer, group, class) in a fixed order (no atomics: bitwise
//     reproducible), applies the weight-group map / north-pole row reversal and accumulates into the fp32 gradients.
// Tensor addresses travel BY VALUE in the kernel arguments (a captured hipGraph keeps them); the plan holds geometry only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "mfma_common.h"

namespace dlwpcs {
static const int g_wb_tags[] = {prof_register_tag("wgrad_batch_kernel"), prof_register_tag("wb_reduce_kernel"),
                                prof_register_tag("wb_reduce_kernel(apply)")};


constexpr int WB_MAX_LAYERS = DLWPCS_WGRAD_BATCH_MAX;
constexpr uint32_t WB_MAGIC = 0x57424c31u;       // 'WBL1'

// instantiations of the segment body: (KS, XV, QX, CT, NT, DV)
enum {
    WB_V_3_8_22 = 0,    // 3x3, 16-B X vectors, 64 ci x 64 co per worker
    WB_V_3_8_21,        // 64 ci x 32 co
    WB_V_3_8_12,        // 32 ci x 64 co
    WB_V_3_8_11,        // 32 ci x 32 co
    WB_V_3_2_8,         // 4-B X vectors, <= 16 input channels (the 14-channel network input)
    WB_V_3_2_16,        // 4-B X vectors, <= 32 input channels (26 = 13 x 2)
    WB_V_1_8_11,        // 1x1, C_out % 8 == 0
    WB_V_1_8_D2,        // 1x1, even C_out <= 16 (the 14-channel head): 4-B dZ vectors
    WB_V_F32_3,         // exact-fp32 mode (round 4): 3x3, 16-B vectors of 4 channels, 32 ci x 32 co per worker, fp32 MFMA (K = 2 pixels)
    WB_V_F32_1,         // ... 1x1
    WB_V_F32_3_X2,      // ... 3x3, even input channel counts that are no multiples of 4 (the 14-channel input): 8-B X loads
    WB_V_F32_3_H2,      // ... 3x3, the output channel count too: 8-B X and dZ loads
    WB_V_F32_1_D2,      // ... 1x1, even output channel count that is no multiple of 4 (the 14-channel head): 8-B dZ loads
    WB_NVARIANTS
};

struct WbLayer {                    // 36 ints
    int32_t B, Nin, No, C0, C1, Cin, Cout, up0;
    int32_t halo, KS, W2, tile_rows_max, pix, nbands, pix_cap, variant;
    uint32_t magicW2, magicNo, magicN, magicB, magicNb;
    int32_t CT, NT, ncit, ncot, cin_logical, flip, want_bias, group_base, has_np, slot_floats, mask;
    float alpha, vmax;              // mask != 0: dz = dy * act'(y) is formed by the producers (ReLU(alpha, vmax))
    int32_t pad0, pad1;
};
static_assert(sizeof(WbLayer) == 144, "WbLayer layout");

struct WbSeg {                      // 8 ints
    int32_t layer, cls, cit, cot, t_first, t_last;
    uint32_t slot_off;              // floats from the workspace base
    int32_t bias;                   // this segment also sums dZ columns (bias gradient)
};

struct WbGroup { uint32_t off, stride; int32_t count, pad; };      // partial slots of one (layer, cit, cot, class)

struct WbHeader {
    uint32_t magic, n_layers, n_segments, n_workers, n_groups, lds_bytes;
    uint32_t off_layers, off_segs, off_groups, total_bytes;
    uint64_t ws_floats;
    uint32_t red_first[WB_MAX_LAYERS + 1];      // first reduction workgroup of every layer
    uint32_t seg_start[257];                     // worker w runs segments [seg_start[w], seg_start[w + 1])
};

struct WbPtrs {
    const void *src0[WB_MAX_LAYERS], *src1[WB_MAX_LAYERS], *dz[WB_MAX_LAYERS], *y[WB_MAX_LAYERS];
    const int32_t *table[WB_MAX_LAYERS];
};
// optimizer fused into the reduction (dlwpcs_wgrad_batch_adam): every reduced gradient element is consumed on the spot
struct WbAdam {
    float *p, *g, *m, *v;           // flat buffers; the items' dw_* / db_* point INTO g, the other three share its offsets
    const int32_t *state;           // {t, ticket}: the weight-gradient launch before this one has already incremented t
    const float *hyper;             // {lr, beta1, beta2, eps, grad_scale}
    int32_t on;
    // apply-only form (dlwpcs_wgrad_batch_apply, the data-parallel step): the partial sums were reduced into g by an earlier
    // launch and g has been summed over the ranks since; this launch adds nothing, it consumes g -- and it moves the step counter
    // on itself: t = state[0] + 1, the last workgroup to finish (ticket state[1]) stores it
    int32_t apply_only;
};
// what the reduction needs of a layer, by value in the kernel arguments (its dependent chain of memory round trips is what
// the reduction's ~20 us are made of: plan header -> layer record -> group record -> partial sums -> optimizer state)
struct WbRedLayer { int32_t KS, Cin, Cout, want_bias, TC, TN, ncot, group_base, flip; };
// optimizer fused in + weight packing fused in: where the layer's packed bf16 operands (dlwpcs_pack_batch outputs) live; the
// reduction writes the updated values into them in place, so the next pass starts without a packing launch
struct WbRedPack { bf16_t *wf, *wb; float *bp; int32_t CGf, NTf, CGb, NTb, f32; };     // (f32: fp32 operands, 4 values per 16-B entry)
struct WbPackArgs { WbRedPack l[WB_MAX_LAYERS]; };
struct WbRedPtrs {
    float *dw_eq[WB_MAX_LAYERS], *dw_pol[WB_MAX_LAYERS], *dw_np[WB_MAX_LAYERS];
    float *db_eq[WB_MAX_LAYERS], *db_pol[WB_MAX_LAYERS], *db_np[WB_MAX_LAYERS];
    uint32_t first[WB_MAX_LAYERS + 1];
    uint32_t live;                               // bit l: layer l is reduced by this launch
    uint32_t adam;                               // bit l: ... and this launch is the LAST one that adds to its destination: the fused
                                                 // optimizer consumes it here (a layer applied twice: its first item only accumulates)
    uint32_t n_layers, off_groups;
    WbRedLayer lay[WB_MAX_LAYERS];
};

// ------------------------------------------------------------------------------------------------------------------
// One segment.  Called by all 512 threads of the workgroup; threads 256.. are the producers.  Both halves execute
// n_items + 3 workgroup barriers.
// ------------------------------------------------------------------------------------------------------------------
// Consumer schedule of one K slab: the 2 * TAPS + 2 transpose reads of the NEXT slab interleaved with the TAPS MFMAs of this
// one (an in-order wave can issue ~5 other instructions in the shadow of a 32x32x16 MFMA; all reads first, then all MFMAs --
// the schedule of wgrad_bf16_kernel -- measured ~600 cycles per slab of 9 MFMAs, twice the matrix time).
#ifndef DLWPCS_WB_SCHED
#define DLWPCS_WB_SCHED 1
#endif
template <int N> __device__ __forceinline__ void wb_sched_pairs() {
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // two LDS transpose reads in its shadow
        wb_sched_pairs<N - 1>();
    }
}
#if DLWPCS_WB_SCHED == 1
#define WB_SCHED() do {                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      /* the dZ fragment of the next slab */ \
        wb_sched_pairs<TAPS>();                                                           \
    } while (0)
#else
#define WB_SCHED() do {                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * TAPS + 2, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);                             \
    } while (0)
#endif

// a plan record -> scalar registers (the plan is the same for every lane; read through a reference the fields would be
// re-fetched with vector loads inside the loops, behind vmcnt(0) waits)
template <typename T> __device__ __forceinline__ T load_uniform(const T &g) {
    T out;
    const int *s = reinterpret_cast<const int *>(&g);
    int *d = reinterpret_cast<int *>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d[i] = __builtin_amdgcn_readfirstlane(s[i]);
    return out;
}

// (noinline on purpose: the kernel carries 22 segment bodies; forced inline -- even of the 8 plain bf16 bodies alone -- hipcc spills
// 266 VGPRs and the launch takes 237 instead of 150 us, EXPERIMENTS.md)
#define WB_SEG_ATTR __attribute__((noinline))
#ifndef DLWPCS_WB_PRODUCER_PRIO
#define DLWPCS_WB_PRODUCER_PRIO 0
#endif
// (side builds only, -DDLWPCS_WB_TL=1, tools/wb_timeline.py: s_memtime marks of the first workers' first producer / consumer wave into
// a library-owned buffer -- [worker][kind][TL_MAX] words (time << 4 | tag), cursor per (worker, kind) behind them)
#ifdef DLWPCS_WB_TL
// phase SUMS in registers, one store per segment (per-mark stores were FLAT stores: hipcc answered them with vmcnt(0) in front of the
// next LDS write, and the "LDS write" phase of the first timeline was really a wait for every load in flight)
constexpr int TL_WORKERS = 256, TL_MAX = 8;
#define WB_TL_DECL(kind_) \
    const bool tl_on = dbg != nullptr && (threadIdx.x & 255) == 0; \
    long long *tl_base = dbg + ((size_t)blockIdx.x * 2 + (kind_)) * TL_MAX; \
    long long tl_a[TL_MAX] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    long long tl_prev = (long long)__builtin_amdgcn_s_memtime();
#define WB_TL(bucket_) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); tl_a[bucket_] += t_ - tl_prev; tl_prev = t_; } while (0)
#define WB_TL_END() do { if (tl_on) { _Pragma("unroll") for (int q_ = 0; q_ < TL_MAX; ++q_) tl_base[q_] += tl_a[q_]; } } while (0)
#else
#define WB_TL_DECL(kind_)
#define WB_TL(tag_)
#define WB_TL_END()
#endif
template <int KS, int XV, int QX, int CT, int NT, int DV, bool MASK = false>
__device__ WB_SEG_ATTR void wb_segment(const WbLayer &Lg, const WbSeg &sgg, const void *src0, const void *src1,
                                           const void *dzp, const void *yp, const int32_t *table, float *ws, char *smem,
                                           long long *dbg) {
    const WbLayer L = load_uniform(Lg);
    const WbSeg sg = load_uniform(sgg);
    constexpr int QD = DV == 8 ? 4 : 8;     // dZ vectors per pixel and 32-channel plane
    constexpr int QDT = QD * NT;            // dZ vectors staged per pixel
    typedef typename VecT<bf16_t, DV>::type DVec;
    typedef typename VecT<bf16_t, XV>::type XVec;
    constexpr int TAPS = KS * KS;
    constexpr int PB = 64;                  // LDS bytes per pixel and plane: 32 channels bf16
    constexpr int NCT = 256;                // consumer threads == producer threads
    constexpr int QXT = QX * CT;            // X vectors staged per tile pixel
    constexpr int IT_X = XV == 8 ? (CT == 2 ? 10 : 8) : 16;      // tile-pixel capacity IT_X * 256 / QXT: 320 | 512 | 512 / 256
    constexpr int CAP_PIX = (CT == 2 || NT == 2) ? 192 : 384;    // item pixels the producers' dZ registers hold
    constexpr int IT_DY = CAP_PIX * QDT / NCT;
    constexpr int NPH = 4 / (CT * NT);      // consumer waves sharing the K slabs of one (ci tile, co tile)
    constexpr int RSLOTS = TAPS - TAPS / NPH;
    constexpr int RED_FLOATS = 4 * RSLOTS * 1024;
    static_assert(CT * NT * NPH == 4, "four consumer waves");
    static_assert(CT == 1 || (XV == 8 && QX == 4), "two ci tiles per worker need full 16-B X vectors");
    static_assert(NCT % QXT == 0 && NCT % QDT == 0, "thread -> channel-vector maps must not depend on the item");

    const int pix_cap = L.pix_cap;
    // LDS planes are as large as the producers' SLOTS (not as the layer's tile): every staged vector has a home, so the producers'
    // LDS writes are unconditional stores at compile-time offsets -- no bounds test, no exec mask, no address arithmetic per write
    // (time-neutral against bounds-tested writes -- tools/wb_timeline.py: the writes + bias sums of an item take a producer wave ~1800
    // cycles either way, a fifth of it the bias sums -- but the producers' code is a third shorter)
    constexpr int XCAP_PIX = (IT_X / CT) * NCT / QX;        // tile pixels a plane can hold
    constexpr int plane_bytes = XCAP_PIX * PB;
    constexpr int x_bytes = CT * plane_bytes;
    constexpr int dzplane_bytes = CAP_PIX * PB;
    constexpr int buf_bytes = x_bytes + NT * dzplane_bytes;

    const int nfaces = sg.cls == 0 ? 4 : 1, fbase = sg.cls == 0 ? 0 : (sg.cls == 1 ? 4 : 5);
    const int n_my = sg.t_last - sg.t_first;
    const int face_pix = L.No * L.No;
    const int tid = threadIdx.x;
    (void)nfaces;

    struct Item { int b, f, combo, m0, npix, y0, nitems; };
    auto item_of = [&](int k) {
        Item it;
        const int t = sg.t_first + max(min(k, n_my - 1), 0);
        it.combo = L.magicB ? __umulhi((uint32_t)t, L.magicB) : t;                       // t / B  (magic 0 <=> divisor 1)
        it.b = t - it.combo * L.B;
        const int fl = L.magicNb ? __umulhi((uint32_t)it.combo, L.magicNb) : it.combo;   // combo / nbands
        const int band = it.combo - fl * L.nbands;
        it.f = fbase + fl;
        it.m0 = band * L.pix;
        it.npix = min(L.pix, face_pix - it.m0);
        it.y0 = __umulhi((uint32_t)it.m0, L.magicNo);
        const int ylast = __umulhi((uint32_t)(it.m0 + it.npix - 1), L.magicNo);
        it.nitems = (ylast - it.y0 + KS) * L.W2 * QXT;
        return it;
    };
    // (the workspace pointer arrives in vector registers -- a noinline call -- and a buffer descriptor built from it is not
    // uniform for the compiler: every store of the epilogue sat in a waterfall loop)
    ws = reinterpret_cast<float *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)ws >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ws));
    float *slot = ws + sg.slot_off;

    if (tid >= NCT) {
        // =========================================== producers ===========================================
        const int ptid = tid - NCT;
        // (s_setprio for the producer waves, as the per-layer kernels and the convolution do it: measured here, priority 2 makes the
        // launch 3 % SLOWER -- the consumers beside them compute 15 % longer, the producers' phases do not shrink.  Default 0.)
        if (DLWPCS_WB_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(DLWPCS_WB_PRODUCER_PRIO);
        WB_TL_DECL(0)
        const int qd = ptid % QDT;                              // this thread's dZ vector inside a pixel: fixed
        const int g0 = L.up0 ? (L.Nin >> 1) : L.Nin;
        const int M = L.Nin + KS - 1;
        const int co = sg.cot * 32 * NT + (qd / QD) * 32 + (qd % QD) * DV;
        const bool co_ok = co < L.Cout;
        const bool want_bias = sg.bias != 0;
        const uint32_t mthr1 = (uint32_t)L.pad0;                // bf16_mask_threshold(vmax) (MASK)
        float bsum[DV];
#pragma unroll
        for (int u = 0; u < DV; ++u) bsum[u] = 0.f;
        // ---- producer pipeline (round 5).  What rounds 3-4 shipped kept ONE item's loads in flight per CU: FLAT loads count in
        // lgkmcnt too, so the LDS wait in front of the barrier also waited for the loads of the next item that had been issued a
        // moment before, and hipcc put vmcnt(0) in front of every issue (read off the ISA; the per-item cost the plan's model was
        // fitted to -- ~3300 cycles + bytes at 23 B/clk -- IS one exposed memory round trip per item).  Now: BUFFER loads (vmcnt
        // only; a lane without work passes an offset beyond num_records and gets zeros: no select, no clamp, no 64-bit address
        // arithmetic per load; the sample / band base lives in the scalar descriptor), two register stages, and the loads of item
        // k + 2 are issued right AFTER item k has gone to LDS: two items in flight, each with a whole period to land.
        // (Measured and dropped, EXPERIMENTS.md: a third stage -- 146.8 against 145.4 us, depth no longer limits the launch; BAND-minor
        // item order with the halo-table entries of every item fetched one issue ahead -- FETCH_SIZE -9 % (vertically adjacent bands
        // share two halo rows through the L2) but 175-186 against 141 us: the per-item offset arithmetic, eight more loads and ~30
        // more registers per thread -- spills in the widest variants -- cost more than the bytes give.)
        auto uni_ptr = [](const void *q) {      // (pointer arguments of a noinline call arrive in vector registers)
            return reinterpret_cast<const char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)q >> 32)) << 32) |
                                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)q));
        };
        const char *const x0_u = uni_ptr(src0), *const x1_u = uni_ptr(src1);
        const char *const dz_u = uni_ptr(dzp), *const y_u = uni_ptr(yp);
        const int32_t *const table_u = reinterpret_cast<const int32_t *>(uni_ptr(table));
        const uint32_t s0_bytes = (uint32_t)((size_t)6 * g0 * g0 * L.C0 * 2), s1_bytes = (uint32_t)((size_t)6 * L.Nin * L.Nin * L.C1 * 2);
        // Thread -> element map of THIS pipeline: load i of a thread serves channel plane i % CT (32 channels each), element
        // ptid + (i / CT) * 256 of that plane's QX vectors per tile pixel -- the plane, and with it the SOURCE of the concatenation
        // (which buffer descriptor), is the same for every lane of a load instruction (the first pipeline gave every thread one fixed
        // plane: a worker whose 64 channels span both sources -- 32 up-sampled + 32 skip channels -- would need two loads per vector).
        constexpr int ITP = IT_X / CT;          // loads per thread and plane
        static_assert(IT_X % CT == 0 && NCT % QX == 0, "plane-major load map");
        const int qx = ptid % QX;               // this thread's vector inside a pixel of a plane: fixed
        // per plane: which source (uniform), stride / channel offset / upsampling of that source.  Plain records picked with
        // compile-time plane indices (small arrays captured by the lambdas below went to scratch memory: a stack load and a
        // vmcnt(0) in front of every use).
        // (a worker with a plane that holds channels of BOTH sources -- C0 no multiple of 32 -- keeps per-thread sources: two loads per
        // vector, one of them with the skip offset; no U-Net layer)
        struct Plane { int cs, stride; bool all0, up, ok, f0; };
        bool straddle = false;
        auto mkplane = [&](int pl) {
            Plane q;
            const int c_lo = sg.cit * 32 * CT + pl * 32, c_hi = min(c_lo + 32, L.Cin);
            const int cxp = c_lo + qx * XV;                     // (per thread)
            q.all0 = c_hi <= L.C0;
            straddle |= c_lo < L.C0 && c_hi > L.C0;
            q.ok = cxp < L.Cin;
            q.f0 = cxp < L.C0;                                  // this thread's source (what counts in a straddling plane)
            q.cs = q.f0 ? cxp : cxp - L.C0;
            q.stride = q.f0 ? L.C0 : L.C1;
            q.up = q.f0 && L.up0;
            return q;
        };
        const Plane P0 = mkplane(0), P1 = mkplane(CT - 1);
        auto nit_plane = [&](const Item &it) { return it.nitems / CT; };
        uint32_t xoff[IT_X];                    // BYTE offsets inside one sample of the load's source; 0xffffffff: nothing to load
        int cur_combo = -1;
        // (the table entries of a rebuild are fetched with ALL loads in flight at once: a uniform branch per entry -- `if (L.halo)` --
        // made them eight serialised memory round trips, vmcnt(0) behind each)
        const rsrc_t rt = make_rsrc(table_u, L.halo ? (uint32_t)(6 * M * M * 4) : 0u);
        auto rebuild = [&](const Item &it) {
            const int nitp = nit_plane(it);
            int tbl[ITP];
#pragma unroll
            for (int j = 0; j < ITP; ++j) {
                const int e = min(ptid + j * NCT, nitp - 1);
                const int pix = e / QX;
                const int ty = __umulhi((uint32_t)pix, L.magicW2);
                const int tx = pix - ty * L.W2;
                tbl[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(rt, (uint32_t)(((it.f * M + it.y0 + ty) * M + tx) * 4), 0, 0);
            }
#pragma unroll
            for (int j = 0; j < ITP; ++j) {
                const int e = min(ptid + j * NCT, nitp - 1);
                const int pix = e / QX;
                const int ty = __umulhi((uint32_t)pix, L.magicW2);
                const int tx = pix - ty * L.W2;
                const int iy = it.y0 + ty;
                const int ii = L.halo ? tbl[j] : (it.f * L.Nin + iy) * L.Nin + tx;      // flat cell on the Nin grid
                const int r = __umulhi((uint32_t)ii, L.magicN);      // row face*Nin + y of the Nin grid -> row r/2 of Nin/2
                const int pix_up = (r >> 1) * g0 + ((ii - r * L.Nin) >> 1);
                {
                    const int spix = P0.up ? pix_up : ii;
                    xoff[j * CT] = (P0.ok && ptid + j * NCT < nitp) ? (uint32_t)(spix * P0.stride + P0.cs) * 2u : 0xffffffffu;
                }
                if constexpr (CT == 2) {
                    const int spix = P1.up ? pix_up : ii;
                    xoff[j * CT + 1] = (P1.ok && ptid + j * NCT < nitp) ? (uint32_t)(spix * P1.stride + P1.cs) * 2u : 0xffffffffu;
                }
            }
            cur_combo = it.combo;
        };
        uint32_t doff[IT_DY];                   // BYTE offsets from the item's first dZ row
#pragma unroll
        for (int i = 0; i < IT_DY; ++i) doff[i] = co_ok ? (uint32_t)(((ptid + i * NCT) / QDT) * L.Cout + co) * 2u : 0xffffffffu;
        struct Stage {
            XVec xv[IT_X];
            DVec dv[IT_DY], yv[MASK ? IT_DY : 1];
        };
        auto bld = [](rsrc_t r, uint32_t off, auto &dst) __attribute__((always_inline)) {
            typedef std::remove_reference_t<decltype(dst)> V;
#ifdef DLWPCS_WB_ABL_P          // (side builds only, wrong numbers: 2 = no global loads)
            if (DLWPCS_WB_ABL_P & 2) { memset(&dst, 0, sizeof(V)); asm volatile("" : "+v"(*reinterpret_cast<uint32_t *>(&dst))); return; }
#endif
            if constexpr (sizeof(V) == 16) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
                dst = make_uint4(q.x, q.y, q.z, q.w);
            } else {
                dst = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
            }
        };
        // `live` (uniform): an item index past the segment's end loads nothing (descriptors of zero bytes)
        auto issue = [&](const Item &it, Stage &st, bool live) {
            if (live && it.combo != cur_combo) rebuild(it);         // uniform, a few times per segment
            const rsrc_t r0 = make_rsrc(x0_u + (size_t)it.b * s0_bytes, live ? s0_bytes : 0u);
            const rsrc_t r1 = make_rsrc(x1_u ? x1_u + (size_t)it.b * s1_bytes : nullptr, live ? s1_bytes : 0u);
            if (!straddle) {
                // one descriptor per plane, chosen as SCALARS (base + size): no branch between the loads
                const rsrc_t rp0 = make_rsrc(P0.all0 ? x0_u + (size_t)it.b * s0_bytes : (x1_u ? x1_u + (size_t)it.b * s1_bytes : nullptr),
                                             live ? (P0.all0 ? s0_bytes : s1_bytes) : 0u);
                const rsrc_t rp1 = make_rsrc(P1.all0 ? x0_u + (size_t)it.b * s0_bytes : (x1_u ? x1_u + (size_t)it.b * s1_bytes : nullptr),
                                             live ? (P1.all0 ? s0_bytes : s1_bytes) : 0u);
#pragma unroll
                for (int i = 0; i < IT_X; ++i) {
                    if (i % CT == 0) bld(rp0, xoff[i], st.xv[i]);       // (compile-time: the loop is unrolled)
                    else bld(rp1, xoff[i], st.xv[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < IT_X; ++i) {
                    XVec a, b;
                    const bool f0 = i % CT == 0 ? P0.f0 : P1.f0;        // this thread's source in plane i % CT
                    bld(r0, f0 ? xoff[i] : 0xffffffffu, a);
                    bld(r1, f0 ? 0xffffffffu : xoff[i], b);
                    st.xv[i] = a | b;
                }
            }
            const size_t rowbase = (((size_t)it.b * 6 + it.f) * face_pix + it.m0) * L.Cout * 2;
            const uint32_t dbytes = live ? (uint32_t)(it.npix * L.Cout * 2) : 0u;     // slot valid <=> its pixel < npix
            const rsrc_t rd = make_rsrc(dz_u + rowbase, dbytes);
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) bld(rd, doff[i], st.dv[i]);
            if constexpr (MASK) {
                const rsrc_t ry = make_rsrc(y_u + rowbase, dbytes);
#pragma unroll
                for (int i = 0; i < IT_DY; ++i) bld(ry, doff[i], st.yv[i]);
            }
        };
        auto commit = [&](const Item &it, int k, Stage &st) {
            char *buf = smem + (k & 1) * buf_bytes;
            WB_TL(0);                                           // bucket 0: set-up + issue of the loads since the last barrier
#ifdef DLWPCS_WB_TL
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IT_X + IT_DY + (MASK ? IT_DY : 0)) : "memory");    // this stage's data (the other stage's loads stay in flight)
            WB_TL(1);                                           // bucket 1: wait for the stage's data
#endif
            // (slots beyond the item's cells hold zeros: their loads were issued with the out-of-range offset)
            char *const xbase = buf + (ptid / QX) * PB + qx * (XV * 2);
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
#ifdef DLWPCS_WB_ABL_P          // (1 = no LDS writes: the data is waited for, then dropped)
                if (DLWPCS_WB_ABL_P & 1) { asm volatile("" :: "v"(*reinterpret_cast<uint32_t *>(&st.xv[i]))); continue; }
#endif
                *reinterpret_cast<XVec *>(xbase + (i % CT) * plane_bytes + (i / CT) * (NCT / QX) * PB) = st.xv[i];
            }
            char *const dbase = buf + x_bytes + (qd / QD) * dzplane_bytes + (ptid / QDT) * PB + (qd % QD) * (DV * 2);
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                if constexpr (MASK) {                                       // dz = dy * act'(y), rounded to bf16 like a stored dz
                    if constexpr (DV == 8) vmask_pk(st.dv[i], st.yv[i], L.alpha, mthr1);
                    else vmask(st.dv[i], st.yv[i], L.alpha, L.vmax);
                }
                const DVec v = st.dv[i];
#ifdef DLWPCS_WB_ABL_P
                if (DLWPCS_WB_ABL_P & 1) { asm volatile("" :: "v"(*reinterpret_cast<const uint32_t *>(&v))); }
                else
#endif
                *reinterpret_cast<DVec *>(dbase + i * (NCT / QDT) * PB) = v;
#ifdef DLWPCS_WB_ABL_P
                if (DLWPCS_WB_ABL_P & 4) continue;      // (4 = no bias sums)
#endif
                if (want_bias) {
                    if constexpr (DV == 8) {
                        bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                        bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
                    } else {
                        bsum[0] += bf_lo(v); bsum[1] += bf_hi(v);
                    }
                }
            }
            // B_k: item k is in LDS.  RAW barrier behind an explicit LDS wait (a __syncthreads() would drain vmcnt(0), i.e. wait
            // for the loads of item k + 1 that are in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WB_TL(2);                                           // bucket 2: LDS writes (+ bias sums)
            __builtin_amdgcn_s_barrier();
            WB_TL(3);                                           // bucket 3: barrier
        };
        {
            Stage A, B;
            Item i0 = item_of(0), i1 = item_of(1);          // (item_of clamps; an index past the end is issued dead)
            issue(i0, A, true);
            issue(i1, B, 1 < n_my);
            // (whole pairs in the loop, an odd last item behind it: with `if (k + 1 >= n_my) break` in the middle of the body hipcc moved
            // the exit check -- and a vmcnt(0) for a value its exit path reads -- between the X loads and the dZ loads of an issue)
            const int npair = n_my >> 1;
            for (int pr = 0; pr < npair; ++pr) {
                const int k = 2 * pr;
                commit(i0, k, A);
                i0 = item_of(k + 2);
                issue(i0, A, k + 2 < n_my);
                commit(i1, k + 1, B);
                i1 = item_of(k + 3);
                issue(i1, B, k + 3 < n_my);
            }
            if (n_my & 1) commit(i0, n_my - 1, A);
        }
        // ---- bias partial: thread (vector qd, 256 / QDT pixel phases) holds sums of DV channels -> fixed-order sum
        __syncthreads();                // E1: consumers are done with the buffers
        float *red = reinterpret_cast<float *>(smem) + RED_FLOATS;   // behind the consumers' reduction slots
        if (want_bias) {
#pragma unroll
            for (int u = 0; u < DV; ++u) red[ptid * DV + u] = bsum[u];
        }
        __syncthreads();                // E2: (the consumers parked their accumulators between the two barriers)
        if (want_bias && ptid < 32 * NT) {
            // channel c = ptid of the 32 * NT staged columns: vector qd = (c / 32) * QD + (c % 32) / DV, element (c % 32) % DV
            float sum = 0.f;
            const int cc = ptid & 31, pl = ptid >> 5;
            if (cc < QD * DV) {
                const int q = pl * QD + cc / DV;
#pragma unroll 4
                for (int ph = 0; ph < NCT / QDT; ++ph) sum += red[(ph * QDT + q) * DV + (cc % DV)];
            }
            slot[TAPS * (32 * CT) * (32 * NT) + ptid] = sum;
        }
        __syncthreads();                // E3: the next segment may overwrite the buffers
        WB_TL(4);                                               // bucket 4: segment epilogue
        WB_TL_END();
        return;
    }

    // ============================================= consumers =============================================
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int li = lane & 15;
    WB_TL_DECL(1)
    const int choff = ((((lane >> 4) & 1) * 16) + (li & 3) * 4) * 2;   // byte offset of this lane's 4 contiguous channels
    const int prow = li >> 2;                                            // which of the 4 pixels of a transpose block
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int nslab = pix_cap / 16;             // K slabs (16 pixels) per item
    const int S = (((nslab + NPH - 1) / NPH) + 1) & ~1;   // slabs per consumer wave, rounded up to even (extras add zero)
    // (the wave index as a SCALAR: which plane, which slabs and whether a slab is alive are then scalar selects, not v_cndmasks)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ct = wave_u % CT, nt = (wave_u / CT) % NT, ph = wave_u / (CT * NT);
    // Addresses carried from slab to slab by ADDITIONS (round 6).  A lane's pixel advances by 16 * NPH per slab of its wave; through
    // round 5 every slab recomputed (row, column) of its two pixels from the flat index -- v_mul_hi / v_mul_lo / v_mad_u64 per pixel,
    // 12 quarter-rate instructions per pair of slabs on the SIMD the producer wave shares (the loop ran ~1030 cycles per 18 MFMAs =
    // 576 cycles of matrix pipe).  Now two divisions per lane and ITEM; per slab the column advances by (16 * NPH) % No with at most one
    // wrap, the tile address by a constant plus W2 - No pixels per wrap.  No clamp to the item's last pixel either: pixels of the
    // 16-pixel round-up and of a band's missing rows hold zeros in the dZ plane (their loads were issued out of range), their X
    // address stays inside the tile plane (rows the producers wrote as zeros).
    constexpr int DSTEP = 16 * NPH;
    const int stq = (int)__umulhi((uint32_t)DSTEP, L.magicNo), str = DSTEP - stq * L.No;
    const int stepA = (stq * L.W2 + str) * PB, wrapA = (L.W2 - L.No) * PB, rowB = L.W2 * PB;
    const int mlane = 16 * ph + 8 * half + prow;        // this lane's pixel of its wave's first slab (jj = 0; jj = 1: + 4)
    for (int k = 0; k < n_my; ++k) {
        WB_TL(1);                               // bucket 1: fragment reads + MFMAs of the previous item (+ set-up)
        __syncthreads();                        // B_k
        WB_TL(0);                               // bucket 0: wait at B_k
        const int buf0 = (k & 1) * buf_bytes;
        const Item it = item_of(k);
        const int x0 = it.m0 - it.y0 * L.No;    // first pixel's column (0: bands are whole rows wherever a face row fits an item)
        int xa[2], ox[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int g = x0 + mlane + 4 * jj;
            const int oy = __umulhi((uint32_t)g, L.magicNo);
            ox[jj] = g - oy * L.No;
            xa[jj] = buf0 + ct * plane_bytes + (oy * L.W2 + ox[jj]) * PB + choff;
        }
        int da = buf0 + x_bytes + nt * dzplane_bytes + mlane * PB + choff;
        // frag(si) must be called for si = 0, 1, 2, ... in order: it reads slab si at the carried addresses and moves them on
        auto frag = [&](int si, uint4 (&a)[TAPS], uint4 &bq) {
            const int s = ph + NPH * si;
            // a slab past the item's pixels reads its dZ fragment 8 MB beyond the allocation: zeros (dlwpcs_lds_oob_probe).  As a
            // select on the LOADED fragment the masking put a wait for these two reads -- issued a moment before -- into every
            // slab: one exposed LDS round trip per 9 MFMAs
            const char *pd = smem + da + (s < nslab ? 0 : (1 << 23));
            const uint2 b0 = lds_tr16(pd), b1 = lds_tr16(pd + 4 * PB);
            bq = make_uint4(b0.x, b0.y, b1.x, b1.y);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const char *p0 = smem + xa[0] + ky * rowB, *p1 = smem + xa[1] + ky * rowB;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
#ifdef DLWPCS_WB_ABL_DS         // (side builds only: wrong numbers -- one X fragment per slab / per tap row instead of nine)
                    if (DLWPCS_WB_ABL_DS == 1 && (ky | kx)) { a[ky * KS + kx] = a[0]; continue; }
                    if (DLWPCS_WB_ABL_DS == 3 && kx) { a[ky * KS + kx] = a[ky * KS]; continue; }
#endif
                    const uint2 a0 = lds_tr16(p0 + kx * PB), a1 = lds_tr16(p1 + kx * PB);
                    a[ky * KS + kx] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                }
            }
            // on to the wave's next slab (frozen once that one is dead: the X address never leaves the tile plane)
            const bool nl = s + NPH < nslab;
            const int sr = nl ? str : 0, sa = nl ? stepA : 0;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int o = ox[jj] + sr;
                const bool w = o >= L.No;
                ox[jj] = w ? o - L.No : o;
                xa[jj] += sa + (w ? wrapA : 0);
            }
            da += DSTEP * PB;
        };
        uint4 fa[2][TAPS], fb[2];
        frag(0, fa[0], fb[0]);
#ifdef DLWPCS_WB_TL
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WB_TL(3);                               // bucket 3: item set-up + the first slab's fragments (of bucket 1's time otherwise)
#endif
        for (int si = 0; si < S; si += 2) {
            frag(si + 1, fa[1], fb[1]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) frag_mma<bf16_t>(acc[tap], fb[0], fa[0][tap]);
            WB_SCHED();
            frag(si + 2, fa[0], fb[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) frag_mma<bf16_t>(acc[tap], fb[1], fa[1][tap]);
            WB_SCHED();
        }
    }
    WB_TL(1);
    __syncthreads();                            // E1: all consumers finished reading the last buffer
    // Cross-wave reduction, one LDS round (see wgrad_bf16_kernel): the NPH waves of a (ci tile, co tile) hold K-split sums
    // of the same (taps, 32, 32) block; tap t belongs to the wave of phase t % NPH, the others park theirs in LDS.
    float4 *red4 = reinterpret_cast<float4 *>(smem);
    const int wgrp = ct * NT + nt;
    auto slot_of = [](int t, int p) { int c = 0; for (int u = 0; u < t; ++u) c += (u % NPH != p) ? 1 : 0; return c; };
    if constexpr (NPH > 1) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (ph != t % NPH) {
                float4 *dst = red4 + (size_t)((wgrp * NPH + ph) * RSLOTS + slot_of(t, ph)) * 256 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[q * 64] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
            }
        }
    }
    __syncthreads();                            // E2
    const rsrc_t pr = make_rsrc(slot, (uint32_t)(TAPS * (32 * CT) * (32 * NT) * 4));
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (ph == t % NPH) {
            f32x16 v[NPH];
#pragma unroll
            for (int p = 0; p < NPH; ++p) {
                if (p == t % NPH) { v[p] = acc[t]; continue; }
                if constexpr (NPH > 1) {
                    const float4 *src = red4 + (size_t)((wgrp * NPH + p) * RSLOTS + slot_of(t, p)) * 256 + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 f = src[q * 64];
                        v[p][4 * q] = f.x; v[p][4 * q + 1] = f.y; v[p][4 * q + 2] = f.z; v[p][4 * q + 3] = f.w;
                    }
                }
            }
            // The MFMAs ran as D[co][ci] (dZ as the A operand, X as the B operand: the two operand layouts are the same function of
            // (lane, register), so the roles swap with the arguments): a lane owns input channel l31 and, per register quad, FOUR
            // CONSECUTIVE output channels 8 q + 4 half + (0..3) -- four 16-B stores into the slot's [tap][ci][co] rows instead of sixteen
            // 4-B ones per tap.  (The launch's WRITE_SIZE -- 62 MB for 23 MB of partial sums -- did not move with it: what the counter
            // sees besides the slots are the callee-saved registers every noinline segment call parks in scratch memory, 71 dwords x 512
            // threads x ~290 calls = 42 MB written and read back per launch.)
            const uint32_t row0 = (uint32_t)(((t * (32 * CT) + ct * 32 + l31) * (32 * NT) + nt * 32 + 4 * half) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sum[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * q + u;
                    if (NPH == 4) sum[u] = (v[0][r] + v[1][r]) + (v[2][r] + v[3][r]);
                    else if (NPH == 2) sum[u] = v[0][r] + v[1][r];
                    else sum[u] = v[0][r];
                }
                bst128(make_uint4(__float_as_uint(sum[0]), __float_as_uint(sum[1]), __float_as_uint(sum[2]), __float_as_uint(sum[3])), pr,
                       row0 + (uint32_t)(8 * q * 4));
            }
        }
    }
    __syncthreads();                            // E3
    WB_TL(2);                                   // bucket 2: segment epilogue (reduction, partial sums)
    WB_TL_END();
}

// ------------------------------------------------------------------------------------------------------------------
// One segment in the exact-fp32 mode (round 4): the main loop of wgrad_mfma_kernel<float, KS, 4, MASK> (conv_mfma.hip) --
// D[ci][co] += sum over pixel pairs of Xpad[pixel + tap][ci] * dZ[pixel][co] on v_mfma_f32_32x32x2_f32, four consumer waves
// splitting the pixel pairs of an item of <= 192 pixels, four producer waves staging the X tile (band + halo rows, 32 input
// channels, fp32: 128 B per pixel) and the dZ tile through two LDS buffers, dZ = dy * act'(y) formed on load where the item carries
// y -- run over the segment's range of work items, one partial sum per segment in the layout wb_reduce_kernel expects.
// Called by all 512 threads; both halves execute n_items + 2 * ceil(TAPS / 3) + 3 workgroup barriers.
// ------------------------------------------------------------------------------------------------------------------
#ifndef DLWPCS_WB_F32_SCHED
#define DLWPCS_WB_F32_SCHED 2
#endif
#if DLWPCS_WB_F32_SCHED == 1
#define WB_F32_SCHED() do {                                                                                  \
        _Pragma("unroll") for (int t_ = 0; t_ < TAPS; ++t_) {                                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA */                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      /* one LDS read of the next step */       \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      /* address arithmetic of the step after */ \
        }                                                                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                    \
    } while (0)
#elif DLWPCS_WB_F32_SCHED == 2
// the reads front-loaded, two behind each of the first MFMAs: the last one has four MFMAs to land before the next step needs it
#define WB_F32_G(nread) do {                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    \
        if (nread) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                    \
    } while (0)
#define WB_F32_SCHED() do {                                                                                  \
        if constexpr (TAPS == 9) {                                                                            \
            WB_F32_G(2); WB_F32_G(2); WB_F32_G(2); WB_F32_G(2); WB_F32_G(2);                                  \
            WB_F32_G(0); WB_F32_G(0); WB_F32_G(0); WB_F32_G(0);                                               \
        } else {                                                                                              \
            WB_F32_G(2);                                                                                      \
        }                                                                                                     \
    } while (0)
#else
#define WB_F32_SCHED() do {                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);                                                 \
    } while (0)
#endif
template <int KS, bool MASK, bool X2 = false, bool D2 = false>
__device__ __attribute__((noinline)) void wb_segment_f32(const WbLayer &Lg, const WbSeg &sgg, const void *src0, const void *src1,
                                                       const void *dzp, const void *yp, const int32_t *table, float *ws, char *smem_raw) {
    const WbLayer L = load_uniform(Lg);
    const WbSeg sg = load_uniform(sgg);
    constexpr int TAPS = KS * KS;
    constexpr int XS = 32;                  // floats per X tile pixel (the 32 input channels of this ci tile)
    constexpr int NCT = 256;
    constexpr int IT_X = 14;                // X vectors per producer thread per item: capacity 448 tile pixels
    constexpr int IT_DY = 6;                // dZ quads per producer thread per item: capacity 192 pixels
    constexpr int TG = TAPS >= 3 ? 3 : 1;   // taps summed across the consumer waves per LDS round of the epilogue
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int pix_cap = (L.pix + 1) & ~1;
    const int x_floats = L.tile_rows_max * L.W2 * XS;
    const int buf_floats = x_floats + pix_cap * 32;
    const int fbase = sg.cls == 0 ? 0 : (sg.cls == 1 ? 4 : 5);
    const int n_my = sg.t_last - sg.t_first;
    const int face_pix = L.No * L.No;
    const int tid = threadIdx.x;
    // 16-B vectors per staged pixel, rounded up to a power of two: a 14-channel input takes 4 lanes per pixel, not 8 (the
    // channels above them keep whatever the LDS held: they only reach rows / columns of the product nobody reads)
    const int cin_grp = min(32, L.Cin - sg.cit * 32), cout_grp = min(32, L.Cout - sg.cot * 32);
    const int lqx = cin_grp > 16 ? 3 : (cin_grp > 8 ? 2 : (cin_grp > 4 ? 1 : 0));
    const int lqd = cout_grp > 16 ? 3 : (cout_grp > 8 ? 2 : (cout_grp > 4 ? 1 : 0));
    struct Item { int b, f, combo, m0, npix, y0, nitems; };
    auto item_of = [&](int k) {
        Item it;
        const int t = sg.t_first + max(min(k, n_my - 1), 0);
        it.combo = L.magicB ? __umulhi((uint32_t)t, L.magicB) : t;
        it.b = t - it.combo * L.B;
        const int fl = L.magicNb ? __umulhi((uint32_t)it.combo, L.magicNb) : it.combo;
        const int band = it.combo - fl * L.nbands;
        it.f = fbase + fl;
        it.m0 = band * L.pix;
        it.npix = min(L.pix, face_pix - it.m0);
        it.y0 = __umulhi((uint32_t)it.m0, L.magicNo);
        const int ylast = __umulhi((uint32_t)(it.m0 + it.npix - 1), L.magicNo);
        it.nitems = ((ylast - it.y0 + KS) * L.W2) << lqx;
        return it;
    };
    ws = reinterpret_cast<float *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)ws >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ws));      // (see wb_segment)
    float *slot = ws + sg.slot_off;

    if (tid >= NCT) {
        // =========================================== producers ===========================================
        const int ptid = tid - NCT;
        const int qx = ptid & ((1 << lqx) - 1), qd = ptid & ((1 << lqd) - 1);
        const int nit_x = (((L.tile_rows_max * L.W2) << lqx) + NCT - 1) / NCT;         // (uniform: iterations that carry work)
        const int nit_d = ((pix_cap << lqd) + NCT - 1) / NCT;
        const int x_cap = (L.tile_rows_max * L.W2) << lqx;
        const int cx = sg.cit * 32 + qx * 4;
        const bool cx_ok = cx < L.Cin;
        const bool hi_ok = cx + 2 < L.Cin;      // (H2: the vector is two 8-B halves, the upper one may lie past the last channel)
        const bool from0 = cx < L.C0;
        const int g0 = L.up0 ? (L.Nin >> 1) : L.Nin;
        const int cs = from0 ? cx : cx - L.C0;
        const int cstride = from0 ? L.C0 : L.C1;
        const bool up = from0 && L.up0;
        const int M = L.Nin + KS - 1;
        const bool want_bias = sg.bias != 0;
        float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
        int xoff[IT_X];
        int cur_combo = -1;
        auto rebuild = [&](const Item &it) {
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
                if (i >= nit_x) { xoff[i] = -1; continue; }
                const int e = min(ptid + i * NCT, it.nitems - 1);
                const int pix = e >> lqx;
                const int ty = __umulhi((uint32_t)pix, L.magicW2);
                const int tx = pix - ty * L.W2;
                const int iy = it.y0 + ty;
                int ii;
                if (L.halo) ii = table[(it.f * M + iy) * M + tx];
                else ii = (it.f * L.Nin + iy) * L.Nin + tx;
                const int r = __umulhi((uint32_t)ii, L.magicN);
                const int pix_up = (r >> 1) * g0 + ((ii - r * L.Nin) >> 1);
                const int spix = up ? pix_up : ii;
                xoff[i] = (cx_ok && ptid + i * NCT < it.nitems) ? spix * cstride + cs : -1;
            }
            cur_combo = it.combo;
        };
        const size_t sample_elems = from0 ? (size_t)6 * g0 * g0 * L.C0 : (size_t)6 * L.Nin * L.Nin * L.C1;
        const float *src_base = reinterpret_cast<const float *>(from0 ? src0 : src1);
        Item cur = item_of(0);
        for (int k = 0; k < n_my; ++k) {
            float *buf = smem + (k & 1) * buf_floats;
            const Item nxt = item_of(k + 1);
            if (cur.combo != cur_combo) rebuild(cur);
            const float *sb = src_base + (size_t)cur.b * sample_elems;
            float4 xv[IT_X];
            bool xok[IT_X];
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
#ifdef WB_F32_ABL
                if (WB_F32_ABL & 2) { xv[i] = make_float4(1.f, 2.f, 3.f, 4.f); xok[i] = true; continue; }
#endif
                const int o = xoff[i];
                if (i >= nit_x) { xok[i] = false; continue; }
                if constexpr (X2) {
                    const float2 lo = *reinterpret_cast<const float2 *>(sb + (uint32_t)max(o, 0));
                    const float2 hi = *reinterpret_cast<const float2 *>(sb + (uint32_t)(hi_ok ? max(o, 0) + 2 : 0));
                    xv[i] = make_float4(lo.x, lo.y, hi_ok ? hi.x : 0.f, hi_ok ? hi.y : 0.f);
                } else {
                    xv[i] = *reinterpret_cast<const float4 *>(sb + (uint32_t)max(o, 0));
                }
                xok[i] = o >= 0;
            }
            const size_t rowbase = (((size_t)cur.b * 6 + cur.f) * face_pix + cur.m0) * L.Cout;
            const float *dyb = reinterpret_cast<const float *>(dzp) + rowbase;
            const float *yb = MASK ? reinterpret_cast<const float *>(yp) + rowbase : nullptr;
            float4 dv[IT_DY], yv[MASK ? IT_DY : 1];
            bool dok[IT_DY];
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                const int e = ptid + i * NCT;
                const int kk = e >> lqd, co = sg.cot * 32 + qd * 4;
                const bool ok = kk < cur.npix && co < L.Cout;
                const size_t o = ok ? (size_t)kk * L.Cout + co : 0;
                if (i >= nit_d) { dv[i] = make_float4(0.f, 0.f, 0.f, 0.f); if (MASK) yv[i] = dv[i]; dok[i] = false; continue; }
#ifdef WB_F32_ABL
                if (WB_F32_ABL & 2) { dv[i] = make_float4(1.f, 2.f, 3.f, 4.f); if (MASK) yv[i] = dv[i]; dok[i] = ok; continue; }
#endif
                if constexpr (D2) {
                    const bool dhi = ok && co + 2 < L.Cout;
                    const float2 lo = *reinterpret_cast<const float2 *>(dyb + o), hi = *reinterpret_cast<const float2 *>(dyb + (dhi ? o + 2 : 0));
                    dv[i] = make_float4(lo.x, lo.y, dhi ? hi.x : 0.f, dhi ? hi.y : 0.f);
                    if (MASK) {
                        const float2 ylo = *reinterpret_cast<const float2 *>(yb + o), yhi = *reinterpret_cast<const float2 *>(yb + (dhi ? o + 2 : 0));
                        yv[i] = make_float4(ylo.x, ylo.y, yhi.x, yhi.y);
                    }
                } else {
                    dv[i] = *reinterpret_cast<const float4 *>(dyb + o);
                    if (MASK) yv[i] = *reinterpret_cast<const float4 *>(yb + o);
                }
                dok[i] = ok;
            }
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                if (MASK) vmask(dv[i], yv[i], L.alpha, L.vmax);
                dv[i] = vsel(dok[i], dv[i]);
            }
#pragma unroll
            for (int i = 0; i < IT_X; ++i) {
                const int e = ptid + i * NCT;
                // (every slot of the tile, zeros beyond the item's cells: the consumers' capped addresses may read them, see pb_max)
                if (i < nit_x && e < x_cap) *reinterpret_cast<float4 *>(buf + (e >> lqx) * XS + qx * 4) = vsel(xok[i], xv[i]);
            }
#pragma unroll
            for (int i = 0; i < IT_DY; ++i) {
                const int e = ptid + i * NCT;
                if (i < nit_d && (e >> lqd) < pix_cap) *reinterpret_cast<float4 *>(buf + x_floats + (e >> lqd) * 32 + qd * 4) = dv[i];
                if (want_bias) { bsum.x += dv[i].x; bsum.y += dv[i].y; bsum.z += dv[i].z; bsum.w += dv[i].w; }
            }
            __syncthreads();            // B_k: item k is in LDS
            cur = nxt;
        }
        __syncthreads();                // E1: consumers are done with the buffers
        if (want_bias) {
            float *red = smem + TG * 4096;          // behind the consumers' reduction scratch
            red[ptid * 4 + 0] = bsum.x; red[ptid * 4 + 1] = bsum.y; red[ptid * 4 + 2] = bsum.z; red[ptid * 4 + 3] = bsum.w;
        }
        __syncthreads();                // E2
        if (want_bias && ptid < 32) {
            float sum = 0.f;
            const float *red = smem + TG * 4096;
            // the threads that staged this channel's quad: ptid % (1 << lqd) == channel / 4
            if ((ptid >> 2) < (1 << lqd))
                for (int t = ptid >> 2; t < NCT; t += 1 << lqd) sum += red[t * 4 + (ptid & 3)];
            slot[TAPS * 32 * 32 + ptid] = sum;
        }
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += TG) { __syncthreads(); __syncthreads(); }
        __syncthreads();                // E3: the next segment may overwrite the buffers
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
    const int nsteps = pix_cap / 2;
    const int S = (((nsteps + 3) / 4) + 1) & ~1;
    // Addresses carried from step to step by additions (round 6, as in wb_segment): a lane's pixel advances by 8 per step of its wave.
    // A v_mfma_f32_32x32x2_f32 takes 64 cycles and the fitted period was 73: the ~13 VALU instructions per step (three of them
    // quarter-rate: v_mul_hi, v_mul_lo, v_mad) issue BETWEEN the MFMAs of an in-order wave, they do not hide behind them.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int stq = (int)__umulhi(8u, L.magicNo), str = 8 - stq * L.No;
    const int stepA = (stq * L.W2 + str) * XS, wrapA = (L.W2 - L.No) * XS;
    // (the X address is capped at the tile's last window: pixels of the round-up to even / of a band's missing rows have dZ = 0, and the
    // producers write every slot of the tile -- zeros beyond the item's cells --, so what they read there is finite)
    const int pb_max = ((L.tile_rows_max - KS) * L.W2 + (L.W2 - KS)) * XS + l31;
    for (int k = 0; k < n_my; ++k) {
        __syncthreads();                // B_k
        const float *lds_x = smem + (k & 1) * buf_floats, *lds_dy = lds_x + x_floats;
        const Item it = item_of(k);
        // Pixel pair `si` of this wave: the LDS offset of its X value under tap (0, 0) and of its dZ value.  The 10 reads of step
        // si + 1 are issued in the shadow of the 9 MFMAs of step si: an in-order wave that first issues all reads, then all MFMAs
        // leaves the matrix pipe idle (measured, nine 3x3 layers of unet2 at B = 32: 816 us that way; 440 us is the pipe's own time).
        const int kk0 = 2 * wave_u + half;
        int ox, pb, db = kk0 * 32 + l31;
        {
            const int g = it.m0 - it.y0 * L.No + kk0;
            const int oy = __umulhi((uint32_t)g, L.magicNo);
            ox = g - oy * L.No;
            pb = (oy * L.W2 + ox) * XS + l31;
        }
        // frag(si) must be called for si = 0, 1, 2, ... in order: it reads step si at the carried addresses and moves them on
        auto frag = [&](int si, float (&a)[TAPS], float &bq) {
            const int s = wave_u + 4 * si;
            // (a step past the item's pixel pairs reads its dZ value 8 MB beyond the allocation: zero -- dlwpcs_lds_oob_probe)
            bq = lds_dy[db + (s < nsteps ? 0 : (1 << 21))];
            const int p = min(pb, pb_max);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) a[tap] = lds_x[p + ((tap / KS) * L.W2 + (tap % KS)) * XS];
            const bool nl = s + 4 < nsteps;
            const int sr = nl ? str : 0, sa = nl ? stepA : 0;
            const int o = ox + sr;
            const bool w = o >= L.No;
            ox = w ? o - L.No : o;
            pb += sa + (w ? wrapA : 0);
            db += 8 * 32;
        };
        float fa[2][TAPS], fb[2];
        frag(0, fa[0], fb[0]);
#ifdef WB_F32_ABL
        if (WB_F32_ABL & 1) continue;
#endif
        for (int si = 0; si < S; si += 2) {
            frag(si + 1, fa[1], fb[1]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][tap], fb[0], acc[tap], 0, 0, 0);
            WB_F32_SCHED();
            frag(si + 2, fa[0], fb[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][tap], fb[1], acc[tap], 0, 0, 0);
            WB_F32_SCHED();
        }
    }
    __syncthreads();                    // E1
    __syncthreads();                    // E2 (the producers stage their bias sums between these two)
    float *red = smem;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TG) {
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
            if (t0 + tt < TAPS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                    red[(tt * 4 + wave) * 1024 + ci * 32 + l31] = acc[t0 + tt][r];
                }
            }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
            if (t0 + tt < TAPS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = tid + i * NCT;
                    const float *rt = red + tt * 4096;
                    slot[(size_t)(t0 + tt) * 1024 + e] = (rt[e] + rt[1024 + e]) + (rt[2048 + e] + rt[3072 + e]);
                }
            }
        __syncthreads();
    }
    __syncthreads();                    // E3
}

__global__ void __launch_bounds__(512) wgrad_batch_kernel(const char *__restrict__ plan, const WbPtrs ptrs, float *__restrict__ ws,
                                                          long long *dbg, int32_t *adam_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WbHeader *H = reinterpret_cast<const WbHeader *>(plan);
    const WbLayer *layers = reinterpret_cast<const WbLayer *>(plan + H->off_layers);
    const WbSeg *segs = reinterpret_cast<const WbSeg *>(plan + H->off_segs);
    const int w = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int s0 = (int)H->seg_start[w], s1 = (int)H->seg_start[w + 1];
    for (int s = s0; s < s1; ++s) {
        const WbSeg &sg = segs[s];
        const WbLayer &L = layers[sg.layer];
        const void *a0 = ptrs.src0[sg.layer], *a1 = ptrs.src1[sg.layer], *dz = ptrs.dz[sg.layer], *yy = ptrs.y[sg.layer];
        const int32_t *tb = ptrs.table[sg.layer];
        if (L.variant >= WB_V_F32_3) {
            switch (L.variant) {
            case WB_V_F32_1: wb_segment_f32<1, false>(L, sg, a0, a1, dz, yy, tb, ws, smem); break;
            case WB_V_F32_1_D2: wb_segment_f32<1, false, false, true>(L, sg, a0, a1, dz, yy, tb, ws, smem); break;
            case WB_V_F32_3_X2:
                if (L.mask) wb_segment_f32<3, true, true>(L, sg, a0, a1, dz, yy, tb, ws, smem);
                else wb_segment_f32<3, false, true>(L, sg, a0, a1, dz, yy, tb, ws, smem);
                break;
            case WB_V_F32_3_H2:
                if (L.mask) wb_segment_f32<3, true, true, true>(L, sg, a0, a1, dz, yy, tb, ws, smem);
                else wb_segment_f32<3, false, true, true>(L, sg, a0, a1, dz, yy, tb, ws, smem);
                break;
            default:
                if (L.mask) wb_segment_f32<3, true>(L, sg, a0, a1, dz, yy, tb, ws, smem);
                else wb_segment_f32<3, false>(L, sg, a0, a1, dz, yy, tb, ws, smem);
            }
            continue;
        }
        if (L.mask) {
            switch (L.variant) {
                case WB_V_3_8_22: wb_segment<3, 8, 4, 2, 2, 8, true>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
                case WB_V_3_8_21: wb_segment<3, 8, 4, 2, 1, 8, true>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
                case WB_V_3_8_12: wb_segment<3, 8, 4, 1, 2, 8, true>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
                case WB_V_3_8_11: wb_segment<3, 8, 4, 1, 1, 8, true>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
                case WB_V_3_2_8:  wb_segment<3, 2, 8, 1, 1, 8, true>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
                default:          wb_segment<3, 2, 16, 1, 1, 8, true>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            }
            continue;
        }
        switch (L.variant) {
            case WB_V_3_8_22: wb_segment<3, 8, 4, 2, 2, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            case WB_V_3_8_21: wb_segment<3, 8, 4, 2, 1, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            case WB_V_3_8_12: wb_segment<3, 8, 4, 1, 2, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            case WB_V_3_8_11: wb_segment<3, 8, 4, 1, 1, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            case WB_V_3_2_8:  wb_segment<3, 2, 8, 1, 1, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            case WB_V_3_2_16: wb_segment<3, 2, 16, 1, 1, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            case WB_V_1_8_11: wb_segment<1, 8, 4, 1, 1, 8>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
            default:          wb_segment<1, 8, 4, 1, 1, 2>(L, sg, a0, a1, dz, yy, tb, ws, smem, dbg); break;
        }
    }
    // optimizer fused into the reduction that follows: the step counter {t, ticket} moves on here (last worker to finish)
    if (adam_state != nullptr && threadIdx.x == 0) {
        const int done = atomicAdd(adam_state + 1, 1);
        if (done == (int)gridDim.x - 1) { adam_state[1] = 0; adam_state[0] += 1; }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Reduction: a workgroup owns 64 groups of VEC consecutive output channels of one (layer, tap, ci) row (or of the bias
// vector) x the 3 face classes (equatorial, face 4, face 5): thread (o, c) adds the partial sums of its (ci group, co group)
// and class in slot order and updates that class's destination -- a fixed order, no atomics.  Where the two pole classes
// share their weights, class 2 hands its sum to class 1 through LDS.  (Three light waves per workgroup, all of them resident
// at once: the first version -- 4 slot phases x 3 classes per thread, 140 VGPRs -- ran in two rounds of workgroups.)
// ------------------------------------------------------------------------------------------------------------------
constexpr int WB_RED_OUT = 64, WB_RED_THREADS = 3 * WB_RED_OUT;

template <int VEC, bool ALIGNED>
__device__ __forceinline__ void wb_reduce_body(const WbRedLayer &L, const WbGroup *__restrict__ groups, const float *__restrict__ ws,
                                               float *dw_eq, float *dw_pol, float *dw_np, float *db_eq, float *db_pol,
                                               float *db_np, int block, const WbAdam &A, const WbRedPack &K) {
    typedef float VT __attribute__((ext_vector_type(VEC)));
    constexpr bool VECIO = ALIGNED || VEC == 1;         // (the four flat buffers share their 16-B alignment)
    float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, gscale = 1.f;
    if (A.on) {
        b1 = A.hyper[1]; b2 = A.hyper[2]; eps = A.hyper[3]; gscale = A.hyper[4];
        lr_t = adam_lr_t(A.hyper[0], b1, b2, (float)(A.state[0] + A.apply_only));
    }
    const int KS = L.KS, TAPS = KS * KS, Cin = L.Cin, Cout = L.Cout;
    const int nW = TAPS * Cin * Cout;
    const int o = (int)threadIdx.x % WB_RED_OUT, c = (int)threadIdx.x / WB_RED_OUT;
    const int e = (block * WB_RED_OUT + o) * VEC;
    const bool is_w = e < nW, is_b = !is_w && L.want_bias && e < nW + Cout;
    const int TC = L.TC, TN = L.TN;
    // this thread's destination; the two pole sums share one where the layer has no north-pole parameters of its own
    float *dst = nullptr;
    bool shared_pole = false;
    if (is_w) { dst = c == 0 ? dw_eq + e : (c == 1 ? dw_pol + e : (dw_np ? dw_np + e : nullptr)); shared_pole = dw_np == nullptr; }
    if (is_b) {
        const int co = e - nW;
        float *base = c == 0 ? db_eq : (c == 1 ? db_pol : db_np);
        dst = base ? base + co : nullptr; shared_pole = db_np == nullptr;
    }
    // what the destination holds (and, with the optimizer fused in, the parameter and its two moments) is fetched BEFORE the
    // partial sums: independent of them, so the round trips overlap
    VT gv = 0.f, pv = 0.f, mv = 0.f, vv = 0.f;
    const size_t ai = (A.on && dst) ? (size_t)(dst - A.g) : 0;
    if (dst != nullptr) {
        if constexpr (VECIO) {
            gv = *reinterpret_cast<const VT *>(dst);
            if (A.on) {
                pv = *reinterpret_cast<const VT *>(A.p + ai); mv = *reinterpret_cast<const VT *>(A.m + ai);
                vv = *reinterpret_cast<const VT *>(A.v + ai);
            }
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                gv[k] = dst[k];
                if (A.on) { pv[k] = A.p[ai + k]; mv[k] = A.m[ai + k]; vv[k] = A.v[ai + k]; }
            }
        }
    }
    VT s = 0.f;
    int w_co = 0, w_ci = 0, w_ty = 0, w_tx = 0;         // (weight element of this thread, for the fused packing below)
    if (A.apply_only) {
        if (is_w) { w_co = e % Cout; w_ci = (e / Cout) % Cin; const int tap = e / (Cout * Cin); w_ty = tap / KS; w_tx = tap - w_ty * KS; }
    } else if (is_w || is_b) {
        int gi;
        size_t off;
        if (is_w) {
            const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cout * Cin);
            const int cit = ci / TC, cot = co / TN, lci = ci - cit * TC, lco = co - cot * TN;
            const int ty = tap / KS, tx = tap - ty * KS;
            w_co = co; w_ci = ci; w_ty = ty; w_tx = tx;
            const int tap5 = L.flip ? (KS - 1 - ty) * KS + tx : tap;    // face 5 ran with the row-reversed kernel
            gi = L.group_base + (cit * L.ncot + cot) * 3 + c;
            off = (size_t)(((c == 2 ? tap5 : tap) * TC + lci) * TN + lco);
        } else {
            const int co = e - nW;
            const int cot = co / TN, lco = co - cot * TN;
            gi = L.group_base + cot * 3 + c;                             // ci group 0 carries the bias sums
            off = (size_t)TAPS * TC * TN + lco;
        }
        const WbGroup g = groups[gi];
        const float *p = ws + g.off + off;
#pragma unroll 8
        for (int j = 0; j < g.count; ++j) s += *reinterpret_cast<const VT *>(p + (size_t)j * g.stride);
    }
    __shared__ VT red[WB_RED_OUT];
    if (c == 2) red[o] = s;
    __syncthreads();
    if (dst == nullptr) return;
    if (c == 1 && shared_pole) s = s + red[o];
    // destination update; with the optimizer fused in, the finished gradient (what the buffer held + the reduced sum) is
    // consumed here: Adam element update (the arithmetic of adam_fused_kernel, same bits), gradient cleared for the next step
    VT out;
    if (A.on) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float pk = pv[k], mk = mv[k], vk = vv[k];
            adam_elem(pk, (gv[k] + s[k]) * gscale, mk, vk, lr_t, b1, b2, eps);
            pv[k] = pk; mv[k] = mk; vv[k] = vk;
        }
        out = 0.f;
    } else {
        out = gv + s;
    }
    // The packed bf16 operands of the layer (dlwpcs_pack_batch layout, see pack_weights_range in conv_mfma.hip) follow the
    // parameter: forward fragments [v][n tile][ci group of 16][tap][half][32 columns = co][8 = ci], data-gradient fragments
    // [v][n tile = ci / 32][co group of 16][tap, flipped][half][32 columns = ci][8 = co]; v = face variant (equatorial, south
    // pole, north pole = the pole weights again with the tap rows reversed when flip_north_pole), biases [v][32 * n tiles] fp32.
    if (A.on && K.wf != nullptr) {
        const int v0 = c, v1 = (c == 1 && shared_pole) ? 2 : -1;
#pragma unroll
        for (int iv = 0; iv < 2; ++iv) {
            const int vv = iv == 0 ? v0 : v1;
            if (vv < 0) continue;
            if (is_w) {
                const int TAPS2 = KS * KS;
                const int tyv = (vv == 2 && L.flip) ? KS - 1 - w_ty : w_ty;
                if (K.f32) {
                    // exact-fp32 mode: the same fragment order with 4 fp32 values per 16-B entry (ci / co groups of 8)
                    float *wf = reinterpret_cast<float *>(K.wf), *wb = reinterpret_cast<float *>(K.wb);
                    {
                        const int cg = w_ci >> 3, hf = (w_ci >> 2) & 1, j = w_ci & 3;
                        const size_t base = (size_t)(((((vv * K.NTf + (w_co >> 5)) * K.CGf + cg) * TAPS2 + tyv * KS + w_tx) * 2 + hf) * 32);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) wf[(base + ((w_co + k) & 31)) * 4 + j] = pv[k];
                    }
                    {
                        const int cg = w_co >> 3, hf = (w_co >> 2) & 1, j = w_co & 3;
                        const int ey = KS - 1 - tyv, ex = KS - 1 - w_tx;
                        float *q = wb + ((size_t)(((((vv * K.NTb + (w_ci >> 5)) * K.CGb + cg) * TAPS2 + ey * KS + ex) * 2 + hf) * 32) +
                                         (w_ci & 31)) * 4 + j;
                        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(q) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                        else q[0] = pv[0];
                    }
                    continue;
                }
                {   // forward fragments: VEC consecutive co of one 32-column tile, 16 B apart
                    const int cg = w_ci >> 4, hf = (w_ci >> 3) & 1, j = w_ci & 7;
                    const size_t base = (size_t)(((((vv * K.NTf + (w_co >> 5)) * K.CGf + cg) * TAPS2 + tyv * KS + w_tx) * 2 + hf) * 32);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) K.wf[(base + ((w_co + k) & 31)) * 8 + j] = f2bf(pv[k]);
                }
                {   // data-gradient fragments: VEC consecutive co = consecutive elements of one 16-B entry
                    const int cg = w_co >> 4, hf = (w_co >> 3) & 1, j = w_co & 7;
                    const int ey = KS - 1 - tyv, ex = KS - 1 - w_tx;
                    bf16_t *q = K.wb + ((size_t)(((((vv * K.NTb + (w_ci >> 5)) * K.CGb + cg) * TAPS2 + ey * KS + ex) * 2 + hf) * 32) +
                                        (w_ci & 31)) * 8 + j;
                    if constexpr (VEC == 4) *reinterpret_cast<uint2 *>(q) = make_uint2(f2bf2(pv[0], pv[1]), f2bf2(pv[2], pv[3]));
                    else q[0] = f2bf(pv[0]);
                }
            } else if (K.bp != nullptr) {
                const int co = e - nW;
#pragma unroll
                for (int k = 0; k < VEC; ++k) K.bp[vv * K.NTf * 32 + co + k] = pv[k];
            }
        }
    }
    if constexpr (VECIO) {
        *reinterpret_cast<VT *>(dst) = out;
        if (A.on) {
            *reinterpret_cast<VT *>(A.p + ai) = pv; *reinterpret_cast<VT *>(A.m + ai) = mv; *reinterpret_cast<VT *>(A.v + ai) = vv;
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            dst[k] = out[k];
            if (A.on) { A.p[ai + k] = pv[k]; A.m[ai + k] = mv[k]; A.v[ai + k] = vv[k]; }
        }
    }
}

// `tail` (optional): the second stage of the fused head's loss reduction rides in one extra workgroup (the launch then has 256
// threads per workgroup, the reduction's use the first 192)
__global__ void __launch_bounds__(256) wb_reduce_kernel(const char *__restrict__ plan, const WbRedPtrs R, const float *__restrict__ ws,
                                                        const WbAdam A, const dlwpcs_loss_tail tail, uint32_t red_blocks,
                                                        const WbPackArgs PK) {
    if (blockIdx.x >= red_blocks) {
        loss_stage2_body(tail.partial, tail.loss_out, tail.nblocks, tail.inv_n, tail.weight, tail.overwrite);
    } else if (threadIdx.x < WB_RED_THREADS) {
        const WbGroup *groups = reinterpret_cast<const WbGroup *>(plan + R.off_groups);
        const uint32_t b = blockIdx.x;
        int l = 0;
#pragma unroll
        for (int k = 1; k < WB_MAX_LAYERS; ++k) l += (k < (int)R.n_layers && b >= R.first[k]) ? 1 : 0;
        if ((R.live >> l) & 1u) {
            const WbRedLayer L = R.lay[l];          // (block-uniform index into the kernel arguments: scalar loads)
            WbAdam Al = A;
            if (!((R.adam >> l) & 1u)) Al.on = 0;
            const int blk = (int)(b - R.first[l]);
            const bool al = ((((uintptr_t)R.dw_eq[l] | (uintptr_t)R.dw_pol[l] | (uintptr_t)R.dw_np[l] | (uintptr_t)R.db_eq[l] |
                               (uintptr_t)R.db_pol[l] | (uintptr_t)R.db_np[l]) & 15) == 0);
            if (L.Cout % 4 == 0 && al)
                wb_reduce_body<4, true>(L, groups, ws, R.dw_eq[l], R.dw_pol[l], R.dw_np[l], R.db_eq[l], R.db_pol[l], R.db_np[l], blk, Al, PK.l[l]);
            else if (L.Cout % 4 == 0)       // (the plan sized this layer's workgroups for 4 outputs per thread)
                wb_reduce_body<4, false>(L, groups, ws, R.dw_eq[l], R.dw_pol[l], R.dw_np[l], R.db_eq[l], R.db_pol[l], R.db_np[l], blk, Al, PK.l[l]);
            else
                wb_reduce_body<1, true>(L, groups, ws, R.dw_eq[l], R.dw_pol[l], R.dw_np[l], R.db_eq[l], R.db_pol[l], R.db_np[l], blk, Al, PK.l[l]);
        }
    }
    // apply-only form: the step counter moves on when the last workgroup is through (every workgroup read state[0] before it
    // took its ticket; the ticket word goes back to 0 for the next launch)
    // (the waves that ran the reduction body / the loss tail; a fourth, idle wave of a reduction workgroup has nothing to wait for)
    if (A.apply_only && (threadIdx.x < WB_RED_THREADS || blockIdx.x >= red_blocks)) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t *st = const_cast<int32_t *>(A.state);
            const int done = atomicAdd(st + 1, 1);
            if (done == (int)gridDim.x - 1) { st[1] = 0; st[0] += 1; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Host: variant choice, item geometry, cost model, plan
// ------------------------------------------------------------------------------------------------------------------
static inline int wb_out_size(const dlwpcs_conv_desc *d) { return d->halo ? d->N : d->N - d->ksize + 1; }

// -1: the layer cannot be an item of the batch
static int wb_variant(const dlwpcs_conv_desc *d, int &CT, int &NT, int &cap_tile_px, int &cap_pix) {
    CT = NT = 1;
    if (d->dtype == DLWPCS_F32) {
        // exact-fp32 mode: 16-B vectors of 4 channels on both operands, one 32 x 32 tile pair per worker
        if (d->B < 1 || (d->ksize != 1 && d->ksize != 3) || (d->halo && d->ksize != 3) || (d->up0 && (d->N % 2))) return -1;
        if (!d->halo && d->N < d->ksize) return -1;
        if ((long)d->N * d->N >= (1l << 16)) return -1;
        if ((long)6 * d->N * d->N * (d->C0 > d->C1 ? d->C0 : d->C1) >= (1l << 31) || (long)6 * d->N * d->N * d->Cout >= (1l << 31)) return -1;
        cap_tile_px = 448; cap_pix = 192;
        if (d->C0 % 2 || d->C1 % 2 || d->Cout % 2 || (d->C1 > 0 && d->C0 % 4)) return -1;     // (a vector never straddles the sources)
        const bool x2 = d->C0 % 4 || d->C1 % 4, d2 = d->Cout % 4 != 0;
        if (d->ksize == 1) return x2 ? -1 : (d2 ? WB_V_F32_1_D2 : WB_V_F32_1);
        return x2 ? (d2 ? WB_V_F32_3_H2 : WB_V_F32_3_X2) : (d2 ? WB_V_F32_3_H2 : WB_V_F32_3);
    }
    if (d->dtype != DLWPCS_BF16 || d->B < 1 || (d->ksize != 1 && d->ksize != 3)) return -1;
    if (d->halo && d->ksize != 3) return -1;
    if (d->up0 && (d->N % 2)) return -1;
    if (!d->halo && d->N < d->ksize) return -1;
    if ((long)d->N * d->N >= (1l << 16)) return -1;                 // 16-bit index arithmetic (magic divisions)
    if ((long)6 * d->N * d->N * (d->C0 > d->C1 ? d->C0 : d->C1) >= (1l << 31) || (long)6 * d->N * d->N * d->Cout >= (1l << 31))
        return -1;                                                  // 32-bit element offsets inside a sample
    const int Cin = d->C0 + d->C1;
    const bool xv8 = d->C0 % 8 == 0 && d->C1 % 8 == 0;
    if (d->ksize == 3) {
        if (d->Cout % 8 != 0) return -1;
        if (xv8) {
            CT = Cin % 64 == 0 ? 2 : 1;
            NT = d->Cout % 64 == 0 ? 2 : 1;
            cap_tile_px = CT == 2 ? 320 : 512;
            cap_pix = (CT == 2 || NT == 2) ? 192 : 384;
            return CT == 2 ? (NT == 2 ? WB_V_3_8_22 : WB_V_3_8_21) : (NT == 2 ? WB_V_3_8_12 : WB_V_3_8_11);
        }
        if (d->C0 % 2 || d->C1 % 2 || Cin > 32) return -1;
        cap_pix = 384;
        if (Cin <= 16) { cap_tile_px = 512; return WB_V_3_2_8; }
        cap_tile_px = 256;
        return WB_V_3_2_16;
    }
    if (!xv8 || d->c0_valid != 0) return -1;
    if (Cin > 32) return -1;                                        // (1x1 layers wider than one ci tile: per-layer path)
    cap_tile_px = 512; cap_pix = 384;
    if (d->Cout % 8 == 0) return d->Cout <= 32 ? WB_V_1_8_11 : -1;
    if (d->Cout % 2 == 0 && d->Cout <= 16) return WB_V_1_8_D2;
    return -1;
}

struct WbGeom {
    WbLayer L;
    size_t lds;
    double cost_item[2];      // estimated cycles per item of a full / the last (possibly narrower) ci group
};

static inline int wb_cin_logical(const dlwpcs_conv_desc *d) { return (d->c0_valid > 0 ? d->c0_valid : d->C0) + d->C1; }

static int wb_geometry(const dlwpcs_conv_desc *d, bool want_bias, bool has_np, bool mask, WbGeom &G) {
    int CT, NT, cap_tile_px, cap_pix;
    const int variant = wb_variant(d, CT, NT, cap_tile_px, cap_pix);
    if (variant < 0) return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: layer (N=%d C0=%d C1=%d Cout=%d k=%d dtype=%d) has no batched kernel",
                                 d->N, d->C0, d->C1, d->Cout, d->ksize, d->dtype);
    const int KS = d->ksize, No = wb_out_size(d), face_pix = No * No, W2 = No + KS - 1;
    int CAP = cap_pix;
    auto pix_of = [&](int cap) { int p = cap; if (No <= cap) p = (cap / No) * No; if (p > face_pix) p = face_pix; return p; };
    while (CAP > 16 && (long)(tile_rows_for(pix_of(CAP), No) + KS - 1) * W2 > cap_tile_px) CAP -= (CAP > 96 ? 96 : 16);
    const int pix = pix_of(CAP);
    const int rows = tile_rows_for(pix, No) + KS - 1;
    if ((long)rows * W2 > cap_tile_px || pix < 1)
        return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: a row of face size %d exceeds the producers' register capacity", No);
    WbLayer &L = G.L;
    memset(&L, 0, sizeof(L));
    L.B = d->B; L.Nin = d->N; L.No = No; L.C0 = d->C0; L.C1 = d->C1; L.Cin = d->C0 + d->C1; L.Cout = d->Cout; L.up0 = d->up0;
    L.halo = d->halo; L.KS = KS; L.W2 = W2; L.tile_rows_max = rows; L.pix = pix; L.nbands = ceil_div(face_pix, pix);
    const bool f32 = variant >= WB_V_F32_3;
    L.pix_cap = f32 ? (pix + 1) & ~1 : (pix + 15) & ~15; L.variant = variant;
    if ((long)d->B * 4 * L.nbands >= (1l << 16))
        return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: %ld work items per face class exceed the 16-bit item arithmetic", (long)d->B * 4 * L.nbands);
    L.magicW2 = div_magic(W2); L.magicNo = div_magic(No); L.magicN = div_magic(d->N);
    L.magicB = d->B > 1 ? div_magic(d->B) : 0; L.magicNb = L.nbands > 1 ? div_magic(L.nbands) : 0;
    L.CT = CT; L.NT = NT;
    L.ncit = ceil_div(L.Cin, 32 * CT); L.ncot = ceil_div(d->Cout, 32 * NT);
    L.cin_logical = wb_cin_logical(d); L.flip = d->flip_north_pole ? 1 : 0; L.want_bias = want_bias ? 1 : 0;
    L.has_np = has_np ? 1 : 0;
    if (mask) {
        if (KS != 3) return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: act' on load is built for the 3x3 kernels (pass a pre-masked dz)");
        if (d->act != DLWPCS_ACT_LEAKY_CLIP || !(d->alpha >= 0.f) || !(d->vmax >= 0.f))
            return fail(DLWPCS_E_INVALID, "wgrad_batch: item with y needs act = LEAKY_CLIP with negative_slope >= 0 and max_value >= 0");
        L.mask = 1; L.alpha = d->alpha; L.vmax = d->vmax; L.pad0 = f32 ? 0 : (int32_t)bf16_mask_threshold(d->vmax);
    }
    const int TAPS = KS * KS;
    L.slot_floats = (int)align_up((size_t)TAPS * 32 * CT * 32 * NT + 32 * NT, 64);
    if (f32) {
        // fp32 tiles: 128 B per X pixel and per dZ pixel, two buffers; the epilogue's scratch (3 taps x 4 waves x 4 KiB + the
        // bias partials) aliases them.  The consumers set the period: 9 (1) v_mfma_f32_32x32x2 of 64 cycles per pixel pair.
        const size_t buf = ((size_t)rows * W2 + L.pix_cap) * 128;
        const size_t need = (size_t)(TAPS >= 3 ? 3 : 1) * 4 * 4096 + 4096;
        G.lds = 2 * buf > need ? 2 * buf : need;
        if (G.lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: LDS tile of %zu bytes exceeds 160 KiB", G.lds);
        // Cost model (ticks per item), fitted to the per-segment s_memtime accounting of tools/wb_segtime.py on MI355X (the
        // eleven layers of unet2 at B = 32, +-3 %): the consumers set the period -- ~73 per v_mfma_f32_32x32x2 (64 is the pipe's
        // own time; 85 before the LDS reads of a step were front-loaded behind its first five MFMAs) plus ~4300 per item
        // (barrier, first fragments), ~6500 for a 1x1 kernel whose few MFMAs no longer hide the producers; act' on load adds
        // ~7.5 per pixel.  The chains are cut at equal cost: a layer whose items run x % over the
        // model makes the whole launch x % longer.
        const double cm[4] = {73.0, 4300.0, 6500.0, 7.5};
        const int nsteps = L.pix_cap / 2, S = (ceil_div(nsteps, 4) + 1) & ~1;
        const double cost = (double)S * TAPS * cm[0] + (KS == 1 ? cm[2] : cm[1]) + (mask ? cm[3] * pix : 0.0);
        G.cost_item[0] = G.cost_item[1] = cost;
        return DLWPCS_OK;
    }
    const size_t buf = (size_t)CT * cap_tile_px * 64 + (size_t)NT * cap_pix * 64;      // planes as large as the producers' slots (wb_segment)
    const int nph = 4 / (CT * NT), rslots = TAPS - TAPS / nph;
    const size_t need = ((size_t)4 * rslots * 1024 + 2048) * 4;
    G.lds = 2 * buf > need ? 2 * buf : need;
    if (G.lds > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: LDS tile of %zu bytes exceeds 160 KiB", G.lds);
    // Cost model (cycles per item and CU), fitted to the s_memtime accounting of tools/wb_timing.py on MI355X (eleven layer
    // shapes of the U-Net, +-7 %): when the producers set the period an item costs ~3300 cycles plus its bytes at 23 B/clk
    // (plus ~45 per 4-B load instruction of a thread); when the consumers do, ~530 cycles per 16-pixel slab of 9 taps (260
    // for a 1x1 kernel; the slab count per wave is rounded up to even) plus ~1200.
    // (round 5: re-swept for the new producers on a side build, tools/wb_sweep.sh, three runs per setting, twice: bytes cheaper -- 35
    // instead of 23 B/clk --, a slab and the consumers' fixed part dearer -- 570 / 2000 instead of 530 / 1200: 145.4-147.0 against
    // 152.7-156.4 us on the same box, step 0.653 -> 0.644 ms; settings that left a worker without a segment were not considered)
    double fix = 3300.0, bpc = 35.0, slab3 = 570.0, slab1 = 260.0, ld4 = 45.0, cfix = 2000.0;
#ifdef DLWPCS_WB_TUNE_ENV       // (side builds only: the six constants from the environment, tools/wb_sweep.sh)
    if (const char *e = getenv("DLWPCS_WB_COST")) sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", &fix, &bpc, &slab3, &slab1, &ld4, &cfix);
#endif
    const int nslab = L.pix_cap / 16;
    const int S = ((ceil_div(nslab, nph)) + 1) & ~1;
    const double mma = (double)S * (KS == 3 ? slab3 : slab1) + cfix;
    const bool x4 = variant == WB_V_3_2_8 || variant == WB_V_3_2_16, d4 = variant == WB_V_1_8_D2;
    for (int last = 0; last < 2; ++last) {
        int cin_grp = 32 * CT;
        if (last) { cin_grp = L.Cin - (L.ncit - 1) * 32 * CT; }
        const double xbytes = (double)rows * W2 * cin_grp * 2.0;
        const double dbytes = (double)pix * (d->Cout < 32 * NT ? d->Cout : 32 * NT) * 2.0 * (mask ? 2.0 : 1.0);
        double ld = fix + (xbytes + dbytes) / bpc;
        if (x4) ld += ld4 * ((double)rows * W2 * (variant == WB_V_3_2_8 ? 8 : 16) / 256.0);
        // a plane that holds channels of BOTH sources of a concatenation (C0 no multiple of 32: the 10 + 2 channel input of the
        // reference scripts' production model) takes two loads per vector, one of them a no-op (wb_segment, `straddle`): measured
        // 42.5 us for that layer alone against 35.2 for the same 12 channels in one source -- and 183 against 131 us for the whole
        // list while the plan priced it like a single-source layer (its workers ran long, everybody else waited): priced at four times
        // its load term now
        if (x4 && d->C1 > 0 && d->C0 % 32 != 0) ld = 4.0 * ld - 3.0 * fix;      // (swept 1 .. 6 on that list: 188 / 166 / 152 / 150 / 152 / 150 us)
        if (d4) ld += ld4 * ((double)pix * 8 / 256.0);
        if (mask) ld += 190.0 * ((double)pix * 4 * NT / 256.0);     // act' arithmetic + the second load stream (measured: 5.85 k -> 7.9 k)
        G.cost_item[last] = ld > mma ? ld : mma;
    }
    return DLWPCS_OK;
}

static double wb_seg_overhead() {
#ifdef DLWPCS_WB_TUNE_ENV
    if (const char *e = getenv("DLWPCS_WB_SEG")) return atof(e);
#endif
    return 20000.0;
}

struct WbPlanOut {
    WbHeader H;
    std::vector<WbLayer> layers;
    std::vector<WbSeg> segs;
    std::vector<WbGroup> groups;
};

static int wb_build(const dlwpcs_wgrad_item *items, int n, int n_workers, WbPlanOut &P) {
    if (n < 1 || n > WB_MAX_LAYERS) return fail(DLWPCS_E_INVALID, "wgrad_batch: %d items (1..%d)", n, WB_MAX_LAYERS);
    if (n_workers < 1 || n_workers > 256) return fail(DLWPCS_E_INVALID, "wgrad_batch: %d workers", n_workers);
    std::vector<WbGeom> G(n);
    size_t lds = 0;
    struct Grp { int layer, cit, cot, cls; long items; double cost_item; };
    std::vector<Grp> grps;
    double total = 0;
    int group_base = 0;
    uint32_t red_blocks = 0;
    memset(&P.H, 0, sizeof(P.H));
    for (int l = 0; l < n; ++l) {
        const dlwpcs_wgrad_item &it = items[l];
        const bool want_bias = it.db_eq || it.db_pol || it.db_np;
        int rc = wb_geometry(&it.d, want_bias, it.dw_np != nullptr, it.y != nullptr, G[l]);
        if (rc) return rc;
        if ((it.db_np != nullptr) != (it.dw_np != nullptr) && it.db_eq)
            return fail(DLWPCS_E_INVALID, "wgrad_batch: item %d: dw_np / db_np must match", l);
        WbLayer &L = G[l].L;
        L.group_base = group_base;
        group_base += L.ncit * L.ncot * 3;
        if (G[l].lds > lds) lds = G[l].lds;
        for (int cit = 0; cit < L.ncit; ++cit)
            for (int cot = 0; cot < L.ncot; ++cot)
                for (int cls = 0; cls < 3; ++cls) {
                    Grp g{l, cit, cot, cls, (long)L.B * (cls == 0 ? 4 : 1) * L.nbands, G[l].cost_item[cit == L.ncit - 1 ? 1 : 0]};
                    grps.push_back(g);
                    total += g.items * g.cost_item;
                }
        const int nout = L.KS * L.KS * L.cin_logical * L.Cout + (want_bias ? L.Cout : 0);
        P.H.red_first[l] = red_blocks;
        red_blocks += (uint32_t)ceil_div(L.Cout % 4 == 0 ? nout / 4 : nout, WB_RED_OUT);
    }
    for (int l = n; l <= WB_MAX_LAYERS; ++l) P.H.red_first[l] = red_blocks;
    // Equal-cost chains: walk the groups in order and cut at worker boundaries; every piece of a group costs its items plus a
    // fixed per-segment overhead (first loads, epilogue).  The smallest per-worker budget that fits n_workers chains is found
    // by bisection (the greedy walk is monotone in the budget).
    const double seg_ovh = wb_seg_overhead();
    std::vector<uint32_t> seg_start(257, 0);
    auto pack = [&](double target, bool emit) -> bool {
        if (emit) P.segs.clear();
        int w = 0;
        double acc = 0;
        uint32_t nseg = 0;
        if (emit) seg_start[0] = 0;
        for (size_t gi = 0; gi < grps.size(); ++gi) {
            const Grp &g = grps[gi];
            long done = 0;
            while (done < g.items) {
                long take = (long)((target - acc - seg_ovh) / g.cost_item);
                if (take < 1) {
                    if (acc > 0) {                                  // this worker is full: open the next chain
                        if (++w >= n_workers) return false;
                        if (emit) seg_start[w] = nseg;
                        acc = 0;
                        continue;
                    }
                    take = 1;                                       // (a single item above the budget)
                }
                if (take > g.items - done) take = g.items - done;
                if (emit) {
                    WbSeg s{};
                    s.layer = g.layer; s.cls = g.cls; s.cit = g.cit; s.cot = g.cot;
                    s.t_first = (int)done; s.t_last = (int)(done + take);
                    s.bias = (G[g.layer].L.want_bias && g.cit == 0) ? 1 : 0;
                    P.segs.push_back(s);
                }
                ++nseg;
                acc += seg_ovh + take * g.cost_item;
                done += take;
            }
        }
        if (emit) for (int k = w + 1; k <= 256; ++k) seg_start[k] = nseg;
        return true;
    };
    double lo = total / n_workers, hi = total + (double)grps.size() * seg_ovh + 1.0;
    for (int it = 0; it < 40 && hi - lo > 1.0; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (pack(mid, false)) hi = mid; else lo = mid;
    }
    if (!pack(hi, true)) return fail(DLWPCS_E_INVALID, "wgrad_batch: internal: the work list does not fit %d workers", n_workers);
    // slots: the segments of a group are consecutive -> one (offset, stride, count) triple per group
    P.groups.assign((size_t)group_base, WbGroup{0, 0, 0, 0});
    uint64_t off = 0;
    for (size_t si = 0; si < P.segs.size(); ++si) {
        WbSeg &s = P.segs[si];
        const WbLayer &L = G[s.layer].L;
        WbGroup &g = P.groups[(size_t)L.group_base + (s.cit * L.ncot + s.cot) * 3 + s.cls];
        if (g.count == 0) { g.off = (uint32_t)off; g.stride = (uint32_t)L.slot_floats; }
        s.slot_off = (uint32_t)off;
        g.count += 1;
        off += (uint64_t)L.slot_floats;
        if (off >= (1ull << 32)) return fail(DLWPCS_E_UNSUPPORTED, "wgrad_batch: partial sums exceed 16 GiB");
    }
    P.layers.resize(n);
    for (int l = 0; l < n; ++l) P.layers[l] = G[l].L;
    WbHeader &H = P.H;
    H.magic = WB_MAGIC; H.n_layers = (uint32_t)n; H.n_segments = (uint32_t)P.segs.size(); H.n_workers = (uint32_t)n_workers;
    H.n_groups = (uint32_t)group_base; H.lds_bytes = (uint32_t)lds;
    H.off_layers = (uint32_t)align_up(sizeof(WbHeader), 256);
    H.off_segs = (uint32_t)align_up(H.off_layers + sizeof(WbLayer) * n, 256);
    H.off_groups = (uint32_t)align_up(H.off_segs + sizeof(WbSeg) * P.segs.size(), 256);
    H.total_bytes = (uint32_t)align_up(H.off_groups + sizeof(WbGroup) * P.groups.size(), 256);
    H.ws_floats = off;
    memcpy(H.seg_start, seg_start.data(), sizeof(H.seg_start));
    return DLWPCS_OK;
}

static int wb_workers() { return 256; }
#ifdef DLWPCS_WB_TL
static long long *g_tl_last = nullptr;
#endif

// upper bound of the plan size that does not depend on where the cuts fall: every group is cut at most once per worker
static size_t wb_plan_bound(int n_groups, int n_layers, int n_workers) {
    return align_up(sizeof(WbHeader), 256) + align_up(sizeof(WbLayer) * n_layers, 256) +
           align_up(sizeof(WbSeg) * (size_t)(n_groups + n_workers + 8), 256) + align_up(sizeof(WbGroup) * (size_t)n_groups, 256) + 256;
}

}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" int dlwpcs_wgrad_batch_supported(const dlwpcs_conv_desc *d) {
    if (!d) return 0;
    int CT, NT, a, b;
    if (wb_variant(d, CT, NT, a, b) < 0) return 0;
    WbGeom G;
    const int rc = wb_geometry(d, true, false, false, G);
    return rc == DLWPCS_OK ? 1 : 0;
}

extern "C" int dlwpcs_wgrad_batch_sizes(const dlwpcs_wgrad_item *items, int n_items, size_t *plan_bytes, size_t *workspace_bytes) {
    if (!items || !plan_bytes || !workspace_bytes) return fail(DLWPCS_E_INVALID, "wgrad_batch_sizes: null pointer");
    WbPlanOut P;
    const int rc = wb_build(items, n_items, wb_workers(), P);
    if (rc) return rc;
    *plan_bytes = wb_plan_bound((int)P.H.n_groups, n_items, wb_workers());
    *workspace_bytes = align_up((size_t)P.H.ws_floats * 4, 256);
    return DLWPCS_OK;
}

extern "C" int dlwpcs_wgrad_batch_plan(const dlwpcs_wgrad_item *items, int n_items, void *plan_host, size_t plan_bytes) {
    if (!items || !plan_host) return fail(DLWPCS_E_INVALID, "wgrad_batch_plan: null pointer");
    WbPlanOut P;
    const int rc = wb_build(items, n_items, wb_workers(), P);
    if (rc) return rc;
    if (plan_bytes < P.H.total_bytes) return fail(DLWPCS_E_WORKSPACE, "wgrad_batch_plan: plan buffer %zu < %u bytes", plan_bytes, P.H.total_bytes);
    char *out = (char *)plan_host;
    memset(out, 0, plan_bytes);
    memcpy(out, &P.H, sizeof(WbHeader));
    memcpy(out + P.H.off_layers, P.layers.data(), sizeof(WbLayer) * P.layers.size());
    memcpy(out + P.H.off_segs, P.segs.data(), sizeof(WbSeg) * P.segs.size());
    memcpy(out + P.H.off_groups, P.groups.data(), sizeof(WbGroup) * P.groups.size());
    return DLWPCS_OK;
}

static int wgrad_batch_impl(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                            void *workspace, size_t workspace_bytes, const WbAdam &adam, int32_t *adam_state, dlwpcs_stream_t stream,
                            const dlwpcs_loss_tail *tail = nullptr, const dlwpcs_pack_item *pack_host = nullptr);

extern "C" int dlwpcs_wgrad_batch(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                  void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream) {
    WbAdam none{};
    return wgrad_batch_impl(items, n_items, plan_host, plan_dev, workspace, workspace_bytes, none, nullptr, stream);
}

static int wgrad_batch_adam_impl(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                 void *workspace, size_t workspace_bytes, float *p, float *g, float *m, float *v, size_t n,
                                 int32_t *state_dev, const float *hyper_dev, dlwpcs_stream_t stream, const dlwpcs_loss_tail *tail,
                                 const dlwpcs_pack_item *pack_host);

extern "C" int dlwpcs_wgrad_batch_adam(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                       void *workspace, size_t workspace_bytes, float *p, float *g, float *m, float *v, size_t n,
                                       int32_t *state_dev, const float *hyper_dev, dlwpcs_stream_t stream) {
    return wgrad_batch_adam_impl(items, n_items, plan_host, plan_dev, workspace, workspace_bytes, p, g, m, v, n, state_dev, hyper_dev,
                                 stream, nullptr, nullptr);
}

extern "C" int dlwpcs_wgrad_batch_adam_tail(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                            void *workspace, size_t workspace_bytes, float *p, float *g, float *m, float *v, size_t n,
                                            int32_t *state_dev, const float *hyper_dev, const dlwpcs_loss_tail *tail,
                                            const dlwpcs_pack_item *pack_items_host, dlwpcs_stream_t stream) {
    if (tail && (!tail->partial || !tail->loss_out || tail->nblocks < 1))
        return fail(DLWPCS_E_INVALID, "wgrad_batch_adam_tail: bad loss tail");
    return wgrad_batch_adam_impl(items, n_items, plan_host, plan_dev, workspace, workspace_bytes, p, g, m, v, n, state_dev, hyper_dev,
                                 stream, tail, pack_items_host);
}

static int wgrad_batch_adam_impl(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                 void *workspace, size_t workspace_bytes, float *p, float *g, float *m, float *v, size_t n,
                                 int32_t *state_dev, const float *hyper_dev, dlwpcs_stream_t stream, const dlwpcs_loss_tail *tail,
                                 const dlwpcs_pack_item *pack_host) {
    if (!items || !p || !g || !m || !v || !state_dev || !hyper_dev) return fail(DLWPCS_E_INVALID, "wgrad_batch_adam: null pointer");
    // every destination must lie inside g, no destination may be named twice (a layer applied twice takes the unfused path)
    for (int l = 0; l < n_items; ++l) {
        const void *ds[6] = {items[l].dw_eq, items[l].dw_pol, items[l].dw_np, items[l].db_eq, items[l].db_pol, items[l].db_np};
        for (int k = 0; k < 6; ++k)
            if (ds[k] && ((const float *)ds[k] < g || (const float *)ds[k] >= g + n))
                return fail(DLWPCS_E_INVALID, "wgrad_batch_adam: item %d: gradient tensor outside the flat gradient buffer", l);
        // (items that share their gradients -- a layer applied twice, integration_steps = 2 in the reference scripts -- are reduced in
        // successive launches; the optimizer consumes a destination in the launch of its LAST item, see wgrad_batch_impl)
        for (int k = 0; k < l; ++k)
            if (items[k].dw_eq == items[l].dw_eq &&
                (items[k].dw_pol != items[l].dw_pol || items[k].dw_np != items[l].dw_np || items[k].db_eq != items[l].db_eq ||
                 items[k].db_pol != items[l].db_pol || items[k].db_np != items[l].db_np))
                return fail(DLWPCS_E_INVALID, "wgrad_batch_adam: items %d and %d share some of their gradient tensors but not all", k, l);
    }
    WbAdam A{};
    A.p = p; A.g = g; A.m = m; A.v = v; A.state = state_dev; A.hyper = hyper_dev; A.on = 1;
    return wgrad_batch_impl(items, n_items, plan_host, plan_dev, workspace, workspace_bytes, A, state_dev, stream, tail, pack_host);
}

extern "C" int dlwpcs_wgrad_batch_apply(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                        float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev, const float *hyper_dev,
                                        const dlwpcs_loss_tail *tail, const dlwpcs_pack_item *pack_items_host, dlwpcs_stream_t stream) {
    if (!items || !p || !g || !m || !v || !state_dev || !hyper_dev) return fail(DLWPCS_E_INVALID, "wgrad_batch_apply: null pointer");
    if (tail && (!tail->partial || !tail->loss_out || tail->nblocks < 1)) return fail(DLWPCS_E_INVALID, "wgrad_batch_apply: bad loss tail");
    for (int l = 0; l < n_items; ++l) {
        const void *ds[6] = {items[l].dw_eq, items[l].dw_pol, items[l].dw_np, items[l].db_eq, items[l].db_pol, items[l].db_np};
        for (int k = 0; k < 6; ++k)
            if (ds[k] && ((const float *)ds[k] < g || (const float *)ds[k] >= g + n))
                return fail(DLWPCS_E_INVALID, "wgrad_batch_apply: item %d: gradient tensor outside the flat gradient buffer", l);
    }
    WbAdam A{};
    A.p = p; A.g = g; A.m = m; A.v = v; A.state = state_dev; A.hyper = hyper_dev; A.on = 1; A.apply_only = 1;
    return wgrad_batch_impl(items, n_items, plan_host, plan_dev, nullptr, 0, A, nullptr, stream, tail, pack_items_host);
}

static int wgrad_batch_impl(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                            void *workspace, size_t workspace_bytes, const WbAdam &adam, int32_t *adam_state, dlwpcs_stream_t stream,
                            const dlwpcs_loss_tail *tail, const dlwpcs_pack_item *pack_host) {
    const bool apply_only = adam.apply_only != 0;
    if (!items || !plan_host || !plan_dev || (!workspace && !apply_only)) return fail(DLWPCS_E_INVALID, "wgrad_batch: null pointer");
    const WbHeader *H = (const WbHeader *)plan_host;
    if (H->magic != WB_MAGIC || (int)H->n_layers != n_items) return fail(DLWPCS_E_INVALID, "wgrad_batch: plan does not match the items");
    if (!apply_only && workspace_bytes < H->ws_floats * 4)
        return fail(DLWPCS_E_WORKSPACE, "wgrad_batch: workspace %zu < %llu bytes", workspace_bytes, (unsigned long long)H->ws_floats * 4);
    const WbLayer *layers = (const WbLayer *)((const char *)plan_host + H->off_layers);
    WbPtrs ptrs{};
    WbRedPtrs R{};
    double flops = 0, bytes = 0;
    for (int l = 0; l < n_items; ++l) {
        const dlwpcs_wgrad_item &it = items[l];
        const WbLayer &L = layers[l];
        if (L.B != it.d.B || L.Nin != it.d.N || L.C0 != it.d.C0 || L.C1 != it.d.C1 || L.Cout != it.d.Cout || L.KS != it.d.ksize ||
            L.halo != it.d.halo || L.up0 != it.d.up0)
            return fail(DLWPCS_E_INVALID, "wgrad_batch: item %d differs from the plan", l);
        if (!it.dw_eq || !it.dw_pol || (!apply_only && (!it.src0 || !it.dz || (it.d.C1 > 0 && !it.src1) || (it.d.halo && !it.table_dev))))
            return fail(DLWPCS_E_INVALID, "wgrad_batch: item %d: null pointer", l);
        if ((L.want_bias != 0) != (it.db_eq || it.db_pol || it.db_np) || (L.has_np != 0) != (it.dw_np != nullptr) ||
            (!apply_only && (L.mask != 0) != (it.y != nullptr)))
            return fail(DLWPCS_E_INVALID, "wgrad_batch: item %d: bias / north-pole / y pointers differ from the plan", l);
        ptrs.y[l] = it.y;
        ptrs.src0[l] = it.src0; ptrs.src1[l] = it.d.C1 > 0 ? it.src1 : it.src0; ptrs.dz[l] = it.dz; ptrs.table[l] = it.table_dev;
        R.dw_eq[l] = (float *)it.dw_eq; R.dw_pol[l] = (float *)it.dw_pol; R.dw_np[l] = (float *)it.dw_np;
        R.db_eq[l] = (float *)it.db_eq; R.db_pol[l] = (float *)it.db_pol; R.db_np[l] = (float *)it.db_np;
        const double n0 = L.up0 ? L.Nin / 2 : L.Nin;
        flops += 2.0 * L.B * 6 * (double)L.No * L.No * L.KS * L.KS * L.cin_logical * L.Cout;
        bytes += (it.d.dtype == DLWPCS_F32 ? 4.0 : 2.0) * L.B * 6.0 * (n0 * n0 * L.C0 + (double)L.Nin * L.Nin * L.C1 + (double)L.No * L.No * L.Cout) +
                 4.0 * L.KS * L.KS * L.cin_logical * L.Cout;
    }
    memcpy(R.first, H->red_first, sizeof(R.first));
    R.n_layers = H->n_layers;
    R.off_groups = H->off_groups;
    for (int l = 0; l < n_items; ++l) {
        const WbLayer &L = layers[l];
        R.lay[l] = WbRedLayer{L.KS, L.cin_logical, L.Cout, L.want_bias, 32 * L.CT, 32 * L.NT, L.ncot, L.group_base, L.flip};
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = H->lds_bytes;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)wgrad_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "wgrad_batch: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    int pidx = -1;
    int rc = DLWPCS_OK;
    if (!apply_only) {
        if (prof_enabled()) pidx = prof_begin("wgrad_batch_kernel", flops, bytes, s);
        long long *dbg = nullptr;
#ifdef DLWPCS_WB_TL
        {
            static long long *g_tl = nullptr;
            const size_t nb = (size_t)TL_WORKERS * 2 * TL_MAX * sizeof(long long);
            if (!g_tl && hipMalloc((void **)&g_tl, nb) != hipSuccess) g_tl = nullptr;
            if (g_tl) (void)hipMemsetAsync(g_tl, 0, nb, s);
            dbg = g_tl;
            g_tl_last = g_tl;
        }
#endif
        hipLaunchKernelGGL(wgrad_batch_kernel, dim3(H->n_workers), dim3(512), lds, s, (const char *)plan_dev, ptrs, (float *)workspace, dbg,
                           adam_state);
        if (pidx >= 0) prof_end(pidx, s);
        rc = check_launch("wgrad_batch");
        if (rc) return rc;
    }
    // reduction rounds: items that share their destination (a layer applied twice) go into successive launches
    uint32_t pending = n_items >= 32 ? 0xffffffffu : ((1u << n_items) - 1u);
    // last[l]: no later item adds to the destination of item l
    uint32_t last = 0;
    for (int l = 0; l < n_items; ++l) {
        bool later = false;
        for (int k = l + 1; k < n_items; ++k) later |= items[k].dw_eq == items[l].dw_eq;
        if (!later) last |= 1u << l;
    }
    if (apply_only) pending &= last;            // (nothing is added: every destination is consumed once)
    while (pending) {
        uint32_t live = 0;
        for (int l = 0; l < n_items; ++l) {
            if (!((pending >> l) & 1u)) continue;
            bool clash = false;
            for (int k = 0; k < l; ++k)
                if (((live >> k) & 1u) && items[k].dw_eq == items[l].dw_eq) clash = true;
            if (!clash) live |= 1u << l;
        }
        R.live = live;
        R.adam = live & last;
        const bool final_round = (pending & ~live) == 0;
        pidx = -1;
        if (prof_enabled()) pidx = prof_begin(apply_only ? "wb_reduce_kernel(apply)" : "wb_reduce_kernel", 0.0, (double)H->ws_floats * 4.0, s);
        const uint32_t rb = H->red_first[WB_MAX_LAYERS];
        dlwpcs_loss_tail tl{};
        if (tail) tl = *tail;
        WbPackArgs PK{};
        if (pack_host && adam.on) {
            for (int l = 0; l < n_items; ++l) {
                const dlwpcs_pack_item &pk = pack_host[l];
                const WbLayer &L = layers[l];
                // the packed operands must belong to the parameters this item's gradients update
                const float *pw = adam.p + ((const float *)items[l].dw_eq - adam.g);
                const bool pf32 = items[l].d.dtype == DLWPCS_F32;
                if (pk.dtype != items[l].d.dtype || pk.ksize != L.KS || pk.Cin != L.cin_logical || pk.Cout != L.Cout ||
                    (const float *)pk.w_eq != pw || !pk.wpk_fwd || !pk.wpk_bwd || (pk.flip_north_pole != 0) != (L.flip != 0) ||
                    (L.want_bias && !pk.bias_pk))
                    return fail(DLWPCS_E_INVALID, "wgrad_batch_adam: pack item %d does not describe the layer of gradient item %d", l, l);
                WbRedPack &K = PK.l[l];
                K.wf = (bf16_t *)pk.wpk_fwd; K.wb = (bf16_t *)pk.wpk_bwd; K.bp = (float *)pk.bias_pk;
                const int cgw = pf32 ? 8 : 16;
                K.CGf = (pk.Cin + cgw - 1) / cgw; K.NTf = (pk.Cout + 31) / 32;
                K.CGb = (pk.Cout + cgw - 1) / cgw; K.NTb = (pk.Cin + 31) / 32;
                K.f32 = pf32 ? 1 : 0;
            }
        }
        const bool with_tail = tail && final_round;     // (the loss is finished once, by the step's last launch)
        hipLaunchKernelGGL(wb_reduce_kernel, dim3(rb + (with_tail ? 1u : 0u)), dim3(with_tail ? 256 : WB_RED_THREADS), 0, s,
                           (const char *)plan_dev, R, (const float *)workspace, adam, tl, rb, PK);
        if (pidx >= 0) prof_end(pidx, s);
        pending &= ~live;
    }
    return check_launch("wgrad_batch_reduce");
}

#ifdef DLWPCS_WB_TL
// side builds only: the timeline of the last dlwpcs_wgrad_batch launch -> host (tools/wb_timeline.py)
extern "C" int dlwpcs_wb_timeline(long long *host, size_t words) {
    const size_t have = (size_t)dlwpcs::TL_WORKERS * 2 * dlwpcs::TL_MAX;
    if (!dlwpcs::g_tl_last || words < have) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(host, dlwpcs::g_tl_last, have * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? (int)have : -2;
}
#endif
