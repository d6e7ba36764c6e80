// Multi-layer persistent convolution chain (gfx950): consecutive fused cubed-sphere convolutions of a forward pass as ONE launch.
//
// Launched layer by layer, each convolution of the DLWP-CS U-Net (Azure/train_cs.py:277-305: pad -> CubeSphereConv2D -> ReLU, ten
// times per pass) pays a dependent kernel boundary, the start-up burst of 256 workgroups requesting their first tiles at once, and
// the first-tile penalty measured in DESIGN.md 9 -- about 9.5 us per launch that do not scale with the batch.  Here the layers of
// a pass are PHASES of one persistent launch:
//   * the batch is cut into sample groups (8 groups when the batch allows: one per XCD, 32 workgroups each -- the dispatcher
//     places workgroup b on XCD b % 8, which is what makes a group's traffic stay in one L2; correctness never depends on it).  A
//     layer's halo couples the six faces of ONE sample only (DLWP/custom.py:1198-1308), so layer k + 1 of a group needs nothing but
//     that group's own output of layer k (and of earlier layers: skip connections): the groups never wait for each other;
//   * inside a group the phases are separated by a GROUP barrier: an arrival counter and a generation word per group (device-scope
//     atomics; no grid-wide barrier, no cooperative launch).  Hand-over protocol (MI355X_MICROARCH.md, "inter-workgroup
//     visibility"): every output store of the convolution body is write-through (sc1) -> each wave drains its stores
//     (s_waitcnt vmcnt(0)) -> workgroup barrier -> one thread arrives, polls the generation word, and issues ONE agent-scope
//     acquire (buffer_inv sc1: this CU's vector L1 forgets everything) -> workgroup barrier -> the next phase loads plainly;
//   * every spin is BOUNDED: a barrier that does not complete within ~1 s (the GPU is shared with another process whose
//     workgroups keep ours from being co-resident) sets the abort word, every workgroup leaves, and the host finds the word set
//     (dlwpcs_conv_chain_status) -- a wrong result that is reported, never a hang.  DLWP.keras.Model keeps chains off when two
//     ranks share a device;
//   * each phase runs the SAME code as the per-layer kernel (conv_ws.h: conv_ws_body) on the same tiles in the same order of
//     arithmetic: results are bit-identical to the layer-by-layer path (tests/test_gpu_chain.py).
#include <stddef.h>
#include "conv_ws.h"

namespace dlwpcs {
static const int g_chain_tag = prof_register_tag("conv_chain_kernel");


constexpr int CHAIN_MAX_PHASES = DLWPCS_CHAIN_MAX;

struct ChainArgs {
    ConvKParams ph[CHAIN_MAX_PHASES];       // per phase, prepared for a batch of Bg samples, pointers at sample 0
    int16_t cfg[CHAIN_MAX_PHASES], gy[CHAIN_MAX_PHASES];
    int32_t nph, ngroups, Bg;
    uint32_t magicBg;
    uint32_t *sync;                         // [group][64]: word 0 = arrivals, word 32 = generation; word 8 * 64 = abort, + 1 = ticket;
                                            // from word CHAIN_FLOW_OFF on: the flow counters [phase][sample]
    uint32_t spin_limit;
    int32_t flow;                           // 1: barrier-free form (per-sample completion counters), one group of all workers
};
constexpr int CHAIN_FLOW_OFF = 1024;
static_assert(sizeof(ChainArgs) <= 4096, "chain arguments must fit the kernel-argument segment");

// One phase boundary of a sample group.  Returns false when the launch has been aborted.
__device__ __forceinline__ bool chain_group_barrier(const ChainArgs &A, int grp, int Gw, int ph, int *s_flag) {
    // the stores of this workgroup's epilogues (write-through) have left before anybody is told that the phase is complete
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t *cnt = A.sync + grp * 64, *gen = cnt + 32, *abortw = A.sync + 8 * 64;
        int ok = 1;
        const uint32_t arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        const bool last_phase = ph == A.nph - 1;
        if (arrived == (uint32_t)(ph + 1) * (uint32_t)Gw) {
            if (last_phase) {
                // everybody of the group is through its last phase: the words go back to zero for the next launch
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gen, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(gen, (uint32_t)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (!last_phase) {
            uint32_t n = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)(ph + 1)) {
                __builtin_amdgcn_s_sleep(8);
                if ((++n & 63u) == 0 && __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (n > A.spin_limit) { __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
        }
        // acquire: this CU's vector L1 forgets everything (buffer_inv sc1) -- what the next phase reads was written by other
        // CUs during this launch; one lane per workgroup, the workgroup barrier below hands the effect to the other waves
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

// One phase = one instantiation of the convolution body as a noinline function.  What was measured on the way here (MI355X):
//   * every instantiation INLINED in one kernel behind a switch: 256 VGPRs with 165-207 spilled;
//   * one kernel per instantiation, the body inlined inside the phase loop: no spills -- but the compiler no longer unswitches the
//     tile loop on the (uniform) layer options: 54 MFMA instructions in the kernel where the per-layer kernel has 267 (five
//     specialised copies of its loop nest), 40 % more branches in the hot path; every phase ran ~20 % slower than the per-layer
//     kernel of the same layer (C96 rollout 13.5 ms against 11.1);
//   * noinline functions receiving a COPY of the parameter block: the copy lived in spilled SGPRs, ~7 us per phase lost.
// Here the function reads the block where it lives -- in the kernel-argument segment, through a constant-address-space pointer made
// uniform with readfirstlane: scalar loads on demand, like the per-layer kernel's own arguments -- and the tile loop is the top-level
// loop of its function again.  The price is the calling convention: each call saves / restores its callee-saved VGPRs through
// scratch memory (~0.3 us per phase and workgroup).
typedef const __attribute__((address_space(4))) ConvKParams *ConvKParamsK;
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int KC, int MT, int WM, int WN, bool TAIL8>
__device__ __attribute__((noinline)) void chain_phase(ConvKParamsK pk, int b0v, int lwv, int Gv, int byv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint64_t a = (uint64_t)(uintptr_t)pk;
    pk = (ConvKParamsK)(uintptr_t)(((uint64_t)(uint32_t)uni((int)(a >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)a));
    const int b0 = uni(b0v), lw = uni(lwv), G = uni(Gv), by = uni(byv);
    const ConvKParams &P = *(const ConvKParams *)pk;        // (the address space is recovered from the cast: scalar loads)
    conv_ws_body<bf16_t, 3, KC, MT, 1, WM, WN, 8, MODE_HALO, false, TAIL8, false, true>(P, smem, (uint32_t)lw, G, by, b0);
}

__global__ void __launch_bounds__(512) conv_chain_kernel(const ChainArgs A) {
    __shared__ int s_flag;
    // (flow form: one group of all workers.  Laying its workers out XCD-aware like the per-layer launch's made it slower still.)
    const int grp = (int)(blockIdx.x % (uint32_t)A.ngroups), w = (int)(blockIdx.x / (uint32_t)A.ngroups);
    const int Gw = (int)(gridDim.x / (uint32_t)A.ngroups);
    const int b0 = grp * A.Bg;
#pragma unroll 1
    for (int ph = 0; ph < A.nph; ++ph) {
        // (taking &A.ph[ph] instead would make the compiler copy all of A to scratch memory first)
        ConvKParamsK pk = (ConvKParamsK)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                         offsetof(ChainArgs, ph) + (size_t)ph * sizeof(ConvKParams));
        const int gy = A.gy[ph];
        const int G = Gw / gy, lw = w % G, by = w / G;
        switch (A.cfg[ph]) {
            case CHAIN_CFG_3_32_3141:    chain_phase<32, 3, 4, 1, false>(pk, b0, lw, G, by); break;
            case CHAIN_CFG_3_32_3122:    chain_phase<32, 3, 2, 2, false>(pk, b0, lw, G, by); break;
            case CHAIN_CFG_3_16_5114:    chain_phase<16, 5, 1, 4, false>(pk, b0, lw, G, by); break;
            case CHAIN_CFG_3_16_3141_T8: chain_phase<16, 3, 4, 1, true>(pk, b0, lw, G, by); break;
            default:                     chain_phase<32, 3, 4, 1, true>(pk, b0, lw, G, by); break;
        }
        if (A.flow) {
            // FLOW: no barrier between the phases -- a tile waits for ITS sample's previous phase only (conv_ws_body); the
            // workgroup's own waves meet here because the next phase's producers reuse the LDS buffers
            __syncthreads();
            continue;
        }
        if (!chain_group_barrier(A, grp, Gw, ph, &s_flag)) return;
    }
    if (A.flow) {
        // the last workgroup to finish puts every completion counter (and the ticket) back to zero for the next launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int s_last;
        if (threadIdx.x == 0) {
            uint32_t *ticket = A.sync + 8 * 64 + 1;
            s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
            if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last) {
            uint32_t *cnt = A.sync + CHAIN_FLOW_OFF;
            for (int i = threadIdx.x; i < A.nph * A.Bg; i += blockDim.x)
                __hip_atomic_store(cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" size_t dlwpcs_conv_chain_sync_bytes(void) { return DLWPCS_CHAIN_SYNC_BYTES; }

// number of sample groups for a batch: one per XCD when the batch allows
static int chain_groups(int B) {
    static int gmax = 0;
    if (!gmax) { const char *e = getenv("DLWPCS_CHAIN_GROUPS"); gmax = e ? atoi(e) : 8; if (gmax < 1 || gmax > 8) gmax = 8; }
    for (int g = gmax; g > 1; g >>= 1)
        if (B % g == 0) return g;
    return 1;
}

static int chain_build(const dlwpcs_chain_item *items, int n_items, ChainArgs &A, size_t &lds, double &flops, double &bytes) {
    if (!items || n_items < 1 || n_items > CHAIN_MAX_PHASES)
        return fail(DLWPCS_E_INVALID, "conv_chain: %d items (1..%d)", n_items, CHAIN_MAX_PHASES);
    const int B = items[0].d.B;
    if (B < 1) return fail(DLWPCS_E_UNSUPPORTED, "conv_chain: empty batch");
    memset(&A, 0, sizeof(A));
    static int flow_on = -1;
    // DLWPCS_CHAIN_FLOW: 0 (default) group barriers between the phases; 1: the barrier-free form -- per-(phase, sample) completion
    // counters, a worker owns one (face, band) and every M-th sample so that whole samples complete rounds before they are needed;
    // 2: the counters with the plain contiguous tile ranges.  Measured on the unet2 step (0.680 ms with per-layer launches):
    // 0.744 (barriers) / 1.07-1.09 (flow) / 0.865 (counters only); an earlier flow form that dealt single tiles sample-major (table
    // gather per tile) 1.139.  Not understood beyond "the dependencies couple the workgroups more tightly than a barrier does".
    if (flow_on < 0) { const char *e = getenv("DLWPCS_CHAIN_FLOW"); flow_on = e ? atoi(e) : 0; }
    A.flow = flow_on && (size_t)(CHAIN_FLOW_OFF + n_items * B) * 4 <= DLWPCS_CHAIN_SYNC_BYTES ? 1 : 0;
    A.ngroups = A.flow ? 1 : chain_groups(B);
    A.Bg = B / A.ngroups;
    A.magicBg = A.Bg > 1 ? div_magic((uint32_t)A.Bg) : 0;
    A.nph = n_items;
    lds = 0; flops = 0; bytes = 0;
    for (int i = 0; i < n_items; ++i) {
        const dlwpcs_chain_item &it = items[i];
        if (it.d.B != B) return fail(DLWPCS_E_INVALID, "conv_chain: item %d has batch %d, item 0 has %d", i, it.d.B, B);
        if (it.d.dtype != DLWPCS_BF16) return fail(DLWPCS_E_UNSUPPORTED, "conv_chain: item %d: bf16 layers only", i);
        dlwpcs_conv_desc dg = it.d;
        dg.B = A.Bg;                            // tiling, tile counts and magics of ONE sample group
        dg.flags |= DLWPCS_CONV_PREPACKED;
        ConvPlanOut po;
        const int rc = conv_fwd_plan(&dg, it.src0, it.src1, it.wpk_fwd, it.bias_pk, it.y, it.y_pooled, it.table_dev, &po);
        if (rc) return rc;
        if (po.cfg < 0)
            return fail(DLWPCS_E_UNSUPPORTED, "conv_chain: item %d (N=%d C0=%d C1=%d Cout=%d k=%d) has no chain phase", i, it.d.N, it.d.C0,
                        it.d.C1, it.d.Cout, it.d.ksize);
        if (256 / A.ngroups % po.gy != 0) return fail(DLWPCS_E_UNSUPPORTED, "conv_chain: item %d: %d N-tile groups do not divide a sample group's workers", i, po.gy);
        A.ph[i] = po.P;
        A.cfg[i] = (int16_t)po.cfg; A.gy[i] = (int16_t)po.gy;
        if (A.flow) {
            ConvKParams &Q = A.ph[i];
            Q.flow_phase = i; Q.flow_bmax = B;
            // arrivals that complete a sample of the PREVIOUS phase: 4 consumer waves x its tiles per sample x its N-tile groups
            Q.flow_need = i > 0 ? 4 * 6 * A.ph[i - 1].nblk_face * A.gy[i - 1] : 0;
            Q.flow_rot = flow_on == 2 ? -1 : (i * 97) % 256;    // (who gets the longer tile list moves from phase to phase)
        }
        if (po.lds + 64 > 160 * 1024) return fail(DLWPCS_E_UNSUPPORTED, "conv_chain: item %d fills the LDS (%zu bytes): no room for the barrier word", i, po.lds);
        if (po.lds > lds) lds = po.lds;
        const double No = it.d.halo ? it.d.N : it.d.N - it.d.ksize + 1, n0 = it.d.up0 ? it.d.N / 2 : it.d.N;
        const double cin = (it.d.c0_valid > 0 ? it.d.c0_valid : it.d.C0) + it.d.C1;
        flops += 2.0 * B * 6 * No * No * it.d.ksize * it.d.ksize * cin * it.d.Cout;
        bytes += 2.0 * B * 6.0 * (n0 * n0 * it.d.C0 + (double)it.d.N * it.d.N * it.d.C1 + No * No * it.d.Cout);
    }
    return DLWPCS_OK;
}

extern "C" int dlwpcs_conv_chain_supported(const dlwpcs_chain_item *items, int n_items) {
    ChainArgs A;
    size_t lds;
    double f, b;
    // (geometry only: placeholder pointers are fine, nothing is dereferenced on the host)
    return chain_build(items, n_items, A, lds, f, b) == DLWPCS_OK ? 1 : 0;
}

extern "C" int dlwpcs_conv_chain_fwd(const dlwpcs_chain_item *items, int n_items, void *sync_dev, dlwpcs_stream_t stream) {
    if (!sync_dev) return fail(DLWPCS_E_INVALID, "conv_chain_fwd: null sync buffer");
    static ChainArgs A;                          // (4 KB: not on the stack of every call; the C ABI is single-threaded per device)
    size_t lds;
    double flops, bytes;
    int rc = chain_build(items, n_items, A, lds, flops, bytes);
    if (rc) return rc;
    for (int i = 0; i < n_items; ++i) {
        const dlwpcs_chain_item &it = items[i];
        if (!it.src0 || !it.wpk_fwd || !it.y || (it.d.C1 > 0 && !it.src1) || (it.d.halo && !it.table_dev))
            return fail(DLWPCS_E_INVALID, "conv_chain_fwd: item %d: null pointer", i);
    }
    A.sync = (uint32_t *)sync_dev;
    static uint32_t spin = 0;
    if (!spin) { const char *e = getenv("DLWPCS_CHAIN_SPIN"); spin = e ? (uint32_t)strtoul(e, nullptr, 0) : 400000u; if (!spin) spin = 1; }
    A.spin_limit = spin;
    if (A.flow)
        for (int i = 0; i < n_items; ++i) {
            A.ph[i].flow_done = A.sync + CHAIN_FLOW_OFF; A.ph[i].flow_abort = A.sync + 8 * 64; A.ph[i].flow_spin = spin;
        }
    hipStream_t s = (hipStream_t)stream;
    const void *kern = (const void *)conv_chain_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv_chain: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    int pidx = -1;
    if (prof_enabled()) pidx = prof_begin("conv_chain_kernel", flops, bytes, s);
    void *kargs[] = {(void *)&A};
    hipError_t le = hipLaunchKernel(kern, dim3(256), dim3(512), kargs, lds, s);
    if (pidx >= 0) prof_end(pidx, s);
    if (le != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv_chain: %s", hipGetErrorString(le));
    return check_launch("conv_chain");
}

extern "C" int dlwpcs_conv_chain_status(const void *sync_dev, int *aborted) {
    if (!sync_dev || !aborted) return fail(DLWPCS_E_INVALID, "conv_chain_status: null pointer");
    uint32_t w = 0;
    hipError_t e = hipMemcpy(&w, (const uint32_t *)sync_dev + 8 * 64, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(DLWPCS_E_LAUNCH, "conv_chain_status: %s", hipGetErrorString(e));
    *aborted = w != 0;
    return DLWPCS_OK;
}
