/*
 * dlwpcs.h -- C ABI of the MI355X-native DLWP-CS cubed-sphere hot path (libdlwpcs.so, HIP, gfx950).
 *
 * This is the drop-in boundary described in SURVEY.md section 8(b): everything the Python layers
 * `DLWP.custom.CubeSpherePadding2D` / `DLWP.custom.CubeSphereConv2D` (and the Keras stock ops wired between them in
 * the DLWP-CS U-Net) need from the device.  Conventions:
 *
 *   - plain C types only; every tensor argument is a raw DEVICE pointer + explicit sizes.  No torch types.
 *   - the CALLER owns every buffer (inputs, outputs, weights, gradients, tables, workspace); the library allocates
 *     nothing and keeps no mutable global state (the opt-in launch profiler at the end of this header excepted), so it
 *     is re-entrant across streams and devices.
 *   - every device entry point enqueues on the caller's `stream` (a hipStream_t) and returns without synchronising.
 *   - return value: 0 = OK, <0 = error code (DLWPCS_E_*); a human-readable message for the calling thread is
 *     available from dlwpcs_last_error().  Nothing throws across the boundary.
 *   - tensors on the hot path are `channels_last`:  x[b][face][row][col][channel]  (face axis = 6,
 *     faces 0-3 equatorial going east, 4 = south pole, 5 = north pole; reference DLWP/custom.py:1057-1070).
 *     `channels_first` (B,C,6,H,W) callers convert with dlwpcs_cf_to_cl / dlwpcs_cl_to_cf.
 *   - dtype: DLWPCS_F32 -- every tensor fp32, arithmetic is exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate.
 *            DLWPCS_BF16 -- mixed precision (the reference trains under TF's AMP graph rewrite, Azure/train_cs.py:429):
 *            ACTIVATIONS and their gradients (src*, y, dy, dsrc*, x/dx of the stock ops, mse y/t/dy) are bfloat16 in
 *            HBM; PARAMETERS and their gradients (w_*, b_*, dw_*, db_*, Adam state) stay fp32 master copies; the
 *            contraction runs on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16, fp32 accumulate), every
 *            elementwise kernel computes in fp32 and rounds once on store (round-to-nearest-even).
 *
 * Reference interfaces replaced (paths relative to the reference repository root):
 *   CubeSpherePadding2D.call        DLWP/custom.py:1082-1308   -> dlwpcs_halo_table, dlwpcs_pad_fwd/_bwd
 *   CubeSphereConv2D.call           DLWP/custom.py:921-1002    -> dlwpcs_conv_fwd / _bwd_data / _bwd_weights
 *   ReLU(0.1,10), AveragePooling3D((1,2,2)), UpSampling3D((1,2,2)), concatenate
 *                                   Azure/train_cs.py:197-199,277-305 -> epilogue/loader flags of dlwpcs_conv_*,
 *                                                                 dlwpcs_act_*, dlwpcs_avgpool2_*, dlwpcs_upsample2_*
 *   loss='mse', Adam()              Azure/train_cs.py:424-430  -> dlwpcs_mse_fwd_bwd, dlwpcs_adam_step, dlwpcs_adam_step_fused
 */
#ifndef DLWPCS_H
#define DLWPCS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLWPCS_VERSION 105            /* 0.1.5: dlwpcs_comm_* / dlwpcs_allreduce_f32, gather-form data gradient by default; the chain launch is gone */

/* error codes */
#define DLWPCS_OK             0
#define DLWPCS_E_INVALID     -1       /* bad argument (shape, flag, null pointer) */
#define DLWPCS_E_UNSUPPORTED -2       /* valid request this build has no kernel for */
#define DLWPCS_E_WORKSPACE   -3       /* workspace too small */
#define DLWPCS_E_LAUNCH      -4       /* HIP launch error */

/* dtype tags */
#define DLWPCS_F32  0
#define DLWPCS_BF16 1                 /* activations / activation gradients stored as bfloat16 (see "dtype" above) */

/* activation tags (epilogue of conv_fwd, mask of the backward kernels) */
#define DLWPCS_ACT_NONE        0
#define DLWPCS_ACT_LEAKY_CLIP  1      /* keras ReLU(negative_slope=alpha, max_value=vmax): Azure/train_cs.py:199 */

typedef void *dlwpcs_stream_t;        /* hipStream_t */

int dlwpcs_version(void);
const char *dlwpcs_last_error(void);

/* ------------------------------------------------------------------------------------------------------------- *
 * Halo tables (HOST functions, no device work).  Replace the slice/reverse/transpose/concat graph of
 * CubeSpherePadding2D.call (DLWP/custom.py:1082-1308) by one gather table.
 * ------------------------------------------------------------------------------------------------------------- */

/* out[6*M*M], M = N+2p: flat source index (face*N + row)*N + col of every padded cell. */
int dlwpcs_halo_table(int N, int p, int32_t *out);

/* inv[6*N*N*4]: for every source cell the up-to-4 EXTRA padded cells (flat index (face*M + i)*M + j) that read
 * it besides its own identity copy at (face, row+p, col+p); unused slots are -1.  (Fan-out <= 5) */
int dlwpcs_halo_inverse_table(int N, int p, int32_t *inv);

/* Gather form of the data gradient (p = 1, 3x3 layers; the adjoint of DLWP/custom.py:1198-1308 folded into the kernel that
 * computes the gradient, SURVEY 2a "cs_conv_dgrad ... with halo scatter folded in").  out[dlwpcs_dgrad_gather_plan_ints(N)]:
 * the inverse table of dlwpcs_halo_inverse_table(N, 1) FIRST (so the buffer serves every call that takes inv_table_dev), then
 * the plan: a data-gradient halo table (the forward's table where the neighbour contributes exactly the term the correlation
 * forms, -1 elsewhere), six source slots per border cell and two weight-id triples per face for the terms it does not (see
 * csrc/halo_table.cpp).  A data-gradient call whose inv_table_dev is such a buffer sets DLWPCS_CONV_DGRAD_GATHER.
 * dlwpcs_dgrad_gather_plan_ints returns 0 for face sizes the plan does not serve (N < 8). */
size_t dlwpcs_dgrad_gather_plan_ints(int N);
int dlwpcs_dgrad_gather_plan(int N, int32_t *out);

/* ------------------------------------------------------------------------------------------------------------- *
 * Stand-alone padding layer, channels_last.  x: (B,6,N,N,C)  y: (B,6,N+2p,N+2p,C)
 * ------------------------------------------------------------------------------------------------------------- */
int dlwpcs_pad_fwd(const void *x, void *y, int B, int N, int C, int p, int dtype,
                   const int32_t *table_dev, dlwpcs_stream_t stream);
/* dx[src] = sum of dy over every padded cell that gathered from src (deterministic inverse gather, no atomics). */
int dlwpcs_pad_bwd(const void *dy, void *dx, int B, int N, int C, int p, int dtype,
                   const int32_t *inv_table_dev, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Fused cubed-sphere convolution.
 *
 * Virtual input V (B,6,N,N,C0+C1) = concat_channels( up0 ? nearest_upsample_x2(src0) : src0 , src1 ).
 *   halo == 1 : V is halo-padded on the fly through `table` (CubeSpherePadding2D(p=(k-1)/2) fused into the load)
 *               and convolved 'valid' -> y (B,6,N,N,Cout).
 *   halo == 0 : V is consumed as is ('valid'): y (B,6,N-k+1,N-k+1,Cout).  (CubeSphereConv2D on an already padded
 *               tensor, or the 1x1 head.)
 * Weight groups follow CubeSphereConv2D.call (DLWP/custom.py:921-1002): w_eq on faces 0-3, w_pol on face 4,
 * face 5 uses w_np when given (independent_north_pole) else w_pol, with the kernel ROWS reversed when
 * flip_north_pole (== flip -> conv -> flip of the reference for stride 1).  Kernels are HWIO (k,k,Cin,Cout) fp32,
 * biases (Cout,) or NULL.  Epilogue: + bias, then `act`.
 * k in {1,3}, stride 1, dilation 1 (the hot-path configuration, Azure/train_cs.py:200-207); anything else returns
 * DLWPCS_E_UNSUPPORTED and is served by dlwpcs_gconv_*.
 * ------------------------------------------------------------------------------------------------------------- */
/* dlwpcs_conv_desc.flags */
#define DLWPCS_CONV_ACCUMULATE_WGRAD 1   /* bwd_weights ADDS to dw_* / db_* (shared layers, flat gradient buffer) */
#define DLWPCS_CONV_PREPACKED        2   /* conv_fwd: w_eq = wpk_fwd, b_eq = bias_pk (or NULL); conv_bwd_data: w_eq = wpk_bwd,
                                          * all produced by dlwpcs_pack_batch; the other kernel / bias pointers are ignored.
                                          * Without the flag every call re-packs its weights into the workspace. */
#define DLWPCS_CONV_REUSE_DZ         4   /* set on BOTH conv_bwd_weights and the conv_bwd_data call that FOLLOWS it on the
                                          * same stream with the same desc and workspace (act != NONE): where the bf16
                                          * weight-gradient kernel applies it leaves dz = dy * act'(y) in the workspace
                                          * and bwd_data reads that instead of dy and y; otherwise the flag is ignored */
#define DLWPCS_CONV_DEFER_RING0     16   /* conv_bwd_data[_masked], halo, source 0 written directly and not masked: the border
                                          * fix-up launch of source 0 is left out -- dsrc0 holds the interior contributions, the
                                          * halo ring stays in the workspace (dlwpcs_conv_ring_info) and the caller's next pass
                                          * over dsrc0 adds it (dlwpcs_avgpool2_bwd_ring).  Only where dlwpcs_conv_ring_info
                                          * returns 1 for the descriptor; ignored otherwise */
#define DLWPCS_CONV_DGRAD_GATHER    64   /* conv_bwd_data[_masked], halo, k = 3, bf16 with 16-B channel vectors, gradient arriving as dz:
                                          * inv_table_dev is a dlwpcs_dgrad_gather_plan buffer and the data gradient is computed in
                                          * GATHER form on the N x N grid -- every border cell sums the terms of the neighbouring
                                          * faces itself (extra MFMAs on gathered dz rows): no halo ring is written, no fix-up launch
                                          * follows, dlwpcs_conv_ring_info returns 0.  Ignored where the kernel does not apply. */
#define DLWPCS_CONV_OUT_PADDED      32   /* conv_fwd of the pointwise bf16 output layer (k = 1, 32 -> even C_out in 8..32, no halo): y has
                                        * C_out rounded up to a multiple of 8 channels per pixel, the padding written as zeros -- the
                                        * layout a following dlwpcs_conv_fwd takes as its source with c0_valid = C_out (an
                                        * autoregressive rollout feeds the output straight back, Azure/train_cs.py:401-406,
                                        * DLWP/model/models.py:446-454); DLWPCS_E_UNSUPPORTED for any other layer */
#define DLWPCS_CONV_DEFER_REDUCE     8   /* conv_bwd_weights: run the weight-gradient kernel only and leave the per-worker
                                          * partial sums in the workspace (dw_* / db_* are not touched); the caller keeps
                                          * that workspace untouched until it has run dlwpcs_wgrad_reduce_batch over the
                                          * item dlwpcs_conv_wgrad_reduce_item describes (one reduction launch for all
                                          * layers of a backward pass instead of one per layer) */

typedef struct dlwpcs_conv_desc {
    int32_t B;              /* batch */
    int32_t N;              /* face size of the virtual input V */
    int32_t C0, C1;         /* channels of src0 / src1 (C1 = 0: no second source) */
    int32_t Cout;
    int32_t ksize;          /* 1 or 3 */
    int32_t halo;           /* 1: fuse the cube-sphere halo gather, 0: plain 'valid' */
    int32_t up0;            /* 1: src0 is (B,6,N/2,N/2,C0), nearest-upsampled x2 on load */
    int32_t flip_north_pole;
    int32_t act;            /* DLWPCS_ACT_* */
    float   alpha, vmax;    /* parameters of DLWPCS_ACT_LEAKY_CLIP.  For 0 <= alpha <= 1 and vmax > 0 (every activation of the
                             * reference's models) the fused convolution evaluates it as min(max(x, alpha*x), vmax): identical
                             * for all finite and infinite x, and a NaN stays a NaN like in keras' ReLU (the min is
                             * gfx950's NaN-propagating v_minimum3_f32; the select form used for other alpha, by
                             * dlwpcs_act_fwd and by the oracle returns NaN as well).  alpha < 0 or vmax < 0: rejected. */
    int32_t dtype;          /* DLWPCS_F32 | DLWPCS_BF16 (dtype of src*, y, dy, dsrc*; parameters are always fp32) */
    int32_t flags;          /* DLWPCS_CONV_* bits */
    int32_t c0_valid;       /* 0: every channel of src0 is real.  > 0 (C1 must be 0): src0 is stored with C0 channels per
                             * pixel but only the first c0_valid are variables, the rest is zero padding up to the vector
                             * width (7 variables in an 8-channel layout: 16-B loads instead of scalar ones).  The HWIO
                             * kernels and their gradients keep c0_valid input rows; dsrc0 is written with C0 channels,
                             * zeros in the padding. */
} dlwpcs_conv_desc;

size_t dlwpcs_conv_workspace_bytes(const dlwpcs_conv_desc *d);      /* max over fwd / bwd_data / bwd_weights */

/* Weight packing hoisted out of the per-call path: the kernels consume the HWIO fp32 weights in matrix-core fragment
 * order (3 face variants, rounded to bf16 in DLWPCS_BF16 mode).  A model packs ALL its layers with one launch per step
 * (after the optimizer update) and passes the packed buffers with DLWPCS_CONV_PREPACKED.
 * dlwpcs_conv_packed_bytes: size of one packed buffer of the layer described by d (only ksize, C0+C1, Cout, dtype matter).
 * dlwpcs_pack_batch: items_dev is a DEVICE array of n_items descriptors (caller-owned; pointers inside are device
 * pointers); wpk_fwd / wpk_bwd / bias_pk may individually be NULL to skip that output. */
#define DLWPCS_PACK_FWD  0
#define DLWPCS_PACK_BWD  1
#define DLWPCS_PACK_BIAS 2
typedef struct dlwpcs_pack_item {
    const void *w_eq, *w_pol, *w_np;      /* HWIO fp32 kernels (w_np NULL unless independent north pole) */
    const void *b_eq, *b_pol, *b_np;      /* fp32 biases or NULL */
    void *wpk_fwd, *wpk_bwd, *bias_pk;    /* outputs */
    int32_t ksize, Cin, Cout, flip_north_pole, dtype, reserved;
} dlwpcs_pack_item;
size_t dlwpcs_conv_packed_bytes(const dlwpcs_conv_desc *d, int which);
int dlwpcs_pack_batch(const dlwpcs_pack_item *items_dev, int n_items, dlwpcs_stream_t stream);

int dlwpcs_conv_fwd(const dlwpcs_conv_desc *d,
                    const void *src0, const void *src1,
                    const void *w_eq, const void *w_pol, const void *w_np,
                    const void *b_eq, const void *b_pol, const void *b_np,
                    void *y, const int32_t *table_dev,
                    void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream);
/* dlwpcs_conv_fwd + the 2x2 average pooling of its output (AveragePooling3D((1,2,2)) behind the block's last convolution,
 * Azure/train_cs.py:282,287) as a SECOND output y_pooled (B,6,N/2,N/2,Cout): written by the convolution's epilogue where the
 * tiling allows (every consumer wave owns whole pairs of rows: the U-Net levels at N = 48 / 24), by dlwpcs_avgpool2_fwd behind
 * it otherwise -- the same bits either way.  Halo convolutions on even face sizes. */
int dlwpcs_conv_fwd_pool(const dlwpcs_conv_desc *d, const void *src0, const void *src1,
                         const void *w_eq, const void *w_pol, const void *w_np,
                         const void *b_eq, const void *b_pol, const void *b_np,
                         void *y, void *y_pooled, const int32_t *table_dev,
                         void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream);
/* Inference: dlwpcs_conv_fwd of layer `d` + the POINTWISE OUTPUT LAYER `dh` behind it (the U-Net's last 3x3 CubeSphereConv2D
 * and its 1x1 head, Azure/train_cs.py:300-305; reference layer: DLWP/custom.py:921-1002 with kernel_size 1) -- y_head =
 * conv1x1(act(conv(...)))).  Both layers take dlwpcs_pack_batch operands (DLWPCS_CONV_PREPACKED in both descriptors: wpk_fwd /
 * bias_pk and head_wpk_fwd / head_bias_pk; a bias pointer may be NULL).  Where the tiling allows -- bf16, 32 output channels of
 * `d`, a head without activation whose stored rows are 32 channels (C_out = 32, or 25..31 with DLWPCS_CONV_OUT_PADDED in dh->flags:
 * the rollout's 26 = 13 variables x 2 steps) -- the head is FOLDED INTO THE EPILOGUE of the convolution: the tile's activated
 * result, rounded to bf16 as it would have been stored, is contracted with the head's fragments straight out of the accumulators,
 * y is never written (113 MB written + 113 MB read back per pass at C96, batch 32, and a launch) and *fused = 1.  Otherwise the
 * two launches of dlwpcs_conv_fwd run (y holds the layer's output, *fused = 0).  y must be a valid (B,6,No,No,Cout) buffer either
 * way; fused may be NULL.  The folded result differs from the two-launch one only by the summation order inside the head's MFMA
 * (fp32 accumulation in both). */
int dlwpcs_conv_fwd_head(const dlwpcs_conv_desc *d, const void *src0, const void *src1, const void *wpk_fwd,
                         const void *bias_pk, const dlwpcs_conv_desc *dh, const void *head_wpk_fwd,
                         const void *head_bias_pk, void *y, void *y_head, const int32_t *table_dev,
                         void *workspace, size_t workspace_bytes, int *fused, dlwpcs_stream_t stream);

/* Gradients w.r.t. the sources.  dy: gradient w.r.t. the (post-activation) output y; y: the saved forward output
 * (needed when act != NONE: dz = dy * act'(.) is applied on load), NULL otherwise.
 * dsrc0: same shape as src0 (the 2x2 block sum of the upsample adjoint is applied when up0), dsrc1 like src1.
 * Either may be NULL to skip it (first layer).  inv_table_dev from dlwpcs_halo_inverse_table (halo==1 only). */
int dlwpcs_conv_bwd_data(const dlwpcs_conv_desc *d,
                         const void *dy, const void *y,
                         const void *w_eq, const void *w_pol, const void *w_np,
                         void *dsrc0, void *dsrc1, const int32_t *inv_table_dev,
                         void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream);

/* Pre-masked gradient convention (training steps of a network whose activations are fused into the convolutions): the
 * gradient w.r.t. the OUTPUT y of an activated layer is multiplied by act'(y) where it is PRODUCED -- by the data-gradient
 * call of the layer that consumes y, by the pooling / loss kernels -- so that the layer itself receives dz = dy * act'(y):
 * neither its data gradient nor its weight gradient reads y again, and no dz hand-over is written.
 *   dz       gradient w.r.t. this layer's PRE-activation output (act / alpha / vmax of d are ignored)
 *   m0, m1   NULL, or the source tensors themselves (src0 / src1 of the forward call: outputs of activated layers with
 *            ReLU(negative_slope = m_alpha, max_value = m_vmax)): dsrc0 / dsrc1 come out multiplied by act'(m0) / act'(m1).
 * The multiply happens in the direct-store epilogue of the matrix-core kernel and in the ring fix-up / inverse-gather
 * kernels; shapes those do not serve get one elementwise launch inside the call.  Otherwise like dlwpcs_conv_bwd_data. */
int dlwpcs_conv_bwd_data_masked(const dlwpcs_conv_desc *d, const void *dz,
                                const void *w_eq, const void *w_pol, const void *w_np,
                                void *dsrc0, void *dsrc1, const void *m0, const void *m1, float m_alpha, float m_vmax,
                                const int32_t *inv_table_dev,
                                void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream);

/* 1 if a data-gradient call on d with DLWPCS_CONV_DEFER_RING0 defers the fix-up of source 0; then *dxv_offset = byte offset of
 * the padded gradient (B,6,N+2,N+2,channels) inside that call's workspace and *channels = C0 + C1 (source 0 = the first C0). */
int dlwpcs_conv_ring_info(const dlwpcs_conv_desc *d, size_t *dxv_offset, int *channels);

/* Gradients w.r.t. kernels and biases (deterministic: fixed-order partial sums, no atomics).
 * dw_*: HWIO like the kernels; db_*: (Cout,) or NULL.  dw_np/db_np NULL unless independent north pole. */
int dlwpcs_conv_bwd_weights(const dlwpcs_conv_desc *d,
                            const void *src0, const void *src1, const void *dy, const void *y,
                            void *dw_eq, void *dw_pol, void *dw_np,
                            void *db_eq, void *db_pol, void *db_np,
                            const int32_t *table_dev,
                            void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream);

/* Deferred, batched reduction of the per-worker partial sums (DLWPCS_CONV_DEFER_REDUCE).
 * dlwpcs_conv_wgrad_reduce_item (host only, launches nothing): fills *item with what the reduction of the layer
 * described by d needs — the same d (flags included: ACCUMULATE_WGRAD decides add vs overwrite), destinations and
 * workspace as the dlwpcs_conv_bwd_weights call that produced the partials.  `nblocks` is the item's share of the launch.
 * dlwpcs_wgrad_reduce_batch: items_dev is a DEVICE copy of the n_items host items (the host copy supplies the launch
 * geometry; both must hold the same items).
 * Items whose destinations overlap (a layer applied twice) must all carry ACCUMULATE_WGRAD: items run concurrently,
 * every destination element is owned by one thread per item, so overlapping items are only safe as atomics-free adds
 * when the caller serialises them — put them in separate launches.  Bitwise reproducible like the per-layer path
 * (same fixed summation order). */
typedef struct dlwpcs_reduce_item {
    const float *partial, *bpartial;                       /* inside the layer's workspace */
    float *dw_eq, *dw_pol, *dw_np, *db_eq, *db_pol, *db_np;
    int32_t ksize, Cin, Cout, CinP, CoutP, n_eq, n_4, n_5, flip_north_pole, accumulate, vec, nblocks;
    int32_t reserved[4];
} dlwpcs_reduce_item;
int dlwpcs_conv_wgrad_reduce_item(const dlwpcs_conv_desc *d,
                                  void *dw_eq, void *dw_pol, void *dw_np,
                                  void *db_eq, void *db_pol, void *db_np,
                                  void *workspace, size_t workspace_bytes, dlwpcs_reduce_item *item);
int dlwpcs_wgrad_reduce_batch(const dlwpcs_reduce_item *items_dev, const dlwpcs_reduce_item *items_host, int n_items,
                              dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Batched weight gradients: ONE persistent launch for every layer of a backward pass (DLWPCS_BF16 hot-path shapes).
 * The weight gradient of a layer needs only the layer's saved inputs and dz, the gradient w.r.t. its PRE-activation
 * output, so nothing in the backward chain waits for it: a training step queues one item per layer and runs them
 * together after the last data gradient.  The host plan cuts the joint work list of all layers into one equal-cost
 * chain per compute unit (a layer gets workers in proportion to its work, few partial sums), the reduction adds the
 * partial sums in a fixed order (bitwise reproducible) and ACCUMULATES into dw_* / db_* (shared layers add up;
 * the caller zeroes the gradient buffer once per step).  Replaces n_items calls of dlwpcs_conv_bwd_weights.
 *
 *   dz          (B,6,No,No,Cout) bf16: dy * act'(y) already applied by whoever produced the gradient
 *               (dlwpcs_conv_bwd_data_masked, dlwpcs_avgpool2_bwd_add_masked, ...); act / alpha / vmax / flags of d are
 *               ignored
 *   plan        built on the HOST from the descriptors (and from which of the bias / north-pole pointers are non-NULL);
 *               holds geometry only -- every tensor address travels by value in the kernel arguments, so one plan (and
 *               its device copy, which the caller makes once) serves every step and a captured hipGraph keeps replaying
 *               the addresses it was captured with.
 * dlwpcs_wgrad_batch_supported: 1 if the layer described by d can be an item (others use dlwpcs_conv_bwd_weights).
 * ------------------------------------------------------------------------------------------------------------- */
#define DLWPCS_WGRAD_BATCH_MAX 24
typedef struct dlwpcs_wgrad_item {
    dlwpcs_conv_desc d;
    const void *src0, *src1;              /* the layer's saved inputs (src1 NULL when C1 == 0) */
    const void *dz;
    const void *y;                        /* NULL: dz is pre-masked.  Else (3x3 kernels, d.act = LEAKY_CLIP): `dz` holds the plain
                                           * gradient dy and y the layer's saved output; the kernel forms dy * act'(y) on load */
    const int32_t *table_dev;             /* halo table (halo == 1) */
    void *dw_eq, *dw_pol, *dw_np;         /* fp32 HWIO, accumulated into; dw_np NULL unless independent north pole */
    void *db_eq, *db_pol, *db_np;         /* fp32 (Cout,) or NULL */
} dlwpcs_wgrad_item;
int dlwpcs_wgrad_batch_supported(const dlwpcs_conv_desc *d);
int dlwpcs_wgrad_batch_sizes(const dlwpcs_wgrad_item *items, int n_items, size_t *plan_bytes, size_t *workspace_bytes);
int dlwpcs_wgrad_batch_plan(const dlwpcs_wgrad_item *items, int n_items, void *plan_host, size_t plan_bytes);
int dlwpcs_wgrad_batch(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                       void *workspace, size_t workspace_bytes, dlwpcs_stream_t stream);
/* The same with the optimizer fused into the reduction (one launch less, the gradients never travel through HBM again):
 * every dw_* / db_* must be a view into the flat gradient buffer g (n floats); p, m, v are the flat parameter / Adam-state
 * buffers with the same offsets.  A reduced element is consumed on the spot: g' = (g[i] + sum of partial sums) *
 * grad_scale -> the Adam update of dlwpcs_adam_step_dev (same arithmetic, same bits) -> g[i] = 0.  Elements of g that no
 * item covers are NOT updated (alignment padding; a caller with parameters outside the items uses dlwpcs_wgrad_batch +
 * dlwpcs_adam_step_dev instead).  state_dev = {t - 1, ticket} as for dlwpcs_adam_step_fused (the weight-gradient launch
 * increments t, the reduction reads it), hyper_dev = {lr, beta1, beta2, eps, grad_scale}.  Items that share their gradient
 * tensors (a layer applied twice: integration_steps = 2, /root/reference/Azure/train_cs.py:391-408) must share ALL of them; they are
 * reduced in successive launches and the optimizer consumes a tensor in the launch of its LAST item (round 6; rounds 3-5 rejected
 * such lists with DLWPCS_E_UNSUPPORTED and the caller ran reduction, optimizer and operand packing as launches of their own). */
int dlwpcs_wgrad_batch_adam(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                            void *workspace, size_t workspace_bytes, float *p, float *g, float *m, float *v, size_t n,
                            int32_t *state_dev, const float *hyper_dev, dlwpcs_stream_t stream);
/* ... and what else ends a training step, in the same launch:
 *  tail != NULL: the deferred second stage of the fused head's loss (DLWPCS_HEAD_DEFER_STAGE2);
 *  pack_items_host != NULL (HOST array, one dlwpcs_pack_item per gradient item, same order; bf16): every updated parameter is also
 *  written -- rounded to bf16 -- into its places in the layer's packed operands (wpk_fwd / wpk_bwd / bias_pk, the outputs of
 *  dlwpcs_pack_batch), so that the next pass needs no packing launch.  The packed buffers must hold a full dlwpcs_pack_batch
 *  result already (the zero padding is not rewritten); w_eq of pack item i must be the parameter tensor that dw_eq of gradient
 *  item i belongs to (same offset in p as dw_eq has in g). */
struct dlwpcs_loss_tail;
int dlwpcs_wgrad_batch_adam_tail(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                                 void *workspace, size_t workspace_bytes, float *p, float *g, float *m, float *v, size_t n,
                                 int32_t *state_dev, const float *hyper_dev, const struct dlwpcs_loss_tail *tail,
                                 const dlwpcs_pack_item *pack_items_host, dlwpcs_stream_t stream);

/* The data-parallel form of the step's last launch (reference counterpart: the multi-GPU model of DLWP/model/models.py:369-374,
 * one optimizer step on the gradient of the GLOBAL batch): dlwpcs_wgrad_batch leaves the finished local gradients in g, the
 * caller sums g over the ranks (RCCL all-reduce of the flat buffer), and this ONE launch then does what dlwpcs_wgrad_batch_adam_tail
 * does after its reduction -- g * grad_scale -> Adam (same arithmetic, same bits) -> g = 0, packed bf16 operands refreshed, loss
 * tail -- without reading any partial sum.  The items name the gradient tensors (dw_* / db_*: views into g) and the plan gives
 * their geometry; src / dz / table pointers are ignored.  state_dev = {t, ticket}: this launch computes with t + 1 and its last
 * workgroup stores it (the form of dlwpcs_adam_step_fused), so no earlier launch has to move the counter. */
int dlwpcs_wgrad_batch_apply(const dlwpcs_wgrad_item *items, int n_items, const void *plan_host, const void *plan_dev,
                             float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev, const float *hyper_dev,
                             const struct dlwpcs_loss_tail *tail, const dlwpcs_pack_item *pack_items_host, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Generic (any kernel size / stride / dilation / 'same') per-face convolution on an ALREADY PADDED channels_last
 * tensor: the off-hot-path options of CubeSphereConv2D (DLWP/custom.py:824-842).  Direct VALU kernels.
 * x: (B,6,H,W,Cin) -> y: (B,6,Ho,Wo,Cout);  pad_{t,l}: zero padding already resolved by the caller ('same').
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct dlwpcs_gconv_desc {
    int32_t B, H, W, Cin, Cout;
    int32_t kh, kw, sh, sw, dh, dw;
    int32_t pad_t, pad_l;   /* zero padding before the first row / column ('same'), 0 for 'valid' */
    int32_t Ho, Wo;
    int32_t flip_north_pole;
    int32_t dtype;
} dlwpcs_gconv_desc;

int dlwpcs_gconv_fwd(const dlwpcs_gconv_desc *d, const void *x,
                     const void *w_eq, const void *w_pol, const void *w_np,
                     const void *b_eq, const void *b_pol, const void *b_np, void *y, dlwpcs_stream_t stream);
int dlwpcs_gconv_bwd_data(const dlwpcs_gconv_desc *d, const void *dy,
                          const void *w_eq, const void *w_pol, const void *w_np, void *dx, dlwpcs_stream_t stream);
int dlwpcs_gconv_bwd_weights(const dlwpcs_gconv_desc *d, const void *x, const void *dy,
                             void *dw_eq, void *dw_pol, void *dw_np, void *db_eq, void *db_pol, void *db_np,
                             dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Keras stock ops between the custom layers (Azure/train_cs.py:197-199), channels_last, n = element count.
 * ------------------------------------------------------------------------------------------------------------- */
int dlwpcs_act_fwd(const void *x, void *y, size_t n, int act, float alpha, float vmax, int dtype,
                   dlwpcs_stream_t stream);
/* dx = dy * act'(.) evaluated from the saved OUTPUT y */
int dlwpcs_act_bwd(const void *dy, const void *y, void *dx, size_t n, int act, float alpha, float vmax, int dtype,
                   dlwpcs_stream_t stream);
/* x: (B,6,N,N,C) -> y: (B,6,N/2,N/2,C), 2x2 mean;  backward spreads dy/4 */
int dlwpcs_avgpool2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream);
int dlwpcs_avgpool2_bwd(const void *dy, void *dx, int B, int N, int C, int dtype, dlwpcs_stream_t stream);
/* dx = dskip + avgpool2_bwd(dy): the pooled tensor also feeds a skip connection whose gradient is dskip (N,N like dx);
 * one pass instead of avgpool2_bwd + add.  dx may alias dskip. */
int dlwpcs_avgpool2_bwd_add(const void *dy, const void *dskip, void *dx, int B, int N, int C, int dtype,
                            dlwpcs_stream_t stream);
/* Pre-masked gradients: dx = act'(m) * (dskip + avgpool2_bwd(dy)); m = the pooled tensor itself (output of an activated layer,
 * ReLU(m_alpha, m_vmax)), dskip may be NULL (no skip connection).  dx may alias dskip. */
int dlwpcs_avgpool2_bwd_masked(const void *dy, const void *dskip, const void *m, void *dx, int B, int N, int C,
                               float m_alpha, float m_vmax, int dtype, dlwpcs_stream_t stream);
/* The same, with dy = the pooled tensor's gradient as a data-gradient call with DLWPCS_CONV_DEFER_RING0 left it: the ring cells
 * (`ring` = that call's workspace + dxv_offset, `ring_channels` per cell, the pooled tensor's window starting at `ring_choff`;
 * inv_half_dev = dlwpcs_halo_inverse_table(N/2, 1)) are added to the border cells of the N/2 grid on the fly.  m may be NULL
 * here (no mask: plain pooling adjoint + fix-up). */
int dlwpcs_avgpool2_bwd_ring(const void *dy, const void *dskip, const void *m, void *dx, int B, int N, int C,
                             float m_alpha, float m_vmax, int dtype, const void *ring, const int32_t *inv_half_dev,
                             int ring_channels, int ring_choff, dlwpcs_stream_t stream);
/* x: (B,6,N,N,C) -> y: (B,6,2N,2N,C), nearest;  backward sums 2x2 blocks */
int dlwpcs_upsample2_fwd(const void *x, void *y, int B, int N, int C, int dtype, dlwpcs_stream_t stream);
int dlwpcs_upsample2_bwd(const void *dy, void *dx, int B, int N, int C, int dtype, dlwpcs_stream_t stream);
/* y[:, :Ca] = a, y[:, Ca:] = b over `rows` pixels;  split is the adjoint */
int dlwpcs_concat2(const void *a, const void *b, void *y, size_t rows, int Ca, int Cb, int dtype,
                   dlwpcs_stream_t stream);
int dlwpcs_split2(const void *y, void *a, void *b, size_t rows, int Ca, int Cb, int dtype, dlwpcs_stream_t stream);
/* Channel padding for dlwpcs_conv_desc.c0_valid: y (rows, Cp) <- x (rows, C), zeros in channels C..Cp-1; slice is the adjoint
 * (x (rows, C) <- the first C channels of y (rows, Cp)). */
int dlwpcs_pad_channels(const void *x, void *y, size_t rows, int C, int Cp, int dtype, dlwpcs_stream_t stream);
int dlwpcs_slice_channels(const void *y, void *x, size_t rows, int Cp, int C, int dtype, dlwpcs_stream_t stream);
/* Rollout state re-injection (replaces the per-step numpy concatenate / transpose / reshape of the reference's
 * TimeSeriesEstimator.predict, DLWP/model/extensions.py:281-299, and the Reshape / Permute / Concatenate chain of
 * Azure/train_cs.py:401-406): out (B,S,T*(V+E)) <- state (B,S,T*V) with extra (B,T,S,E) appended as the last E channels of
 * every one of the T time steps.  All three tensors have element type `dtype`. */
int dlwpcs_state_repack(const void *state, const void *extra, void *out, int B, size_t S, int T, int V, int E,
                        int dtype, dlwpcs_stream_t stream);
/* layout converters for data_format='channels_first' callers: (B,C,S) <-> (B,S,C), S = 6*H*W */
int dlwpcs_cf_to_cl(const void *x, void *y, int B, int C, size_t S, int dtype, dlwpcs_stream_t stream);
int dlwpcs_cl_to_cf(const void *x, void *y, int B, int C, size_t S, int dtype, dlwpcs_stream_t stream);
/* y = a + b (gradient accumulation at skip connections / shared weights) */
int dlwpcs_add(const void *a, const void *b, void *y, size_t n, int dtype, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Training-step tail (Azure/train_cs.py:424-430): keras 'mse' with a loss weight, metric 'mae', and Adam.
 * ------------------------------------------------------------------------------------------------------------- */
/* loss_out[0] += weight*mean((y-t)^2), loss_out[1] += mean(|y-t|);  dy = weight*2*(y-t)/n  (dy may be NULL).
 * scratch: >= dlwpcs_mse_scratch_bytes() bytes.  Deterministic two-stage reduction, fp32 arithmetic.
 * dtype = dtype of y and dy; t has the same dtype unless DLWPCS_MSE_TARGET_F32 is OR-ed in (bf16 prediction scored
 * against the fp32 target, as TF's AMP does: the loss is computed in fp32). */
#define DLWPCS_MSE_TARGET_F32 0x100
#define DLWPCS_MSE_OVERWRITE  0x200   /* OR-ed into dtype: loss_out[0..1] = ... instead of +=  (loss_out need not be zeroed) */
size_t dlwpcs_mse_scratch_bytes(void);
int dlwpcs_mse_fwd_bwd(const void *y, const void *t, void *dy, float *loss_out, size_t n, float weight, int dtype,
                       void *scratch, dlwpcs_stream_t stream);
/* TF2.1-keras Adam on flat fp32 buffers: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps).
 * `step_dev` points to a device int32 holding t-1 (incremented by the kernel), so the call is graph-capturable.
 * grad_scale multiplies g first (1/world_size after a sum all-reduce). */
/* Fused training tail (bf16): pointwise output layer (k = 1, 32 -> even C_out in 8..32, Azure/train_cs.py:228) + 'mse' loss
 * (+ 'mae') against the fp32 target + loss gradient dy (bf16, what the head's weight gradient reads) + the head's data
 * gradient dx (bf16, (B,6,N,N,32)) in one pass: replaces dlwpcs_conv_fwd + dlwpcs_mse_fwd_bwd + dlwpcs_conv_bwd_data of the
 * output layer; the prediction itself is not written.  wpk_fwd / bias_pk / wpk_bwd are dlwpcs_pack_batch outputs;
 * loss_out[0] = weight * mse, loss_out[1] = mae (overwrite != 0: assigned, else added); scratch >= dlwpcs_head_mse_scratch_bytes(). */
size_t dlwpcs_head_mse_scratch_bytes(void);
/* OR-ed into `overwrite`: the launch that finishes the loss (second stage of its reduction) is left to the caller --
 * dlwpcs_head_mse_tail describes it; dlwpcs_loss_tail_run runs it as a launch of its own, dlwpcs_wgrad_batch_adam_tail as one
 * extra workgroup of the weight-gradient reduction that ends the training step (same code, same bits). */
#define DLWPCS_HEAD_DEFER_STAGE2 2
typedef struct dlwpcs_loss_tail {
    const float *partial;        /* the scratch of dlwpcs_head_mse_step: [nblocks][2] workgroup sums */
    float *loss_out;
    int nblocks;
    float inv_n, weight;
    int overwrite;
} dlwpcs_loss_tail;
int dlwpcs_head_mse_tail(const dlwpcs_conv_desc *d, float weight, int overwrite, void *scratch, float *loss_out,
                         dlwpcs_loss_tail *tail);
int dlwpcs_loss_tail_run(const dlwpcs_loss_tail *tail, dlwpcs_stream_t stream);
int dlwpcs_head_mse_step(const dlwpcs_conv_desc *d, const void *x, const void *wpk_fwd, const void *bias_pk,
                         const void *wpk_bwd, const float *target, float weight, void *dy, void *dx, float *loss_out,
                         int overwrite, void *scratch, dlwpcs_stream_t stream);

/* The same with dx multiplied by act'(x; m_alpha, m_vmax): x is the output of an activated layer that expects its gradient
 * pre-masked (see dlwpcs_conv_bwd_data_masked). */
int dlwpcs_head_mse_step_masked(const dlwpcs_conv_desc *d, const void *x, const void *wpk_fwd, const void *bias_pk,
                                const void *wpk_bwd, const float *target, float weight, void *dy, void *dx, float *loss_out,
                                int overwrite, void *scratch, float m_alpha, float m_vmax, dlwpcs_stream_t stream);

int dlwpcs_adam_step(float *p, const float *g, float *m, float *v, size_t n, int32_t *step_dev,
                     float lr, float beta1, float beta2, float eps, float grad_scale, dlwpcs_stream_t stream);
/* Same update as one launch: `state_dev` points to TWO device int32 {t-1, 0}; the second is a ticket counter (must be 0
 * between calls, the kernel leaves it 0): the last workgroup to finish increments t-1, so no second launch is needed.
 * DLWPCS_ADAM_ZERO_GRAD: g[i] = 0 after it has been consumed (the next step's gradient accumulation starts from zeros
 * without a fill launch). */
#define DLWPCS_ADAM_ZERO_GRAD 1
int dlwpcs_adam_step_fused(float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev,
                           float lr, float beta1, float beta2, float eps, float grad_scale, int flags,
                           dlwpcs_stream_t stream);
/* Same kernel, hyper-parameters read from DEVICE memory: hyper_dev -> five floats {lr, beta1, beta2, eps, grad_scale}.
 * A step captured in a hipGraph then honours `optimizer.lr = x` / learning-rate schedules between replays (by-value
 * arguments are frozen at capture).  Same element arithmetic as dlwpcs_adam_step_fused -> same bits for equal values. */
int dlwpcs_adam_step_dev(float *p, float *g, float *m, float *v, size_t n, int32_t *state_dev,
                         const float *hyper_dev, int flags, dlwpcs_stream_t stream);

/* Weight regularizers / constraints of CubeSphereConv2D (DLWP/custom.py:837-842 -> add_weight(regularizer=, constraint=), :898-914).
 * dlwpcs_l1l2_regularize: keras regularizers.L1L2 on one fp32 weight: *penalty += l1 * sum|w| + l2 * sum w^2 (penalty may be NULL)
 * and g[i] += (l1 * sign(w[i]) + 2 * l2 * w[i]) * inv_grad_scale (g may be NULL; inv_grad_scale = 1 / the optimizer's grad_scale, so
 * that the penalty's gradient is not averaged over the ranks).  One workgroup, fixed summation order.
 * dlwpcs_weight_constraint: keras constraints on one fp32 weight viewed as a (rows, cols) matrix -- one norm per column, i.e. over
 * the weight's leading axes (axis=[0,1,2] of a (k,k,Cin,Cout) kernel: rows = k*k*Cin, cols = Cout) -- applied in place. */
#define DLWPCS_CONSTRAINT_MAX_NORM     1   /* a = max_value */
#define DLWPCS_CONSTRAINT_NON_NEG      2
#define DLWPCS_CONSTRAINT_UNIT_NORM    3
#define DLWPCS_CONSTRAINT_MIN_MAX_NORM 4   /* a = min_value, b = max_value, rate */
int dlwpcs_l1l2_regularize(const float *w, float *g, size_t n, float l1, float l2, float inv_grad_scale, float *penalty,
                           dlwpcs_stream_t stream);
int dlwpcs_weight_constraint(float *w, int rows, int cols, int kind, float a, float b, float rate, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Batch feed (reference ArrayDataGenerator.generate, DLWP/model/generators.py:872-984): with the whole data array
 * (T, V, S) fp32 resident in HBM (S = flattened space, e.g. 6*N*N), one launch assembles a batch window
 *   channels_last : out[b][s][c_off + n*c_stride + j] = array[samples[b] + t_off + n*t_stride][var_idx[j]][s]
 *   channels_first: out[b][c_off + n*c_stride + j][s] = (same)
 * for n < n_steps, j < nv.  out is (B, S, Ctot) or (B, Ctot, S) of `dtype` (fp32 source rounded to bf16 on the fly).
 * The caller guarantees samples[b] + t_off + (n_steps-1)*t_stride < T.  Predictors, insolation channels and targets
 * of a batch are three calls with different (array, var_idx, c_off, c_stride, t_off).
 * ------------------------------------------------------------------------------------------------------------- */
int dlwpcs_batch_gather(const void *array, int64_t T, int V, int64_t S, const int32_t *samples_dev, int B,
                        const int32_t *var_idx_dev, int nv, int n_steps, int t_off, int t_stride, void *out,
                        int Ctot, int c_off, int c_stride, int channels_last, int dtype, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Data-parallel exchange: ONE in-place sum of the flat fp32 gradient buffer over the ranks, through an RCCL
 * communicator the caller owns, enqueued on the CALLER's stream -- inside a captured training step the collective is a plain
 * node of the step's graph between dlwpcs_wgrad_batch (reduction into the buffer) and dlwpcs_wgrad_batch_apply.
 * Reference counterpart: keras.utils.multi_gpu_model, DLWP/model/models.py:369-374 (host-side averaging over replicas).
 *   dlwpcs_comm_load(path):    dlopen RCCL (the librccl.so PyTorch ships: one RCCL per process); once per process
 *   dlwpcs_comm_unique_id(id): 128 bytes from ncclGetUniqueId -- rank 0 calls it and hands the bytes to the other ranks
 *   dlwpcs_comm_init(&comm, id, rank, world): ncclCommInitRank on the calling thread's current device (collective: every rank)
 *   dlwpcs_allreduce_f32(comm, buf, n, stream): ncclAllReduce(sum, in place); capturable; no host synchronisation
 *   dlwpcs_comm_destroy(comm)
 * ------------------------------------------------------------------------------------------------------------- */
int dlwpcs_comm_load(const char *librccl_path);
int dlwpcs_comm_unique_id(void *id128);
int dlwpcs_comm_init(void **comm, const void *id128, int rank, int world);
int dlwpcs_comm_destroy(void *comm);
int dlwpcs_allreduce_f32(void *comm, float *buf, size_t n, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Hardware self-check of the gather-form data gradient (DLWPCS_CONV_DGRAD_GATHER): its kernel masks MFMA operands by
 * address -- lanes that must add nothing read LDS beyond the workgroup's allocation, which returns zeros on gfx950.
 * Writes to *nonzero_dev (device int32, caller-owned) the number of non-zero dwords such reads returned at the offsets
 * the kernel uses: 0 = the assumption holds on this device.
 * ------------------------------------------------------------------------------------------------------------- */
int dlwpcs_lds_oob_probe(int32_t *nonzero_dev, dlwpcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------- *
 * Opt-in launch profiler (bench.py `roofline`): when enabled, every MFMA convolution kernel launch is bracketed by two
 * HIP events recorded on the launch stream.  Off by default.  Enabled while a stream is being graph-captured, the events
 * become external event-record nodes of that graph (tag suffix "@graph"): every replay records them again and
 * dlwpcs_prof_get returns the duration of the launch inside the last completed replay; destroy the graph before
 * dlwpcs_prof_reset.
 * dlwpcs_prof_get: tag = kernel name as rocprofv3 prints it (template arguments included), ms = event-elapsed time,
 * flops / bytes = ALGORITHMIC work of that launch (2*B*6*N^2*k^2*Cin*Cout; unpadded tensors touched once + weights).
 * ------------------------------------------------------------------------------------------------------------- */
int dlwpcs_prof_enable(int on);
int dlwpcs_prof_reset(void);
int dlwpcs_prof_count(void);
int dlwpcs_prof_get(int i, char *tag, int tag_len, double *ms, double *flops, double *bytes);
/* Host only, no device needed: the tags the profiler can report -- one per kernel instantiation the library carries, registered
 * when the library is loaded.  A tag is the kernel's name exactly as the code object / rocprofv3 / `nm -C` spell it (a suffix in
 * parentheses names a mode of the same kernel).  dlwpcs_prof_known_tags: how many; dlwpcs_prof_known_tag: the i-th. */
int dlwpcs_prof_known_tags(void);
int dlwpcs_prof_known_tag(int i, char *tag, int tag_len);

#ifdef __cplusplus
}
#endif
#endif /* DLWPCS_H */
