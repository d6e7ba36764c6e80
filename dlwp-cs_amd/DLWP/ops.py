"""
torch.autograd glue over the C ABI (include/dlwpcs.h).  Every function here launches hand-written HIP kernels from
libdlwpcs.so on torch's current stream; torch only provides storage and the autograd tape.  All tensors on this level
are channels_last `(B, 6, H, W, C)` on a HIP device, float32 or bfloat16 (activations only; parameters, their gradients
and the optimizer state are always float32 -- see the dtype note in include/dlwpcs.h).
"""
import ctypes
import os

import torch

from . import _native as nat
from .options import option
from ._native import ConvDesc, GConvDesc, check, lib, ptr, require_device, stream_ptr

# When True, convolution backward accumulates weight gradients directly into the parameters' preset `.grad` buffers
# (DLWP.keras.Model turns this on around its training step; plain autograd use keeps the standard semantics).
DIRECT_PARAM_GRADS = False

# Pre-packed weights (dlwpcs_pack_batch): id(equatorial kernel tensor) -> (dtype tag, wpk_fwd, bias_pk | None, wpk_bwd).
# DLWP.keras.Model fills this around its forward pass after packing every layer with ONE launch; a convolution whose
# kernel is not listed (or listed for another dtype) packs its weights itself, per call, into the workspace.
PREPACKED = {}

# When True (DLWP.keras.Model training step), weight-gradient kernels are enqueued on a second HIP stream: they depend
# only on (src, dy) and nothing before the optimizer consumes them, so they overlap the data-gradient chain of the layers
# below (fills launch gaps and the tails of the persistent kernels).  Whoever sets this must call join_side_stream()
# before the gradients are read.
WGRAD_SIDE_STREAM = False
_side_streams = {}


def side_stream(device):
    key = str(device)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


def join_side_stream(device):
    """Make the current stream wait for everything enqueued on the weight-gradient side stream."""
    st = _side_streams.get(str(device))
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)


# workspace: one growing byte buffer per (device, role) (caller-owned from the library's point of view)
_workspaces = {}
_retired_workspaces = []    # outgrown buffers stay allocated: hipGraphs captured earlier have their addresses baked in


def _workspace(nbytes, device, role='main'):
    key = (str(device), role)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise nat.NativeError('workspace would have to grow during graph capture; run one eager step first')
        if ws is not None:
            _retired_workspaces.append(ws)
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# Deferred weight-gradient reduction (DLWPCS_CONV_DEFER_REDUCE): while the flag is on, every fused-conv backward node
# that accumulates straight into preset .grad buffers leaves its per-worker partials in a workspace of its own and
# queues a dlwpcs_reduce_item; flush_deferred_reduce() then sums ALL layers' partials with one launch (one per
# application round of shared layers) instead of one small launch per layer.  Whoever sets the flag must flush before
# the gradients are read.
DEFER_WGRAD_REDUCE = False
_deferred = []              # [(ReduceItem, workspace tensor)] of the backward pass in flight
_reduce_tables = {}         # (device, raw item bytes) -> device copy of the item table


def drop_deferred_reduce():
    del _deferred[:]


def flush_deferred_reduce(device):
    if not _deferred:
        return
    pending = list(_deferred)
    del _deferred[:]
    # items whose destinations coincide (a layer applied twice) go into successive launches
    rounds, seen = [], {}
    for item, ws in pending:
        k = seen.get(item.dw_eq, 0)
        seen[item.dw_eq] = k + 1
        while len(rounds) <= k:
            rounds.append([])
        rounds[k].append(item)
    for items in rounds:
        host = (nat.ReduceItem * len(items))(*items)
        raw = bytes(host)
        key = (str(device), raw)
        table = _reduce_tables.get(key)
        if table is None:
            if torch.cuda.is_current_stream_capturing():
                raise nat.NativeError('reduce-item table would have to be uploaded during graph capture; run one eager '
                                      'step first')
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
            _reduce_tables[key] = table
        check(lib().dlwpcs_wgrad_reduce_batch(ptr(table), ctypes.addressof(host), len(items), stream_ptr()),
              'dlwpcs_wgrad_reduce_batch')


# ------------------------------------------------------------------------------------------------------------------ #
# Batched weight gradients (dlwpcs_wgrad_batch): ONE persistent launch + one reduction for all layers of a backward pass
# ------------------------------------------------------------------------------------------------------------------ #
_wb_plans = {}              # geometry key -> (host plan buffer, device plan tensor, workspace bytes)


# When True (DLWP.keras.Model training step) a convolution's backward node does not launch its weight gradient: it queues
# (descriptor, inputs, dz, gradient buffers) and flush_wgrad_batch() runs ALL queued layers with one launch after the last
# data gradient (the weight gradients are off the critical path of the backward pass).  Needs DIRECT_PARAM_GRADS (the
# gradients are accumulated into the parameters' preset .grad buffers); whoever sets the flag must flush.
WGRAD_BATCH = False
_wb_pending = []


def wgrad_batch_supported(d):
    return bool(lib().dlwpcs_wgrad_batch_supported(ctypes.byref(d)))


_wb_flushed = []            # the entries flush_wgrad_batch() has run since the last drop (apply_wgrad_batch names their gradients)


def drop_wgrad_batch():
    del _wb_pending[:]
    del _wb_flushed[:]


# The fused head's loss is finished (second stage of its reduction) by the LAST launch of the training step when that is the
# fused weight-gradient reduction + optimizer (dlwpcs_wgrad_batch_adam_tail): DEFER_LOSS_TAIL is set by DLWP.keras.Model for such
# steps, _HeadMSE.forward then leaves a dlwpcs_loss_tail here; whoever ends the step without that launch runs finish_loss_tail().
DEFER_LOSS_TAIL = False
_pending_tail = []


def finish_loss_tail():
    """Run a deferred loss tail as a launch of its own (no fused reduction + optimizer launch took it)."""
    while _pending_tail:
        tail, keep = _pending_tail.pop()
        check(lib().dlwpcs_loss_tail_run(ctypes.byref(tail), stream_ptr()), 'dlwpcs_loss_tail_run')


PACK_FUSED = False          # did the last flush_wgrad_batch refresh the packed operands inside its optimizer launch?


def flush_wgrad_batch(adam=None, pack_lookup=None):
    """Run the queued layers.  adam = (flat params, flat grads, m, v, state {t-1, ticket}, hyper {lr, b1, b2, eps, scale},
    number of parameter elements) asks for the optimizer fused into the reduction (dlwpcs_wgrad_batch_adam): done -- and True
    returned -- when the queued layers cover every parameter exactly once; otherwise the gradients are left in the flat buffer
    as usual and the caller runs its optimizer launch.  pack_lookup (with adam): address of a layer's equatorial-kernel gradient
    -> its make_pack_items entry; when every queued layer has one, the launch also refreshes the packed bf16 operands
    (PACK_FUSED says whether it did)."""
    global PACK_FUSED
    PACK_FUSED = False
    if not _wb_pending:
        return False
    pending = list(_wb_pending)
    del _wb_pending[:]
    _wb_flushed.extend(pending)
    if adam is not None and len(pending) <= nat.WGRAD_BATCH_MAX:
        p, g, m, v, state, hyper, n_elems = adam
        if _wb_covers(pending, n_elems):
            packs = None
            if pack_lookup is not None:
                packs = [pack_lookup(ent[5][0].data_ptr()) for ent in pending]
                if any(pk is None for pk in packs):
                    packs = None
            wgrad_batch(pending, adam=(p, g, m, v, state, hyper), packs=packs)
            PACK_FUSED = packs is not None
            return True
    wgrad_batch(pending)
    return False


def _wb_covers(entries, n_elems):
    """do the gradient tensors of `entries` cover n_elems parameter elements?  A layer applied more than once (integration_steps
    = 2 in the reference scripts) names its tensors once per application: the library reduces such items in successive launches
    and lets the optimizer consume a tensor in the launch of its last item."""
    seen, covered = set(), 0
    for ent in entries:
        for t in ent[5]:
            if t is not None and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                covered += t.numel()
    return covered == n_elems


_wb_anchor = {}


def flushed_wgrad_entries():
    """the entries run by flush_wgrad_batch() since the last drop, for a caller that applies the update after the backward pass's
    clean-up (apply_wgrad_batch / prepare_wgrad_plan) -- WITHOUT the layers' saved inputs, outputs and output gradients: the
    apply launch names the layers' descriptors and gradient views only, and a model that kept the full entries would hold one
    step's activations until the end of the next one (twice the peak memory of eager steps).  dz / y are replaced by a one-element
    tensor of the same device (the plan key asks which device, and whether a layer masks on load)."""
    out = []
    for ent in _wb_flushed:
        dev = ent[3].device
        a = _wb_anchor.get(dev)
        if a is None:
            a = _wb_anchor[dev] = torch.zeros(1, dtype=torch.uint8, device=dev)
        slim = (ent[0], None, None, a, ent[4], ent[5])
        if len(ent) > 6:
            slim += ((a if ent[6] is not None else None),)
        out.append(slim)
    return out


def apply_wgrad_batch(adam, pack_lookup=None, entries=None):
    """Data-parallel tail of a training step (dlwpcs_wgrad_batch_apply): the layers flush_wgrad_batch() has run since the last
    drop left their finished gradients in the flat buffer, the caller has summed it over the ranks; ONE launch now applies
    grad_scale + Adam, clears the gradients, refreshes the packed bf16 operands (pack_lookup as for flush_wgrad_batch) and
    finishes a deferred loss.  adam = (p, g, m, v, state {t, ticket}, hyper, number of parameter elements).  Returns False --
    nothing launched -- when those layers do not cover every parameter exactly once (the caller runs its optimizer launch)."""
    global PACK_FUSED
    PACK_FUSED = False
    entries = list(_wb_flushed) if entries is None else list(entries)
    p, g, m, v, state, hyper, n_elems = adam
    if not entries or len(entries) > nat.WGRAD_BATCH_MAX or not _wb_covers(entries, n_elems):
        return False
    dev = entries[0][3].device
    arr, key = _wb_items(entries)
    hit = _wb_plan(arr, len(entries), (str(dev), key), dev)
    host, plan_dev, _ = hit
    packs = None
    if pack_lookup is not None:
        packs = [pack_lookup(ent[5][0].data_ptr()) for ent in entries]
        if any(pk is None for pk in packs):
            packs = None
    tail = None
    if len(_pending_tail) == 1:
        tail, keep = _pending_tail.pop()                        # the step's loss is finished by this launch
    check(lib().dlwpcs_wgrad_batch_apply(arr, len(entries), host, ptr(plan_dev), ptr(p), ptr(g), ptr(m), ptr(v), g.numel(),
                                         ptr(state), ptr(hyper), ctypes.byref(tail) if tail is not None else None,
                                         _pack_array(packs) if packs is not None else None, stream_ptr()),
          'dlwpcs_wgrad_batch_apply')
    PACK_FUSED = packs is not None
    return True


def prepare_wgrad_plan(entries, n_elems):
    """Build (and upload) the plan of an item list outside a graph capture, so that a captured apply_wgrad_batch() of the same
    layers finds it (the two-bucket step flushes its halves separately: their union is first needed by the captured update)."""
    entries = list(entries)
    if not entries or len(entries) > nat.WGRAD_BATCH_MAX or not _wb_covers(entries, n_elems):
        return
    dev = entries[0][3].device
    arr, key = _wb_items(entries)
    _wb_plan(arr, len(entries), (str(dev), key), dev)


def _pack_array(packs):
    pk_arr = (nat.PackItem * len(packs))()
    for it, (we, wp, wn, be, bp, bn, bufs, ksize, flip, tag) in zip(pk_arr, packs):
        it.w_eq, it.w_pol, it.w_np = ptr(we), ptr(wp), ptr(wn)
        it.b_eq, it.b_pol, it.b_np = ptr(be), ptr(bp), ptr(bn)
        it.wpk_fwd, it.bias_pk, it.wpk_bwd = ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2])
        it.ksize, it.Cin, it.Cout = int(ksize), int(we.shape[2]), int(we.shape[3])
        it.flip_north_pole, it.dtype, it.reserved = int(flip), int(tag), 0
    return pk_arr


def _wb_plan(arr, n, key, dev):
    """(host plan buffer, device plan tensor, workspace bytes) of an item list, built once per geometry"""
    hit = _wb_plans.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            raise nat.NativeError('the weight-gradient plan would have to be uploaded during graph capture; run one eager '
                                  'step first')
        pb, wb = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().dlwpcs_wgrad_batch_sizes(arr, n, ctypes.byref(pb), ctypes.byref(wb)), 'dlwpcs_wgrad_batch_sizes')
        host = (ctypes.c_char * pb.value)()
        check(lib().dlwpcs_wgrad_batch_plan(arr, n, host, pb.value), 'dlwpcs_wgrad_batch_plan')
        plan_dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)
        hit = (host, plan_dev, wb.value)
        _wb_plans[key] = hit
    return hit


def _wb_items(entries):
    arr = (nat.WgradItem * len(entries))()
    key = []
    for it, ent in zip(arr, entries):
        d, src0, src1, dz, table, grads = ent[:6]
        y = ent[6] if len(ent) > 6 else None
        it.d = d
        it.src0, it.src1, it.dz, it.y, it.table_dev = ptr(src0), ptr(src1), ptr(dz), ptr(y), ptr(table)
        it.dw_eq, it.dw_pol, it.dw_np, it.db_eq, it.db_pol, it.db_np = [ptr(g) for g in grads]
        key.append((d.B, d.N, d.C0, d.C1, d.Cout, d.ksize, d.halo, d.up0, d.flip_north_pole, d.dtype, d.c0_valid,
                    tuple(g is not None for g in grads), None if y is None else (d.act, d.alpha, d.vmax)))
    return arr, tuple(key)


def wgrad_batch(entries, adam=None, packs=None):
    """entries: [(ConvDesc, src0, src1 | None, dz, halo table | None, (dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np)[, y])] with
    fp32 gradient tensors that are ACCUMULATED into (None where the layer has no such parameter).  dz is the gradient
    w.r.t. the layer's pre-activation output (already masked) -- or, with the optional 7th element y (the layer's saved
    output, desc.act = LEAKY_CLIP), the plain gradient dy, masked by the kernel on load.  The plan of a layer list is built
    once and cached."""
    if not entries:
        return
    dev = entries[0][3].device
    for lo in range(0, len(entries), nat.WGRAD_BATCH_MAX):
        chunk = entries[lo:lo + nat.WGRAD_BATCH_MAX]
        arr, key = _wb_items(chunk)
        host, plan_dev, ws_bytes = _wb_plan(arr, len(chunk), (str(dev), key), dev)
        ws = _workspace(ws_bytes, dev, 'wgrad_batch')
        if adam is not None:
            p, g, m, v, state, hyper = adam
            tail = None
            if len(_pending_tail) == 1:
                tail, keep = _pending_tail.pop()            # the step's loss is finished by this launch
            if tail is not None or packs is not None:
                pk_arr = _pack_array(packs[lo:lo + len(chunk)]) if packs is not None else None
                check(lib().dlwpcs_wgrad_batch_adam_tail(arr, len(chunk), host, ptr(plan_dev), ptr(ws), ws.numel(), ptr(p), ptr(g),
                                                         ptr(m), ptr(v), g.numel(), ptr(state), ptr(hyper),
                                                         ctypes.byref(tail) if tail is not None else None, pk_arr, stream_ptr()),
                      'dlwpcs_wgrad_batch_adam_tail')
                continue
            check(lib().dlwpcs_wgrad_batch_adam(arr, len(chunk), host, ptr(plan_dev), ptr(ws), ws.numel(), ptr(p), ptr(g), ptr(m),
                                                ptr(v), g.numel(), ptr(state), ptr(hyper), stream_ptr()),
                  'dlwpcs_wgrad_batch_adam')
            continue
        check(lib().dlwpcs_wgrad_batch(arr, len(chunk), host, ptr(plan_dev), ptr(ws), ws.numel(), stream_ptr()),
              'dlwpcs_wgrad_batch')


# Deferred ring fix-up (DLWPCS_CONV_DEFER_RING0): a data-gradient call whose source 0 is a pooled tensor with no other
# consumer leaves the halo ring of that source in its workspace; the pooling adjoint that receives the gradient next adds it
# while it spreads the gradient (dlwpcs_avgpool2_bwd_ring) -- one launch less per pooling level.  Keyed by the address of the
# gradient tensor the convolution's backward returns; DLWP.keras.Model asks for it only where its plan guarantees that the next
# reader of that tensor is the pooling node, and clears the table around every backward pass.
_pending_ring = {}
# Pooled by-products of the convolutions that ran with want_pool (dlwpcs_conv_fwd_pool), keyed by the address of the full-size
# output; the pooling node that follows takes its entry (DLWP.keras.Model asks for it only when the next reader of that
# output is such a node, and clears the table at the start of every forward pass).
_POOLED = {}


def drop_pending_rings():
    _pending_ring.clear()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------------------------ #
# CubeSpherePadding2D  (reference DLWP/custom.py:1082-1308)
# ------------------------------------------------------------------------------------------------------------------ #

class _CSPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        require_device(x, 'cs_pad')
        x = _c(x)
        B, F6, N, N2, C = x.shape
        if F6 != 6 or N != N2:
            raise ValueError('cs_pad: expected (B, 6, N, N, C), got %s' % (tuple(x.shape),))
        table, inv = nat.halo_tables(N, p, x.device)
        y = torch.empty((B, 6, N + 2 * p, N + 2 * p, C), dtype=x.dtype, device=x.device)
        check(lib().dlwpcs_pad_fwd(ptr(x), ptr(y), B, N, C, p, nat.dtype_tag(x), ptr(table), stream_ptr()),
              'dlwpcs_pad_fwd')
        ctx.p, ctx.shape = p, (B, N, C)
        ctx.inv = inv
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, C = ctx.shape
        dy = _c(dy)
        dx = torch.empty((B, 6, N, N, C), dtype=dy.dtype, device=dy.device)
        check(lib().dlwpcs_pad_bwd(ptr(dy), ptr(dx), B, N, C, ctx.p, nat.dtype_tag(dy), ptr(ctx.inv), stream_ptr()),
              'dlwpcs_pad_bwd')
        return dx, None


def cs_pad(x, p):
    """Halo-pad a cubed-sphere tensor (B,6,N,N,C) -> (B,6,N+2p,N+2p,C)."""
    return _CSPad.apply(x, int(p))


# ------------------------------------------------------------------------------------------------------------------ #
# Fused cubed-sphere convolution (reference DLWP/custom.py:921-1002 + :1082-1308 + Keras ReLU/UpSampling3D/concatenate)
# ------------------------------------------------------------------------------------------------------------------ #

def _make_desc(B, N, C0, C1, Cout, ksize, halo, up0, flip, act, alpha, vmax, dtype=nat.F32, c0_valid=0):
    return ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=ksize, halo=int(halo), up0=int(up0),
                    flip_north_pole=int(flip), act=int(act), alpha=float(alpha), vmax=float(vmax), dtype=int(dtype),
                    flags=0, c0_valid=int(c0_valid))


class _PadChannels(torch.autograd.Function):
    """(…, C) -> (…, Cp): zero channels appended (dlwpcs_pad_channels); the backward slices them off again."""

    @staticmethod
    def forward(ctx, x, cp):
        require_device(x, 'pad_channels')
        x = _c(x)
        C = x.shape[-1]
        rows = x.numel() // C if C else 0
        y = torch.empty(tuple(x.shape[:-1]) + (cp,), dtype=x.dtype, device=x.device)
        check(lib().dlwpcs_pad_channels(ptr(x), ptr(y), rows, C, cp, nat.dtype_tag(x), stream_ptr()), 'dlwpcs_pad_channels')
        ctx.c = C
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        cp = dy.shape[-1]
        rows = dy.numel() // cp
        dx = torch.empty(tuple(dy.shape[:-1]) + (ctx.c,), dtype=dy.dtype, device=dy.device)
        check(lib().dlwpcs_slice_channels(ptr(dy), ptr(dx), rows, cp, ctx.c, nat.dtype_tag(dy), stream_ptr()),
              'dlwpcs_slice_channels')
        return dx, None


def pad_channels(x, cp):
    """Append zero channels up to `cp` (no-op when the tensor already has cp channels)."""
    if x.shape[-1] == cp:
        return x
    if x.shape[-1] > cp:
        raise ValueError('pad_channels: tensor has %d channels, asked for %d' % (x.shape[-1], cp))
    return _PadChannels.apply(x, int(cp))


def channel_vector(dtype):
    """channels per 16-B vector of an activation dtype: the alignment the fast kernel paths want"""
    return 8 if dtype == torch.bfloat16 else 4


def padded_channels(c, dtype, even_too=None):
    """Physical channel count the engine stores a c-channel NETWORK INPUT with: odd counts (7 variables) are always padded
    to the next 16-B vector (the scalar-load kernel paths spill registers and run at a fraction of the vector paths); even
    counts that are not vector multiples (14 = 7 x 2) only when asked to (they have 4-B-vector / shifted-tail paths)."""
    v = channel_vector(dtype)
    if c % v == 0:
        return c
    if c % 2 == 1 or even_too:
        return (c + v - 1) // v * v
    return c


def _f32_param(t, what):
    if t is not None and t.dtype != torch.float32:
        raise TypeError('%s: parameters must be float32 master copies, got %s' % (what, t.dtype))


def _weight_gradients(d, src0, src1, dy, y, params, table, ws, nbytes, direct, defer, need, has_np, has_bias, has_bnp,
                      batch_mask_ok=False):
    """Weight / bias gradients of one fused convolution (shared by _CSConv.backward and the fused head + loss step).
    direct: accumulate straight into the parameters' preset .grad buffers (views of the model's flat gradient buffer, zeroed
    once per step): no temporaries, no AccumulateGrad add kernels; shared layers simply accumulate twice.  Returns the six
    gradient tensors (None in direct mode)."""
    dev = dy.device
    w_eq, w_pol, w_np = params[0], params[1], params[2]
    dw_eq = dw_pol = dw_np = db_eq = db_pol = db_np = None
    if (direct and WGRAD_BATCH and d.B > 0 and not WGRAD_SIDE_STREAM and wgrad_batch_supported(d)
            and (d.act == nat.ACT_NONE or (batch_mask_ok and d.ksize == 3 and y is not None))):
        # dy IS dz (no activation, or the gradient arrived pre-masked) -- or the batched kernel masks on load (a layer whose
        # inputs need no gradient: nobody else wants dz): queue the layer for the batched launch
        pe, pp, pn, be, bp, bn = params
        d2 = ConvDesc.from_buffer_copy(d)
        _wb_pending.append((d2, src0, src1, dy, table,
                            (pe.grad, pp.grad, None if pn is None else pn.grad, None if be is None else be.grad,
                             None if bp is None else bp.grad, None if bn is None else bn.grad),
                            None if d.act == nat.ACT_NONE else y))
        return dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np
    if direct:
        pe, pp, pn, be, bp, bn = params
        d2 = ConvDesc.from_buffer_copy(d)
        d2.flags = d.flags | nat.CONV_ACCUMULATE_WGRAD | (nat.CONV_DEFER_REDUCE if defer else 0)
        grads = (ptr(pe.grad), ptr(pp.grad), ptr(None if pn is None else pn.grad),
                 ptr(None if be is None else be.grad), ptr(None if bp is None else bp.grad),
                 ptr(None if bn is None else bn.grad))

        def launch(wsx):
            check(lib().dlwpcs_conv_bwd_weights(ctypes.byref(d2), ptr(src0), ptr(src1), ptr(dy), ptr(y), *grads,
                                                ptr(table), ptr(wsx), wsx.numel(), stream_ptr()),
                  'dlwpcs_conv_bwd_weights')
            if defer:
                item = nat.ReduceItem()
                check(lib().dlwpcs_conv_wgrad_reduce_item(ctypes.byref(d2), *grads, ptr(wsx), wsx.numel(),
                                                          ctypes.byref(item)), 'dlwpcs_conv_wgrad_reduce_item')
                _deferred.append((item, wsx))
        if WGRAD_SIDE_STREAM:
            side = side_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))      # dy and the saved activations are ready
            ws2 = _workspace(nbytes, dev, 'side')                 # own workspace: the main stream keeps using `ws`
            with torch.cuda.stream(side):
                launch(ws2)
            for t in (src0, src1, dy, y):                         # keep their memory until the side stream is done
                if t is not None:
                    t.record_stream(side)
        else:
            launch(ws)
    elif any(need):
        dw_eq, dw_pol = torch.empty_like(w_eq), torch.empty_like(w_pol)
        dw_np = torch.empty_like(w_np) if has_np else None
        if has_bias:
            db_eq = torch.empty(d.Cout, dtype=torch.float32, device=dev)
            db_pol = torch.empty(d.Cout, dtype=torch.float32, device=dev)
            db_np = torch.empty(d.Cout, dtype=torch.float32, device=dev) if has_bnp else None
        check(lib().dlwpcs_conv_bwd_weights(ctypes.byref(d), ptr(src0), ptr(src1), ptr(dy), ptr(y), ptr(dw_eq),
                                            ptr(dw_pol), ptr(dw_np), ptr(db_eq), ptr(db_pol), ptr(db_np),
                                            ptr(table), ptr(ws), ws.numel(), stream_ptr()),
              'dlwpcs_conv_bwd_weights')
    return dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np


class _CSConv(torch.autograd.Function):
    """
    y = act(conv(halo_pad(concat(up?(src0), src1))) + bias); see dlwpcs_conv_fwd in include/dlwpcs.h.
    Inputs that may be None: src1, w_np, b_eq, b_pol, b_np.
    """

    @staticmethod
    def forward(ctx, src0, src1, w_eq, w_pol, w_np, b_eq, b_pol, b_np, ksize, halo, up0, flip, act, alpha, vmax,
                c0_valid=0, premask0=None, premask1=None, dy_premasked=False, defer_ring0=False, want_pool=False,
                out_padded=False):
        """out_padded (inference only, the pointwise bf16 output layer): y gets ceil8(C_out) channels per pixel, the padding
        zero (DLWPCS_CONV_OUT_PADDED) -- what the first layer takes back as a padded source in a rollout.
        want_pool: the 2x2 average pooling of y is produced as a by-product (dlwpcs_conv_fwd_pool) and parked in _POOLED for
        the pooling node that follows (avgpool2_skip), which then launches nothing in its forward pass.
        defer_ring0: the gradient of source 0 goes to a pooling node that adds the halo ring itself (see _pending_ring).
        premask0 / premask1 = (negative_slope, max_value) | None: src0 / src1 is the output of an activated layer that expects
        its gradient PRE-MASKED (multiplied by act'(src)): the backward applies it to dsrc0 / dsrc1.  dy_premasked: the gradient
        THIS node receives is already dz = dy * act'(y) (every consumer of y honours premask).  See dlwpcs_conv_bwd_data_masked."""
        require_device(src0, 'cs_conv')
        src0 = _c(src0)
        B = src0.shape[0]
        if src0.dim() != 5 or src0.shape[1] != 6 or src0.shape[2] != src0.shape[3]:
            raise ValueError('cs_conv: expected (B, 6, N, N, C), got %s' % (tuple(src0.shape),))
        N = src0.shape[2] * (2 if up0 else 1)
        C0 = src0.shape[4]
        C1 = 0
        if src1 is not None:
            require_device(src1, 'cs_conv')
            src1 = _c(src1)
            if tuple(src1.shape[:4]) != (B, 6, N, N):
                raise ValueError('cs_conv: src1 shape %s does not match (B,6,%d,%d,*)' % (tuple(src1.shape), N, N))
            C1 = src1.shape[4]
            if src1.dtype != src0.dtype:
                raise TypeError('cs_conv: src0 is %s but src1 is %s' % (src0.dtype, src1.dtype))
        for prm in (w_eq, w_pol, w_np, b_eq, b_pol, b_np):
            _f32_param(prm, 'cs_conv')
        kh, kw, cin, Cout = w_eq.shape
        if kh != ksize or kw != ksize or cin != (c0_valid or C0) + C1:
            raise ValueError('cs_conv: kernel shape %s does not match ksize=%d, C_in=%d' % (tuple(w_eq.shape), ksize,
                                                                                          (c0_valid or C0) + C1))
        w_eq, w_pol = _c(w_eq), _c(w_pol)
        w_np = _c(w_np) if w_np is not None else None
        d = _make_desc(B, N, C0, C1, Cout, ksize, halo, up0, flip, act, alpha, vmax, nat.dtype_tag(src0), c0_valid)
        No = N if halo else N - ksize + 1
        if out_padded:
            if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (src0, src1, w_eq)):
                raise RuntimeError('cs_conv: out_padded is an inference-only layout')
            d.flags |= nat.CONV_OUT_PADDED
        y = torch.empty((B, 6, No, No, (Cout + 7) // 8 * 8 if out_padded else Cout), dtype=src0.dtype, device=src0.device)
        table = inv = None
        if halo:
            table, inv = nat.halo_tables(N, (ksize - 1) // 2, src0.device)
        nbytes = lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
        ws = _workspace(nbytes, src0.device)
        packed = PREPACKED.get(id(w_eq))
        if packed is not None and packed[0] != d.dtype:
            packed = None
        yp = None
        if want_pool and halo and No % 2 == 0:
            yp = torch.empty((B, 6, No // 2, No // 2, Cout), dtype=src0.dtype, device=src0.device)
        wargs = ((ptr(packed[1]), 0, 0, ptr(packed[2]) if b_eq is not None else 0, 0, 0) if packed is not None else
                 (ptr(w_eq), ptr(w_pol), ptr(w_np), ptr(b_eq), ptr(b_pol), ptr(b_np)))
        if packed is not None:
            d.flags |= nat.CONV_PREPACKED
        if yp is not None:
            check(lib().dlwpcs_conv_fwd_pool(ctypes.byref(d), ptr(src0), ptr(src1), *wargs, ptr(y), ptr(yp), ptr(table), ptr(ws),
                                             ws.numel(), stream_ptr()), 'dlwpcs_conv_fwd_pool')
            _POOLED[y.data_ptr()] = yp
        else:
            check(lib().dlwpcs_conv_fwd(ctypes.byref(d), ptr(src0), ptr(src1), *wargs, ptr(y), ptr(table), ptr(ws),
                                        ws.numel(), stream_ptr()), 'dlwpcs_conv_fwd')
        if premask0 is not None and premask1 is not None and tuple(premask0) != tuple(premask1):
            raise ValueError('cs_conv: both sources must share the activation parameters of their masks')
        ctx.premask = (premask0, premask1)
        ctx.dy_premasked = bool(dy_premasked) and act != nat.ACT_NONE
        ctx.defer_ring0 = bool(defer_ring0)
        ctx.packed = packed
        ctx.desc = d
        ctx.tables = (table, inv)
        ctx.has = (src1 is not None, w_np is not None, b_eq is not None, b_np is not None)
        # parameter objects (leaves whose .grad may be a view into the model's flat gradient buffer, see backward)
        ctx.params = (w_eq, w_pol, w_np, b_eq, b_pol, b_np)
        ctx.save_for_backward(src0, src1, w_eq, w_pol, w_np, y if act != nat.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        src0, src1, w_eq, w_pol, w_np, y = ctx.saved_tensors
        d = ctx.desc
        table, inv = ctx.tables
        has_src1, has_np, has_bias, has_bnp = ctx.has
        dy = _c(dy)
        dev = dy.device
        nbytes = lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
        need = ctx.needs_input_grad
        dsrc0 = torch.empty_like(src0) if need[0] else None
        dsrc1 = torch.empty_like(src1) if (has_src1 and need[1]) else None
        want_w = any(need[2:8])
        direct = DIRECT_PARAM_GRADS and (need[2] or need[3]) and all(
            p is None or (p.is_leaf and p.grad is not None and p.grad.is_contiguous()) for p in ctx.params)
        pm0, pm1 = ctx.premask
        if dsrc0 is None:
            pm0 = None
        if dsrc1 is None:
            pm1 = None
        masked_io = ctx.dy_premasked or pm0 is not None or pm1 is not None
        if masked_io:
            # Pre-masked gradient convention.  dy_premasked: dy is dz already -> both gradient kernels run as for a layer
            # without activation and never read y.  Otherwise (only the sources want masks) dz is formed once, elementwise.
            if ctx.dy_premasked or d.act == nat.ACT_NONE:
                dz = dy
            else:
                dz = torch.empty_like(dy)
                check(lib().dlwpcs_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), nat.ACT_LEAKY_CLIP, d.alpha, d.vmax,
                                           nat.dtype_tag(dy), stream_ptr()), 'dlwpcs_act_bwd')
            dn = ConvDesc.from_buffer_copy(d)
            dn.act = nat.ACT_NONE
            dn.flags = (d.flags & nat.CONV_PREPACKED) | _gather_flag(d, dev)
            batching = (direct and WGRAD_BATCH and d.B > 0 and not WGRAD_SIDE_STREAM and wgrad_batch_supported(dn))
            defer = direct and DEFER_WGRAD_REDUCE and not WGRAD_SIDE_STREAM and d.B > 0 and not batching
            ws = _workspace(nbytes, dev, 'defer%d' % len(_deferred)) if defer else _workspace(nbytes, dev)
            ring = None
            if ctx.defer_ring0 and dsrc0 is not None and pm0 is None and halo_ring_info(dn) is not None:
                ring = halo_ring_info(dn)
                dn.flags |= nat.CONV_DEFER_RING0
                ws = _workspace(nbytes, dev, 'ring%d' % len(_pending_ring))      # stays intact until the pooling adjoint ran
            if dsrc0 is not None or dsrc1 is not None:
                wq = ctx.packed[3] if ctx.packed is not None else w_eq
                pm = pm0 if pm0 is not None else pm1
                ma, mv = (float(pm[0]), float(pm[1])) if pm is not None else (0.0, 0.0)
                check(lib().dlwpcs_conv_bwd_data_masked(ctypes.byref(dn), ptr(dz), ptr(wq), ptr(w_pol), ptr(w_np), ptr(dsrc0),
                                                        ptr(dsrc1), ptr(src0 if pm0 is not None else None),
                                                        ptr(src1 if pm1 is not None else None), ma, mv, ptr(inv), ptr(ws),
                                                        ws.numel(), stream_ptr()), 'dlwpcs_conv_bwd_data_masked')
                if ring is not None:
                    _pending_ring[dsrc0.data_ptr()] = (ws, ring[0], ring[1], nat.halo_tables(d.N, 1, dev)[1])
                    dn.flags &= ~nat.CONV_DEFER_RING0
            dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np = _weight_gradients(
                dn, src0, src1, dz, None, ctx.params, table, ws, nbytes, direct, defer, need[2:8], has_np, has_bias, has_bnp)
            return (dsrc0, dsrc1, dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np) + (None,) * 14
        no_dgrad = dsrc0 is None and dsrc1 is None        # (first layer: the batched kernel applies act' itself)
        # (exact-fp32 mode: the batched kernel forms dz = dy * act'(y) on load in every 3x3 layer -- exact in fp32, so the
        #  data-gradient kernel forming its own copy changes no bits -- and no dz hand-over is written at all)
        mask_on_load = d.ksize == 3 and (no_dgrad or d.dtype == nat.F32)
        batching = (direct and WGRAD_BATCH and d.B > 0 and not WGRAD_SIDE_STREAM and wgrad_batch_supported(d)
                    and (d.act == nat.ACT_NONE or mask_on_load))
        defer = direct and DEFER_WGRAD_REDUCE and not WGRAD_SIDE_STREAM and d.B > 0 and not batching
        # deferred reduction: the partials (and the dz hand-over next to them) live in this node's own workspace
        ws = _workspace(nbytes, dev, 'defer%d' % len(_deferred)) if defer else _workspace(nbytes, dev)
        reuse_dz = ((dsrc0 is not None or dsrc1 is not None) and want_w and d.act != nat.ACT_NONE
                    and not WGRAD_SIDE_STREAM and not batching)
        if reuse_dz:
            # the weight-gradient kernel computes dz = dy * act'(y) anyway: launched FIRST, it leaves dz in the workspace
            # for the data-gradient kernel right behind it (which then reads neither y nor does the act' arithmetic)
            d.flags |= nat.CONV_REUSE_DZ

        def run_bwd_data():
            if dsrc0 is not None or dsrc1 is not None:
                # (d.flags carries CONV_PREPACKED from the forward when packed buffers were used)
                wq = ctx.packed[3] if ctx.packed is not None else w_eq
                d.flags |= _gather_flag(d, dev)         # (honoured where the gradient needs no mask on load: layers without activation)
                check(lib().dlwpcs_conv_bwd_data(ctypes.byref(d), ptr(dy), ptr(y), ptr(wq), ptr(w_pol), ptr(w_np),
                                                 ptr(dsrc0), ptr(dsrc1), ptr(inv), ptr(ws), ws.numel(), stream_ptr()),
                      'dlwpcs_conv_bwd_data')
                d.flags &= ~nat.CONV_DGRAD_GATHER
        if not reuse_dz:
            run_bwd_data()
        dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np = _weight_gradients(
            d, src0, src1, dy, y, ctx.params, table, ws, nbytes, direct, defer, need[2:8], has_np, has_bias, has_bnp,
            batch_mask_ok=mask_on_load)
        if reuse_dz:
            run_bwd_data()
        return (dsrc0, dsrc1, dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np) + (None,) * 14


_ring_info_cache = {}

# Data gradient in gather form (dlwpcs.h: DLWPCS_CONV_DGRAD_GATHER; the adjoint of CubeSpherePadding2D, DLWP/custom.py:1198-1308, folded
# into the data-gradient kernel): bf16 3x3 halo layers whose gradient arrives as dz compute every border cell completely inside the
# kernel -- no halo ring is materialised, no fix-up launch, one rounding per cell (tests/test_gpu_dgrad_gather.py: <= 1 bf16 ulp against
# the fp64 oracle); window_src 2 x 2 sums for upsampled sources.  Default since round 5 (weight substitution inside the matrix phase:
# as fast as the padded-grid kernel + the fix-up launches it replaces, 24 instead of 31 launches per unet2 step); faces whose tile
# holds both edge rows (N <= 16) and the exact-fp32 mode keep the padded-grid path.  Engine option dgrad_gather=0 (DLWP/options.py)
# restores it everywhere.


def _gather_flag(d, dev):
    if option('dgrad_gather') and d.halo and d.ksize == 3 and d.dtype == nat.BF16 and nat.dgrad_gather_ready(d.N, 1, dev):
        return nat.CONV_DGRAD_GATHER
    return 0


def halo_ring_info(d):
    """(byte offset of the padded gradient in the workspace, its channel count) if a data-gradient call on `d` with
    CONV_DEFER_RING0 leaves the fix-up of source 0 to the caller, else None (dlwpcs_conv_ring_info)."""
    key = (d.B, d.N, d.C0, d.C1, d.Cout, d.ksize, d.halo, d.up0, d.dtype, d.c0_valid, d.flags & nat.CONV_DGRAD_GATHER)
    if key not in _ring_info_cache:
        off, ch = ctypes.c_size_t(), ctypes.c_int()
        ok = lib().dlwpcs_conv_ring_info(ctypes.byref(d), ctypes.byref(off), ctypes.byref(ch))
        _ring_info_cache[key] = (off.value, ch.value) if ok else None
    return _ring_info_cache[key]


def conv_packed_buffers(ksize, cin, cout, dtype_tag, device, bias=True):
    """Allocate (wpk_fwd, bias_pk | None, wpk_bwd) for one layer; sizes from dlwpcs_conv_packed_bytes."""
    d = _make_desc(1, max(ksize, 2), cin, 0, cout, ksize, False, False, True, nat.ACT_NONE, 0.0, 0.0, dtype_tag)
    out = []
    for which in (nat.PACK_FWD, nat.PACK_BIAS, nat.PACK_BWD):
        n = lib().dlwpcs_conv_packed_bytes(ctypes.byref(d), which)
        if n == 0:
            check(-1, 'dlwpcs_conv_packed_bytes')
        out.append(torch.empty(n, dtype=torch.uint8, device=device))
    if not bias:
        out[1] = None
    return tuple(out)


def make_pack_items(entries, device):
    """
    entries: list of (w_eq, w_pol, w_np, b_eq, b_pol, b_np, (wpk_fwd, bias_pk, wpk_bwd), ksize, flip, dtype_tag).
    Returns the device-resident dlwpcs_pack_item array (uint8 tensor) for dlwpcs_pack_batch.
    """
    arr = (nat.PackItem * len(entries))()
    for it, (we, wp, wn, be, bp, bn, bufs, ksize, flip, tag) in zip(arr, entries):
        it.w_eq, it.w_pol, it.w_np = ptr(we), ptr(wp), ptr(wn)
        it.b_eq, it.b_pol, it.b_np = ptr(be), ptr(bp), ptr(bn)
        it.wpk_fwd, it.bias_pk, it.wpk_bwd = ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2])
        it.ksize, it.Cin, it.Cout = int(ksize), int(we.shape[2]), int(we.shape[3])
        it.flip_north_pole, it.dtype, it.reserved = int(flip), int(tag), 0
    raw = bytes(arr)
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8) if raw else torch.empty(0, dtype=torch.uint8)
    return host.to(device)


def pack_batch(items_dev, n_items):
    check(lib().dlwpcs_pack_batch(ptr(items_dev), int(n_items), stream_ptr()), 'dlwpcs_pack_batch')


def cs_conv(src0, w_eq, w_pol, w_np=None, b_eq=None, b_pol=None, b_np=None, src1=None, ksize=3, halo=True, up0=False,
            flip_north_pole=True, act=nat.ACT_NONE, alpha=0.0, vmax=0.0, premask0=None, premask1=None, dy_premasked=False,
            defer_ring0=False, want_pool=False, out_padded=False):
    if (w_np is None) != (b_np is None) and b_eq is not None:
        raise ValueError('cs_conv: north-pole kernel and bias must be given together')
    # Network inputs with a channel count that is not a multiple of the 16-B vector (7 variables; optionally 14 = 7 x 2):
    # stored with zero channels up to the vector width (dlwpcs_conv_desc.c0_valid), so that every kernel takes its vector
    # path.  A caller that already holds the padded layout (Model pads its static input buffers once, the device batch feed
    # gathers straight into it) passes it as is.
    c0_valid = 0
    cin_w = w_eq.shape[2]
    if src1 is None and not up0:
        c_phys = src0.shape[-1]
        if c_phys == cin_w:
            cp = padded_channels(c_phys, src0.dtype)
            if cp != c_phys:
                src0 = pad_channels(src0, cp)
                c0_valid = cin_w
        elif cin_w < c_phys <= (cin_w + channel_vector(src0.dtype) - 1) // channel_vector(src0.dtype) * channel_vector(src0.dtype):
            c0_valid = cin_w
    return _CSConv.apply(src0, src1, w_eq, w_pol, w_np, b_eq, b_pol, b_np, int(ksize), bool(halo), bool(up0),
                         bool(flip_north_pole), int(act), float(alpha), float(vmax), int(c0_valid), premask0, premask1,
                         bool(dy_premasked), bool(defer_ring0), bool(want_pool), bool(out_padded))


def cs_conv_head_applicable(src0, src1, w_eq, head_w_eq, out_padded):
    """True when dlwpcs_conv_fwd_head can serve (last 3x3 layer, pointwise head) of an inference pass: bf16 device tensors, both
    layers' operands packed for this pass (PREPACKED), 32 channels between them and head rows of 32 channels."""
    if torch.is_grad_enabled() or not src0.is_cuda or src0.dtype != torch.bfloat16:
        return False
    pk, hk = PREPACKED.get(id(w_eq)), PREPACKED.get(id(head_w_eq))
    if pk is None or hk is None or pk[0] != nat.BF16 or hk[0] != nat.BF16:
        return False
    cout2 = head_w_eq.shape[3]
    rows = (cout2 + 7) // 8 * 8 if out_padded else cout2
    c0, c1 = src0.shape[-1], (src1.shape[-1] if src1 is not None else 0)
    if c0 % 8 or c1 % 8 or c0 + c1 != w_eq.shape[2]:
        return False
    return (tuple(w_eq.shape[:2]) == (3, 3) and w_eq.shape[3] == 32 and tuple(head_w_eq.shape[:3]) == (1, 1, 32) and rows == 32
            and cout2 % 2 == 0 and cout2 >= 8)


def cs_conv_head(src0, w_eq, b_eq, head_w_eq, head_b_eq, src1=None, up0=False, flip_north_pole=True, act=nat.ACT_NONE,
                 alpha=0.0, vmax=0.0, out_padded=False):
    """Inference: y_head = conv1x1(act(conv3x3(halo_pad(concat(up?(src0), src1))) + b)) + b_head through dlwpcs_conv_fwd_head --
    the pointwise output layer folded into the epilogue of the convolution in front of it (Azure/train_cs.py:300-305).  The
    weights are identified by the layers' equatorial kernels (their packed operands of this pass: PREPACKED).  No autograd."""
    require_device(src0, 'cs_conv_head')
    src0 = _c(src0)
    B = src0.shape[0]
    N = src0.shape[2] * (2 if up0 else 1)
    C0, C1 = src0.shape[4], 0
    if src1 is not None:
        src1 = _c(src1)
        C1 = src1.shape[4]
    Cout, cout2 = w_eq.shape[3], head_w_eq.shape[3]
    pk, hk = PREPACKED[id(w_eq)], PREPACKED[id(head_w_eq)]
    d = _make_desc(B, N, C0, C1, Cout, 3, True, up0, flip_north_pole, act, alpha, vmax, nat.BF16, 0)
    d.flags |= nat.CONV_PREPACKED
    dh = _make_desc(B, N, Cout, 0, cout2, 1, False, False, flip_north_pole, nat.ACT_NONE, 0.0, 0.0, nat.BF16, 0)
    dh.flags |= nat.CONV_PREPACKED | (nat.CONV_OUT_PADDED if out_padded else 0)
    rows = (cout2 + 7) // 8 * 8 if out_padded else cout2
    y = torch.empty((B, 6, N, N, Cout), dtype=src0.dtype, device=src0.device)
    yh = torch.empty((B, 6, N, N, rows), dtype=src0.dtype, device=src0.device)
    table = nat.halo_tables(N, 1, src0.device)[0]
    nbytes = max(lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d)), lib().dlwpcs_conv_workspace_bytes(ctypes.byref(dh)))
    ws = _workspace(nbytes, src0.device)
    fused = ctypes.c_int(0)
    check(lib().dlwpcs_conv_fwd_head(ctypes.byref(d), ptr(src0), ptr(src1), ptr(pk[1]), ptr(pk[2]) if b_eq is not None else 0,
                                     ctypes.byref(dh), ptr(hk[1]), ptr(hk[2]) if head_b_eq is not None else 0, ptr(y), ptr(yh),
                                     ptr(table), ptr(ws), ws.numel(), ctypes.byref(fused), stream_ptr()), 'dlwpcs_conv_fwd_head')
    global HEAD_FOLDED
    HEAD_FOLDED = bool(fused.value)
    return yh


HEAD_FOLDED = False     # (what the last cs_conv_head call did: tests / diagnostics)


# ------------------------------------------------------------------------------------------------------------------ #
# Generic per-face convolution for the off-hot-path layer options (strides, dilation, 'same')
# ------------------------------------------------------------------------------------------------------------------ #

def _same_pads(n, k, s, dil):
    out = -(-n // s)
    total = max((out - 1) * s + (k - 1) * dil + 1 - n, 0)
    return total // 2, out


class _GConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_eq, w_pol, w_np, b_eq, b_pol, b_np, strides, padding, dilation, flip):
        require_device(x, 'cs_gconv')
        for prm in (w_eq, w_pol, w_np, b_eq, b_pol, b_np):
            _f32_param(prm, 'cs_gconv')
        x, w_eq, w_pol = _c(x), _c(w_eq), _c(w_pol)
        w_np = _c(w_np) if w_np is not None else None
        B, F6, H, W, Cin = x.shape
        kh, kw, cin, Cout = w_eq.shape
        if F6 != 6 or cin != Cin:
            raise ValueError('cs_gconv: bad shapes x=%s kernel=%s' % (tuple(x.shape), tuple(w_eq.shape)))
        sh, sw = strides
        dh, dw = dilation
        if padding == 'same':
            pad_t, Ho = _same_pads(H, kh, sh, dh)
            pad_l, Wo = _same_pads(W, kw, sw, dw)
        elif padding == 'valid':
            pad_t = pad_l = 0
            Ho = (H - (kh - 1) * dh - 1) // sh + 1
            Wo = (W - (kw - 1) * dw - 1) // sw + 1
        else:
            raise ValueError('padding must be "valid" or "same"')
        if Ho < 1 or Wo < 1:
            raise ValueError('cs_gconv: empty output')
        d = GConvDesc(B=B, H=H, W=W, Cin=Cin, Cout=Cout, kh=kh, kw=kw, sh=sh, sw=sw, dh=dh, dw=dw, pad_t=pad_t,
                      pad_l=pad_l, Ho=Ho, Wo=Wo, flip_north_pole=int(flip), dtype=nat.dtype_tag(x))
        y = torch.empty((B, 6, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
        check(lib().dlwpcs_gconv_fwd(ctypes.byref(d), ptr(x), ptr(w_eq), ptr(w_pol), ptr(w_np), ptr(b_eq), ptr(b_pol),
                                     ptr(b_np), ptr(y), stream_ptr()), 'dlwpcs_gconv_fwd')
        ctx.desc = d
        ctx.has = (w_np is not None, b_eq is not None, b_np is not None)
        ctx.save_for_backward(x, w_eq, w_pol, w_np)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_eq, w_pol, w_np = ctx.saved_tensors
        d = ctx.desc
        has_np, has_bias, has_bnp = ctx.has
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib().dlwpcs_gconv_bwd_data(ctypes.byref(d), ptr(dy), ptr(w_eq), ptr(w_pol), ptr(w_np), ptr(dx),
                                              stream_ptr()), 'dlwpcs_gconv_bwd_data')
        dw_eq, dw_pol = torch.empty_like(w_eq), torch.empty_like(w_pol)
        dw_np = torch.empty_like(w_np) if has_np else None
        db_eq = db_pol = db_np = None
        if has_bias:
            db_eq = torch.empty(d.Cout, dtype=torch.float32, device=dy.device)
            db_pol = torch.empty_like(db_eq)
            db_np = torch.empty_like(db_eq) if has_bnp else None
        check(lib().dlwpcs_gconv_bwd_weights(ctypes.byref(d), ptr(x), ptr(dy), ptr(dw_eq), ptr(dw_pol), ptr(dw_np),
                                             ptr(db_eq), ptr(db_pol), ptr(db_np), stream_ptr()),
              'dlwpcs_gconv_bwd_weights')
        return dx, dw_eq, dw_pol, dw_np, db_eq, db_pol, db_np, None, None, None, None


def cs_gconv(x, w_eq, w_pol, w_np=None, b_eq=None, b_pol=None, b_np=None, strides=(1, 1), padding='valid',
             dilation=(1, 1), flip_north_pole=True):
    return _GConv.apply(x, w_eq, w_pol, w_np, b_eq, b_pol, b_np, tuple(strides), padding, tuple(dilation),
                        bool(flip_north_pole))


# ------------------------------------------------------------------------------------------------------------------ #
# Keras stock ops of the U-Net (Azure/train_cs.py:197-199)
# ------------------------------------------------------------------------------------------------------------------ #

class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha, vmax):
        require_device(x, 'leaky_clip_relu')
        x = _c(x)
        y = torch.empty_like(x)
        check(lib().dlwpcs_act_fwd(ptr(x), ptr(y), x.numel(), nat.ACT_LEAKY_CLIP, alpha, vmax, nat.dtype_tag(x),
                                   stream_ptr()),
              'dlwpcs_act_fwd')
        ctx.alpha, ctx.vmax = alpha, vmax
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().dlwpcs_act_bwd(ptr(dy), ptr(y), ptr(dx), dy.numel(), nat.ACT_LEAKY_CLIP, ctx.alpha, ctx.vmax,
                                   nat.dtype_tag(dy), stream_ptr()), 'dlwpcs_act_bwd')
        return dx, None, None


def leaky_clip_relu(x, negative_slope=0.0, max_value=None):
    vmax = float('inf') if max_value is None else float(max_value)
    return _Act.apply(x, float(negative_slope), vmax)


def _bnc(x, what):
    require_device(x, what)
    if x.dim() != 5 or x.shape[1] != 6 or x.shape[2] != x.shape[3]:
        raise ValueError('%s: expected (B, 6, N, N, C), got %s' % (what, tuple(x.shape)))
    return x.shape[0], x.shape[2], x.shape[4]


class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, premask=None):
        B, N, C = _bnc(x, 'avgpool2')
        if N % 2:
            raise ValueError('avgpool2: odd face size %d' % N)
        x = _c(x)
        y = torch.empty((B, 6, N // 2, N // 2, C), dtype=x.dtype, device=x.device)
        check(lib().dlwpcs_avgpool2_fwd(ptr(x), ptr(y), B, N, C, nat.dtype_tag(x), stream_ptr()), 'dlwpcs_avgpool2_fwd')
        ctx.shape = (B, N, C)
        ctx.premask = premask
        if premask is not None:
            ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, C = ctx.shape
        dy = _c(dy)
        dx = torch.empty((B, 6, N, N, C), dtype=dy.dtype, device=dy.device)
        if ctx.premask is not None:
            (x,) = ctx.saved_tensors
            check(lib().dlwpcs_avgpool2_bwd_masked(ptr(dy), 0, ptr(x), ptr(dx), B, N, C, float(ctx.premask[0]),
                                                   float(ctx.premask[1]), nat.dtype_tag(dy), stream_ptr()),
                  'dlwpcs_avgpool2_bwd_masked')
            return dx, None
        check(lib().dlwpcs_avgpool2_bwd(ptr(dy), ptr(dx), B, N, C, nat.dtype_tag(dy), stream_ptr()),
              'dlwpcs_avgpool2_bwd')
        return dx, None


class _AvgPool2Skip(torch.autograd.Function):
    """(pooled, alias of x): x feeds the pooling AND later consumers (a U-Net skip connection).  Routing those consumers
    through the alias hands this node both gradients at once, so the backward is ONE pass
    dx = d_alias + avgpool2_bwd(d_pooled) instead of avgpool2_bwd + autograd's elementwise add."""

    @staticmethod
    def forward(ctx, x, premask=None):
        """premask = (negative_slope, max_value) | None: x is the output of an activated layer that expects its gradient
        pre-masked; the backward then writes act'(x) * (d_alias + avgpool2_bwd(d_pooled)) (the alias itself is NOT premask: its
        consumers hand back plain gradients)."""
        B, N, C = _bnc(x, 'avgpool2')
        if N % 2:
            raise ValueError('avgpool2: odd face size %d' % N)
        x = _c(x)
        y = _POOLED.pop(x.data_ptr(), None)         # the producing convolution pooled in its epilogue (want_pool)
        if y is None or tuple(y.shape) != (B, 6, N // 2, N // 2, C) or y.dtype != x.dtype:
            y = torch.empty((B, 6, N // 2, N // 2, C), dtype=x.dtype, device=x.device)
            check(lib().dlwpcs_avgpool2_fwd(ptr(x), ptr(y), B, N, C, nat.dtype_tag(x), stream_ptr()), 'dlwpcs_avgpool2_fwd')
        ctx.shape = (B, N, C)
        ctx.dtype = x.dtype
        ctx.premask = premask
        if premask is not None:
            ctx.save_for_backward(x)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        B, N, C = ctx.shape
        if ctx.premask is not None:
            (x,) = ctx.saved_tensors
            if dy is None:
                dx = _c(dskip).clone()
                check(lib().dlwpcs_act_bwd(ptr(dx), ptr(x), ptr(dx), dx.numel(), nat.ACT_LEAKY_CLIP, float(ctx.premask[0]),
                                           float(ctx.premask[1]), nat.dtype_tag(dx), stream_ptr()), 'dlwpcs_act_bwd')
                return dx, None
            ring = _pending_ring.pop(dy.data_ptr(), None)
            dy = _c(dy)
            dskip = _c(dskip) if dskip is not None else None
            dx = torch.empty((B, 6, N, N, C), dtype=dy.dtype, device=dy.device)
            if ring is not None:
                # dy holds the interior contributions of its producer's data gradient; the halo ring is still in that call's
                # workspace: added here, no fix-up launch
                rws, roff, rch, rinv = ring
                check(lib().dlwpcs_avgpool2_bwd_ring(ptr(dy), ptr(dskip), ptr(x), ptr(dx), B, N, C, float(ctx.premask[0]),
                                                     float(ctx.premask[1]), nat.dtype_tag(dy), rws.data_ptr() + roff, ptr(rinv),
                                                     rch, 0, stream_ptr()), 'dlwpcs_avgpool2_bwd_ring')
                return dx, None
            check(lib().dlwpcs_avgpool2_bwd_masked(ptr(dy), ptr(dskip), ptr(x), ptr(dx), B, N, C, float(ctx.premask[0]),
                                                   float(ctx.premask[1]), nat.dtype_tag(dy), stream_ptr()),
                  'dlwpcs_avgpool2_bwd_masked')
            return dx, None
        if dy is None:
            return dskip, None
        dy = _c(dy)
        dx = torch.empty((B, 6, N, N, C), dtype=dy.dtype, device=dy.device)
        if dskip is None:
            check(lib().dlwpcs_avgpool2_bwd(ptr(dy), ptr(dx), B, N, C, nat.dtype_tag(dy), stream_ptr()),
                  'dlwpcs_avgpool2_bwd')
        else:
            dskip = _c(dskip)
            check(lib().dlwpcs_avgpool2_bwd_add(ptr(dy), ptr(dskip), ptr(dx), B, N, C, nat.dtype_tag(dy), stream_ptr()),
                  'dlwpcs_avgpool2_bwd_add')
        return dx, None


def avgpool2_skip(x, premask=None):
    """-> (avgpool2(x), x') where x' aliases x and must be used by x's remaining consumers."""
    return _AvgPool2Skip.apply(x, premask)


class _Detour(torch.autograd.Function):
    """Identity with a node of its own in the autograd graph, no launch in either direction.  The split backward pass of the
    two-bucket exchange (DLWP.keras.Model) stops at the skip tensors; they are the SECOND output of the pooling node, and
    torch.autograd.grad(..., inputs=[that output]) would run every node that reaches the pooling node through its FIRST output
    as well (needed-ness is per node, not per edge) -- the whole encoder.  Behind a detour the capture point is a node only the
    decoder reaches."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def detour(x):
    return _Detour.apply(x)


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, N, C = _bnc(x, 'upsample2')
        x = _c(x)
        y = torch.empty((B, 6, 2 * N, 2 * N, C), dtype=x.dtype, device=x.device)
        check(lib().dlwpcs_upsample2_fwd(ptr(x), ptr(y), B, N, C, nat.dtype_tag(x), stream_ptr()),
              'dlwpcs_upsample2_fwd')
        ctx.shape = (B, N, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, C = ctx.shape
        dy = _c(dy)
        dx = torch.empty((B, 6, N, N, C), dtype=dy.dtype, device=dy.device)
        check(lib().dlwpcs_upsample2_bwd(ptr(dy), ptr(dx), B, N, C, nat.dtype_tag(dy), stream_ptr()),
              'dlwpcs_upsample2_bwd')
        return dx


def avgpool2(x, premask=None):
    return _AvgPool2.apply(x, premask)


def upsample2(x):
    return _Upsample2.apply(x)


class _Concat2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        require_device(a, 'concat2')
        require_device(b, 'concat2')
        if a.shape[:-1] != b.shape[:-1]:
            raise ValueError('concat2: leading shapes differ: %s vs %s' % (tuple(a.shape), tuple(b.shape)))
        if a.dtype != b.dtype:
            raise TypeError('concat2: dtypes differ: %s vs %s' % (a.dtype, b.dtype))
        a, b = _c(a), _c(b)
        Ca, Cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // Ca
        y = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=a.dtype, device=a.device)
        check(lib().dlwpcs_concat2(ptr(a), ptr(b), ptr(y), rows, Ca, Cb, nat.dtype_tag(a), stream_ptr()),
              'dlwpcs_concat2')
        ctx.c = (Ca, Cb, rows, a.shape, b.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        Ca, Cb, rows, sa, sb = ctx.c
        dy = _c(dy)
        da = torch.empty(sa, dtype=dy.dtype, device=dy.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(sb, dtype=dy.dtype, device=dy.device) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            check(lib().dlwpcs_split2(ptr(dy), ptr(da), ptr(db), rows, Ca, Cb, nat.dtype_tag(dy), stream_ptr()),
                  'dlwpcs_split2')
        return da, db


def concat_channels(tensors):
    out = tensors[0]
    for t in tensors[1:]:
        out = _Concat2.apply(out, t)
    return out


def state_repack(state, extra, n_time):
    """Next main input of a forecast step: `state` (B, *space, T*V) with `extra` (B, T, *space, E) appended as the last E
    channels of each of the T time steps -> (B, *space, T*(V+E)).  One launch (dlwpcs_state_repack); inference only."""
    require_device(state, 'state_repack')
    require_device(extra, 'state_repack')
    if state.dtype != extra.dtype:
        raise TypeError('state_repack: state is %s but extra is %s' % (state.dtype, extra.dtype))
    state, extra = _c(state), _c(extra)
    B, T = state.shape[0], int(n_time)
    space = tuple(state.shape[1:-1])
    if state.shape[-1] % T or tuple(extra.shape[:2]) != (B, T) or tuple(extra.shape[2:-1]) != space:
        raise ValueError('state_repack: state %s / extra %s do not match %d time steps'
                         % (tuple(state.shape), tuple(extra.shape), T))
    V, E = state.shape[-1] // T, extra.shape[-1]
    S = 1
    for v in space:
        S *= int(v)
    out = torch.empty((B,) + space + (T * (V + E),), dtype=state.dtype, device=state.device)
    check(lib().dlwpcs_state_repack(ptr(state), ptr(extra), ptr(out), B, S, T, V, E, nat.dtype_tag(state), stream_ptr()),
          'dlwpcs_state_repack')
    return out


class _Transpose(torch.autograd.Function):
    """(B, C, S...) <-> (B, S..., C) layout conversion for data_format='channels_first' callers."""

    @staticmethod
    def forward(ctx, x, to_last):
        require_device(x, 'layout')
        x = _c(x)
        B = x.shape[0]
        if to_last:
            C, spatial = x.shape[1], tuple(x.shape[2:])
            S = x.numel() // (B * C) if B * C else 0
            y = torch.empty((B,) + spatial + (C,), dtype=x.dtype, device=x.device)
            check(lib().dlwpcs_cf_to_cl(ptr(x), ptr(y), B, C, S, nat.dtype_tag(x), stream_ptr()), 'dlwpcs_cf_to_cl')
        else:
            C, spatial = x.shape[-1], tuple(x.shape[1:-1])
            S = x.numel() // (B * C) if B * C else 0
            y = torch.empty((B, C) + spatial, dtype=x.dtype, device=x.device)
            check(lib().dlwpcs_cl_to_cf(ptr(x), ptr(y), B, C, S, nat.dtype_tag(x), stream_ptr()), 'dlwpcs_cl_to_cf')
        ctx.to_last = to_last
        return y

    @staticmethod
    def backward(ctx, dy):
        return _Transpose.apply(dy, not ctx.to_last), None


def channels_first_to_last(x):
    return _Transpose.apply(x, True)


def channels_last_to_first(x):
    return _Transpose.apply(x, False)


# ------------------------------------------------------------------------------------------------------------------ #
# Loss and optimizer (Azure/train_cs.py:424-430)
# ------------------------------------------------------------------------------------------------------------------ #

_mse_scratch = {}
_unit_seeds = {}


def unit_seed(device):
    """The cached ones(2) tensor DLWP.keras.Model seeds the backward of every loss node with: _MSE.backward recognises it
    by address and returns the gradient the forward kernel already wrote; any other upstream gradient is applied."""
    key = str(device)
    t = _unit_seeds.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('first training step must run eagerly (graph capture allocates nothing)')
        t = torch.ones(2, dtype=torch.float32, device=device)
        _unit_seeds[key] = t
    return t


class _MSE(torch.autograd.Function):
    """returns a (2,) tensor: [weight * mse, mae]; gradient flows through element 0 only."""

    @staticmethod
    def forward(ctx, y, t, weight):
        require_device(y, 'mse')
        require_device(t, 'mse')
        if y.shape != t.shape:
            raise ValueError('mse: shapes differ: %s vs %s' % (tuple(y.shape), tuple(t.shape)))
        tag = nat.dtype_tag(y)
        if y.dtype != t.dtype:
            if t.dtype != torch.float32:
                raise TypeError('mse: prediction is %s but target is %s' % (y.dtype, t.dtype))
            tag |= nat.MSE_TARGET_F32           # bf16 prediction scored against the fp32 target
        y, t = _c(y), _c(t)
        key = str(y.device)
        scratch = _mse_scratch.get(key)
        if scratch is None:
            scratch = torch.empty(lib().dlwpcs_mse_scratch_bytes(), dtype=torch.uint8, device=y.device)
            _mse_scratch[key] = scratch
        out = torch.empty(2, dtype=torch.float32, device=y.device)
        dy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        tag |= nat.MSE_OVERWRITE                # out = ..., no zero-fill launch
        check(lib().dlwpcs_mse_fwd_bwd(ptr(y), ptr(t), ptr(dy), ptr(out), y.numel(), weight, tag, ptr(scratch),
                                       stream_ptr()), 'dlwpcs_mse_fwd_bwd')
        ctx.save_for_backward(dy)
        return out

    @staticmethod
    def backward(ctx, dout):
        (dy,) = ctx.saved_tensors
        # The forward kernel already wrote dy = weight * 2 (y - t) / n.  DLWP.keras.Model seeds the backward with
        # unit_seed() (upstream gradient of out[0] exactly 1): no extra pass.  Any other upstream gradient (a scaled or
        # combined loss built on this op) is applied here; out[1] (mae) carries no gradient.
        seed = _unit_seeds.get(str(dout.device))
        if seed is not None and dout.data_ptr() == seed.data_ptr():
            return dy, None, None
        return (dy.float() * dout[0]).to(dy.dtype), None, None


def mse_mae(y, t, weight=1.0):
    return _MSE.apply(y, t, float(weight))


_head_scratch = {}


def head_mse_applicable(x, w_eq, ksize, act, target):
    """True when the fused training tail (dlwpcs_head_mse_step) serves this output layer + loss: bf16 activations, pointwise
    kernel on 32 input channels, even C_out in 8..32, no activation, fp32 target, weights packed by the model's pack launch."""
    if not (DIRECT_PARAM_GRADS and x.is_cuda and x.dtype == torch.bfloat16 and ksize == 1 and act == nat.ACT_NONE):
        return False
    packed = PREPACKED.get(id(w_eq))
    cout = w_eq.shape[3]
    return (packed is not None and packed[0] == nat.BF16 and x.dim() == 5 and x.shape[1] == 6 and x.shape[4] == 32
            and w_eq.shape[2] == 32 and cout % 2 == 0 and 8 <= cout <= 32 and (x.shape[2] * x.shape[3]) % 16 == 0
            and target.dtype == torch.float32 and tuple(target.shape) == tuple(x.shape[:4]) + (cout,)
            and x.numel() < (1 << 31) and x.requires_grad)


class _HeadMSE(torch.autograd.Function):
    """stats = [weight * mse, mae] of the pointwise output layer applied to x, against `target`; the forward launch also
    leaves dy and dx behind (dlwpcs_head_mse_step), the backward only runs the layer's weight gradient."""

    @staticmethod
    def forward(ctx, x, target, w_eq, w_pol, b_eq, b_pol, weight, flip, premask=None):
        x, target = _c(x), _c(target)
        B, _, N, _, C0 = x.shape
        Cout = w_eq.shape[3]
        d = _make_desc(B, N, C0, 0, Cout, 1, False, False, flip, nat.ACT_NONE, 0.0, 0.0, nat.BF16)
        packed = PREPACKED[id(w_eq)]
        d.flags |= nat.CONV_PREPACKED
        key = str(x.device)
        scratch = _head_scratch.get(key)
        if scratch is None:
            scratch = torch.empty(lib().dlwpcs_head_mse_scratch_bytes(), dtype=torch.uint8, device=x.device)
            _head_scratch[key] = scratch
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        dy = torch.empty((B, 6, N, N, Cout), dtype=x.dtype, device=x.device)
        dx = torch.empty_like(x)
        defer = DEFER_LOSS_TAIL and not _pending_tail          # (one scratch buffer per device: one deferred tail at a time)
        ow = 1 | (nat.HEAD_DEFER_STAGE2 if defer else 0)
        if premask is not None:
            check(lib().dlwpcs_head_mse_step_masked(ctypes.byref(d), ptr(x), ptr(packed[1]),
                                                    ptr(packed[2]) if b_eq is not None else 0, ptr(packed[3]), ptr(target),
                                                    float(weight), ptr(dy), ptr(dx), ptr(out), ow, ptr(scratch),
                                                    float(premask[0]), float(premask[1]), stream_ptr()),
                  'dlwpcs_head_mse_step_masked')
        else:
            check(lib().dlwpcs_head_mse_step(ctypes.byref(d), ptr(x), ptr(packed[1]), ptr(packed[2]) if b_eq is not None else 0,
                                             ptr(packed[3]), ptr(target), float(weight), ptr(dy), ptr(dx), ptr(out), ow,
                                             ptr(scratch), stream_ptr()), 'dlwpcs_head_mse_step')
        if defer:
            tail = nat.LossTail()
            check(lib().dlwpcs_head_mse_tail(ctypes.byref(d), float(weight), 1, ptr(scratch), ptr(out), ctypes.byref(tail)),
                  'dlwpcs_head_mse_tail')
            _pending_tail.append((tail, (scratch, out)))
        ctx.desc = d
        ctx.params = (w_eq, w_pol, None, b_eq, b_pol, None)
        ctx.save_for_backward(x, dy, dx)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, dy, dx = ctx.saved_tensors
        seed = _unit_seeds.get(str(dout.device))
        if seed is None or dout.data_ptr() != seed.data_ptr():
            raise RuntimeError('the fused head + loss step is seeded by DLWP.keras.Model only (upstream gradient 1)')
        d = ctx.desc
        dev = dy.device
        nbytes = lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
        need = ctx.needs_input_grad
        direct = DIRECT_PARAM_GRADS and all(p is None or (p.is_leaf and p.grad is not None and p.grad.is_contiguous())
                                            for p in ctx.params)
        if not direct:
            raise RuntimeError('the fused head + loss step accumulates straight into the model\'s flat gradient buffer')
        batching = WGRAD_BATCH and not WGRAD_SIDE_STREAM and wgrad_batch_supported(d)
        defer = DEFER_WGRAD_REDUCE and not WGRAD_SIDE_STREAM and not batching
        ws = _workspace(nbytes, dev, 'defer%d' % len(_deferred)) if defer else _workspace(nbytes, dev)
        _weight_gradients(d, x, None, dy, None, ctx.params, None, ws, nbytes, True, defer, need[2:6],
                          False, ctx.params[3] is not None, False)
        return (dx if need[0] else None), None, None, None, None, None, None, None, None


def head_mse(x, target, w_eq, w_pol, b_eq, b_pol, weight=1.0, flip_north_pole=True, premask=None):
    return _HeadMSE.apply(x, target, w_eq, w_pol, b_eq, b_pol, float(weight), bool(flip_north_pole), premask)


def adam_step(p, g, m, v, step_dev, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, zero_grads=False):
    """In-place TF2.1-keras Adam on flat fp32 device buffers.  `step_dev`: int32 device tensor; with two elements
    {t-1, ticket = 0} the update and the step increment are ONE launch (dlwpcs_adam_step_fused; zero_grads clears g
    once it has been consumed), with one element the two-launch dlwpcs_adam_step."""
    for t in (p, g, m, v):
        require_device(t, 'adam_step')
        _f32_param(t, 'adam_step')
    if step_dev.numel() >= 2:
        check(lib().dlwpcs_adam_step_fused(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(step_dev), lr, beta1, beta2,
                                           eps, grad_scale, nat.ADAM_ZERO_GRAD if zero_grads else 0, stream_ptr()),
              'dlwpcs_adam_step_fused')
        return
    check(lib().dlwpcs_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(step_dev), lr, beta1, beta2, eps,
                                 grad_scale, stream_ptr()), 'dlwpcs_adam_step')
    if zero_grads:
        g.zero_()


def adam_step_dev(p, g, m, v, state_dev, hyper_dev, zero_grads=False):
    """dlwpcs_adam_step_dev: the fused Adam launch with {lr, beta1, beta2, eps, grad_scale} read from the 5-float device
    tensor `hyper_dev` -- the form a captured hipGraph needs to honour learning-rate changes between replays."""
    for t in (p, g, m, v, hyper_dev):
        require_device(t, 'adam_step_dev')
        _f32_param(t, 'adam_step_dev')
    if hyper_dev.numel() < 5 or state_dev.numel() < 2:
        raise ValueError('adam_step_dev: hyper_dev needs 5 floats, state_dev 2 int32')
    check(lib().dlwpcs_adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(state_dev), ptr(hyper_dev),
                                     nat.ADAM_ZERO_GRAD if zero_grads else 0, stream_ptr()), 'dlwpcs_adam_step_dev')


def add(a, b):
    require_device(a, 'add')
    require_device(b, 'add')
    a, b = _c(a), _c(b)
    y = torch.empty_like(a)
    check(lib().dlwpcs_add(ptr(a), ptr(b), ptr(y), a.numel(), nat.dtype_tag(a), stream_ptr()), 'dlwpcs_add')
    return y


# ------------------------------------------------------------------------------------------------------------------ #
# Batch feed (reference DLWP/model/generators.py:872-984): gather a batch window out of the HBM-resident data array
# ------------------------------------------------------------------------------------------------------------------ #

def batch_gather(array, samples, var_idx, out, n_steps, t_off, t_stride, c_off, c_stride, channels_last=True):
    """
    array (T, V, *space) fp32 device tensor; samples (B,) / var_idx (nv,) int32 device tensors; out (B, *space, Ctot) or
    (B, Ctot, *space), float32 or bfloat16, written in place: channel c_off + n*c_stride + j <- array[samples + t_off +
    n*t_stride, var_idx[j]].
    """
    for t in (array, out):
        require_device(t, 'batch_gather')
    if array.dtype != torch.float32 or not array.is_contiguous() or not out.is_contiguous():
        raise TypeError('batch_gather: array must be contiguous float32, out contiguous')
    T, V = int(array.shape[0]), int(array.shape[1])
    S = int(array[0, 0].numel())
    B = int(samples.numel())
    Ctot = int(out.shape[-1] if channels_last else out.shape[1])
    if out.numel() != B * S * Ctot:
        raise ValueError('batch_gather: out shape %s does not match batch %d, space %d' % (tuple(out.shape), B, S))
    check(lib().dlwpcs_batch_gather(ptr(array), T, V, S, ptr(samples), B, ptr(var_idx), int(var_idx.numel()),
                                    int(n_steps), int(t_off), int(t_stride), ptr(out), Ctot, int(c_off), int(c_stride),
                                    1 if channels_last else 0, nat.dtype_tag(out), stream_ptr()), 'dlwpcs_batch_gather')
    return out
