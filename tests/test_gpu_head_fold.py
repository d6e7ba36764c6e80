"""
GPU tests of dlwpcs_conv_fwd_head (include/dlwpcs.h): the pointwise output layer of the U-Net (Azure/train_cs.py:300-305, a
CubeSphereConv2D with kernel_size 1, DLWP/custom.py:921-1002) folded into the epilogue of the 3x3 CubeSphereConv2D in front of it
on inference passes.

Checkers: (1) the fp64 oracle (oracle/cs_oracle.py) of the two layers evaluated on the bf16-rounded operands, the intermediate
tensor rounded to bf16 where the two-launch path stores it; (2) the two launches of dlwpcs_conv_fwd themselves (pinned to the oracle
in tests/test_gpu_bf16.py): the folded result may differ from theirs only by the summation order inside the head's MFMA (fp32
accumulation in both), i.e. by one bf16 rounding step of the stored value.  Shapes the fold does not serve (head rows that are
not 32 channels) must run the two launches and give THEIR bits.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -8
ALPHA, VMAX = 0.1, 10.0


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def _f32(t):
    return t.detach().float().cpu().numpy()


def _bf_round(a):
    return torch.tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def _layers(rng, cin, cmid, cout2, dev):
    """fp32 parameters of (3x3 layer cin -> cmid, pointwise head cmid -> cout2), bf16-representable values"""
    mk = lambda *s, sc=1.0: torch.tensor(_bf_round(rng.standard_normal(s) * sc), dtype=torch.float32, device=dev)
    w = [mk(3, 3, cin, cmid, sc=(9 * cin) ** -0.5), mk(3, 3, cin, cmid, sc=(9 * cin) ** -0.5), mk(cmid, sc=0.2), mk(cmid, sc=0.2)]
    h = [mk(1, 1, cmid, cout2, sc=cmid ** -0.5), mk(1, 1, cmid, cout2, sc=cmid ** -0.5), mk(cout2, sc=0.2), mk(cout2, sc=0.2)]
    return w, h


def _prepack(layers, dev):
    """what DLWP.keras.Model does before a pass: one dlwpcs_pack_batch launch, ops.PREPACKED keyed by the equatorial kernels"""
    from DLWP import _native as nat
    from DLWP import ops
    entries, table = [], {}
    for (we, wp, be, bp), k in layers:
        bufs = ops.conv_packed_buffers(k, we.shape[2], we.shape[3], nat.BF16, dev, bias=be is not None)
        entries.append((we, wp, None, be, bp, None, bufs, k, True, nat.BF16))
        table[id(we)] = (nat.BF16, bufs[0], bufs[1], bufs[2])
    items = ops.make_pack_items(entries, dev)
    ops.pack_batch(items, len(entries))
    torch.cuda.synchronize()
    return table, entries, items


def _oracle(x, x1, up0, w, h, act):
    """fp64: head(bf16(act(conv(halo_pad(concat(up?(x), x1)))))) on bf16-rounded operands"""
    t = lambda a: torch.tensor(_f32(a), dtype=torch.float64)
    v = t(x)
    if up0:
        v = orc.upsample_122(v)
    if x1 is not None:
        v = torch.cat([v, t(x1)], dim=-1)
    y = orc.cs_conv2d(orc.cs_pad(v, 1), t(w[0]), t(w[1]), None, t(w[2]), t(w[3]))
    if act:
        y = orc.relu_leaky_clip(y, ALPHA, VMAX)
    yb = torch.tensor(_bf_round(y.numpy()), dtype=torch.float64)
    return orc.cs_conv2d(yb, t(h[0]), t(h[1]), None, t(h[2]), t(h[3])).numpy()


@pytest.mark.parametrize('B,N,C0,C1,up0,cout2,padded,act', [
    (3, 24, 32, 0, False, 26, True, True),       # the rollout's head: 26 = 13 variables x 2 steps, rows padded to 32
    (2, 48, 32, 0, False, 32, False, True),      # 32 output channels, plain rows
    (2, 16, 64, 0, False, 26, True, True),       # one-tile faces, two channel chunks
    (2, 24, 32, 32, True, 30, True, False),      # upsampled + skip source in front, no activation between
    (1, 96, 32, 0, False, 26, True, True),       # BASELINE config 5's face size
])
def test_folded_head_matches_the_two_launches_and_the_oracle(B, N, C0, C1, up0, cout2, padded, act):
    from DLWP import _native as nat
    from DLWP import ops
    dev = _dev()
    rng = np.random.default_rng(100 + N + cout2)
    w, h = _layers(rng, C0 + C1, 32, cout2, dev)
    n0 = N // 2 if up0 else N
    x = torch.tensor(rng.standard_normal((B, 6, n0, n0, C0)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    x1 = torch.tensor(rng.standard_normal((B, 6, N, N, C1)), dtype=torch.float32, device=dev).to(torch.bfloat16) if C1 else None
    table, keep, items = _prepack([(w, 3), (h, 1)], dev)
    ops.PREPACKED = table
    try:
        with torch.no_grad():
            assert ops.cs_conv_head_applicable(x, x1, w[0], h[0], padded)
            a = nat.ACT_LEAKY_CLIP if act else nat.ACT_NONE
            yh = ops.cs_conv_head(x, w[0], w[2], h[0], h[2], src1=x1, up0=up0, act=a, alpha=ALPHA, vmax=VMAX, out_padded=padded)
            assert ops.HEAD_FOLDED, 'this shape is served by the folded epilogue'
            y = ops.cs_conv(x, w[0], w[1], None, w[2], w[3], src1=x1, ksize=3, halo=True, up0=up0, act=a, alpha=ALPHA, vmax=VMAX)
            y2 = ops.cs_conv(y, h[0], h[1], None, h[2], h[3], ksize=1, halo=False, out_padded=padded)
        torch.cuda.synchronize()
    finally:
        ops.PREPACKED = {}
    rows = 32 if (padded or cout2 == 32) else cout2
    assert tuple(yh.shape) == (B, 6, N, N, rows) and tuple(y2.shape) == tuple(yh.shape)
    got, two = _f32(yh), _f32(y2)
    if rows > cout2:
        assert np.abs(got[..., cout2:]).max() == 0.0, 'padding channels are zero'
    ref = _oracle(x, x1, up0, w, h, act)
    scale = np.abs(ref).max()
    assert scale > 0.5
    # one bf16 rounding step of the stored value between the two forms; the oracle's intermediate tensor rounds to bf16 from the
    # exact sum (the kernels from an fp32 sum): a flipped rounding of y moves the head's result by |w_head| * ulp(y) -- two steps
    assert np.abs(got[..., :cout2] - two[..., :cout2]).max() <= 1.0 * EPS * scale
    assert np.abs(got[..., :cout2] - ref).max() <= 2.0 * EPS * scale
    assert np.abs(two[..., :cout2] - ref).max() <= 2.0 * EPS * scale
    # not a near miss of a systematic error: the mean deviation from the oracle is a fraction of a rounding step
    assert np.abs(got[..., :cout2] - ref).mean() <= 0.2 * EPS * scale


@pytest.mark.parametrize('cout2,padded', [(14, False), (14, True), (26, False)])
def test_head_rows_that_are_not_32_channels_run_the_two_launches(cout2, padded):
    """the contract's fallback: *fused = 0, y holds the layer's own output, y_head the bits of dlwpcs_conv_fwd on it"""
    from DLWP import _native as nat
    from DLWP import ops
    dev = _dev()
    rng = np.random.default_rng(7 + cout2)
    B, N = 2, 24
    w, h = _layers(rng, 32, 32, cout2, dev)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, 32)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    table, keep, items = _prepack([(w, 3), (h, 1)], dev)
    ops.PREPACKED = table
    try:
        with torch.no_grad():
            assert not ops.cs_conv_head_applicable(x, None, w[0], h[0], padded)
            yh = ops.cs_conv_head(x, w[0], w[2], h[0], h[2], act=nat.ACT_LEAKY_CLIP, alpha=ALPHA, vmax=VMAX, out_padded=padded)
            assert not ops.HEAD_FOLDED
            y = ops.cs_conv(x, w[0], w[1], None, w[2], w[3], ksize=3, halo=True, act=nat.ACT_LEAKY_CLIP, alpha=ALPHA, vmax=VMAX)
            y2 = ops.cs_conv(y, h[0], h[1], None, h[2], h[3], ksize=1, halo=False, out_padded=padded)
        torch.cuda.synchronize()
    finally:
        ops.PREPACKED = {}
    assert torch.equal(yh, y2)


def test_conv_fwd_head_rejects_inconsistent_descriptors():
    from DLWP import _native as nat
    from DLWP import ops
    dev = _dev()
    lib = nat.lib()
    B, N = 1, 16
    d = ops._make_desc(B, N, 32, 0, 32, 3, True, False, True, nat.ACT_NONE, 0., 0., nat.BF16, 0)
    dh = ops._make_desc(B, N, 32, 0, 26, 1, False, False, True, nat.ACT_NONE, 0., 0., nat.BF16, 0)
    buf = torch.zeros(1 << 22, dtype=torch.uint8, device=dev)
    p = buf.data_ptr()
    fused = ctypes.c_int(-1)
    call = lambda: lib.dlwpcs_conv_fwd_head(ctypes.byref(d), p, 0, p, p, ctypes.byref(dh), p, p, p, p, p, p, buf.numel(),
                                            ctypes.byref(fused), 0)
    assert call() == -1 and b'PREPACKED' in lib.dlwpcs_last_error()      # raw kernels are not accepted
    d.flags |= nat.CONV_PREPACKED
    dh.flags |= nat.CONV_PREPACKED
    dh.N = N + 2
    assert call() == -1 and b'does not consume' in lib.dlwpcs_last_error()
    dh.N = N
    dh.C0 = 16
    assert call() == -1
    assert fused.value == 0


def test_rollout_model_folds_its_head_and_matches_the_unfolded_passes():
    """bf16 rollout of a 26-channel unet2 (BASELINE config 5's layout, small face): passes with the head folded into the last
    convolution's epilogue (engine option fold_head, default) against passes without -- within one bf16 rounding step per pass of
    the state's scale, and the engine really folded (ops.HEAD_FOLDED); a training step of the same model is untouched by the option"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    from DLWP import ops
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 16, 26, 2
    rng = np.random.default_rng(19)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    outs = []
    for flag in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'fold_head=' + flag
        try:
            backend.set_compute_dtype('bfloat16')
            try:
                np.random.seed(3)
                model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
            finally:
                backend.set_compute_dtype('float32')
            assert model._head_fold, 'the plan found the (last convolution, pointwise head) pair'
            ops.HEAD_FOLDED = False
            s = model.predict_on_device(x, repack=True, padded_io=True)
            torch.cuda.synchronize()
            assert ops.HEAD_FOLDED == (flag == '1')
            assert s.shape[-1] == 32 and float(s[..., 26:].float().abs().max()) == 0.0
            outs.append(_f32(s[..., :C]))
            # without the padded state the head's rows are 26 channels: never folded, whatever the option says
            ops.HEAD_FOLDED = False
            s26 = model.predict_on_device(x, repack=False, padded_io=False)
            torch.cuda.synchronize()
            assert not ops.HEAD_FOLDED and s26.shape[-1] == 26
            if flag == '0':
                assert np.array_equal(_f32(s26), outs[0])
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    scale = np.abs(outs[0]).max()
    assert scale > 1e-3
    assert np.abs(outs[0] - outs[1]).max() <= 1.0 * EPS * scale
