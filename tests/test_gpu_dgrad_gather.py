"""
GPU parity of the data gradient in GATHER form (include/dlwpcs.h: DLWPCS_CONV_DGRAD_GATHER, csrc/conv_ws.h EDGE): every cell of the
gradient -- border cells included -- is complete when the kernel stores it; no halo ring, no fix-up launch.

Checker: autograd (fp64, torch-CPU) of the oracle's forward -- nearest upsample / concat / CubeSpherePadding2D gather / per-face
correlation with the pole kernels (DLWP/custom.py:1198-1308, :921-1002; Azure/train_cs.py:196-228) -- on the bf16-rounded operands
the device consumes, times act'(source) where a mask is asked for.  Tolerance: the device accumulates in fp32 and rounds the result
to bf16 ONCE (1 ulp of max|ref| covers rounding + accumulation order); the upsampled source sums four rounded values (3 ulp).  The
padded-grid path of rounds 1-3 (flags without the bit) is run beside it: same numbers within its own 3-ulp border tolerance.
"""
import ctypes

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


def _bf(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32).to(torch.bfloat16)


def _slope(y):
    return torch.where(y < 0, torch.full_like(y, ALPHA), torch.where((y > 0) & (y < VMAX), torch.ones_like(y), torch.zeros_like(y)))


def _oracle(src0, src1, up0, w_eq, w_pol, dz, mask0, mask1):
    """fp64 gradients w.r.t. the sources of  conv(pad(concat(up(src0), src1)))"""
    s0 = src0.double().clone().requires_grad_(True)
    s1 = src1.double().clone().requires_grad_(True) if src1 is not None else None
    x = orc.upsample_122(s0) if up0 else s0
    if s1 is not None:
        x = torch.cat([x, s1], dim=-1)
    z = orc.cs_conv2d(orc.cs_pad(x, 1), w_eq.double(), w_pol.double(), flip_north_pole=True)
    z.backward(dz.double())
    g0 = s0.grad * (_slope(src0.double()) if mask0 else 1.0)
    g1 = None
    if s1 is not None:
        g1 = s1.grad * (_slope(src1.double()) if mask1 else 1.0)
    return g0, g1


def _run(case, gather, seed):
    from DLWP import _native as nat
    B, N, C0, C1, up0, Cout, mask0, mask1 = case
    rng = np.random.default_rng(seed)
    dev = _dev()
    n0 = N // 2 if up0 else N
    src0 = _bf(rng.standard_normal((B, 6, n0, n0, C0)) * 6.0)
    src1 = _bf(rng.standard_normal((B, 6, N, N, C1)) * 6.0) if C1 else None
    dz = _bf(rng.standard_normal((B, 6, N, N, Cout)))
    w = [_bf(rng.standard_normal((3, 3, C0 + C1, Cout)) / np.sqrt(9 * (C0 + C1))).float() for _ in range(2)]
    d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=3, halo=1, up0=int(up0), flip_north_pole=1, act=0,
                     alpha=0., vmax=0., dtype=nat.BF16, flags=nat.CONV_DGRAD_GATHER if gather else 0, c0_valid=0)
    nbytes = nat.lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    assert nat.dgrad_gather_ready(N, 1, dev)
    inv = nat.halo_tables(N, 1, dev)[1]
    s0d, s1d, dzd = src0.to(dev), (src1.to(dev) if C1 else None), dz.to(dev)
    wd = [t.to(dev) for t in w]
    g0 = torch.full_like(s0d, float('nan'))
    g1 = torch.full_like(s1d, float('nan')) if C1 else None
    nat.check(nat.lib().dlwpcs_conv_bwd_data_masked(ctypes.byref(d), nat.ptr(dzd), nat.ptr(wd[0]), nat.ptr(wd[1]), 0,
                                                    nat.ptr(g0), nat.ptr(g1), nat.ptr(s0d if mask0 else None),
                                                    nat.ptr(s1d if mask1 else None), ALPHA, VMAX, nat.ptr(inv),
                                                    nat.ptr(ws), nbytes, nat.stream_ptr()), 'conv_bwd_data_masked')
    torch.cuda.synchronize()
    return (g0.float().cpu(), g1.float().cpu() if C1 else None), (src0, src1, w, dz)


# (B, N, C0, C1, up0, Cout, mask0, mask1)
CASES = [
    (2, 48, 32, 0, 0, 32, True, False),       # the U-Net's 32 -> 32 layers at N = 48: 384-pixel tiles, four consumer waves of 2 rows
    (2, 48, 32, 0, 0, 32, False, False),
    (2, 24, 64, 0, 0, 64, True, False),       # 64 gradient channels (two N tiles)
    (2, 12, 128, 0, 0, 64, True, False),      # 128 gradient channels, the whole face in one tile (MT = 5)
    (2, 12, 64, 0, 0, 128, True, False),      # 128 dz channels: eight operand groups per edge term
    (2, 24, 64, 0, 0, 32, True, False),
    (2, 24, 32, 0, 0, 64, False, False),
    (2, 24, 64, 64, 1, 64, True, False),      # decoder: upsampled source through the workspace + 2 x 2 sum, skip source direct
    (2, 24, 64, 64, 1, 64, True, True),
    (2, 48, 32, 32, 1, 32, False, False),
    (2, 16, 32, 32, 0, 32, True, True),       # two directly written sources
    (1, 8, 8, 0, 0, 8, False, False),         # smallest face the plan serves
    (3, 10, 16, 0, 0, 24, True, False),       # 24 dz channels: the second operand group is half empty
    (1, 96, 32, 0, 0, 32, True, False),       # N = 96: a consumer wave owns one row
    (33, 24, 32, 0, 0, 32, False, False),     # more samples than a workgroup's run of one (face, band)
]


@pytest.mark.parametrize('case', CASES)
def test_gather_form_matches_the_oracle(case):
    B, N, C0, C1, up0, Cout, mask0, mask1 = case
    seed = abs(hash(case)) % (2 ** 31)
    (g0, g1), (src0, src1, w, dz) = _run(case, True, seed)
    r0, r1 = _oracle(src0.float(), src1.float() if src1 is not None else None, up0, w[0], w[1], dz.float(), mask0, mask1)
    # (N <= 16: a tile holds both edge rows of a face -> the library keeps the padded-grid path, whose border cells round twice)
    one = 1.0 if N >= 20 else 3.0
    for g, r, ulps in ((g0, r0, 3.0 if up0 else one), (g1, r1, one)):
        if g is None:
            continue
        assert torch.isfinite(g).all(), case
        err = (g.double() - r).abs().max().item()
        assert err <= ulps * EPS * r.abs().max().item(), (case, err / (EPS * r.abs().max().item()))
        # border cells are where the two forms differ: look at them on their own as well
        n = g.shape[2]
        bm = torch.zeros(n, n, dtype=torch.bool)
        bm[0] = bm[-1] = True
        bm[:, 0] = bm[:, -1] = True
        eb = (g.double() - r)[:, :, bm].abs().max().item()
        assert eb <= ulps * EPS * r.abs().max().item(), (case, 'border')


@pytest.mark.parametrize('case', [CASES[0], CASES[2], CASES[8], CASES[10]])
def test_gather_form_against_the_padded_grid_path(case):
    seed = 11
    (a0, a1), _ = _run(case, True, seed)
    (b0, b1), _ = _run(case, False, seed)
    for a, b in ((a0, b0), (a1, b1)):
        if a is None:
            continue
        assert (a - b).abs().max().item() <= 5 * EPS * b.abs().max().item(), case
        # interior cells of directly written sources: one accumulation order in both forms up to the fp32 sum of the nine taps
        assert (a - b)[:, :, 1:-1, 1:-1].abs().max().item() <= 2 * EPS * b.abs().max().item(), case


def test_gather_form_is_reproducible_bit_for_bit():
    case = CASES[0]
    (a0, _), _ = _run(case, True, 5)
    (b0, _), _ = _run(case, True, 5)
    assert torch.equal(a0, b0)


@pytest.mark.parametrize('N', [16, 48])
def test_unet2_training_in_gather_form(N):
    """A bf16 `unet2` trains through the gather-form data gradients (engine option dgrad_gather): no ring fix-up / inverse-gather launch
    on the levels the form serves, the same loss and update direction as the padded-grid path, eager steps and hipGraph replays."""
    from DLWP import ops
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    C, B = 14, 4
    rng = np.random.default_rng(N)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev)
    w0, out = None, []
    import os
    try:
        for on in (False, True):
            os.environ['DLWPCS_OPTIONS'] = 'dgrad_gather=%d' % int(on)
            backend.set_compute_dtype('bfloat16')
            try:
                np.random.seed(5)
                model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
            finally:
                backend.set_compute_dtype('float32')
            model.compile(optimizer='adam', loss='mse', metrics=['mae'])
            if w0 is None:
                w0 = model.get_weights()
            model.set_weights(w0)
            for _ in range(5):                      # eager warm-up, capture, replays
                stats = model.train_on_device_batch([x], [t])
            torch.cuda.synchronize()
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
    finally:
        os.environ.pop('DLWPCS_OPTIONS', None)
    (p_off, s_off), (p_on, s_on) = out
    flat0 = np.concatenate([w.ravel() for w in w0])
    d_off, d_on = p_off - flat0, p_on - flat0
    cos = float(np.dot(d_off, d_on) / (np.linalg.norm(d_off) * np.linalg.norm(d_on)))
    assert np.isfinite(p_on).all() and cos > 0.99, cos
    assert abs(s_on[0, 0] - s_off[0, 0]) <= 5e-3 * abs(s_off[0, 0])


def test_lds_reads_beyond_the_allocation_return_zero():
    """The gather form masks MFMA operands by address: lanes that must add nothing read LDS beyond the workgroup's allocation
    (csrc/conv_ws.h, EDGE).  The hardware returns zeros there; dlwpcs_lds_oob_probe checks it on THIS device."""
    from DLWP import _native as nat
    nz = torch.full((1,), -1, dtype=torch.int32, device=_dev())
    nat.check(nat.lib().dlwpcs_lds_oob_probe(nat.ptr(nz), nat.stream_ptr()), 'lds_oob_probe')
    torch.cuda.synchronize()
    assert int(nz.item()) == 0


def test_a_device_that_fails_the_probe_trains_on_the_fallback_kernels(monkeypatch):
    """DLWP._native.lds_oob_reads_zero gates the gather form and the batched weight gradient: with the probe's answer forced to
    'non-zero words' a bf16 `unet2` step runs the padded-grid data gradient + per-layer weight gradients (one warning) -- BITWISE the
    step of the engine options dgrad_gather=0,wgrad_batch=0 (the same kernels, chosen by the gate instead of by hand) -- and moves
    the weights the way the default kernels do."""
    import os
    import warnings
    from DLWP import _native as nat
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 24, 14, 4
    rng = np.random.default_rng(11)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev)
    w0, out = None, []
    try:
        for mode in ('default', 'probe_fails', 'options_off'):
            monkeypatch.setattr(nat, '_lds_probe', {})
            monkeypatch.setattr(nat, '_run_lds_oob_probe', (lambda device: 3) if mode == 'probe_fails' else (lambda device: 0))
            os.environ['DLWPCS_OPTIONS'] = 'dgrad_gather=0,wgrad_batch=0' if mode == 'options_off' else ''
            backend.set_compute_dtype('bfloat16')
            try:
                np.random.seed(5)
                model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
            finally:
                backend.set_compute_dtype('float32')
            model.compile(optimizer='adam', loss='mse', metrics=['mae'])
            if w0 is None:
                w0 = model.get_weights()
            model.set_weights(w0)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                for _ in range(4):
                    stats = model.train_on_device_batch([x], [t])
                torch.cuda.synchronize()
            warned = [m for m in w if 'dlwpcs_lds_oob_probe' in str(m.message)]
            assert len(warned) == (1 if mode == 'probe_fails' else 0)
            assert model.batch_wgrad == (mode == 'default')
            if mode != 'options_off':
                assert nat.dgrad_gather_ready(N, 1, dev) == (mode == 'default')
            out.append((np.concatenate([p.ravel() for p in model.get_weights()]), stats.cpu().numpy().copy()))
    finally:
        os.environ.pop('DLWPCS_OPTIONS', None)
        monkeypatch.setattr(nat, '_lds_probe', {})
    (p_ok, s_ok), (p_fb, s_fb), (p_off, s_off) = out
    assert np.array_equal(p_fb, p_off) and np.array_equal(s_fb, s_off)
    flat0 = np.concatenate([p.ravel() for p in w0])
    a, b = p_ok - flat0, p_fb - flat0
    cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))
    # (four Adam steps normalise every entry's update to ~lr: entries whose gradient is bf16 noise flip freely, the direction of the
    # rest must agree; the loss after the steps pins the size)
    assert np.isfinite(p_fb).all() and cos > 0.9, cos
    assert abs(s_fb[0, 0] - s_ok[0, 0]) <= 5e-3 * abs(s_ok[0, 0])
