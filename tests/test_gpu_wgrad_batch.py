"""
GPU parity tests of the batched weight gradient (dlwpcs_wgrad_batch: one persistent launch + one reduction for all layers of
a backward pass, csrc/wgrad_batch.hip) through the C ABI.  The checker is the fp64 oracle (oracle/cs_oracle.py: the
reference's CubeSpherePadding2D + CubeSphereConv2D, DLWP/custom.py:921-1002,1082-1308, under torch autograd) evaluated on
exactly the bf16 numbers the device reads (inputs and dz rounded to bf16 first): what is left is fp32 accumulation order.

Tolerance: max|dW - ref| <= 2e-5 * max|ref| per tensor (fp32 accumulation of up to ~10^5 bf16 products per output; the
partial sums are combined in a fixed order, so results are also bitwise reproducible -- checked).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def _bf(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32).to(torch.bfloat16).to(_dev())


def _f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32).to(_dev())


def _f64(t):
    return t.detach().to(torch.float64).cpu()


def rel_err(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    d = np.abs(ref).max()
    return np.abs(a - ref).max() / (d if d > 0 else 1.0)


class Layer(object):
    """one item: device tensors + the oracle's gradients"""

    def __init__(self, rng, B, N, C0, C1, up0, Cout, k, halo, flip=True, indep=False, bias=True, c0_valid=0, f32=False):
        from DLWP import _native as nat
        _bf = _f32 if f32 else globals()['_bf']
        self.cfg = (B, N, C0, C1, up0, Cout, k, halo, flip, indep, bias, c0_valid)
        n0 = N // 2 if up0 else N
        x0 = rng.standard_normal((B, 6, n0, n0, C0)) * 2.0
        if c0_valid:
            x0[..., c0_valid:] = 0.0
        self.x0 = _bf(x0)
        self.x1 = _bf(rng.standard_normal((B, 6, N, N, C1))) if C1 else None
        No = N if halo else N - k + 1
        self.dz = _bf(rng.standard_normal((B, 6, No, No, Cout)))
        cin = (c0_valid or C0) + C1
        names = ['eq', 'pol'] + (['np'] if indep else [])
        self.dw = {n: torch.zeros((k, k, cin, Cout), dtype=torch.float32, device=_dev()) for n in names}
        self.db = {n: torch.zeros((Cout,), dtype=torch.float32, device=_dev()) for n in names} if bias else {}
        self.d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=int(halo), up0=int(up0),
                              flip_north_pole=int(flip), act=0, alpha=0., vmax=0., dtype=nat.F32 if f32 else nat.BF16, flags=0,
                              c0_valid=c0_valid)
        self.table = nat.halo_tables(N, 1, _dev())[0] if halo else None

    def entry(self):
        g = (self.dw['eq'], self.dw['pol'], self.dw.get('np'), self.db.get('eq'), self.db.get('pol'), self.db.get('np'))
        return (self.d, self.x0, self.x1, self.dz, self.table, g)

    def reference(self):
        B, N, C0, C1, up0, Cout, k, halo, flip, indep, bias, c0_valid = self.cfg
        t = _f64(self.x0)
        if c0_valid:
            t = t[..., :c0_valid]
        if up0:
            t = orc.upsample_122(t)
        if C1:
            t = torch.cat([t, _f64(self.x1)], dim=-1)
        if halo:
            t = torch.as_tensor(orc.cs_pad(t.numpy(), 1, 'channels_last'))
        cin = t.shape[-1]
        w = {n: torch.zeros((k, k, cin, Cout), dtype=torch.float64, requires_grad=True) for n in ('eq', 'pol', 'np')}
        b = {n: torch.zeros((Cout,), dtype=torch.float64, requires_grad=True) for n in ('eq', 'pol', 'np')}
        z = orc.cs_conv2d(t, w['eq'], w['pol'], w['np'] if indep else None, b['eq'], b['pol'], b['np'] if indep else None,
                          data_format='channels_last', flip_north_pole=flip, independent_north_pole=indep)
        z.backward(_f64(self.dz))
        return {n: w[n].grad.numpy() for n in self.dw}, {n: b[n].grad.numpy() for n in self.db}

    def check(self, scale=1.0):
        rw, rb = self.reference()
        for n in self.dw:
            e = rel_err(_f64(self.dw[n]).numpy(), scale * rw[n])
            assert e <= TOL, 'dW_%s of %r: %.3g' % (n, self.cfg, e)
        for n in self.db:
            e = rel_err(_f64(self.db[n]).numpy(), scale * rb[n])
            assert e <= TOL, 'db_%s of %r: %.3g' % (n, self.cfg, e)


# (N, C0, C1, up0, Cout, k, halo): the eleven convolutions of `unet2` (Azure/train_cs.py:277-305) with 14 channels in / out
UNET2 = [(48, 14, 0, 0, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (24, 32, 0, 0, 64, 3, 1), (24, 64, 0, 0, 64, 3, 1),
         (12, 64, 0, 0, 128, 3, 1), (12, 128, 0, 0, 64, 3, 1), (24, 64, 64, 1, 64, 3, 1), (24, 64, 0, 0, 32, 3, 1),
         (48, 32, 32, 1, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (48, 32, 0, 0, 14, 1, 0)]

# (B, N, C0, C1, up0, Cout, k, halo): every instantiation of the segment body, ragged bands, both sources, upsampling
CASES = [
    (2, 24, 64, 0, 0, 64, 3, 1),      # 64 x 64 per worker
    (3, 12, 64, 0, 0, 128, 3, 1),     # ... two output groups, whole-face items
    (2, 12, 128, 0, 0, 64, 3, 1),     # ... two input groups
    (2, 24, 64, 64, 1, 64, 3, 1),     # ... decoder: upsampled source + skip source
    (2, 24, 64, 0, 0, 32, 3, 1),      # 64 x 32
    (1, 48, 32, 32, 1, 32, 3, 1),     # ... decoder at N = 48
    (2, 24, 32, 0, 0, 64, 3, 1),      # 32 x 64
    (2, 48, 32, 0, 0, 32, 3, 1),      # 32 x 32, 384-pixel items
    (3, 10, 16, 0, 0, 24, 3, 1),      # ... partial channel tiles, ragged band
    (2, 20, 40, 8, 0, 48, 3, 1),      # ... 48 input channels from two sources (two ci groups of 32)
    (2, 24, 40, 24, 0, 64, 3, 1),     # 64 x 64 per worker, the second 32-channel plane holds channels of BOTH sources
    (2, 24, 32, 32, 0, 32, 3, 1),     # 64 x 32, one plane per source, no upsampling
    (2, 48, 14, 0, 0, 32, 3, 1),      # 4-B X vectors: the 14-channel network input
    (1, 16, 26, 0, 0, 32, 3, 1),      # ... 26 channels
    (2, 48, 32, 0, 0, 14, 1, 0),      # 1x1 head, 14 outputs (4-B dZ vectors)
    (2, 16, 32, 0, 0, 32, 1, 0),      # 1x1, 32 outputs
    (2, 14, 32, 0, 0, 32, 3, 0),      # 3x3 'valid' on an already padded tensor (no halo)
]


@pytest.mark.parametrize('case', CASES)
def test_single_layer_matches_oracle(case):
    from DLWP import ops
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    lay = Layer(rng, *case)
    assert ops.wgrad_batch_supported(lay.d)
    ops.wgrad_batch([lay.entry()])
    lay.check()


def test_layer_options():
    """independent north pole, no north-pole flip, no bias, 7 variables in an 8-channel layout"""
    from DLWP import ops
    rng = np.random.default_rng(5)
    lays = [Layer(rng, 2, 12, 32, 0, 0, 32, 3, 1, flip=True, indep=True),
            Layer(rng, 2, 12, 32, 0, 0, 64, 3, 1, flip=False, indep=False),
            Layer(rng, 2, 12, 64, 0, 0, 64, 3, 1, flip=False, indep=True, bias=False),
            Layer(rng, 3, 16, 8, 0, 0, 32, 3, 1, c0_valid=7)]
    ops.wgrad_batch([l.entry() for l in lays])
    for l in lays:
        l.check()


@pytest.mark.parametrize('case', [(2, 48, 14, 0, 0, 32, 3, 1), (3, 16, 8, 0, 0, 32, 3, 1), (2, 24, 64, 0, 0, 64, 3, 1),
                                  (2, 12, 32, 0, 0, 64, 3, 1)])
def test_mask_on_load(case):
    """item with y: `dz` is the plain gradient, the producers form dy * act'(y) (rounded to bf16 like a stored dz)"""
    from DLWP import _native as nat
    from DLWP import ops
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    lay = Layer(rng, *case)
    y = _bf(rng.standard_normal(tuple(lay.dz.shape)) * 6.0)
    dy = lay.dz
    yf, gf = y.float().cpu().numpy(), dy.float().cpu().numpy()
    sl = np.where(yf < 0, np.float32(0.1), np.where((yf > 0) & (yf < 10.0), np.float32(1.0), np.float32(0.0)))
    lay.dz = torch.tensor(gf * sl).to(torch.bfloat16).to(_dev())        # what the reference sees
    d = nat.ConvDesc.from_buffer_copy(lay.d)
    d.act, d.alpha, d.vmax = nat.ACT_LEAKY_CLIP, 0.1, 10.0
    e = lay.entry()
    ops.wgrad_batch([(d, e[1], e[2], dy, e[4], e[5], y)])
    lay.check()


# exact-fp32 mode (round 4): 16-B vectors of 4 channels, one 32 x 32 tile pair per worker, v_mfma_f32_32x32x2_f32
F32_CASES = [
    (2, 48, 32, 0, 0, 32, 3, 1),      # 192-pixel items (4 rows of 48)
    (2, 24, 64, 0, 0, 64, 3, 1),      # two ci x two co groups
    (2, 24, 64, 64, 1, 64, 3, 1),     # decoder: upsampled source + skip source
    (3, 12, 64, 0, 0, 128, 3, 1),     # whole-face items
    (2, 48, 16, 0, 0, 32, 3, 1),      # the padded 14-channel input (16 channels)
    (3, 10, 20, 0, 0, 24, 3, 1),      # partial channel tiles, ragged bands
    (2, 20, 40, 8, 0, 48, 3, 1),      # 48 input channels from two sources
    (2, 16, 32, 0, 0, 32, 1, 0),      # 1x1
    (2, 14, 32, 0, 0, 32, 3, 0),      # 3x3 'valid', no halo
    (2, 48, 14, 0, 0, 32, 3, 1),      # 8-B loads: the 14-channel network input
    (2, 48, 32, 0, 0, 14, 1, 0),      # ... the 1x1 head with 14 outputs
    (2, 12, 8, 6, 0, 10, 3, 1),       # ... two sources, 10 outputs
]


@pytest.mark.parametrize('case', F32_CASES)
def test_f32_single_layer_matches_oracle(case):
    from DLWP import ops
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    lay = Layer(rng, *case, f32=True)
    assert ops.wgrad_batch_supported(lay.d)
    ops.wgrad_batch([lay.entry()])
    lay.check()


def test_f32_options_and_mask_on_load():
    """fp32 items: independent north pole / no flip / no bias / 14 of 16 channels valid, and an item with y (act' formed on
    load, exact in fp32 -- also through the 8-B loads); odd channel counts have no batched fp32 kernel"""
    from DLWP import _native as nat
    from DLWP import ops
    rng = np.random.default_rng(23)
    lays = [Layer(rng, 2, 12, 32, 0, 0, 32, 3, 1, flip=True, indep=True, f32=True),
            Layer(rng, 2, 12, 64, 0, 0, 64, 3, 1, flip=False, indep=True, bias=False, f32=True),
            Layer(rng, 3, 48, 16, 0, 0, 32, 3, 1, c0_valid=14, f32=True),
            Layer(rng, 2, 24, 32, 0, 0, 64, 3, 1, f32=True),
            Layer(rng, 2, 24, 14, 0, 0, 30, 3, 1, f32=True)]
    entries = [l.entry() for l in lays]
    for k in (3, 4):
        m = lays[k]
        y = _f32(rng.standard_normal(tuple(m.dz.shape)) * 6.0)
        dy = m.dz
        yf, gf = y.cpu().numpy(), dy.cpu().numpy()
        sl = np.where(yf < 0, np.float32(0.1), np.where((yf > 0) & (yf < 10.0), np.float32(1.0), np.float32(0.0)))
        m.dz = _f32(gf * sl)
        d = nat.ConvDesc.from_buffer_copy(m.d)
        d.act, d.alpha, d.vmax = nat.ACT_LEAKY_CLIP, 0.1, 10.0
        e = entries[k]
        entries[k] = (d, e[1], e[2], dy, e[4], e[5], y)
    ops.wgrad_batch(entries)
    for l in lays:
        l.check()
    odd = Layer(rng, 1, 12, 32, 0, 0, 7, 1, 0, f32=True)
    assert not ops.wgrad_batch_supported(odd.d)


def test_f32_layer_list_is_reproducible_and_matches_the_per_layer_kernel():
    from DLWP import _native as nat
    from DLWP import ops
    rng = np.random.default_rng(29)
    lays = [Layer(rng, 2, *cfg, f32=True) for cfg in UNET2]
    ops.wgrad_batch([l.entry() for l in lays])
    first = [[g.clone() for g in list(l.dw.values()) + list(l.db.values())] for l in lays]
    for l in lays:
        l.check()
    ops.wgrad_batch([l.entry() for l in lays])
    for l, f in zip(lays, first):
        for g, g1 in zip(list(l.dw.values()) + list(l.db.values()), f):
            assert torch.equal(g, g1 + g1)
    lay = lays[3]
    d = lay.d
    nbytes = nat.lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=_dev())
    ref = {n: torch.empty_like(v) for n, v in lay.dw.items()}
    rb = {n: torch.empty_like(v) for n, v in lay.db.items()}
    nat.check(nat.lib().dlwpcs_conv_bwd_weights(ctypes.byref(d), nat.ptr(lay.x0), 0, nat.ptr(lay.dz), 0, nat.ptr(ref['eq']),
                                                nat.ptr(ref['pol']), 0, nat.ptr(rb['eq']), nat.ptr(rb['pol']), 0,
                                                nat.ptr(lay.table), nat.ptr(ws), nbytes, nat.stream_ptr()), 'conv_bwd_weights')
    for n in ref:
        assert rel_err(_f64(lay.dw[n]).numpy(), 2.0 * _f64(ref[n]).numpy()) <= TOL
        assert rel_err(_f64(lay.db[n]).numpy(), 2.0 * _f64(rb[n]).numpy()) <= TOL


def test_unet2_layer_list_accumulates_and_is_reproducible():
    """the eleven convolutions of `unet2` as ONE batch; a second run adds to the gradients (x 2), bit for bit the same sum"""
    from DLWP import ops
    rng = np.random.default_rng(9)
    lays = [Layer(rng, 2, *cfg) for cfg in UNET2]
    ops.wgrad_batch([l.entry() for l in lays])
    first = [[g.clone() for g in list(l.dw.values()) + list(l.db.values())] for l in lays]
    for l in lays:
        l.check()
    ops.wgrad_batch([l.entry() for l in lays])
    for l, f in zip(lays, first):
        for g, g1 in zip(list(l.dw.values()) + list(l.db.values()), f):
            assert torch.equal(g, g1 + g1)
    for l in lays:
        l.check(scale=2.0)


def test_shared_destination_adds_both_applications():
    """a layer applied twice (integration_steps = 2): two items with the same gradient tensors"""
    from DLWP import ops
    rng = np.random.default_rng(13)
    a = Layer(rng, 2, 12, 32, 0, 0, 32, 3, 1)
    b = Layer(rng, 2, 12, 32, 0, 0, 32, 3, 1)
    ra, rba = a.reference()
    rb, rbb = b.reference()
    b.dw, b.db = a.dw, a.db
    ops.wgrad_batch([a.entry(), b.entry()])
    for n in a.dw:
        assert rel_err(_f64(a.dw[n]).numpy(), ra[n] + rb[n]) <= TOL
        assert rel_err(_f64(a.db[n]).numpy(), rba[n] + rbb[n]) <= TOL


def test_matches_the_per_layer_kernel():
    """same numbers as dlwpcs_conv_bwd_weights (per-layer launch) up to fp32 summation order"""
    from DLWP import _native as nat
    from DLWP import ops
    rng = np.random.default_rng(17)
    lay = Layer(rng, 4, 24, 64, 0, 0, 64, 3, 1)
    ops.wgrad_batch([lay.entry()])
    d = lay.d
    nbytes = nat.lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=_dev())
    ref = {n: torch.empty_like(v) for n, v in lay.dw.items()}
    rb = {n: torch.empty_like(v) for n, v in lay.db.items()}
    nat.check(nat.lib().dlwpcs_conv_bwd_weights(ctypes.byref(d), nat.ptr(lay.x0), 0, nat.ptr(lay.dz), 0, nat.ptr(ref['eq']),
                                                nat.ptr(ref['pol']), 0, nat.ptr(rb['eq']), nat.ptr(rb['pol']), 0,
                                                nat.ptr(lay.table), nat.ptr(ws), nbytes, nat.stream_ptr()), 'conv_bwd_weights')
    for n in ref:
        assert rel_err(_f64(lay.dw[n]).numpy(), _f64(ref[n]).numpy()) <= TOL
        assert rel_err(_f64(lay.db[n]).numpy(), _f64(rb[n]).numpy()) <= TOL


def _slice_of(lay, sl):
    """the same layer restricted to the samples `sl` (views of the inputs, gradient tensors of its own)"""
    B, N, C0, C1, up0, Cout, k, halo, flip, indep, bias, c0_valid = lay.cfg
    from DLWP import _native as nat
    part = Layer.__new__(Layer)
    nb = len(range(*sl.indices(B)))
    part.cfg = (nb,) + lay.cfg[1:]
    part.x0 = lay.x0[sl]
    part.x1 = None if lay.x1 is None else lay.x1[sl]
    part.dz = lay.dz[sl]
    part.dw = {n: torch.zeros_like(v) for n, v in lay.dw.items()}
    part.db = {n: torch.zeros_like(v) for n, v in lay.db.items()}
    part.d = nat.ConvDesc.from_buffer_copy(lay.d)
    part.d.B = nb
    part.table = lay.table
    return part


def test_full_size_batch_32():
    """BASELINE config 3 geometry: the unet2 layer list at batch 32 (plan with 256 chains), ALL eleven layers: the 32-sample
    launch equals the sum of four 8-sample launches (the weight gradient is linear in the samples; fp32 summation order is all
    that differs), and every layer of the first 8-sample launch is checked against the fp64 oracle."""
    from DLWP import ops
    rng = np.random.default_rng(21)
    lays = [Layer(rng, 32, *cfg) for cfg in UNET2]
    ops.wgrad_batch([l.entry() for l in lays])
    torch.cuda.synchronize()
    acc = [({n: torch.zeros_like(v, dtype=torch.float64) for n, v in l.dw.items()},
            {n: torch.zeros_like(v, dtype=torch.float64) for n, v in l.db.items()}) for l in lays]
    for s in range(0, 32, 8):
        parts = [_slice_of(l, slice(s, s + 8)) for l in lays]
        ops.wgrad_batch([q.entry() for q in parts])
        torch.cuda.synchronize()
        for q, (aw, ab) in zip(parts, acc):
            for n in aw:
                aw[n] += q.dw[n].double()
            for n in ab:
                ab[n] += q.db[n].double()
        if s == 0:
            for q in parts:
                q.check()                       # fp64 oracle, 8 samples, every layer
    for l, (aw, ab) in zip(lays, acc):
        for n in aw:
            e = rel_err(_f64(l.dw[n]).numpy(), aw[n].cpu().numpy())
            assert e <= TOL, 'dW_%s of %r: %.3g' % (n, l.cfg, e)
        for n in ab:
            e = rel_err(_f64(l.db[n]).numpy(), ab[n].cpu().numpy())
            assert e <= TOL, 'db_%s of %r: %.3g' % (n, l.cfg, e)
