"""
GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI (ctypes, DLWP.ops), against
the CPU oracle (oracle/cs_oracle.py, fp64) and the golden vectors generated from the reference's own layer code.

Bars: integer/byte-exact for pure data movement (halo gather, pooling layout, concat); fp32 kernels within 1e-5 relative
(`max|d| / max|ref|`, the tolerance BASELINE.json's north_star states) of the fp64 oracle.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def rel_err(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    denom = np.abs(ref).max()
    return np.abs(a - ref).max() / (denom if denom > 0 else 1.0)


def to_dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=_dev())


def test_native_library_loaded():
    from DLWP import _native as nat
    assert nat.lib().dlwpcs_version() == 105
    # the loaded shared object is the in-tree one
    assert os.path.samefile(nat.LIB_PATH, os.path.join(os.path.dirname(nat.__file__), '..', 'lib', 'libdlwpcs.so'))


# ---------------------------------------------------------------------------------------------------------------------
# halo gather
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('N,p,C', [(4, 1, 1), (8, 1, 3), (8, 2, 4), (8, 3, 5), (12, 1, 8), (48, 1, 4), (96, 1, 2)])
def test_pad_forward_exact(N, p, C):
    from DLWP import ops
    rng = np.random.default_rng(N * 10 + p)
    x = rng.standard_normal((2, 6, N, N, C)).astype(np.float32)
    y = ops.cs_pad(to_dev(x), p).cpu().numpy()
    assert np.array_equal(y, orc.cs_pad(x, p, 'channels_last'))


def test_pad_golden(golden_dir):
    from DLWP.custom import CubeSpherePadding2D
    g = np.load(os.path.join(golden_dir, 'g2_padding.npz'))
    x = g['x']
    for p in (1, 2):
        y = CubeSpherePadding2D(p, data_format='channels_last')(to_dev(x)).cpu().numpy()
        assert np.array_equal(y, g['cl_p%d' % p])
        xcf = np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3))
        y = CubeSpherePadding2D(p, data_format='channels_first')(to_dev(xcf)).cpu().numpy()
        assert np.array_equal(y, g['cf_p%d' % p])


@pytest.mark.parametrize('N,p,C', [(8, 1, 3), (8, 2, 4), (12, 1, 8), (24, 3, 2)])
def test_pad_backward_is_adjoint(N, p, C):
    from DLWP import ops
    rng = np.random.default_rng(7)
    x = to_dev(rng.standard_normal((2, 6, N, N, C))).requires_grad_(True)
    gy = rng.standard_normal((2, 6, N + 2 * p, N + 2 * p, C)).astype(np.float32)
    y = ops.cs_pad(x, p)
    y.backward(to_dev(gy))
    # oracle: scatter-add of gy through the table (fp64)
    T = orc.halo_table(N, p).reshape(-1)
    ref = np.zeros((2, 6 * N * N, C))
    np.add.at(ref, (slice(None), T), gy.reshape(2, -1, C).astype(np.float64))
    assert rel_err(x.grad.cpu().numpy().reshape(2, -1, C), ref) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# fused convolution
# ---------------------------------------------------------------------------------------------------------------------

def _rand_conv_params(rng, k, cin, cout, indep=False):
    w = {n: rng.standard_normal((k, k, cin, cout)) * (1.0 / np.sqrt(k * k * cin)) for n in ('eq', 'pol', 'np')}
    b = {n: rng.standard_normal((cout,)) * 0.1 for n in ('eq', 'pol', 'np')}
    if not indep:
        w['np'] = b['np'] = None
    return w, b


def _oracle_conv(x0, x1, up0, halo, w, b, k, flip, indep, act):
    t = torch.tensor(x0, dtype=torch.float64)
    if up0:
        t = orc.upsample_122(t)
    if x1 is not None:
        t = torch.cat([t, torch.tensor(x1, dtype=torch.float64)], dim=-1)
    t.requires_grad_(False)
    if halo:
        t = orc.cs_pad(t, (k - 1) // 2, 'channels_last')
    tw = {n: (None if v is None else torch.tensor(v, dtype=torch.float64)) for n, v in w.items()}
    tb = {n: (None if v is None else torch.tensor(v, dtype=torch.float64)) for n, v in b.items()}
    y = orc.cs_conv2d(t, tw['eq'], tw['pol'], tw['np'], tb['eq'], tb['pol'], tb['np'], data_format='channels_last',
                      flip_north_pole=flip, independent_north_pole=indep)
    if act:
        y = orc.relu_leaky_clip(y, 0.1, 10.0)
    return y


CONV_CASES = [
    # B, N, C0, C1, Cout, k, halo, up0, flip, indep, act
    (1, 48, 4, 0, 4, 3, True, False, True, False, False),      # BASELINE cfg 1 shape
    (2, 8, 3, 0, 4, 3, True, False, True, False, True),        # scalar (C % 4 != 0) loader
    (2, 12, 8, 0, 40, 3, True, False, True, False, True),      # 2 N tiles, partial
    (2, 12, 64, 0, 128, 3, True, False, True, False, True),    # U-Net bottom: 4 N tiles, small face
    (2, 24, 32, 0, 64, 3, True, False, True, False, True),
    (2, 24, 16, 16, 32, 3, True, True, True, False, True),     # decoder: upsample + concat fused
    (1, 48, 32, 32, 32, 3, True, True, True, False, True),
    (2, 16, 12, 0, 20, 3, True, False, False, True, False),    # no flip, independent north pole
    (2, 16, 12, 0, 20, 3, True, False, True, True, True),      # flip + independent north pole
    (2, 10, 8, 0, 8, 3, False, False, True, False, False),     # plain 'valid' on an already padded tensor
    (2, 48, 32, 0, 14, 1, False, False, True, False, False),   # 1x1 head
    (1, 20, 5, 0, 7, 1, False, False, True, False, True),      # odd sizes everywhere
    (3, 9, 6, 0, 33, 3, True, False, True, False, True),       # odd face size, partial N tile
    (1, 96, 8, 0, 32, 3, True, False, True, False, True),      # C96
    (2, 12, 16, 0, 96, 3, True, False, True, False, True),     # 3 N tiles under a 4-N-tile workgroup
    (2, 24, 96, 0, 16, 3, True, False, True, False, True),     # ... and on the data-gradient side
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_and_backward(case):
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP, ACT_NONE
    B, N, C0, C1, Cout, k, halo, up0, flip, indep, act = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    n0 = N // 2 if up0 else N
    x0 = rng.standard_normal((B, 6, n0, n0, C0))
    x1 = rng.standard_normal((B, 6, N, N, C1)) if C1 else None
    w, b = _rand_conv_params(rng, k, C0 + C1, Cout, indep)
    # scale up so that the clip at 10 and the negative slope are both exercised
    x0 *= 3.0
    No = N if halo else N - k + 1
    gy = rng.standard_normal((B, 6, No, No, Cout))

    # oracle (fp64 autograd)
    t0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    t1 = torch.tensor(x1, dtype=torch.float64, requires_grad=True) if C1 else None
    tw = {n: (None if v is None else torch.tensor(v, dtype=torch.float64, requires_grad=True)) for n, v in w.items()}
    tb = {n: (None if v is None else torch.tensor(v, dtype=torch.float64, requires_grad=True)) for n, v in b.items()}
    t = orc.upsample_122(t0) if up0 else t0
    if C1:
        t = torch.cat([t, t1], dim=-1)
    if halo:
        t = orc.cs_pad(t, (k - 1) // 2, 'channels_last')
    yref = orc.cs_conv2d(t, tw['eq'], tw['pol'], tw['np'], tb['eq'], tb['pol'], tb['np'], data_format='channels_last',
                         flip_north_pole=flip, independent_north_pole=indep)
    if act:
        yref = orc.relu_leaky_clip(yref, 0.1, 10.0)
    yref.backward(torch.tensor(gy, dtype=torch.float64))

    # device
    d0 = to_dev(x0).requires_grad_(True)
    d1 = to_dev(x1).requires_grad_(True) if C1 else None
    dw = {n: (None if v is None else to_dev(v).requires_grad_(True)) for n, v in w.items()}
    db = {n: (None if v is None else to_dev(v).requires_grad_(True)) for n, v in b.items()}
    y = ops.cs_conv(d0, dw['eq'], dw['pol'], dw['np'], db['eq'], db['pol'], db['np'], src1=d1, ksize=k, halo=halo,
                    up0=up0, flip_north_pole=flip, act=ACT_LEAKY_CLIP if act else ACT_NONE, alpha=0.1, vmax=10.0)
    assert rel_err(y.detach().cpu().numpy(), yref.detach().numpy()) < RTOL
    y.backward(to_dev(gy))
    assert rel_err(d0.grad.cpu().numpy(), t0.grad.numpy()) < RTOL
    if C1:
        assert rel_err(d1.grad.cpu().numpy(), t1.grad.numpy()) < RTOL
    for n in ('eq', 'pol', 'np'):
        if dw[n] is not None:
            assert rel_err(dw[n].grad.cpu().numpy(), tw[n].grad.numpy()) < RTOL, 'dW ' + n
            assert rel_err(db[n].grad.cpu().numpy(), tb[n].grad.numpy()) < RTOL, 'db ' + n


@pytest.mark.parametrize('alpha,vmax', [(0.1, 10.0), (1.0, 4.0), (0.0, 2.5), (1.5, 6.0), (4.0, 3.0)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_activation_parameter_forms(alpha, vmax, dtype):
    """keras ReLU(negative_slope=alpha, max_value=vmax) in the fused epilogue: 0 <= alpha <= 1 takes the one-instruction
    med3(x, alpha*x, vmax) form, alpha > 1 the compare / select form; both must be the oracle's function, forward and
    through act'(y) in both gradient kernels.  A negative slope is rejected like in Keras."""
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP
    B, N, C0, Cout = 2, 12, 16, 32
    rng = np.random.default_rng(int(1000 * abs(alpha) + 10 * vmax))
    bf = dtype == torch.bfloat16
    rnd = (lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()) if bf else (lambda a: a)
    x0 = rnd(3.0 * rng.standard_normal((B, 6, N, N, C0)))
    gy = rnd(rng.standard_normal((B, 6, N, N, Cout)))
    w, b = _rand_conv_params(rng, 3, C0, Cout, False)
    wr = {n: (None if v is None else rnd(v)) for n, v in w.items()}       # the matrix cores see bf16-rounded kernels
    t0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    tw = {n: (None if v is None else torch.tensor(v, dtype=torch.float64, requires_grad=True)) for n, v in wr.items()}
    tb = {n: (None if v is None else torch.tensor(v, dtype=torch.float64, requires_grad=True)) for n, v in b.items()}
    yref = orc.cs_conv2d(orc.cs_pad(t0, 1, 'channels_last'), tw['eq'], tw['pol'], None, tb['eq'], tb['pol'], None,
                         data_format='channels_last', flip_north_pole=True, independent_north_pole=False)
    yref = orc.relu_leaky_clip(yref, alpha, vmax)
    yref.backward(torch.tensor(gy, dtype=torch.float64))
    d0 = to_dev(x0).to(dtype).requires_grad_(True)
    dw = {n: (None if v is None else to_dev(v).requires_grad_(True)) for n, v in w.items()}
    db = {n: (None if v is None else to_dev(v).requires_grad_(True)) for n, v in b.items()}
    y = ops.cs_conv(d0, dw['eq'], dw['pol'], None, db['eq'], db['pol'], None, ksize=3, halo=True, flip_north_pole=True,
                    act=ACT_LEAKY_CLIP, alpha=alpha, vmax=vmax)
    eps = 2.0 ** -8
    assert rel_err(y.detach().float().cpu().numpy(), yref.detach().numpy()) < (eps if bf else RTOL)
    # both regions of the activation must occur in the data, or the test says nothing
    yn = yref.detach().numpy()
    assert (yn >= vmax).any() and ((yn < 0).any() or alpha <= 0.0)
    with pytest.raises(ValueError):
        ops.cs_conv(d0.detach(), dw['eq'].detach(), dw['pol'].detach(), None, None, None, None, ksize=3, halo=True,
                    act=ACT_LEAKY_CLIP, alpha=-0.25, vmax=vmax)
    if bf:
        return          # act'(y) is evaluated on the bf16-rounded y: elements at a kink flip; fp32 checks the gradients
    y.backward(to_dev(gy).to(dtype))
    assert rel_err(d0.grad.float().cpu().numpy(), t0.grad.numpy()) < RTOL
    for n in ('eq', 'pol'):
        assert rel_err(dw[n].grad.cpu().numpy(), tw[n].grad.numpy()) < RTOL, 'dW ' + n
        assert rel_err(db[n].grad.cpu().numpy(), tb[n].grad.numpy()) < RTOL, 'db ' + n


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_nan_propagates_like_keras_relu(dtype):
    """Non-finite pre-activations through the fused convolution + keras ReLU(0.1, 10) (Azure/train_cs.py:199): a NaN stays a NaN
    (until round 3 the IEEE-minNum epilogue turned it into max_value), +inf clips to max_value, -inf stays -inf -- exactly where the
    oracle says so -- and a NaN in a network's input reaches the loss as NaN."""
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP
    B, N, C0, Cout = 2, 12, 16, 32
    rng = np.random.default_rng(5)
    bf = dtype == torch.bfloat16
    rnd = (lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()) if bf else (lambda a: a)
    x0 = rnd(rng.standard_normal((B, 6, N, N, C0)))
    x0[0, 0, 3, 4, 2] = np.nan          # interior cell
    x0[0, 4, 0, 0, 5] = np.nan          # pole-face corner: reaches other faces through the halo
    x0[1, 2, 6, 11, 1] = np.inf         # edge cell
    x0[1, 5, 9, 2, 7] = -np.inf
    w, b = _rand_conv_params(rng, 3, C0, Cout, False)
    wr = {n: (None if v is None else rnd(v)) for n, v in w.items()}
    t = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64)
    yref = orc.cs_conv2d(orc.cs_pad(t(x0), 1, 'channels_last'), t(wr['eq']), t(wr['pol']), None, t(b['eq']), t(b['pol']), None,
                         data_format='channels_last', flip_north_pole=True, independent_north_pole=False)
    yref = orc.relu_leaky_clip(yref, 0.1, 10.0).numpy()
    dw = {n: (None if v is None else to_dev(v)) for n, v in w.items()}
    db = {n: (None if v is None else to_dev(v)) for n, v in b.items()}
    y = ops.cs_conv(to_dev(x0).to(dtype), dw['eq'], dw['pol'], None, db['eq'], db['pol'], None, ksize=3, halo=True,
                    flip_north_pole=True, act=ACT_LEAKY_CLIP, alpha=0.1, vmax=10.0).float().cpu().numpy()
    assert np.isnan(yref).sum() > 50 and np.isinf(yref).any() and (yref == 10.0).any()
    assert np.array_equal(np.isnan(y), np.isnan(yref))
    assert np.array_equal(np.isposinf(y), np.isposinf(yref)) and np.array_equal(np.isneginf(y), np.isneginf(yref))
    fin = np.isfinite(yref)
    assert np.abs(y[fin] - yref[fin]).max() < (2.0 ** -8 if bf else RTOL) * np.abs(yref[fin]).max()

    # a NaN in the input of a network reaches the loss as NaN (both the inference and the training path)
    from DLWP.keras import mixed_precision
    mixed_precision.set_policy('mixed_bfloat16' if bf else 'float32')
    try:
        model, convs = _build_unet2(8, 8, 8, 8)
        model.compile(optimizer='adam', loss='mse')
        xs = rng.standard_normal((2, 6, 8, 8, 8)).astype(np.float32)
        xs[1, 3, 4, 4, 0] = np.nan
        ys = rng.standard_normal((2, 6, 8, 8, 8)).astype(np.float32)
        pred = model.predict(xs)
        assert np.isfinite(pred[0]).all() and np.isnan(pred[1]).any()
        hist = model.fit(xs, ys, batch_size=2, epochs=1, verbose=0, shuffle=False)
        assert np.isnan(hist.history['loss'][0])
    finally:
        mixed_precision.set_policy('float32')


def test_data_gradient_tile_interleave_is_bitwise_neutral(tmp_path):
    """Round 4 (TUNE_CONV_ILV, csrc/conv_ws.h): in the data-gradient kernel the M tiles of a tile are dealt to the consumer waves
    round-robin, short tiles skip the M tiles they do not have and the tile list is cut by cost.  Which wave computes a pixel
    changes, the arithmetic per pixel does not: input gradients bitwise equal with the bit on (default) and off, fp32 and bf16,
    face sizes whose padded grid ends in a short tile (N = 48: 2500 = 6 x 384 + 196) and others, one and two sources."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP import ops
from DLWP._native import ACT_LEAKY_CLIP
dev = torch.device('cuda', 0)
out = {}
for name, (B, N, C0, C1, Cout, dt) in {'a': (5, 48, 32, 0, 32, torch.bfloat16), 'b': (3, 48, 32, 0, 64, torch.float32),
                                        'c': (4, 24, 64, 0, 64, torch.bfloat16), 'd': (2, 48, 16, 16, 32, torch.bfloat16),
                                        'e': (7, 12, 64, 0, 128, torch.bfloat16), 'f': (2, 96, 32, 0, 32, torch.bfloat16)}.items():
    g = torch.Generator(device=dev).manual_seed(B * 1000 + N)
    x0 = torch.randn(B, 6, N, N, C0, device=dev, generator=g).to(dt).requires_grad_(True)
    x1 = torch.randn(B, 6, N, N, C1, device=dev, generator=g).to(dt).requires_grad_(True) if C1 else None
    w = [torch.randn(3, 3, C0 + C1, Cout, device=dev, generator=g) * 0.1 for _ in range(2)]
    b = [torch.randn(Cout, device=dev, generator=g) * 0.1 for _ in range(2)]
    y = ops.cs_conv(x0, w[0], w[1], None, b[0], b[1], None, src1=x1, ksize=3, halo=True, act=ACT_LEAKY_CLIP, alpha=0.1, vmax=10.0)
    gy = torch.randn(y.shape, device=dev, generator=g).to(dt)
    y.backward(gy)
    out[name + '0'] = x0.grad.float().cpu().numpy()
    if x1 is not None:
        out[name + '1'] = x1.grad.float().cpu().numpy()
np.savez(sys.argv[1], **out)
''' % root
    res = []
    for tune in ('13175', '12663'):          # default bits with / without TUNE_CONV_ILV (512)
        f = str(tmp_path / ('dx_%s.npz' % tune))
        r = subprocess.run([sys.executable, '-c', code, f], cwd=root, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DLWPCS_TUNE=tune))
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(f))
    assert sorted(res[0].files) == sorted(res[1].files) and len(res[0].files) == 7
    for k in res[0].files:
        assert np.isfinite(res[0][k]).all() and np.abs(res[0][k]).max() > 0
        assert np.array_equal(res[0][k], res[1][k]), k


def test_fp32_forward_tile_interleave_is_bitwise_neutral(tmp_path):
    """Round 6 (TUNE_CONV_ILV_FWD, csrc/conv_launch.h): the exact-fp32 FORWARD pass takes tiles of fewer M tiles, dealt to the consumer
    waves round-robin, with the tile list cut by cost, where that shortens the longest list (N = 24, 64 channels: 128-pixel tiles that
    are no whole rows).  Outputs AND input gradients bitwise equal with the bit on (default) and off; face sizes with and without a
    short last tile, one and two sources, an up-sampled source, pooled second output (which keeps the plain tiling)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP import ops
from DLWP._native import ACT_LEAKY_CLIP
dev = torch.device('cuda', 0)
out = {}
for name, (B, N, C0, C1, Cout, up0) in {'a': (32, 24, 32, 0, 64, False), 'b': (5, 24, 64, 0, 64, False), 'c': (3, 48, 32, 0, 32, False),
                                        'd': (4, 20, 64, 64, 64, True), 'e': (2, 36, 16, 16, 64, False), 'f': (7, 12, 64, 0, 128, False),
                                        'g': (3, 28, 64, 0, 96, False)}.items():
    g = torch.Generator(device=dev).manual_seed(B * 1000 + N)
    n0 = N // 2 if up0 else N
    x0 = torch.randn(B, 6, n0, n0, C0, device=dev, generator=g).requires_grad_(True)
    x1 = torch.randn(B, 6, N, N, C1, device=dev, generator=g).requires_grad_(True) if C1 else None
    w = [torch.randn(3, 3, C0 + C1, Cout, device=dev, generator=g) * 0.1 for _ in range(2)]
    b = [torch.randn(Cout, device=dev, generator=g) * 0.1 for _ in range(2)]
    y = ops.cs_conv(x0, w[0], w[1], None, b[0], b[1], None, src1=x1, ksize=3, halo=True, up0=up0, act=ACT_LEAKY_CLIP, alpha=0.1, vmax=10.0)
    gy = torch.randn(y.shape, device=dev, generator=g)
    y.backward(gy)
    out[name + 'y'] = y.detach().cpu().numpy()
    out[name + '0'] = x0.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
''' % root
    res = []
    for tune in ('16247', '14199'):          # default bits with / without TUNE_CONV_ILV_FWD (2048)
        f = str(tmp_path / ('fw_%s.npz' % tune))
        r = subprocess.run([sys.executable, '-c', code, f], cwd=root, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DLWPCS_TUNE=tune))
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(f))
    assert sorted(res[0].files) == sorted(res[1].files) and len(res[0].files) == 14
    for k in res[0].files:
        assert np.isfinite(res[0][k]).all() and np.abs(res[0][k]).max() > 0
        assert np.array_equal(res[0][k], res[1][k]), k


def test_conv_cfg1_golden(golden_dir):
    """BASELINE config 1 against the vector produced by the reference layers."""
    from DLWP import ops
    g = np.load(os.path.join(golden_dir, 'cfg1.npz'))
    y = ops.cs_conv(to_dev(g['x']), to_dev(g['w_eq']), to_dev(g['w_pol']), None, to_dev(g['b_eq']), to_dev(g['b_pol']),
                    None, ksize=3, halo=True)
    assert rel_err(y.cpu().numpy(), g['y']) < RTOL


def test_conv_wgrad_is_deterministic():
    from DLWP import ops
    rng = np.random.default_rng(11)
    x = to_dev(rng.standard_normal((4, 6, 24, 24, 32)))
    w, b = _rand_conv_params(rng, 3, 32, 64)
    gy = to_dev(rng.standard_normal((4, 6, 24, 24, 64)))
    grads = []
    for _ in range(3):
        dw = {n: to_dev(v).requires_grad_(True) for n, v in w.items() if v is not None}
        db = {n: to_dev(v).requires_grad_(True) for n, v in b.items() if v is not None}
        y = ops.cs_conv(x, dw['eq'], dw['pol'], None, db['eq'], db['pol'], None, ksize=3, halo=True)
        y.backward(gy)
        grads.append((dw['eq'].grad.clone(), dw['pol'].grad.clone(), db['eq'].grad.clone()))
    for g in grads[1:]:
        for a, bb in zip(g, grads[0]):
            assert torch.equal(a, bb)       # bitwise: fixed-order reductions, no atomics


def test_layers_golden_g3(golden_dir):
    """Every CubeSphereConv2D option combination of the golden set, through the layer class."""
    from DLWP.custom import CubeSphereConv2D
    g = np.load(os.path.join(golden_dir, 'g3_conv.npz'))
    x = g['x']
    for name in g['case_names']:
        name = str(name)
        parts = name.split('_')
        flip, indep = parts[1] == 'flip1', parts[2] == 'indep1'
        df = 'channels_last' if parts[3] == 'cl' else 'channels_first'
        use_bias, dil, stride, padding = parts[4] == 'bias1', int(parts[5][3:]), int(parts[6][1:]), parts[7]
        lay = CubeSphereConv2D(4, 3, strides=stride, padding=padding, data_format=df, dilation_rate=dil,
                               use_bias=use_bias, flip_north_pole=flip, independent_north_pole=indep)
        xin = x if df == 'channels_last' else np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3))
        lay.build(xin.shape)
        weights = [g['w_eq'], g['w_pol']] + ([g['w_np']] if indep else [])
        if use_bias:
            weights += [g['b_eq'], g['b_pol']] + ([g['b_np']] if indep else [])
        lay.set_weights(weights)
        y = lay(to_dev(xin)).detach().cpu().numpy()
        assert y.shape == tuple(lay.compute_output_shape(xin.shape))
        assert rel_err(y, g[name]) < RTOL, name


def test_gconv_backward():
    from DLWP import ops
    rng = np.random.default_rng(21)
    x = rng.standard_normal((2, 6, 11, 11, 3))
    w, b = _rand_conv_params(rng, 3, 3, 5, indep=True)
    for (stride, padding, dil, flip) in [(2, 'same', 1, True), (1, 'valid', 2, True), (2, 'valid', 1, False)]:
        t0 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        tw = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in w.items()}
        tb = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in b.items()}
        yref = orc.cs_conv2d(t0, tw['eq'], tw['pol'], tw['np'], tb['eq'], tb['pol'], tb['np'],
                             strides=(stride, stride), padding=padding, dilation=(dil, dil),
                             flip_north_pole=flip, independent_north_pole=True)
        gy = rng.standard_normal(tuple(yref.shape))
        yref.backward(torch.tensor(gy))
        d0 = to_dev(x).requires_grad_(True)
        dw = {n: to_dev(v).requires_grad_(True) for n, v in w.items()}
        db = {n: to_dev(v).requires_grad_(True) for n, v in b.items()}
        y = ops.cs_gconv(d0, dw['eq'], dw['pol'], dw['np'], db['eq'], db['pol'], db['np'], strides=(stride, stride),
                         padding=padding, dilation=(dil, dil), flip_north_pole=flip)
        assert rel_err(y.detach().cpu().numpy(), yref.detach().numpy()) < RTOL
        y.backward(to_dev(gy))
        assert rel_err(d0.grad.cpu().numpy(), t0.grad.numpy()) < RTOL
        for n in ('eq', 'pol', 'np'):
            assert rel_err(dw[n].grad.cpu().numpy(), tw[n].grad.numpy()) < RTOL
            assert rel_err(db[n].grad.cpu().numpy(), tb[n].grad.numpy()) < RTOL


# ---------------------------------------------------------------------------------------------------------------------
# stock ops
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('C', [3, 8])
def test_pool_upsample_act_concat(C):
    from DLWP import ops
    rng = np.random.default_rng(31 + C)
    x = rng.standard_normal((2, 6, 8, 8, C)).astype(np.float32) * 6
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    dx = to_dev(x).requires_grad_(True)
    # avgpool
    yr = orc.avgpool_122(tx)
    y = ops.avgpool2(dx)
    assert rel_err(y.detach().cpu().numpy(), yr.detach().numpy()) < 1e-6
    g = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    yr.backward(torch.tensor(g, dtype=torch.float64))
    y.backward(to_dev(g))
    assert rel_err(dx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
    # upsample
    tx.grad = None
    dx.grad = None
    yr = orc.upsample_122(tx)
    y = ops.upsample2(dx)
    assert np.array_equal(y.detach().cpu().numpy(), yr.detach().numpy().astype(np.float32))
    g = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    yr.backward(torch.tensor(g, dtype=torch.float64))
    y.backward(to_dev(g))
    assert rel_err(dx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
    # activation
    tx.grad = None
    dx.grad = None
    yr = orc.relu_leaky_clip(tx, 0.1, 10.0)
    y = ops.leaky_clip_relu(dx, 0.1, 10.0)
    assert rel_err(y.detach().cpu().numpy(), yr.detach().numpy()) < 1e-6
    assert (y.detach().cpu().numpy() == 10.0).any() and (y.detach().cpu().numpy() < 0).any()
    g = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    yr.backward(torch.tensor(g, dtype=torch.float64))
    y.backward(to_dev(g))
    assert rel_err(dx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
    # concat
    x2 = rng.standard_normal((2, 6, 8, 8, C + 4)).astype(np.float32)
    d2 = to_dev(x2).requires_grad_(True)
    dx.grad = None
    y = ops.concat_channels([dx, d2])
    assert np.array_equal(y.detach().cpu().numpy(), np.concatenate([x, x2], axis=-1))
    g = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(to_dev(g))
    assert np.array_equal(dx.grad.cpu().numpy(), g[..., :C])
    assert np.array_equal(d2.grad.cpu().numpy(), g[..., C:])


def test_layout_converters_roundtrip():
    from DLWP import ops
    rng = np.random.default_rng(41)
    x = rng.standard_normal((2, 5, 6, 7, 7)).astype(np.float32)
    y = ops.channels_first_to_last(to_dev(x))
    assert np.array_equal(y.cpu().numpy(), x.transpose(0, 2, 3, 4, 1))
    assert np.array_equal(ops.channels_last_to_first(y).cpu().numpy(), x)


def test_mse_and_adam():
    from DLWP import ops
    rng = np.random.default_rng(51)
    y = rng.standard_normal((3, 6, 8, 8, 5)).astype(np.float32)
    t = rng.standard_normal((3, 6, 8, 8, 5)).astype(np.float32)
    dy_ = to_dev(y).requires_grad_(True)
    out = ops.mse_mae(dy_, to_dev(t), 0.5)
    out.backward(torch.ones(2, device=out.device))
    d = y.astype(np.float64) - t
    assert abs(out[0].item() - 0.5 * (d ** 2).mean()) < 1e-6
    assert abs(out[1].item() - np.abs(d).mean()) < 1e-6
    assert rel_err(dy_.grad.cpu().numpy(), 0.5 * 2 * d / d.size) < 1e-6
    # Adam, 3 steps against the oracle
    n = 1000
    p0 = rng.standard_normal(n).astype(np.float32)
    p = to_dev(p0)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int32, device=p.device)
    pr = torch.tensor(p0, dtype=torch.float64)
    mr, vr = torch.zeros_like(pr), torch.zeros_like(pr)
    for it in range(3):
        g = rng.standard_normal(n).astype(np.float32)
        ops.adam_step(p, to_dev(g), m, v, step)
        orc.adam_step(pr, torch.tensor(g, dtype=torch.float64), mr, vr, it + 1)
    assert step.item() == 3
    assert rel_err(p.cpu().numpy(), pr.numpy()) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# end to end: tiny U-Net through DLWP.keras.Model
# ---------------------------------------------------------------------------------------------------------------------

def _build_unet2(N, c_in, c_out, base):
    from DLWP.custom import CubeSphereConv2D, CubeSpherePadding2D
    from DLWP.keras.layers import AveragePooling3D, Input, ReLU, UpSampling3D, concatenate
    from DLWP.keras.models import Model
    kw = dict(dilation_rate=1, padding='valid', activation='linear', data_format='channels_last',
              independent_north_pole=False, flip_north_pole=True)
    main_input = Input(shape=(6, N, N, c_in), name='main_input')
    pad = CubeSpherePadding2D(1, data_format='channels_last')
    pool = AveragePooling3D((1, 2, 2), data_format='channels_last')
    up = UpSampling3D((1, 2, 2), data_format='channels_last')
    relu = ReLU(negative_slope=0.1, max_value=10.)
    plan = orc.unet2_channel_plan(c_in, c_out, base)
    convs = [CubeSphereConv2D(co, k, **kw) for (_, co, k) in plan]
    x0 = relu(convs[0](pad(main_input)))
    x0 = relu(convs[1](pad(x0)))
    x1 = pool(x0)
    x1 = relu(convs[2](pad(x1)))
    x1 = relu(convs[3](pad(x1)))
    x2 = pool(x1)
    x2 = relu(convs[4](pad(x2)))
    x2 = relu(convs[5](pad(x2)))
    x2 = up(x2)
    x = concatenate([x2, x1], axis=-1)
    x = relu(convs[6](pad(x)))
    x = relu(convs[7](pad(x)))
    x = up(x)
    x = concatenate([x, x0], axis=-1)
    x = relu(convs[8](pad(x)))
    x = relu(convs[9](pad(x)))
    y = convs[10](x)
    return Model(inputs=main_input, outputs=y), convs


def _set_params(convs, params):
    for lay, prm in zip(convs, params):
        lay.set_weights([prm['equatorial_kernel'].numpy(), prm['polar_kernel'].numpy(),
                         prm['equatorial_bias'].numpy(), prm['polar_bias'].numpy()])


def test_unet2_tiny_forward_golden_and_train_step(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_unet2_tiny.npz'))
    model, convs = _build_unet2(8, 3, 3, 4)
    assert model.n_fused == 10
    params = orc.make_unet2_params(3, 3, base=4, seed=1)
    _set_params(convs, params)
    y = model.predict(g['x'].astype(np.float32))
    assert rel_err(y, g['y']) < RTOL

    # one Adam step on (x, target) against the oracle in fp64
    rng = np.random.default_rng(61)
    tgt = rng.standard_normal(g['y'].shape).astype(np.float32)
    model.compile(optimizer='adam', loss='mse', metrics=['mae'])
    _set_params(convs, params)
    pr = [{k: v.clone().requires_grad_(True) for k, v in prm.items()} for prm in params]
    yr = orc.unet2_forward(torch.tensor(g['x']), pr)
    loss = orc.mse_loss(yr, torch.tensor(tgt, dtype=torch.float64))
    loss.backward()
    hist = model.fit(g['x'].astype(np.float32), tgt, batch_size=2, epochs=1, verbose=0, shuffle=False)
    assert abs(hist.history['loss'][0] - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    # gradients left in the flat buffer by the step
    k = 0
    for lay, prm in zip(convs, pr):
        for w, name in zip(lay.weights, ('equatorial_kernel', 'polar_kernel', 'equatorial_bias', 'polar_bias')):
            assert rel_err(w.grad.cpu().numpy(), prm[name].grad.numpy()) < 5e-5, (lay.name, name)
            k += 1
    # parameters after the step
    for lay, prm in zip(convs, pr):
        for w, name in zip(lay.get_weights(), ('equatorial_kernel', 'polar_kernel', 'equatorial_bias', 'polar_bias')):
            p = prm[name].detach().clone()
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            orc.adam_step(p, prm[name].grad, m, v, 1)
            assert np.abs(w - p.numpy()).max() < 2e-5, (lay.name, name)


def test_graph_replay_matches_eager():
    """hipGraph-captured training steps give bitwise the same parameters as eager steps."""
    rng = np.random.default_rng(71)
    x = rng.standard_normal((4, 6, 8, 8, 3)).astype(np.float32)
    t = rng.standard_normal((4, 6, 8, 8, 3)).astype(np.float32)
    params = orc.make_unet2_params(3, 3, base=4, seed=2)
    results = []
    for use_graphs in (False, True):
        model, convs = _build_unet2(8, 3, 3, 4)
        model.use_graphs = use_graphs
        model.compile(optimizer='adam', loss='mse')
        _set_params(convs, params)
        dx, dt = [to_dev(x)], [to_dev(t)]
        for _ in range(4):
            model.train_on_device_batch(dx, dt)
        torch.cuda.synchronize()
        results.append(np.concatenate([w.ravel() for w in model.get_weights()]))
    assert np.array_equal(results[0], results[1])


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_deferred_batched_reduce_is_bitwise_identical(dtype):
    """DLWPCS_CONV_DEFER_REDUCE + dlwpcs_wgrad_reduce_batch (one reduction launch for all layers) sums in the same fixed
    order as the per-layer launches: parameters after 4 Adam steps are bitwise equal, eager and hipGraph-replayed."""
    from DLWP.keras import backend
    rng = np.random.default_rng(72)
    x = rng.standard_normal((4, 6, 16, 16, 6)).astype(np.float32)
    t = rng.standard_normal((4, 6, 16, 16, 6)).astype(np.float32)
    params = orc.make_unet2_params(6, 6, base=8, seed=3)
    tdt = torch.bfloat16 if dtype == 'bfloat16' else torch.float32
    results = []
    for defer, use_graphs in ((False, False), (True, False), (True, True)):
        backend.set_compute_dtype(dtype)
        try:
            model, convs = _build_unet2(16, 6, 6, 8)
        finally:
            backend.set_compute_dtype('float32')
        model.use_graphs = use_graphs
        model.defer_wgrad_reduce = defer
        model.compile(optimizer='adam', loss='mse')
        _set_params(convs, params)
        dx, dt = [to_dev(x).to(tdt)], [to_dev(t)]
        for _ in range(4):
            model.train_on_device_batch(dx, dt)
        torch.cuda.synchronize()
        results.append(np.concatenate([w.ravel() for w in model.get_weights()]))
    assert np.array_equal(results[0], results[1])
    assert np.array_equal(results[0], results[2])


def test_deferred_reduce_with_a_layer_applied_twice():
    """Shared layers (integration_steps = 2 in the reference scripts): both applications accumulate into the same
    gradient; their reduce items go into successive launches.  Gradient against the fp64 oracle and against the
    per-layer path (bitwise)."""
    from DLWP.custom import CubeSphereConv2D, CubeSpherePadding2D
    from DLWP.keras import Input, Model
    rng = np.random.default_rng(73)
    x = rng.standard_normal((3, 6, 12, 12, 8)).astype(np.float32)
    t = rng.standard_normal((3, 6, 12, 12, 8)).astype(np.float32)
    w, b = _rand_conv_params(rng, 3, 8, 8)
    grads = []
    for defer in (False, True):
        pad = CubeSpherePadding2D(1, data_format='channels_last')
        conv = CubeSphereConv2D(8, 3, data_format='channels_last')
        inp = Input(shape=(6, 12, 12, 8), name='main_input')
        model = Model(inputs=inp, outputs=conv(pad(conv(pad(inp)))))
        model.compile(optimizer='adam', loss='mse')
        conv.set_weights([w['eq'].astype(np.float32), w['pol'].astype(np.float32), b['eq'].astype(np.float32),
                          b['pol'].astype(np.float32)])
        model.use_graphs = False
        model.defer_wgrad_reduce = defer
        model.fit(x, t, batch_size=3, epochs=1, verbose=0, shuffle=False)
        grads.append([g.grad.cpu().numpy().copy() for g in conv.weights])
    for a, bb in zip(*grads):
        assert np.array_equal(a, bb)
    tw = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in w.items() if v is not None}
    tb = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in b.items() if v is not None}
    h = torch.tensor(x, dtype=torch.float64)
    for _ in range(2):
        h = orc.cs_conv2d(orc.cs_pad(h, 1, 'channels_last'), tw['eq'], tw['pol'], None, tb['eq'], tb['pol'], None,
                          data_format='channels_last', flip_north_pole=True, independent_north_pole=False)
    orc.mse_loss(h, torch.tensor(t, dtype=torch.float64)).backward()
    for got, ref in zip(grads[1], (tw['eq'], tw['pol'], tb['eq'], tb['pol'])):
        assert rel_err(got, ref.grad.numpy()) < RTOL


def test_adam_fused_single_launch_matches_two_launch_and_oracle():
    """dlwpcs_adam_step_fused: ticket-counter step increment and DLWPCS_ADAM_ZERO_GRAD; bitwise equal to dlwpcs_adam_step."""
    from DLWP import ops
    rng = np.random.default_rng(52)
    _adam_fused_case(rng, 300001)       # many workgroups, ragged tail: scalar path
    _adam_fused_case(rng, 300000)       # 16-B vectors
    _adam_fused_case(rng, 8)            # one workgroup


def _adam_fused_case(rng, n):
    from DLWP import ops
    k = min(n, 1000)
    p0 = rng.standard_normal(n).astype(np.float32)
    pa, pb = to_dev(p0), to_dev(p0)
    ma, va, mb, vb = (torch.zeros_like(pa) for _ in range(4))
    step1 = torch.zeros(1, dtype=torch.int32, device=pa.device)
    step2 = torch.zeros(2, dtype=torch.int32, device=pa.device)
    pr = torch.tensor(p0[:k], dtype=torch.float64)
    mr, vr = torch.zeros_like(pr), torch.zeros_like(pr)
    for it in range(4):
        g = rng.standard_normal(n).astype(np.float32)
        ga, gb = to_dev(g), to_dev(g)
        ops.adam_step(pa, ga, ma, va, step1, grad_scale=0.5)
        ops.adam_step(pb, gb, mb, vb, step2, grad_scale=0.5, zero_grads=(it % 2 == 0))
        orc.adam_step(pr, torch.tensor(0.5 * g[:k].astype(np.float64)), mr, vr, it + 1)
        assert torch.equal(gb, torch.zeros_like(gb) if it % 2 == 0 else ga)
        assert step2.tolist() == [it + 1, 0]
    assert step1.item() == 4
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert rel_err(pb[:k].cpu().numpy(), pr.numpy()) < 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_avgpool_skip_backward_is_one_pass(dtype):
    """ops.avgpool2_skip: (pooled, alias); gradient = d_alias + avgpool2_bwd(d_pooled) in one kernel, equal to the two
    separate ops (fp32: bitwise; bf16: the fused kernel rounds once, the separate path twice -> 1 bf16 ulp)."""
    from DLWP import ops
    gen = torch.Generator(device=_dev()).manual_seed(5)
    x = torch.randn((3, 6, 12, 12, 16), generator=gen, device=_dev()).to(dtype)
    g_pool = torch.randn((3, 6, 6, 6, 16), generator=gen, device=_dev()).to(dtype)
    g_skip = torch.randn((3, 6, 12, 12, 16), generator=gen, device=_dev()).to(dtype)
    xa = x.clone().requires_grad_(True)
    y, alias = ops.avgpool2_skip(xa)
    assert torch.equal(alias, x) and alias.data_ptr() == xa.data_ptr()
    torch.autograd.backward([y, alias], [g_pool, g_skip])
    xb = x.clone().requires_grad_(True)
    yb = ops.avgpool2(xb)
    assert torch.equal(y, yb)
    torch.autograd.backward([yb, xb * 1.0], [g_pool, g_skip])
    if dtype == torch.float32:
        assert torch.equal(xa.grad, xb.grad)
    else:
        ref = g_skip.float() + 0.25 * g_pool.float().repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        assert torch.equal(xa.grad, ref.to(torch.bfloat16))
    # pooled output only
    xc = x.clone().requires_grad_(True)
    yc, _ = ops.avgpool2_skip(xc)
    yc.backward(g_pool)
    xd = x.clone().requires_grad_(True)
    ops.avgpool2(xd).backward(g_pool)
    assert torch.equal(xc.grad, xd.grad)


def test_graph_and_eager_steps_interleave():
    """Graph replays rely on the optimizer launch clearing the gradient buffer; eager steps (other batch shapes) in
    between leave gradients behind.  A mixed sequence must give bitwise the parameters of the all-eager run."""
    rng = np.random.default_rng(74)
    xs = {b: rng.standard_normal((b, 6, 8, 8, 3)).astype(np.float32) for b in (4, 2)}
    ts = {b: rng.standard_normal((b, 6, 8, 8, 3)).astype(np.float32) for b in (4, 2)}
    params = orc.make_unet2_params(3, 3, base=4, seed=4)
    seq = [4, 4, 4, 2, 4, 2, 2, 4, 4]        # per shape: eager, capture + replay, replay ...
    results = []
    for use_graphs in (False, True):
        model, convs = _build_unet2(8, 3, 3, 4)
        model.use_graphs = use_graphs
        model.compile(optimizer='adam', loss='mse')
        _set_params(convs, params)
        dev = {b: ([to_dev(xs[b])], [to_dev(ts[b])]) for b in xs}
        losses = []
        for b in seq:
            losses.append(model.train_on_device_batch(*dev[b]).clone())
        torch.cuda.synchronize()
        results.append((np.concatenate([w.ravel() for w in model.get_weights()]), torch.stack(losses).cpu().numpy()))
        assert model.optimizer.iterations == len(seq)
    assert np.array_equal(results[0][0], results[1][0])
    assert np.array_equal(results[0][1], results[1][1])
