"""
Seeded random sweep of the fused cubed-sphere convolution (forward + all gradients) against the fp64 oracle, in both
dtypes, over shapes the hand-picked cases do not cover: batch 1..5, face sizes 4..26 (odd and even), channel counts that
are odd / even-only / multiples of 8 / of 64, with and without the fused upsample + concat, halo or plain 'valid', 1x1 and
3x3, flip / independent north pole, with and without the activation.  The kernels pick different code paths by shape
(vector widths, matrix-core vs fallback weight gradient, dz hand-over, direct data-gradient writes, ring fix-up, LDS patch
or quad stores), so this is mostly a dispatch-consistency test.  Tolerances as in test_gpu_parity.py / test_gpu_bf16.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu
EPS = 2.0 ** -8


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 3, 3]))
        halo = bool(k == 3 and rng.random() < 0.8)
        up0 = bool(halo and rng.random() < 0.35)
        # (faces >= 24 with channel counts in multiples of 8 take the round-5 default data gradient -- the gather form -- in bf16)
        N = int(rng.choice([4, 6, 8, 10, 12, 16, 20, 24, 26, 32] if up0 else [4, 5, 6, 7, 8, 9, 12, 13, 16, 20, 24, 25, 27, 32]))
        c0 = int(rng.choice([1, 2, 3, 6, 8, 14, 16, 24, 32, 64]))
        c1 = int(rng.choice([0, 0, 2, 8, 10, 32])) if halo else 0
        if c0 + c1 > 72:
            c1 = 0
        cout = int(rng.choice([1, 2, 7, 8, 14, 16, 24, 32, 40, 64]))
        B = int(rng.integers(1, 6))
        flip, indep, act = bool(rng.random() < 0.7), bool(rng.random() < 0.3), bool(rng.random() < 0.7)
        out.append((B, N, c0, c1, cout, k, halo, up0, flip, indep, act))
    return out


def _run(case, bf16):
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP, ACT_NONE
    B, N, C0, C1, Cout, k, halo, up0, flip, indep, act = case
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    rnd = (lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()) if bf16 else (lambda a: a)
    n0 = N // 2 if up0 else N
    x0 = rng.standard_normal((B, 6, n0, n0, C0)) * 3.0
    # every case also has inputs beyond +-100 (pre-activations far above max_value and alpha * x above it for the negative
    # side would be a different kernel bug each): the activation's clip region and its zero-gradient branch are always hit
    x0.reshape(-1)[::97] *= 60.0
    x0 = rnd(x0)
    x1 = rnd(rng.standard_normal((B, 6, N, N, C1))) if C1 else None
    w = {n: (rng.standard_normal((k, k, C0 + C1, Cout)) / np.sqrt(k * k * (C0 + C1))).astype(np.float32) for n in ('eq', 'pol', 'np')}
    b = {n: (rng.standard_normal((Cout,)) * 0.1).astype(np.float32) for n in ('eq', 'pol', 'np')}
    if not indep:
        w['np'] = b['np'] = None
    No = N if halo else N - k + 1
    if No < 1:
        pytest.skip('empty output')
    gy = rnd(rng.standard_normal((B, 6, No, No, Cout)))
    t0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    t1 = torch.tensor(x1, dtype=torch.float64, requires_grad=True) if C1 else None
    tw = {n: (None if v is None else torch.tensor(rnd(v), dtype=torch.float64, requires_grad=True)) for n, v in w.items()}
    tb = {n: (None if v is None else torch.tensor(v, dtype=torch.float64, requires_grad=True)) for n, v in b.items()}
    t = orc.upsample_122(t0) if up0 else t0
    if C1:
        t = torch.cat([t, t1], dim=-1)
    if halo:
        t = orc.cs_pad(t, 1, 'channels_last')
    zref = orc.cs_conv2d(t, tw['eq'], tw['pol'], tw['np'], tb['eq'], tb['pol'], tb['np'], data_format='channels_last',
                         flip_north_pole=flip, independent_north_pole=indep)
    yref = orc.relu_leaky_clip(zref, 0.1, 10.0) if act else zref
    adt = torch.bfloat16 if bf16 else torch.float32
    d0 = torch.tensor(x0, dtype=torch.float32).to(adt).to(dev).requires_grad_(True)
    d1 = torch.tensor(x1, dtype=torch.float32).to(adt).to(dev).requires_grad_(True) if C1 else None
    dw = {n: (None if v is None else torch.tensor(v, device=dev).requires_grad_(True)) for n, v in w.items()}
    db = {n: (None if v is None else torch.tensor(v, device=dev).requires_grad_(True)) for n, v in b.items()}
    y = ops.cs_conv(d0, dw['eq'], dw['pol'], dw['np'], db['eq'], db['pol'], db['np'], src1=d1, ksize=k, halo=halo, up0=up0,
                    flip_north_pole=flip, act=ACT_LEAKY_CLIP if act else ACT_NONE, alpha=0.1, vmax=10.0)

    def err(a, ref, floor=0.0):
        a, ref = a.detach().to(torch.float64).cpu().numpy(), ref.detach().numpy() if isinstance(ref, torch.Tensor) else ref
        den = max(np.abs(ref).max(), floor)
        return np.abs(a - ref).max() / (den if den > 0 else 1.0)
    # a bias gradient is a sum of B*6*No^2 terms of magnitude ~1 that cancel: with very few output channels max|ref| can be
    # far below the natural scale sqrt(#terms) of the fp32 summation error, so that scale is the floor of the denominator
    bias_floor = float(np.sqrt(B * 6 * No * No))
    assert err(y, yref) <= (EPS if bf16 else 1e-5)
    if bf16 and act:      # the device derives act' from ITS stored output and rounds dz; feed the oracle the same dz
        yd = y.detach().to(torch.float64).cpu().numpy()
        slope = np.where(yd < 0, 0.1, np.where((yd > 0) & (yd < 10.0), 1.0, 0.0))
        zref.backward(torch.tensor(rnd(gy * slope), dtype=torch.float64))
    else:
        yref.backward(torch.tensor(gy, dtype=torch.float64))
    y.backward(torch.tensor(gy, dtype=torch.float32).to(adt).to(dev))
    tol_x = ((5 if up0 else 3) * EPS) if bf16 else 1e-5
    # bf16: the oracle is fed exactly the bf16 x and dz the device multiplies, partial sums are fp32 -> what is left is the
    # fp32 summation order (2e-5 of max|ref|, like the fp32 mode's 1e-5 plus the rounding of the bf16-rounded kernels)
    tol_w = 2e-5 if bf16 else 1e-5
    assert err(d0.grad, t0.grad) <= tol_x
    if C1:
        assert err(d1.grad, t1.grad) <= tol_x
    for n in ('eq', 'pol', 'np'):
        if dw[n] is not None:
            assert err(dw[n].grad, tw[n].grad) <= tol_w, 'dW ' + n
            assert err(db[n].grad, tb[n].grad, bias_floor) <= tol_w, 'db ' + n


_N = int(os.environ.get('DLWPCS_FUZZ_N', '28'))                 # DLWPCS_FUZZ_N / DLWPCS_FUZZ_SEED widen or move the sweep
_S = int(os.environ.get('DLWPCS_FUZZ_SEED', '0'))


@pytest.mark.parametrize('case', _cases(_N, 101 + _S))
def test_conv_random_shapes_f32(case):
    _run(case, bf16=False)


@pytest.mark.parametrize('case', _cases(_N, 202 + _S))
def test_conv_random_shapes_bf16(case):
    _run(case, bf16=True)
