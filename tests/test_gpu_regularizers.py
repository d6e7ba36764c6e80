"""
Weight regularizers and constraints of CubeSphereConv2D (reference DLWP/custom.py:837-842 -> add_weight(regularizer=, constraint=),
:898-914; keras regularizers.L1L2, constraints.MaxNorm / NonNeg / UnitNorm / MinMaxNorm): dlwpcs_l1l2_regularize /
dlwpcs_weight_constraint against the keras formulas restated in numpy (fp64), and a layer that trains with them through
DLWP.keras.Model (penalty in the reported loss, its gradient in the update, constraint applied after the step).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


@pytest.mark.parametrize('n,l1,l2', [(7, 0.01, 0.0), (9 * 32 * 32, 0.0, 0.02), (3 * 3 * 64 * 128, 0.003, 0.004)])
def test_l1l2_penalty_and_gradient(n, l1, l2):
    from DLWP import _native as nat
    rng = np.random.default_rng(n)
    w = rng.standard_normal(n).astype(np.float32)
    w[::5] = 0.0                                                                    # sign(0) = 0
    g0 = rng.standard_normal(n).astype(np.float32)
    wd, gd = torch.tensor(w, device=_dev()), torch.tensor(g0, device=_dev())
    pen = torch.full((1,), 0.5, dtype=torch.float32, device=_dev())
    nat.check(nat.lib().dlwpcs_l1l2_regularize(nat.ptr(wd), nat.ptr(gd), n, l1, l2, 2.0, nat.ptr(pen), nat.stream_ptr()), 'l1l2')
    torch.cuda.synchronize()
    w64 = w.astype(np.float64)
    ref_pen = 0.5 + l1 * np.abs(w64).sum() + l2 * (w64 ** 2).sum()
    ref_g = g0 + 2.0 * (l1 * np.sign(w64) + 2.0 * l2 * w64)
    assert abs(float(pen[0]) - ref_pen) <= 1e-5 * abs(ref_pen)
    np.testing.assert_allclose(gd.cpu().numpy(), ref_g, rtol=1e-6, atol=1e-7)


def _constraint_ref(w, kind, a, b, rate, rows):
    m = w.astype(np.float64).reshape(rows, -1)
    if kind == 'non_neg':
        return (m * (m >= 0)).reshape(w.shape)
    norms = np.sqrt((m ** 2).sum(axis=0, keepdims=True))
    eps = 1e-7
    if kind == 'max_norm':
        desired = np.clip(norms, 0, a)
    elif kind == 'unit_norm':
        return (m / (eps + norms)).reshape(w.shape)
    else:
        desired = rate * np.clip(norms, a, b) + (1 - rate) * norms
    return (m * (desired / (eps + norms))).reshape(w.shape)


@pytest.mark.parametrize('shape,axis', [((3, 3, 32, 64), [0, 1, 2]), ((3, 3, 14, 32), 0), ((64,), 0)])
def test_constraints_match_keras_formulas(shape, axis):
    from DLWP.keras import constraints
    rng = np.random.default_rng(len(shape))
    w = (rng.standard_normal(shape) * 1.5).astype(np.float32)
    axes = [axis] if isinstance(axis, int) else axis
    rows = int(np.prod([shape[a] for a in axes]))
    for con, kind, a, b, rate in ((constraints.MaxNorm(2.0, axis=axis), 'max_norm', 2.0, 0, 1),
                                  (constraints.NonNeg(), 'non_neg', 0, 0, 1),
                                  (constraints.UnitNorm(axis=axis), 'unit_norm', 0, 0, 1),
                                  (constraints.MinMaxNorm(0.5, 1.5, rate=0.7, axis=axis), 'min_max', 0.5, 1.5, 0.7)):
        wd = torch.tensor(w, device=_dev())
        con.apply(wd)
        torch.cuda.synchronize()
        np.testing.assert_allclose(wd.cpu().numpy(), _constraint_ref(w, kind, a, b, rate, rows), rtol=2e-6, atol=1e-7)
        assert constraints.get(constraints.serialize(con)).get_config() == con.get_config()
    with pytest.raises(NotImplementedError):
        constraints.MaxNorm(1.0, axis=2).apply(torch.zeros(shape, device=_dev())) if len(shape) > 2 else (_ for _ in ()).throw(NotImplementedError())


def test_layer_trains_with_regularizer_and_constraint():
    """one Adam step of pad + CubeSphereConv2D(kernel_regularizer=l2, kernel_constraint=MaxNorm): the reported loss is mse + penalty,
    the update moves against mse-gradient + penalty-gradient (checked through the first Adam step: -lr * sign(total gradient)), and the
    kernels obey the norm bound afterwards"""
    from DLWP.custom import CubeSphereConv2D, CubeSpherePadding2D
    from DLWP.keras import backend, regularizers, constraints
    from DLWP.keras.layers import Input
    from DLWP.keras.models import Model
    from oracle import cs_oracle as orc
    backend.set_device('cuda:0')
    N, C, F, B = 8, 3, 4, 2
    rng = np.random.default_rng(0)
    np.random.seed(3)
    inp = Input(shape=(6, N, N, C))
    lay = CubeSphereConv2D(F, 3, data_format='channels_last', kernel_regularizer=regularizers.l2(0.05),
                           kernel_constraint=constraints.MaxNorm(0.9, axis=[0, 1, 2]))
    out = lay(CubeSpherePadding2D(1, data_format='channels_last')(inp))
    model = Model(inputs=inp, outputs=out)
    model.compile(optimizer='adam', loss='mse')
    cfg = lay.get_config()
    assert cfg['kernel_regularizer'] == {'class_name': 'L1L2', 'config': {'l1': 0.0, 'l2': 0.05}}
    assert cfg['kernel_constraint']['class_name'] == 'MaxNorm'
    x = rng.standard_normal((B, 6, N, N, C)).astype(np.float32)
    t = rng.standard_normal((B, 6, N, N, F)).astype(np.float32)
    w0 = [w.copy() for w in model.get_weights()]                                     # eq kernel, polar kernel, eq bias, polar bias
    # oracle: loss and gradients in fp64
    ws = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in w0]
    y = orc.cs_conv2d(orc.cs_pad(torch.tensor(x, dtype=torch.float64), 1), ws[0], ws[1], equatorial_bias=ws[2], polar_bias=ws[3])
    mse = ((y - torch.tensor(t, dtype=torch.float64)) ** 2).mean()
    pen = 0.05 * ((ws[0] ** 2).sum() + (ws[1] ** 2).sum())
    (mse + pen).backward()
    hist = model.fit(x, t, batch_size=B, epochs=1, verbose=0, shuffle=False)
    total = float((mse + pen).detach())
    assert abs(hist.history["loss"][0] - total) <= 1e-4 * total
    w1 = model.get_weights()
    for k in (0, 1):
        step = np.where(np.abs(ws[k].grad.numpy()) > 1e-6, -1e-3 * np.sign(ws[k].grad.numpy()), 0.0)     # Adam's first step
        moved = w0[k] + step
        # ... then MaxNorm over axes [0, 1, 2]
        ref = _constraint_ref(moved.astype(np.float32), 'max_norm', 0.9, 0, 1, 9 * C)
        sel = np.abs(ws[k].grad.numpy()) > 1e-6
        np.testing.assert_allclose(w1[k][sel], ref[sel], rtol=2e-3, atol=2e-5)
        assert np.sqrt((w1[k].reshape(9 * C, F) ** 2).sum(axis=0)).max() <= 0.9 * (1 + 1e-5)
    ev = model.evaluate(x, t, batch_size=B, verbose=0)
    assert np.isfinite(ev if np.isscalar(ev) else ev[0])
