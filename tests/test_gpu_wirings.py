"""
N4 (`-m gpu`): the other U-Net wirings of the reference training script and the bf16 generic convolution.

* `basic`, `unet`, `unet2`, `unet3`, `unet4` through the engine (DLWP.model.cs_unet + the fusion plan) against outputs of the
  REFERENCE's own model functions executed over the reference's own layers (tests/golden/g9_wirings.npz, generator
  tests/golden/gen_golden_wirings.py; /root/reference/Azure/train_cs.py:233-388): fp32 <= 1e-5, bf16 <= 3e-2 of the range;
  plus one training step per wiring (finite loss, every weight receives a gradient, hipGraph replay == eager).
* dlwpcs_gconv_* in bf16 (strides / dilation / 'same'): forward and all gradients against the fp64 oracle evaluated on the
  bf16-rounded operands.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu
WIRINGS = ['basic', 'unet', 'unet2', 'unet3', 'unet4']


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def rel_err(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def _build(wiring, dtype, g):
    from DLWP.keras import Input, Model, backend
    from DLWP.model.cs_unet import CubeSphereNet
    backend.set_device('cuda:0')
    backend.set_compute_dtype(dtype)
    n = g[wiring + '/x'].shape[2]
    try:
        net = CubeSphereNet(3, 4, wiring)
        inp = Input(shape=(6, n, n, 3), name='main_input')
        model = Model(inputs=inp, outputs=net(inp))
    finally:
        backend.set_compute_dtype('float32')
    used = [str(n) for n in g[wiring + '/layers']]
    for name in used:
        lay = getattr(net, name)
        lay.set_weights([g['%s/%s/%s' % (wiring, name, k)] for k in
                         ('equatorial_kernel', 'polar_kernel', 'equatorial_bias', 'polar_bias')])
    return model, net, used


@pytest.mark.parametrize('wiring', WIRINGS)
@pytest.mark.parametrize('dtype,tol', [('float32', 1e-5), ('bfloat16', 3e-2)])
def test_wiring_forward_matches_reference_functions(golden_dir, wiring, dtype, tol):
    g = np.load(os.path.join(golden_dir, 'g9_wirings.npz'))
    model, net, used = _build(wiring, dtype, g)
    n_conv = sum(1 for l in model.layers if l.__class__.__name__ == 'CubeSphereConv2D')
    assert n_conv == len(used)                                 # the wiring touches exactly the reference's layers
    y = model.predict(g[wiring + '/x'])
    assert rel_err(y, g[wiring + '/y']) < tol


@pytest.mark.parametrize('wiring', ['basic', 'unet', 'unet3', 'unet4'])
def test_wiring_training_step(golden_dir, wiring):
    g = np.load(os.path.join(golden_dir, 'g9_wirings.npz'))
    x = g[wiring + '/x']
    t = np.random.default_rng(5).standard_normal(x.shape).astype(np.float32)
    finals = []
    for use_graphs in (False, True):
        model, net, used = _build(wiring, 'float32', g)
        model.use_graphs = use_graphs
        model.compile(optimizer='adam', loss='mse')
        w0 = np.concatenate([w.ravel() for w in model.get_weights()])
        hist = model.fit(x, t, batch_size=2, epochs=4, verbose=0, shuffle=False)
        assert np.isfinite(hist.history['loss']).all() and hist.history['loss'][-1] < hist.history['loss'][0]
        w1 = np.concatenate([w.ravel() for w in model.get_weights()])
        finals.append(w1)
        moved = np.abs(w1 - w0)
        k = 0
        for lay in model._weight_layers():
            for w in lay._weights:
                assert moved[k:k + w.numel()].max() > 0, (wiring, lay.name)       # every tensor received a gradient
                k += w.numel()
    assert np.array_equal(finals[0], finals[1])


@pytest.mark.parametrize('stride,padding,dil,flip', [(2, 'same', 1, True), (1, 'valid', 2, True), (2, 'valid', 1, False),
                                                     (1, 'same', 1, True)])
def test_gconv_bf16(stride, padding, dil, flip):
    from DLWP import ops
    rng = np.random.default_rng(31)
    bf = lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16)                  # noqa: E731
    x = bf(rng.standard_normal((2, 6, 11, 11, 4)))
    w = {n: (rng.standard_normal((3, 3, 4, 6)) * 0.3).astype(np.float32) for n in ('eq', 'pol', 'np')}
    b = {n: rng.standard_normal(6).astype(np.float32) for n in ('eq', 'pol', 'np')}
    t0 = x.double().requires_grad_(True)
    tw = {n: bf(v).double().requires_grad_(True) for n, v in w.items()}            # the kernels round the weights to bf16
    tb = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in b.items()}
    yref = orc.cs_conv2d(t0, tw['eq'], tw['pol'], tw['np'], tb['eq'], tb['pol'], tb['np'], strides=(stride, stride),
                         padding=padding, dilation=(dil, dil), flip_north_pole=flip, independent_north_pole=True)
    gy = bf(rng.standard_normal(tuple(yref.shape)))
    yref.backward(gy.double())
    d0 = x.to(_dev()).requires_grad_(True)
    dw = {n: torch.tensor(v, device=_dev()).requires_grad_(True) for n, v in w.items()}
    db = {n: torch.tensor(v, device=_dev()).requires_grad_(True) for n, v in b.items()}
    y = ops.cs_gconv(d0, dw['eq'], dw['pol'], dw['np'], db['eq'], db['pol'], db['np'], strides=(stride, stride),
                     padding=padding, dilation=(dil, dil), flip_north_pole=flip)
    assert y.dtype == torch.bfloat16
    assert rel_err(y.detach().float().cpu().numpy(), yref.detach().numpy()) <= 2.0 ** -8       # one bf16 rounding
    y.backward(gy.to(_dev()))
    assert d0.grad.dtype == torch.bfloat16
    assert rel_err(d0.grad.float().cpu().numpy(), t0.grad.numpy()) <= 2.0 ** -8
    for n in ('eq', 'pol', 'np'):
        assert dw[n].grad.dtype == torch.float32
        assert rel_err(dw[n].grad.cpu().numpy(), tw[n].grad.numpy()) < 1e-5           # fp32 accumulation of exact products
        assert rel_err(db[n].grad.cpu().numpy(), tb[n].grad.numpy()) < 1e-5


def test_layer_options_in_bf16_through_the_layer_class():
    """CubeSphereConv2D with strides / dilation / 'same' inside a bf16 model (was NotImplementedError in round 1)."""
    from DLWP.custom import CubeSphereConv2D
    from DLWP.keras import Input, Model, backend
    backend.set_device('cuda:0')
    backend.set_compute_dtype('bfloat16')
    try:
        inp = Input(shape=(6, 12, 12, 4), name='main_input')
        h = CubeSphereConv2D(8, 3, strides=2, padding='same', data_format='channels_last')(inp)
        out = CubeSphereConv2D(4, 3, dilation_rate=2, padding='same', data_format='channels_last')(h)
        model = Model(inputs=inp, outputs=out)
    finally:
        backend.set_compute_dtype('float32')
    model.compile(optimizer='adam', loss='mse')
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3, 6, 12, 12, 4)).astype(np.float32)
    t = rng.standard_normal((3, 6, 6, 6, 4)).astype(np.float32)
    hist = model.fit(x, t, batch_size=3, epochs=3, verbose=0)
    assert np.isfinite(hist.history['loss']).all() and hist.history['loss'][-1] < hist.history['loss'][0]


def test_channels_first_unet2_runs_channels_last_inside():
    """A uniformly channels_first graph (the layers' default data_format, reference DLWP/custom.py:1085-1196) is converted ONCE
    at the inputs and once at the outputs: the same fused launch list as its channels_last twin plus two transposes, the same
    numbers (bitwise: the arithmetic is the channels_last kernels'), training included."""
    import ctypes
    from DLWP import _native as nat, ops
    from DLWP.custom import CubeSphereConv2D, CubeSpherePadding2D
    from DLWP.keras.layers import AveragePooling3D, Input, ReLU, UpSampling3D, concatenate
    from DLWP.keras.models import Model

    def build(fmt, N, cin, cout, base):
        np.random.seed(5)
        cl = fmt == 'channels_last'
        kw = dict(dilation_rate=1, padding='valid', activation='linear', data_format=fmt)
        inp = Input(shape=(6, N, N, cin) if cl else (cin, 6, N, N), name='main_input')
        pad = CubeSpherePadding2D(1, data_format=fmt)
        pool = AveragePooling3D((1, 2, 2), data_format=fmt)
        up = UpSampling3D((1, 2, 2), data_format=fmt)
        relu = ReLU(negative_slope=0.1, max_value=10.)
        ax = -1 if cl else 1
        x0 = relu(CubeSphereConv2D(base, 3, **kw)(pad(inp)))
        x0 = relu(CubeSphereConv2D(base, 3, **kw)(pad(x0)))
        x1 = pool(x0)
        x1 = relu(CubeSphereConv2D(2 * base, 3, **kw)(pad(x1)))
        x = concatenate([up(x1), x0], axis=ax)
        x = relu(CubeSphereConv2D(base, 3, **kw)(pad(x)))
        y = CubeSphereConv2D(cout, 1, **kw)(x)
        m = Model(inputs=inp, outputs=y)
        m.compile(optimizer='adam', loss='mse', metrics=['mae'])
        return m

    def tags_of(fn):
        lib = nat.lib()
        lib.dlwpcs_prof_reset()
        lib.dlwpcs_prof_enable(1)
        try:
            fn()
            torch.cuda.synchronize()
        finally:
            lib.dlwpcs_prof_enable(0)
        out, tag = [], ctypes.create_string_buffer(160)
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        for i in range(lib.dlwpcs_prof_count()):
            nat.check(lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), 'prof_get')
            out.append(tag.value.decode())
        lib.dlwpcs_prof_reset()
        return out

    N, cin, cout, base = 8, 4, 4, 8
    rng = np.random.default_rng(12)
    x_cl = rng.standard_normal((4, 6, N, N, cin)).astype(np.float32)
    t_cl = rng.standard_normal((4, 6, N, N, cout)).astype(np.float32)
    x_cf = np.ascontiguousarray(np.transpose(x_cl, (0, 4, 1, 2, 3)))
    t_cf = np.ascontiguousarray(np.transpose(t_cl, (0, 4, 1, 2, 3)))
    m_cl, m_cf = build('channels_last', N, cin, cout, base), build('channels_first', N, cin, cout, base)
    assert m_cf._cf_model and not m_cl._cf_model and m_cf.n_fused == m_cl.n_fused == 4
    m_cf.set_weights(m_cl.get_weights())
    y_cl = m_cl.predict(x_cl)
    y_cf = m_cf.predict(x_cf)
    assert y_cf.shape == (4, cout, 6, N, N)
    assert np.array_equal(np.transpose(y_cf, (0, 2, 3, 4, 1)), y_cl)
    # the launch lists: the same convolution launches, plus one transpose in and one out
    d_cl = torch.tensor(x_cl, device='cuda')
    d_cf = torch.tensor(x_cf, device='cuda')
    l_cl = tags_of(lambda: m_cl.predict_on_device(d_cl))
    l_cf = tags_of(lambda: m_cf.predict_on_device(d_cf))
    assert [t for t in l_cf if t.startswith('conv')] == [t for t in l_cl if t.startswith('conv')]
    # training: same losses, same parameters
    h_cl = m_cl.fit(x_cl, t_cl, batch_size=4, epochs=3, verbose=0, shuffle=False)
    h_cf = m_cf.fit(x_cf, t_cf, batch_size=4, epochs=3, verbose=0, shuffle=False)
    assert np.array_equal(np.array(h_cl.history['loss']), np.array(h_cf.history['loss']))
    assert np.array_equal(np.array(h_cl.history['mean_absolute_error']), np.array(h_cf.history['mean_absolute_error']))
    for a, b in zip(m_cl.get_weights(), m_cf.get_weights()):
        assert np.array_equal(a, b)
    # engine option cf_model=0 semantics (per-layer transposes) stay available: same numbers
    os.environ['DLWPCS_OPTIONS'] = 'cf_model=0'
    try:
        m_old = build('channels_first', N, cin, cout, base)
    finally:
        os.environ.pop('DLWPCS_OPTIONS', None)
    assert not m_old._cf_model
    m_old.set_weights(m_cf.get_weights())
    assert np.array_equal(m_old.predict(x_cf), m_cf.predict(x_cf))
