"""
GPU parity tests of the bfloat16 mode (DLWPCS_BF16: bf16 activations, fp32 master weights, fp32 accumulation; BASELINE
configs 3-5 name bf16 compute).  The checker is the fp64 oracle evaluated on EXACTLY the numbers the kernels consume:
inputs, upstream gradients and kernels are rounded to bf16 first (what the device reads / what pack_weights feeds the
matrix cores), so the only differences left are fp32-vs-fp64 accumulation and the final round-to-nearest-even of each
stored bf16 value.

Tolerances (written out, per tensor kind; eps = 2^-8 = one bf16 ulp at 1.0):
  * pure data movement (halo gather, upsample, concat, layout): bit-exact;
  * elementwise arithmetic with a single rounding (pool, add, activation): bit-exact against the same fp32 expression;
  * bf16-stored results of a contraction (y):  max|d| <= eps * max|ref|  (rounding to nearest = eps/2 of the element,
    plus a possible 1-ulp flip where fp32 and fp64 accumulation straddle a rounding boundary); dsrc: 3 eps (5 eps under
    the fused 2x2 upsample adjoint), because the halo / upsample adjoint sums several bf16-stored terms;
  * fp32-stored results of a contraction over bf16 operands (dW, db): 2e-5 relative to max|ref| -- dz = dy * act'(y) is rounded
    to bf16 before it enters the matrix product, and the oracle is fed exactly that dz, so only the fp32 summation order differs.
"""
import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -8


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def bf16_round(a):
    """fp64/fp32 numpy -> the nearest bf16 value (RNE), returned as float64."""
    t = torch.tensor(np.asarray(a), dtype=torch.float32).to(torch.bfloat16)
    return t.to(torch.float64).numpy()


def to_bf(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32).to(torch.bfloat16).to(_dev())


def to_f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=_dev())


def host(t):
    return t.detach().to(torch.float64).cpu().numpy()


def rel_err(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    denom = np.abs(ref).max()
    return np.abs(a - ref).max() / (denom if denom > 0 else 1.0)


# ---------------------------------------------------------------------------------------------------------------------
# data movement and single-rounding elementwise kernels: exact
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('N,p,C', [(8, 1, 3), (8, 2, 4), (12, 1, 8), (48, 1, 14), (24, 1, 32)])
def test_pad_forward_exact_bf16(N, p, C):
    from DLWP import ops
    rng = np.random.default_rng(N + p + C)
    x = bf16_round(rng.standard_normal((2, 6, N, N, C)))
    y = ops.cs_pad(to_bf(x), p)
    assert y.dtype == torch.bfloat16
    assert np.array_equal(host(y), orc.cs_pad(x, p, 'channels_last'))


@pytest.mark.parametrize('N,p,C', [(8, 1, 3), (12, 1, 8), (24, 2, 2)])
def test_pad_backward_bf16(N, p, C):
    from DLWP import ops
    rng = np.random.default_rng(7)
    x = to_bf(rng.standard_normal((2, 6, N, N, C))).requires_grad_(True)
    gy = bf16_round(rng.standard_normal((2, 6, N + 2 * p, N + 2 * p, C)))
    ops.cs_pad(x, p).backward(to_bf(gy))
    T = orc.halo_table(N, p).reshape(-1)
    ref = np.zeros((2, 6 * N * N, C))
    np.add.at(ref, (slice(None), T), gy.reshape(2, -1, C))
    # <= 5 addends summed in fp32, one rounding
    assert rel_err(host(x.grad).reshape(2, -1, C), ref) <= EPS


@pytest.mark.parametrize('C', [3, 6, 8, 32])
def test_stock_ops_bf16(C):
    from DLWP import ops
    rng = np.random.default_rng(C)
    N = 8
    x = bf16_round(rng.standard_normal((2, 6, N, N, C)) * 4)
    dx = to_bf(x).requires_grad_(True)

    # AveragePooling3D((1,2,2)): ((a+b)+(c+d))*0.25 in fp32, one rounding
    y = ops.avgpool2(dx)
    xf = torch.tensor(x, dtype=torch.float32)
    ref = ((xf[:, :, 0::2, 0::2] + xf[:, :, 0::2, 1::2]) + (xf[:, :, 1::2, 0::2] + xf[:, :, 1::2, 1::2])) * 0.25
    assert torch.equal(y.cpu(), ref.to(torch.bfloat16))
    gy = bf16_round(rng.standard_normal(tuple(y.shape)))
    y.backward(to_bf(gy))
    gref = (torch.tensor(gy, dtype=torch.float32) * 0.25).to(torch.bfloat16)
    gref = gref.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert torch.equal(dx.grad.cpu(), gref)

    # UpSampling3D((1,2,2)) and its adjoint
    dx.grad = None
    u = ops.upsample2(dx)
    assert torch.equal(u.cpu(), to_bf(x).cpu().repeat_interleave(2, dim=2).repeat_interleave(2, dim=3))
    gu = bf16_round(rng.standard_normal(tuple(u.shape)))
    u.backward(to_bf(gu))
    gf = torch.tensor(gu, dtype=torch.float32)
    gref = ((gf[:, :, 0::2, 0::2] + gf[:, :, 0::2, 1::2]) + (gf[:, :, 1::2, 0::2] + gf[:, :, 1::2, 1::2]))
    assert torch.equal(dx.grad.cpu(), gref.to(torch.bfloat16))

    # ReLU(0.1, 10)
    dx.grad = None
    a = ops.leaky_clip_relu(dx, 0.1, 10.0)
    aref = torch.where(xf >= 0, torch.clamp(xf, max=10.0), xf * 0.1).to(torch.bfloat16)
    assert torch.equal(a.cpu(), aref)

    # concatenate / split
    x2 = bf16_round(rng.standard_normal((2, 6, N, N, C + 2)))
    d2 = to_bf(x2).requires_grad_(True)
    dx.grad = None
    c = ops.concat_channels([dx, d2])
    assert np.array_equal(host(c), np.concatenate([x, x2], axis=-1))
    gc = bf16_round(rng.standard_normal(tuple(c.shape)))
    c.backward(to_bf(gc))
    assert np.array_equal(host(dx.grad), gc[..., :C]) and np.array_equal(host(d2.grad), gc[..., C:])

    # layout converters
    cf = ops.channels_last_to_first(to_bf(x))
    assert np.array_equal(host(cf), x.transpose(0, 4, 1, 2, 3))
    assert np.array_equal(host(ops.channels_first_to_last(cf)), x)


def test_mse_bf16_prediction_fp32_target():
    from DLWP import ops
    rng = np.random.default_rng(5)
    y = bf16_round(rng.standard_normal((3, 6, 8, 8, 5)))
    t = rng.standard_normal(y.shape).astype(np.float32)
    dy = to_bf(y).requires_grad_(True)
    out = ops.mse_mae(dy, to_f32(t), 0.5)
    out.backward(torch.ones(2, device=_dev()))
    d = y - t.astype(np.float64)
    assert abs(out[0].item() - 0.5 * np.mean(d * d)) < 1e-5 * np.mean(d * d)
    assert abs(out[1].item() - np.mean(np.abs(d))) < 1e-5 * np.mean(np.abs(d))
    assert rel_err(host(dy.grad), 0.5 * 2.0 * d / d.size) <= EPS


# ---------------------------------------------------------------------------------------------------------------------
# fused convolution
# ---------------------------------------------------------------------------------------------------------------------

BF16_CONV_CASES = [
    # B, N, C0, C1, Cout, k, halo, up0, flip, indep, act
    (1, 48, 8, 0, 8, 3, True, False, True, False, False),
    (2, 8, 3, 0, 4, 3, True, False, True, False, True),        # odd channel count: 2-byte loads
    (2, 48, 14, 0, 32, 3, True, False, True, False, True),     # first U-Net layer (7 vars x 2 steps): 4-byte loads
    (2, 12, 8, 0, 40, 3, True, False, True, False, True),      # 2 N tiles, partial
    (2, 12, 64, 0, 128, 3, True, False, True, False, True),    # U-Net bottom: 4 N tiles, small face
    (2, 24, 32, 0, 64, 3, True, False, True, False, True),
    (2, 24, 64, 0, 64, 3, True, False, True, False, True),
    (2, 24, 16, 16, 32, 3, True, True, True, False, True),     # decoder: upsample + concat fused
    (1, 48, 32, 32, 32, 3, True, True, True, False, True),
    (2, 16, 16, 0, 24, 3, True, False, False, True, False),    # no flip, independent north pole
    (2, 16, 16, 0, 24, 3, True, False, True, True, True),      # flip + independent north pole
    (2, 10, 8, 0, 8, 3, False, False, True, False, False),     # plain 'valid' on an already padded tensor
    (2, 48, 32, 0, 14, 1, False, False, True, False, False),   # 1x1 head
    (1, 20, 5, 0, 7, 1, False, False, True, False, True),      # odd sizes everywhere
    (3, 9, 6, 0, 33, 3, True, False, True, False, True),       # odd face size, partial N tile
    (1, 96, 8, 0, 32, 3, True, False, True, False, True),      # C96
    (1, 24, 26, 0, 32, 3, True, False, True, False, True),     # 13 vars x 2 steps (cfg 5): even channels, 16 x 4-B vectors
    (2, 16, 6, 10, 16, 3, True, True, True, False, True),      # even channels from two sources (upsample + concat)
    (1, 96, 26, 0, 32, 3, True, False, True, False, True),     # C96 first layer of cfg 5
    (2, 12, 16, 0, 96, 3, True, False, True, False, True),     # 3 N tiles under a 4-N-tile workgroup
    (2, 24, 96, 0, 16, 3, True, False, True, False, True),     # ... and on the data-gradient side
    (2, 16, 32, 0, 16, 1, False, False, True, False, True),    # pointwise head kernels: activation (forward only)
    (1, 12, 32, 0, 8, 1, False, False, False, True, False),    # ... independent north pole, 8 channels, fwd + dgrad
    (3, 8, 32, 0, 10, 1, False, False, True, False, False),    # ... 10 channels (partial last k-group / store)
    (1, 16, 32, 0, 26, 1, False, False, True, False, False),   # ... 26 channels (config 5 head): two output tiles
    (2, 8, 32, 0, 32, 1, False, False, True, False, True),     # ... 32 channels + activation
    (1, 12, 32, 0, 20, 1, False, False, False, True, False),   # ... 20 channels, independent north pole
    (2, 12, 16, 0, 14, 3, True, False, True, False, True),     # 4-B vector loader on the data-gradient side (dy has 14 channels)
    (1, 16, 8, 0, 26, 3, True, False, True, False, True),      # ... 26 channels
    (2, 10, 10, 0, 16, 3, True, False, True, False, True),     # 4-B vector loader, 10 input channels
    (1, 12, 22, 0, 8, 1, False, False, True, False, True),     # ... 1x1, 22 input channels
]


def _rand_conv_params(rng, k, cin, cout, indep=False):
    w = {n: rng.standard_normal((k, k, cin, cout)) * (1.0 / np.sqrt(k * k * cin)) for n in ('eq', 'pol', 'np')}
    b = {n: rng.standard_normal((cout,)) * 0.1 for n in ('eq', 'pol', 'np')}
    if not indep:
        w['np'] = b['np'] = None
    return w, b


@pytest.mark.parametrize('case', BF16_CONV_CASES)
def test_conv_forward_and_backward_bf16(case):
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP, ACT_NONE
    B, N, C0, C1, Cout, k, halo, up0, flip, indep, act = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    n0 = N // 2 if up0 else N
    x0 = bf16_round(rng.standard_normal((B, 6, n0, n0, C0)) * 3.0)
    x1 = bf16_round(rng.standard_normal((B, 6, N, N, C1))) if C1 else None
    w, b = _rand_conv_params(rng, k, C0 + C1, Cout, indep)
    w = {n: (None if v is None else v.astype(np.float32)) for n, v in w.items()}        # fp32 master weights
    b = {n: (None if v is None else v.astype(np.float32)) for n, v in b.items()}
    No = N if halo else N - k + 1
    gy = bf16_round(rng.standard_normal((B, 6, No, No, Cout)))

    # ---- oracle (fp64 autograd) on the numbers the device consumes: bf16-rounded kernels, fp32 biases
    t0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    t1 = torch.tensor(x1, dtype=torch.float64, requires_grad=True) if C1 else None
    tw = {n: (None if v is None else torch.tensor(bf16_round(v), dtype=torch.float64, requires_grad=True))
          for n, v in w.items()}
    tb = {n: (None if v is None else torch.tensor(v, dtype=torch.float64, requires_grad=True)) for n, v in b.items()}
    t = orc.upsample_122(t0) if up0 else t0
    if C1:
        t = torch.cat([t, t1], dim=-1)
    if halo:
        t = orc.cs_pad(t, (k - 1) // 2, 'channels_last')
    zref = orc.cs_conv2d(t, tw['eq'], tw['pol'], tw['np'], tb['eq'], tb['pol'], tb['np'], data_format='channels_last',
                         flip_north_pole=flip, independent_north_pole=indep)
    yref = orc.relu_leaky_clip(zref, 0.1, 10.0) if act else zref

    # ---- device forward
    d0 = to_bf(x0).requires_grad_(True)
    d1 = to_bf(x1).requires_grad_(True) if C1 else None
    dw = {n: (None if v is None else to_f32(v).requires_grad_(True)) for n, v in w.items()}
    db = {n: (None if v is None else to_f32(v).requires_grad_(True)) for n, v in b.items()}
    y = ops.cs_conv(d0, dw['eq'], dw['pol'], dw['np'], db['eq'], db['pol'], db['np'], src1=d1, ksize=k, halo=halo,
                    up0=up0, flip_north_pole=flip, act=ACT_LEAKY_CLIP if act else ACT_NONE, alpha=0.1, vmax=10.0)
    assert y.dtype == torch.bfloat16
    assert rel_err(host(y), yref.detach().numpy()) <= EPS

    # ---- backward: the device derives act'(.) from ITS stored (bf16) output, and rounds dz = dy*act' to bf16; feed
    # the oracle exactly that dz so that the comparison isolates the two contractions
    yd = host(y)
    if act:
        slope = np.where(yd < 0, 0.1, np.where((yd > 0) & (yd < 10.0), 1.0, 0.0))
        dz = bf16_round(gy * slope)
    else:
        dz = gy
    zref.backward(torch.tensor(dz, dtype=torch.float64))
    y.backward(to_bf(gy))
    # the padded-input gradient is stored in bf16 (one rounding per term), then <= 5 halo terms (x4 under the 2x2
    # upsample adjoint) are summed in fp32 and rounded once more
    assert rel_err(host(d0.grad), t0.grad.numpy()) <= (5 if up0 else 3) * EPS
    if C1:
        assert rel_err(host(d1.grad), t1.grad.numpy()) <= 3 * EPS
    for n in ('eq', 'pol', 'np'):
        if dw[n] is not None:
            assert dw[n].grad.dtype == torch.float32
            # the oracle multiplies exactly the bf16 x and dz the device does; partial sums are fp32: only the fp32
            # summation order is left (measured <= 6e-6 on these cases; the bias sum of few channels gets its natural floor)
            assert rel_err(host(dw[n].grad), tw[n].grad.numpy()) < 2e-5, 'dW ' + n
            bref = tb[n].grad.numpy()
            den = max(np.abs(bref).max(), np.sqrt(B * 6 * No * No))
            assert np.abs(host(db[n].grad) - bref).max() / den < 2e-5, 'db ' + n


@pytest.mark.parametrize('C0,Cout', [(14, 32), (32, 32), (5, 8)])
def test_conv_weight_gradient_only_bf16(C0, Cout):
    """First-layer situation: the input needs no gradient, so the weight-gradient kernel applies act' itself (no dz
    hand-over to a data-gradient kernel) -- must agree with the path that hands dz over."""
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP
    rng = np.random.default_rng(C0)
    x = to_bf(rng.standard_normal((3, 6, 16, 16, C0)) * 3)
    w, b = _rand_conv_params(rng, 3, C0, Cout)
    gy = to_bf(rng.standard_normal((3, 6, 16, 16, Cout)))
    grads = []
    for need_x in (False, True):
        xx = x.clone().requires_grad_(need_x)
        dw = {n: to_f32(v).requires_grad_(True) for n, v in w.items() if v is not None}
        db = {n: to_f32(v).requires_grad_(True) for n, v in b.items() if v is not None}
        y = ops.cs_conv(xx, dw['eq'], dw['pol'], None, db['eq'], db['pol'], None, ksize=3, halo=True, act=ACT_LEAKY_CLIP,
                        alpha=0.1, vmax=10.0)
        y.backward(gy)
        grads.append([dw['eq'].grad, dw['pol'].grad, db['eq'].grad, db['pol'].grad])
    for a, bb in zip(*grads):
        assert torch.equal(a, bb)


def test_conv_wgrad_is_deterministic_bf16():
    from DLWP import ops
    rng = np.random.default_rng(11)
    x = to_bf(rng.standard_normal((4, 6, 24, 24, 32)))
    w, b = _rand_conv_params(rng, 3, 32, 64)
    gy = to_bf(rng.standard_normal((4, 6, 24, 24, 64)))
    grads = []
    for _ in range(3):
        dw = {n: to_f32(v).requires_grad_(True) for n, v in w.items() if v is not None}
        db = {n: to_f32(v).requires_grad_(True) for n, v in b.items() if v is not None}
        y = ops.cs_conv(x, dw['eq'], dw['pol'], None, db['eq'], db['pol'], None, ksize=3, halo=True)
        y.backward(gy)
        grads.append((dw['eq'].grad.clone(), dw['pol'].grad.clone(), db['eq'].grad.clone()))
    for g in grads[1:]:
        for a, bb in zip(g, grads[0]):
            assert torch.equal(a, bb)


# ---------------------------------------------------------------------------------------------------------------------
# whole model under the mixed-precision policy
# ---------------------------------------------------------------------------------------------------------------------

def _build_unet2(N, cin, cout, base):
    from DLWP.model.cs_unet import CubeSphereNet
    from DLWP.keras import Input, Model
    net = CubeSphereNet(base_filter_number=base, output_channels=cout)
    inp = Input(shape=(6, N, N, cin), name='main_input')
    model = Model(inputs=inp, outputs=net.unet2(inp))
    convs = [l for l in model.layers if l.__class__.__name__ == 'CubeSphereConv2D']
    return model, convs


def _set_params(convs, params):
    for lay, prm in zip(convs, params):
        lay.set_weights([prm['equatorial_kernel'].numpy().astype(np.float32), prm['polar_kernel'].numpy().astype(np.float32),
                         prm['equatorial_bias'].numpy().astype(np.float32), prm['polar_bias'].numpy().astype(np.float32)])


def test_unet2_train_step_bf16_tracks_fp64_oracle():
    """Mixed-precision U-Net step: loss within 2 %, flat gradient cosine > 0.999 against the fp64 oracle."""
    from DLWP.keras import mixed_precision
    rng = np.random.default_rng(61)
    N, C, base = 16, 6, 8
    x = rng.standard_normal((4, 6, N, N, C)).astype(np.float32)
    tgt = rng.standard_normal((4, 6, N, N, C)).astype(np.float32)
    params = orc.make_unet2_params(C, C, base=base, seed=1)
    mixed_precision.set_policy('mixed_bfloat16')
    try:
        model, convs = _build_unet2(N, C, C, base)
    finally:
        mixed_precision.set_policy('float32')
    assert model.compute_dtype == 'bfloat16' and model.n_fused == 10
    model.compile(optimizer='adam', loss='mse', metrics=['mae'])
    _set_params(convs, params)
    y = model.predict(x)
    assert y.dtype == np.float32
    pr = [{k: v.clone().requires_grad_(True) for k, v in prm.items()} for prm in params]
    yr = orc.unet2_forward(torch.tensor(x, dtype=torch.float64), pr)
    assert rel_err(y, yr.detach().numpy()) < 3e-2
    loss = orc.mse_loss(yr, torch.tensor(tgt, dtype=torch.float64))
    loss.backward()
    model.use_graphs = False
    hist = model.fit(x, tgt, batch_size=4, epochs=1, verbose=0, shuffle=False)
    assert abs(hist.history['loss'][0] - loss.item()) < 2e-2 * abs(loss.item())
    g_dev = np.concatenate([w.grad.to(torch.float64).cpu().numpy().ravel() for lay in convs for w in lay.weights])
    g_ref = np.concatenate([prm[n].grad.numpy().ravel() for prm in pr
                            for n in ('equatorial_kernel', 'polar_kernel', 'equatorial_bias', 'polar_bias')])
    cos = float(np.dot(g_dev, g_ref) / (np.linalg.norm(g_dev) * np.linalg.norm(g_ref)))
    assert cos > 0.999, cos
    assert all(w.dtype == torch.float32 for lay in convs for w in lay.weights)        # fp32 master weights


def test_graph_replay_matches_eager_bf16():
    from DLWP.keras import backend
    rng = np.random.default_rng(71)
    x = rng.standard_normal((4, 6, 8, 8, 4)).astype(np.float32)
    t = rng.standard_normal((4, 6, 8, 8, 4)).astype(np.float32)
    params = orc.make_unet2_params(4, 4, base=8, seed=2)
    results = []
    for use_graphs in (False, True):
        backend.set_compute_dtype('bfloat16')
        try:
            model, convs = _build_unet2(8, 4, 4, 8)
        finally:
            backend.set_compute_dtype('float32')
        model.use_graphs = use_graphs
        model.compile(optimizer='adam', loss='mse')
        _set_params(convs, params)
        dx, dt = [to_bf(x)], [to_f32(t)]
        for _ in range(4):
            model.train_on_device_batch(dx, dt)
        torch.cuda.synchronize()
        results.append(np.concatenate([w.ravel() for w in model.get_weights()]))
    assert np.array_equal(results[0], results[1])


def test_predict_timeseries_bf16_equals_stepwise_predict():
    """device-resident rollout (state stays in HBM, bf16) == feeding predict()'s output back by hand"""
    from DLWP.keras import Input, Model, backend
    from DLWP.model import DLWPFunctional
    from DLWP.model.cs_unet import CubeSphereNet
    backend.set_compute_dtype('bfloat16')
    try:
        inp = Input(shape=(6, 8, 8, 4), name='main_input')
        model = Model(inputs=inp, outputs=CubeSphereNet(base_filter_number=8, output_channels=4).unet2(inp))
    finally:
        backend.set_compute_dtype('float32')
    dlwp = DLWPFunctional(is_convolutional=True, time_dim=1)
    dlwp.build_model(model, loss='mse', optimizer='adam')
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 6, 8, 8, 4)).astype(np.float32)
    series = dlwp.predict_timeseries(x, 4)
    assert series.shape == (4, 3, 6, 8, 8, 4) and series.dtype == np.float32 and np.isfinite(series).all()
    state = x
    for t in range(4):
        state = dlwp.predict(state)
        assert np.array_equal(series[t], state), t


def test_bf16_training_tracks_fp32_training():
    """60 Adam steps on a learnable synthetic map, same initial weights and batches: the mixed-precision loss curve must
    follow the fp32 one (a systematic gradient error -- wrong hand-over, wrong mask -- shows up as divergence)."""
    from DLWP.keras import backend
    rng = np.random.default_rng(123)
    N, C, base, B = 16, 8, 8, 8
    x = rng.standard_normal((B, 6, N, N, C)).astype(np.float32)
    t = (0.5 * np.roll(x, 1, axis=3) - 0.25 * x + 0.1 * rng.standard_normal(x.shape)).astype(np.float32)
    params = orc.make_unet2_params(C, C, base=base, seed=4)
    curves = {}
    for dtype in ('float32', 'bfloat16'):
        backend.set_compute_dtype(dtype)
        try:
            model, convs = _build_unet2(N, C, C, base)
        finally:
            backend.set_compute_dtype('float32')
        from DLWP.keras.optimizers import Adam
        model.compile(optimizer=Adam(learning_rate=4e-3), loss='mse')
        _set_params(convs, params)
        dx = [to_f32(x).to(backend.torch_dtype(dtype))]
        dt = [to_f32(t)]
        losses = []
        for _ in range(60):
            losses.append(float(model.train_on_device_batch(dx, dt)[0, 0].item()))
        curves[dtype] = np.array(losses)
    f, h = curves['float32'], curves['bfloat16']
    assert f[-1] < 0.8 * f[0], (f[0], f[-1])                    # it learns
    assert np.all(np.abs(h - f) <= 0.03 * f + 1e-3), np.abs(h - f).max()
