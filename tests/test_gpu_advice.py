"""
GPU tests (`-m gpu`) for behaviour that must not depend on the execution mode:
  * optimizer hyper-parameters changed after hipGraph capture take effect on the next replay (they live in device memory);
  * ops.mse_mae applies a non-unit upstream gradient;
  * the HBM-resident batch feed rejects out-of-range sample indices like the host path;
  * an optimizer returned by enable_mixed_precision_graph_rewrite() switches a model built BEFORE the call (the order of the
    reference script: Model at Azure/train_cs.py:411, rewrite at :429, compile at :430).
"""
import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def to_dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=_dev())


def _build(N=8, C=3, base=4):
    from DLWP.keras import backend
    backend.set_device('cuda:0')
    from DLWP.model.cs_unet import build_cs_model
    np.random.seed(5)
    return build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=base)


def _flat(model):
    return np.concatenate([w.ravel() for w in model.get_weights()])


def test_lr_change_after_graph_capture_is_honoured():
    """4 steps at lr = 1e-3, then lr = 5e-3 for 3 more: the graph-replayed model must track the eager model bitwise (same
    kernels, hyper-parameters read from device memory), and the change must be visible in the parameter delta."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((4, 6, 8, 8, 3)).astype(np.float32)
    t = rng.standard_normal((4, 6, 8, 8, 3)).astype(np.float32)
    finals, deltas = [], []
    w0 = None
    for use_graphs in (False, True):
        model = _build()
        model.use_graphs = use_graphs
        model.compile(optimizer='adam', loss='mse')
        if w0 is None:
            w0 = model.get_weights()
        model.set_weights(w0)
        dx, dt = [to_dev(x)], [to_dev(t)]
        for _ in range(4):
            model.train_on_device_batch(dx, dt)
        if use_graphs:
            assert model._graphs, 'the step should have been captured by now'
        before = _flat(model)
        model.optimizer.lr = 5e-3
        model.train_on_device_batch(dx, dt)
        torch.cuda.synchronize()
        after = _flat(model)
        deltas.append(np.abs(after - before).max())
        for _ in range(2):
            model.train_on_device_batch(dx, dt)
        torch.cuda.synchronize()
        finals.append(_flat(model))
    assert np.array_equal(finals[0], finals[1])
    # Adam's per-step displacement is ~lr: the step after the change must be ~5x an lr = 1e-3 step
    assert deltas[1] == deltas[0] and deltas[1] > 2.5e-3


def test_adam_step_dev_matches_by_value_kernel():
    from DLWP import ops
    rng = np.random.default_rng(12)
    n = 4096 + 12
    p0 = rng.standard_normal(n).astype(np.float32)
    g0 = rng.standard_normal(n).astype(np.float32)
    outs = []
    for dev_hyper in (False, True):
        p, g = to_dev(p0), to_dev(g0)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        state = torch.zeros(2, dtype=torch.int32, device=_dev())
        for it in range(3):
            g.copy_(to_dev(g0) * (it + 1))
            if dev_hyper:
                hyper = torch.tensor([2e-3, 0.9, 0.999, 1e-7, 0.5], dtype=torch.float32, device=_dev())
                ops.adam_step_dev(p, g, m, v, state, hyper, zero_grads=True)
            else:
                ops.adam_step(p, g, m, v, state, lr=2e-3, grad_scale=0.5, zero_grads=True)
        torch.cuda.synchronize()
        assert int(state[0].item()) == 3 and int(state[1].item()) == 0 and float(g.abs().max().item()) == 0.0
        outs.append((p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mse_backward_applies_upstream_gradient(dtype):
    from DLWP import ops
    rng = np.random.default_rng(13)
    y0 = rng.standard_normal((2, 6, 4, 4, 3)).astype(np.float32)
    t0 = rng.standard_normal((2, 6, 4, 4, 3)).astype(np.float32)
    y = to_dev(y0).to(dtype).requires_grad_(True)
    out = ops.mse_mae(y, to_dev(t0), 1.0)
    (3.0 * out[0]).backward()
    yy = y.detach().float().cpu().numpy().astype(np.float64)
    ref = 3.0 * 2.0 * (yy - t0) / yy.size
    got = y.grad.float().cpu().numpy()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # the unit seed of DLWP.keras.Model keeps the zero-cost path and the same numbers / 3
    y2 = to_dev(y0).to(dtype).requires_grad_(True)
    out2 = ops.mse_mae(y2, to_dev(t0), 1.0)
    torch.autograd.backward([out2], [ops.unit_seed(_dev())])
    got1 = y2.grad.float().cpu().numpy()
    assert np.abs(got1 * 3.0 - ref).max() <= tol * np.abs(ref).max()


def test_device_generator_rejects_out_of_range_samples():
    from DLWP.model import DLWPFunctional
    from DLWP.model.generators import ArrayDataGenerator
    rng = np.random.default_rng(14)
    arr = rng.standard_normal((12, 2, 6, 4, 4)).astype(np.float32)
    dlwp = DLWPFunctional(is_convolutional=True, time_dim=2)
    gen = ArrayDataGenerator(dlwp, arr, rank=3, batch_size=2, input_time_steps=2, output_time_steps=2, channels_last=True,
                             device='cuda:0')
    p, t = gen.generate(np.array([0, 8]))            # 8 + 3 = 11 is the last valid row
    assert p.is_cuda and tuple(p.shape) == (2, 6, 4, 4, 4)
    with pytest.raises(IndexError):
        gen.generate(np.array([0, 9]))
    with pytest.raises(IndexError):
        gen.generate(np.array([-1, 2]))


def test_mixed_precision_optimizer_switches_a_model_built_earlier():
    from DLWP.keras import mixed_precision
    from DLWP.keras.optimizers import Adam
    model = _build()
    assert model.compute_dtype == 'float32'
    try:
        opt = mixed_precision.enable_mixed_precision_graph_rewrite(Adam())
    finally:
        mixed_precision.disable_mixed_precision_graph_rewrite()
    model.compile(optimizer=opt, loss='mse')
    assert model.compute_dtype == 'bfloat16'
    rng = np.random.default_rng(15)
    x = rng.standard_normal((2, 6, 8, 8, 3)).astype(np.float32)
    hist = model.fit(x, x, batch_size=2, epochs=2, verbose=0)
    assert np.isfinite(hist.history['loss']).all()
    # a plain Adam() leaves an fp32 model alone
    m2 = _build()
    m2.compile(optimizer=Adam(), loss='mse')
    assert m2.compute_dtype == 'float32'
    assert orc is not None


@pytest.mark.parametrize('cout,use_graphs', [(14, False), (14, True), (26, False), (8, False)])
def test_fused_head_loss_step_equals_the_unfused_step(cout, use_graphs):
    """dlwpcs_head_mse_step (output layer + mse/mae + dy + the layer's data gradient in one launch) against the three-launch
    path: same dy / dx bits -> bitwise equal parameters after 3 Adam steps; loss and mae agree to fp32 summation order."""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    backend.set_device('cuda:0')
    rng = np.random.default_rng(17)
    N, B = 8, 3
    x = rng.standard_normal((B, 6, N, N, cout)).astype(np.float32)
    t = rng.standard_normal((B, 6, N, N, cout)).astype(np.float32)
    res = []
    w0 = None
    for fuse in (False, True):
        backend.set_compute_dtype('bfloat16')
        try:
            np.random.seed(5)
            model = build_cs_model((6, N, N, cout), cout, 'unet2', base_filter_number=32)
        finally:
            backend.set_compute_dtype('float32')
        model.fuse_head_loss = fuse
        model.use_graphs = use_graphs
        model.compile(optimizer='adam', loss='mse', metrics=['mae'])
        if w0 is None:
            w0 = model.get_weights()
        model.set_weights(w0)
        dx = [to_dev(x).to(torch.bfloat16)]
        dt = [to_dev(t)]
        stats = None
        for _ in range(3):
            stats = model.train_on_device_batch(dx, dt)
        torch.cuda.synchronize()
        res.append((_flat(model), stats.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.allclose(res[0][1], res[1][1], rtol=1e-5)
    assert res[1][1][0, 0] > 0 and res[1][1][0, 1] > 0
