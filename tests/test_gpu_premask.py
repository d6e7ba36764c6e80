"""
GPU tests of the pre-masked gradient convention (include/dlwpcs.h: dlwpcs_conv_bwd_data_masked, dlwpcs_avgpool2_bwd_masked,
dlwpcs_head_mse_step_masked; DLWP.keras.Model._plan_premask): the gradient w.r.t. the output y of an activated layer is
multiplied by act'(y) where it is produced, so that the layer receives dz = dy * act'(y) (keras ReLU(negative_slope,
max_value), Azure/train_cs.py:199; act' evaluated from y like the reference's backward: slope for y < 0, 1 for 0 < y < max,
0 otherwise).

Checker: the same gradient computed WITHOUT the convention by the kernels that are pinned to the fp64 oracle elsewhere
(tests/test_gpu_bf16.py), multiplied by act'(m) in fp32 and rounded to bf16 -- the masked kernels round in a different place
(interior cells before the multiply, ring cells after), so the comparison allows 3 bf16 ulp of max|ref|; where the order is
the same (interior cells of the direct-store epilogue) it is bit-exact.  The whole-model tests compare training steps with
the convention switched on and off against each other and against the fp64 oracle network.
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


def _bf(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32).to(torch.bfloat16).to(_dev())


def _f32(t):
    return t.detach().float().cpu().numpy()


def slope(y, alpha=ALPHA, vmax=VMAX):
    y = np.asarray(y, dtype=np.float32)
    return np.where(y < 0, np.float32(alpha), np.where((y > 0) & (y < vmax), np.float32(1.0), np.float32(0.0)))


def masked_ref(g, m, alpha=ALPHA, vmax=VMAX):
    """bf16(g * act'(m)) with g, m bf16 tensors"""
    r = torch.tensor(_f32(g) * slope(_f32(m), alpha, vmax)).to(torch.bfloat16)
    return r.float().numpy()


def close(a, ref, ulps):
    d = np.abs(ref).max()
    return np.abs(a - ref).max() <= ulps * EPS * (d if d > 0 else 1.0)


def _conv_grads(B, N, C0, C1, up0, Cout, k, halo, mask0, mask1, seed):
    """(masked dsrc0, dsrc1) and (plain dsrc0, dsrc1), (src0, src1)"""
    from DLWP import _native as nat
    rng = np.random.default_rng(seed)
    dev = _dev()
    n0 = N // 2 if up0 else N
    # sources with values on both sides of 0 and beyond max_value so that all three branches of act' occur
    src0 = _bf(rng.standard_normal((B, 6, n0, n0, C0)) * 6.0)
    src1 = _bf(rng.standard_normal((B, 6, N, N, C1)) * 6.0) if C1 else None
    No = N if halo else N - k + 1
    dz = _bf(rng.standard_normal((B, 6, No, No, Cout)))
    w = [torch.tensor(rng.standard_normal((k, k, C0 + C1, Cout)) / np.sqrt(k * k * (C0 + C1)), dtype=torch.float32, device=dev)
         for _ in range(2)]
    d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=int(halo), up0=int(up0), flip_north_pole=1, act=0,
                     alpha=0., vmax=0., dtype=nat.BF16, flags=0, c0_valid=0)
    nbytes = nat.lib().dlwpcs_conv_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    inv = nat.halo_tables(N, 1, dev)[1] if halo else None
    out = []
    for masked in (True, False):
        g0 = torch.empty_like(src0)
        g1 = torch.empty_like(src1) if C1 else None
        if masked:
            nat.check(nat.lib().dlwpcs_conv_bwd_data_masked(ctypes.byref(d), nat.ptr(dz), nat.ptr(w[0]), nat.ptr(w[1]), 0,
                                                            nat.ptr(g0), nat.ptr(g1), nat.ptr(src0 if mask0 else None),
                                                            nat.ptr(src1 if mask1 else None), ALPHA, VMAX, nat.ptr(inv),
                                                            nat.ptr(ws), nbytes, nat.stream_ptr()), 'conv_bwd_data_masked')
        else:
            nat.check(nat.lib().dlwpcs_conv_bwd_data(ctypes.byref(d), nat.ptr(dz), 0, nat.ptr(w[0]), nat.ptr(w[1]), 0,
                                                     nat.ptr(g0), nat.ptr(g1), nat.ptr(inv), nat.ptr(ws), nbytes,
                                                     nat.stream_ptr()), 'conv_bwd_data')
        torch.cuda.synchronize()
        out.append((g0, g1))
    return out[0], out[1], (src0, src1)


# (B, N, C0, C1, up0, Cout, k, halo, mask0, mask1)
CONV_CASES = [
    (2, 48, 32, 0, 0, 32, 3, 1, True, False),      # direct-store epilogue + ring fix-up, 384-pixel tiles
    (2, 24, 64, 0, 0, 64, 3, 1, True, False),      # ... 64 -> 64 (two 32-channel groups)
    (2, 12, 128, 0, 0, 64, 3, 1, True, False),     # ... 128 gradient channels (64 per workgroup)
    (2, 24, 64, 0, 0, 32, 3, 1, True, False),
    (2, 24, 64, 64, 1, 64, 3, 1, True, False),     # decoder: upsampled source masked by the inverse-gather kernel
    (2, 24, 64, 64, 1, 64, 3, 1, True, True),      # ... and the skip source by epilogue + ring fix-up
    (2, 16, 32, 32, 0, 32, 3, 1, True, True),      # two directly written sources
    (2, 16, 32, 32, 0, 32, 3, 1, False, True),
    (3, 10, 12, 0, 0, 8, 3, 1, True, False),       # 4-channel vectors: routing kernels / elementwise finish
    (2, 12, 7, 0, 0, 5, 3, 1, True, False),        # odd channel counts
    (2, 16, 32, 0, 0, 14, 1, 0, True, False),      # pointwise head kernel, gradient written in place
    (2, 14, 16, 0, 0, 16, 3, 0, True, False),      # 'valid' 3x3 on a padded tensor, written in place
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_masked_data_gradient(case):
    B, N, C0, C1, up0, Cout, k, halo, mask0, mask1 = case
    (m0, m1), (p0, p1), (s0, s1) = _conv_grads(*case, seed=abs(hash(case)) % (2 ** 31))
    ulps = 5 if up0 else 3
    for g_m, g_p, src, on in ((m0, p0, s0, mask0), (m1, p1, s1, mask1)):
        if g_m is None:
            continue
        if on:
            assert close(_f32(g_m), masked_ref(g_p, src), ulps), case
        else:
            assert torch.equal(g_m, g_p), case


def test_masked_epilogue_is_exact_on_interior_cells():
    """interior cells take one path in both runs: round(acc) -> * act' -> round"""
    case = (2, 48, 32, 0, 0, 32, 3, 1, True, False)
    (m0, _), (p0, _), (s0, _) = _conv_grads(*case, seed=3)
    a, ref = _f32(m0)[:, :, 1:-1, 1:-1], masked_ref(p0, s0)[:, :, 1:-1, 1:-1]
    assert np.array_equal(a, ref)
    sl = slope(_f32(s0))
    assert (sl == 0).any() and (sl == 1).any() and (sl == np.float32(ALPHA)).any()


@pytest.mark.parametrize('B,N,C,skip', [(2, 24, 32, True), (2, 8, 64, True), (3, 12, 6, True), (2, 16, 32, False)])
def test_masked_pool_backward(B, N, C, skip):
    from DLWP import _native as nat
    rng = np.random.default_rng(N + C)
    x = _bf(rng.standard_normal((B, 6, N, N, C)) * 6.0)
    dy = _bf(rng.standard_normal((B, 6, N // 2, N // 2, C)))
    dskip = _bf(rng.standard_normal((B, 6, N, N, C))) if skip else None
    dx = torch.empty_like(x)
    nat.check(nat.lib().dlwpcs_avgpool2_bwd_masked(nat.ptr(dy), nat.ptr(dskip), nat.ptr(x), nat.ptr(dx), B, N, C, ALPHA, VMAX,
                                                   nat.BF16, nat.stream_ptr()), 'avgpool2_bwd_masked')
    g = 0.25 * np.repeat(np.repeat(_f32(dy), 2, axis=2), 2, axis=3)
    if skip:
        g = _f32(dskip) + g
    ref = torch.tensor(g * slope(_f32(x))).to(torch.bfloat16).float().numpy()
    assert np.array_equal(_f32(dx), ref)            # one fp32 expression, one rounding


def test_ops_level_chain_matches_the_plain_path():
    """conv+ReLU -> pool (skip) -> conv+ReLU -> decoder conv on [up, skip]: gradients of all parameters and of the input with
    the convention on (consumers mask, producers take dz) and off"""
    from DLWP import ops
    from DLWP._native import ACT_LEAKY_CLIP
    rng = np.random.default_rng(2)
    dev = _dev()
    B, N = 2, 16
    x = _bf(rng.standard_normal((B, 6, N, N, 8)) * 2)

    def params(cin, cout):
        return [torch.tensor(rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin), dtype=torch.float32, device=dev)
                for _ in range(2)] + [torch.tensor(rng.standard_normal(cout) * 0.1, dtype=torch.float32, device=dev)
                                      for _ in range(2)]
    P1, P2, P3 = params(8, 32), params(32, 32), params(64, 16)
    gy = _bf(rng.standard_normal((B, 6, N, N, 16)))
    res = []
    for on in (False, True):
        pm = (ALPHA, VMAX) if on else None
        xx = x.clone().requires_grad_(True)
        ps = [[p.clone().requires_grad_(True) for p in P] for P in (P1, P2, P3)]

        def conv(src0, p, src1=None, up0=False, premask0=None, premask1=None, dyp=False):
            return ops.cs_conv(src0, p[0], p[1], None, p[2], p[3], None, src1=src1, ksize=3, halo=True, up0=up0,
                               act=ACT_LEAKY_CLIP, alpha=ALPHA, vmax=VMAX, premask0=premask0, premask1=premask1,
                               dy_premasked=dyp)
        a = conv(xx, ps[0], dyp=on)                              # consumers: pooling node (masks), decoder via the alias
        pooled, alias = ops.avgpool2_skip(a, pm)
        b = conv(pooled, ps[1], dyp=on)                          # consumer: decoder conv as upsampled source 0 (masks)
        y = conv(b, ps[2], src1=alias, up0=True, premask0=pm)    # the alias hands plain gradients back to the pooling node
        y.backward(gy)
        res.append([xx.grad] + [p.grad for P in ps for p in P])
    for g_off, g_on in zip(*res):
        a, b = _f32(g_off), _f32(g_on)
        assert np.abs(a - b).max() <= 4 * EPS * np.abs(a).max()


@pytest.mark.parametrize('graphs', [False, True])
def test_unet2_training_with_and_without_the_convention(graphs):
    """three Adam steps of a bf16 `unet2`: pre-masked gradients + batched weight gradients against the plain per-layer path"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 16, 14, 4
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev)
    w0, out = None, []
    for on in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'premask=%s,wgrad_batch=%s' % (on, on)
        try:
            backend.set_compute_dtype('bfloat16')
            try:
                np.random.seed(5)
                model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
            finally:
                backend.set_compute_dtype('float32')
            model.use_graphs = graphs
            model.compile(optimizer='adam', loss='mse', metrics=['mae'])
            if w0 is None:
                w0 = model.get_weights()
            model.set_weights(w0)
            if on == '1':
                assert len(model._premask) == 9         # every activated layer but the first (no data gradient there)
            stats = None
            for _ in range(3):
                stats = model.train_on_device_batch([x], [t])
            torch.cuda.synchronize()
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    (p_off, s_off), (p_on, s_on) = out
    # Adam's first steps move every weight by ~lr regardless of the gradient's size: compare the UPDATES
    d_off, d_on = p_off - np.concatenate([w.ravel() for w in w0]), p_on - np.concatenate([w.ravel() for w in w0])
    cos = float(np.dot(d_off, d_on) / (np.linalg.norm(d_off) * np.linalg.norm(d_on)))
    assert cos > 0.995, cos
    assert abs(s_on[0, 0] - s_off[0, 0]) <= 2e-3 * abs(s_off[0, 0])


def test_optimizer_fused_into_the_reduction_gives_the_same_bits():
    """dlwpcs_wgrad_batch_adam (hipGraph-replayed steps) against reduction + dlwpcs_adam_step_dev: same element arithmetic"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 16, 14, 4
    rng = np.random.default_rng(1)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev)
    w0, out = None, []
    for fuse in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'fuse_adam=' + fuse
        try:
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
            for _ in range(5):                      # eager warm-up, capture, three replays
                stats = model.train_on_device_batch([x], [t])
            torch.cuda.synchronize()
            assert model._update_done == (fuse == '1')
            assert model.optimizer.iterations == 5
            assert float(model._flat_grads.abs().max()) == 0.0
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('dtype', ['bfloat16', 'float32'])
def test_optimizer_fused_into_the_reduction_with_layers_applied_twice(dtype):
    """The reference scripts' production wiring (integration_steps = 2, shared weights, solar + constants inputs,
    /root/reference/Azure/train_cs.py:391-430): every layer names its gradient tensors twice.  The fused optimizer consumes a tensor in
    the reduction launch of its LAST item; parameters and statistics are bitwise those of reduction + dlwpcs_adam_step_dev."""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    N, V, ITS, K, B, base = 16, 4, 2, 2, 4, 8
    c_main, c_out = (V + 1) * ITS, V * ITS
    rng = np.random.default_rng(3)
    adt = torch.bfloat16 if dtype == 'bfloat16' else torch.float32
    mk = lambda *sh: torch.tensor(rng.standard_normal(sh), dtype=torch.float32, device=dev)
    xs = [mk(B, 6, N, N, c_main).to(adt), mk(B, ITS, 6, N, N, 1).to(adt), mk(B, 6, N, N, K).to(adt)]
    ts = [mk(B, 6, N, N, c_out), mk(B, 6, N, N, c_out)]
    w0, out = None, []
    for fuse in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'fuse_adam=' + fuse
        try:
            backend.set_compute_dtype(dtype)
            try:
                np.random.seed(5)
                model = build_cs_model((6, N, N, c_main), c_out, 'unet2', base_filter_number=base, integration_steps=2, io_time_steps=ITS,
                                       insolation_shape=(ITS, 6, N, N, 1), constants_shape=(6, N, N, K))
            finally:
                backend.set_compute_dtype('float32')
            model.compile(optimizer='adam', loss='mse', loss_weights=[0.5, 0.5], metrics=['mae'])
            if w0 is None:
                w0 = model.get_weights()
            model.set_weights(w0)
            for _ in range(5):                      # eager warm-up, capture, three replays
                stats = model.train_on_device_batch(xs, ts)
            torch.cuda.synchronize()
            # (fp32: the 10 + 2 channel first layer has no batched fp32 kernel -- a 16-B vector would straddle the sources -- so the
            # list does not cover every parameter and the optimizer stays a launch of its own: same numbers either way)
            assert model._update_done == (fuse == '1' and dtype == 'bfloat16')
            assert model.optimizer.iterations == 5
            assert float(model._flat_grads.abs().max()) == 0.0
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[0][0] - np.concatenate([w.ravel() for w in w0])).max() > 0


@pytest.mark.parametrize('B,N,Cin,Cout', [(2, 24, 32, 64), (3, 12, 64, 128), (2, 16, 32, 32)])
def test_ring_fix_folded_into_the_pooling_adjoint(B, N, Cin, Cout):
    """dlwpcs_conv_bwd_data_masked with DLWPCS_CONV_DEFER_RING0 + dlwpcs_avgpool2_bwd_ring against the same call with its own
    ring fix-up launch + dlwpcs_avgpool2_bwd_masked: the same bits (the ring is added to the bf16 interior value in the same
    order and rounded to bf16 before the pooling adjoint uses it)"""
    import ctypes
    from DLWP import _native as nat, ops
    dev = _dev()
    rng = np.random.default_rng(N * 7 + Cin)
    lib = nat.lib()
    d = nat.ConvDesc(B=B, N=N, C0=Cin, C1=0, Cout=Cout, ksize=3, halo=1, up0=0, flip_north_pole=1, act=0, alpha=0., vmax=0.,
                     dtype=nat.BF16, flags=0, c0_valid=0)
    info = ops.halo_ring_info(d)
    assert info is not None
    dz = _bf(rng.standard_normal((B, 6, N, N, Cout)))
    w = [torch.tensor(rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin), dtype=torch.float32, device=dev)
         for _ in range(2)]
    xfull = _bf(rng.standard_normal((B, 6, 2 * N, 2 * N, Cin)) * 6.0)       # the pooling node's input (mask) ...
    dskip = _bf(rng.standard_normal((B, 6, 2 * N, 2 * N, Cin)))             # ... and the gradient of its alias
    table, inv = nat.halo_tables(N, 1, dev)
    nbytes = lib.dlwpcs_conv_workspace_bytes(ctypes.byref(d))
    out = []
    for fold in (False, True):
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        dsrc = torch.empty((B, 6, N, N, Cin), dtype=torch.bfloat16, device=dev)
        dx = torch.empty_like(xfull)
        d.flags = nat.CONV_DEFER_RING0 if fold else 0
        nat.check(lib.dlwpcs_conv_bwd_data_masked(ctypes.byref(d), nat.ptr(dz), nat.ptr(w[0]), nat.ptr(w[1]), None,
                                                  nat.ptr(dsrc), None, None, None, 0.0, 0.0, nat.ptr(inv), nat.ptr(ws),
                                                  ws.numel(), nat.stream_ptr()), 'bwd_data_masked')
        if fold:
            nat.check(lib.dlwpcs_avgpool2_bwd_ring(nat.ptr(dsrc), nat.ptr(dskip), nat.ptr(xfull), nat.ptr(dx), B, 2 * N, Cin,
                                                   ALPHA, VMAX, nat.BF16, ws.data_ptr() + info[0], nat.ptr(inv), info[1], 0,
                                                   nat.stream_ptr()), 'avgpool2_bwd_ring')
        else:
            nat.check(lib.dlwpcs_avgpool2_bwd_masked(nat.ptr(dsrc), nat.ptr(dskip), nat.ptr(xfull), nat.ptr(dx), B, 2 * N, Cin,
                                                     ALPHA, VMAX, nat.BF16, nat.stream_ptr()), 'avgpool2_bwd_masked')
        torch.cuda.synchronize()
        out.append((_f32(dx), _f32(dsrc)))
    assert np.array_equal(out[0][0], out[1][0])
    # the deferred call really left the ring out (otherwise this test proves nothing)
    edge = np.zeros((N, N), bool)
    edge[0, :] = edge[-1, :] = edge[:, 0] = edge[:, -1] = True
    assert not np.array_equal(out[0][1][:, :, edge], out[1][1][:, :, edge])
    assert np.array_equal(out[0][1][:, :, ~edge], out[1][1][:, :, ~edge])


def test_unet2_training_with_the_ring_folded_gives_the_same_bits():
    """engine option fold_ring on/off on a bf16 unet2 (eager warm-up, capture, replays): two ring fix-up launches less, same weights"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    from DLWP import _native as nat
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 16, 14, 4
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev)
    w0, out = None, []
    for fold in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'fold_ring=' + fold
        try:
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
            assert len(model._defer_ring) == 2          # the first convolution of each of the two lower levels
            for _ in range(5):
                stats = model.train_on_device_batch([x], [t])
            torch.cuda.synchronize()
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('B,N,C0,C1,up0,Cout', [(2, 48, 32, 0, 0, 32), (3, 24, 64, 0, 0, 64), (2, 24, 32, 0, 0, 64),
                                                 (2, 48, 14, 0, 0, 32), (2, 12, 64, 0, 0, 128), (2, 24, 64, 64, 1, 64),
                                                 (1, 20, 16, 0, 0, 32), (2, 16, 8, 0, 0, 24), (2, 96, 32, 0, 0, 32),
                                                 (1, 96, 26, 0, 0, 32), (1, 96, 64, 0, 0, 64)])
def test_pooling_as_a_second_output_of_the_convolution(dtype, B, N, C0, C1, up0, Cout):
    """dlwpcs_conv_fwd_pool against dlwpcs_conv_fwd + dlwpcs_avgpool2_fwd: the same bits in both outputs, whether the tiling lets
    the epilogue pool (the U-Net levels at N = 48 / 24) or the call falls back to the pooling launch"""
    import ctypes
    from DLWP import _native as nat
    dev = _dev()
    rng = np.random.default_rng(N + C0 + Cout)
    lib = nat.lib()
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=3, halo=1, up0=up0, flip_north_pole=1, act=nat.ACT_LEAKY_CLIP,
                     alpha=ALPHA, vmax=VMAX, dtype=nat.BF16 if dtype == 'bf16' else nat.F32, flags=0, c0_valid=0)
    n0 = N // 2 if up0 else N
    s0 = torch.tensor(rng.standard_normal((B, 6, n0, n0, C0)) * 3, dtype=torch.float32, device=dev).to(tdt)
    s1 = torch.tensor(rng.standard_normal((B, 6, N, N, C1)) * 3, dtype=torch.float32, device=dev).to(tdt) if C1 else None
    cin = C0 + C1
    w = [torch.tensor(rng.standard_normal((3, 3, cin, Cout)) / np.sqrt(9 * cin), dtype=torch.float32, device=dev) for _ in range(2)]
    b = [torch.tensor(rng.standard_normal(Cout) * 0.2, dtype=torch.float32, device=dev) for _ in range(2)]
    table = nat.halo_tables(N, 1, dev)[0]
    ws = torch.empty(lib.dlwpcs_conv_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    y0 = torch.empty((B, 6, N, N, Cout), dtype=tdt, device=dev)
    p0 = torch.empty((B, 6, N // 2, N // 2, Cout), dtype=tdt, device=dev)
    y1, p1 = torch.full_like(y0, 7.0), torch.full_like(p0, 7.0)
    args = (nat.ptr(s0), nat.ptr(s1), nat.ptr(w[0]), nat.ptr(w[1]), None, nat.ptr(b[0]), nat.ptr(b[1]), None)
    nat.check(lib.dlwpcs_conv_fwd(ctypes.byref(d), *args, nat.ptr(y0), nat.ptr(table), nat.ptr(ws), ws.numel(), nat.stream_ptr()),
              'conv_fwd')
    nat.check(lib.dlwpcs_avgpool2_fwd(nat.ptr(y0), nat.ptr(p0), B, N, Cout, d.dtype, nat.stream_ptr()), 'avgpool2_fwd')
    nat.check(lib.dlwpcs_conv_fwd_pool(ctypes.byref(d), *args, nat.ptr(y1), nat.ptr(p1), nat.ptr(table), nat.ptr(ws), ws.numel(),
                                       nat.stream_ptr()), 'conv_fwd_pool')
    torch.cuda.synchronize()
    assert np.array_equal(_f32(y0), _f32(y1))
    assert np.array_equal(_f32(p0), _f32(p1))
    assert np.abs(_f32(p0)).max() > 0.1


def test_unet2_training_with_the_pooling_fused_gives_the_same_bits():
    """engine option fuse_pool on/off on a bf16 unet2 at N = 48 (where the epilogue can pool): no pooling launches in the forward pass,
    same weights"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    from DLWP import _native as nat
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 48, 14, 2
    rng = np.random.default_rng(4)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev)
    w0, out = None, []
    lib = nat.lib()
    for fuse in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'fuse_pool=' + fuse
        try:
            backend.set_compute_dtype('bfloat16')
            try:
                np.random.seed(5)
                model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
            finally:
                backend.set_compute_dtype('float32')
            model.use_graphs = False
            model.compile(optimizer='adam', loss='mse', metrics=['mae'])
            if w0 is None:
                w0 = model.get_weights()
            model.set_weights(w0)
            assert len(model._pool_producers) == 2
            lib.dlwpcs_prof_reset()
            lib.dlwpcs_prof_enable(1)
            for _ in range(3):
                stats = model.train_on_device_batch([x], [t])
            torch.cuda.synchronize()
            lib.dlwpcs_prof_enable(0)
            lib.dlwpcs_prof_reset()
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


def test_rollout_with_padded_state_gives_the_same_series():
    """bf16 rollout of a 26-channel unet2 (BASELINE config 5's layout, small face): the state padded to 32 channels between the
    passes (DLWPCS_CONV_OUT_PADDED head, c0_valid first layer) against the plain 26-channel state -- the same bits, through
    predict_on_device and through DLWPFunctional-style rollout_on_device"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    N, C, B = 16, 26, 2
    rng = np.random.default_rng(9)
    backend.set_compute_dtype('bfloat16')
    try:
        np.random.seed(3)
        model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
    finally:
        backend.set_compute_dtype('float32')
    x = torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    outs = []
    # (fold_head=0: the padded state would otherwise take its head through the epilogue of the last convolution, another summation
    # order inside the head's MFMA -- tests/test_gpu_head_fold.py compares that form; here the two layouts must give the same bits)
    os.environ['DLWPCS_OPTIONS'] = 'fold_head=0'
    try:
        np.random.seed(3)
        backend.set_compute_dtype('bfloat16')
        try:
            model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=32)
        finally:
            backend.set_compute_dtype('float32')
    finally:
        os.environ.pop('DLWPCS_OPTIONS', None)
    for padded in (False, True):
        s = x
        for i in range(3):
            s = model.predict_on_device(s, repack=(i == 0), padded_io=padded)
        torch.cuda.synchronize()
        assert s.shape[-1] == (32 if padded else 26)
        if padded:
            assert float(s[..., 26:].float().abs().max()) == 0.0
        outs.append(_f32(s[..., :26]))
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0]).max() > 1e-3
    series = []
    xh = x.float().cpu().numpy()
    for flag in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'padded_io=' + flag
        try:
            out = np.full((3, B, 6, N, N, C), np.nan, dtype=np.float32)
            model.rollout_on_device(xh, 3, 1, out)
            series.append(out)
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(series[0], series[1])
    assert np.array_equal(series[0][-1], outs[0])


def _packed_snapshot(model, dev):
    st = model._pack_state(dev)
    return [b.clone() for ent in st['keep'] for b in ent[6] if b is not None]


def test_packed_operands_refreshed_by_the_optimizer_launch():
    """engine option fuse_pack: hipGraph-replayed steps start without the packing launch, the reduction + optimizer launch writes the
    updated parameters into the packed bf16 operands.  (1) after replays the packed buffers equal what dlwpcs_pack_batch makes
    of the current parameters, bit for bit; (2) the same weights as with the packing launch, also when eager steps (another
    batch size: an optimizer launch of its own) and set_weights come between replays"""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    from DLWP import ops
    dev = _dev()
    backend.set_device('cuda:0')
    N, C = 16, 14
    rng = np.random.default_rng(11)
    mk = lambda B: (torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(torch.bfloat16),
                    torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev))
    (x4, t4), (x2, t2) = mk(4), mk(2)
    w0, out = None, []
    for fuse in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'fuse_pack=' + fuse
        try:
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
            for _ in range(4):                                  # eager, capture, two replays
                model.train_on_device_batch([x4], [t4])
            g = next(iter(model._graphs.values()))
            assert g['no_pack'] == (fuse == '1')
            if fuse == '1':
                torch.cuda.synchronize()
                assert model._packed_ok
                have = _packed_snapshot(model, dev)
                st = model._pack_state(dev)
                ops.pack_batch(st['items'], st['n'])
                torch.cuda.synchronize()
                for a, b in zip(have, _packed_snapshot(model, dev)):
                    assert torch.equal(a, b)
            model.train_on_device_batch([x2], [t2])             # another shape: eager step, optimizer launch of its own
            if fuse == '1':
                assert not model._packed_ok
            model.train_on_device_batch([x4], [t4])             # replay: must repack first
            model.set_weights([w * 0.5 for w in model.get_weights()])
            stats = None
            for _ in range(2):
                stats = model.train_on_device_batch([x4], [t4])
            torch.cuda.synchronize()
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), stats.cpu().numpy().copy()))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('B,N,Cout', [(3, 16, 14), (2, 48, 8), (5, 12, 32), (1, 24, 26)])
def test_pointwise_data_gradient_masks_inside_the_launch(B, N, Cout):
    """Round 6: the pointwise layer's data gradient with a pre-masked source (dlwpcs_conv_bwd_data_masked on a 1 x 1 layer:
    pw_dgrad_kernel<true>) multiplies by act'(source) inside the launch -- the rounded gradient times the slope, rounded again:
    bitwise what the unmasked launch followed by the masking pass gave (the production model's first-step head, 14.7 us per step)."""
    from DLWP import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(100 * N + Cout)
    x = (torch.randn(B, 6, N, N, 32, device=dev, generator=g) * 6).to(torch.bfloat16)
    w = [torch.randn(1, 1, 32, Cout, device=dev, generator=g) * 0.2 for _ in range(2)]
    b = [torch.randn(Cout, device=dev, generator=g) * 0.1 for _ in range(2)]
    gy = torch.randn(B, 6, N, N, Cout, device=dev, generator=g).to(torch.bfloat16)
    grads = []
    for pm in ((ALPHA, VMAX), None):
        xi = x.clone().requires_grad_(True)
        y = ops.cs_conv(xi, w[0], w[1], None, b[0], b[1], None, ksize=1, halo=False, premask0=pm)
        y.backward(gy)
        grads.append(xi.grad)
    ref = masked_ref(grads[1], x)
    got = _f32(grads[0])
    assert np.isfinite(got).all() and np.abs(got).max() > 0
    assert (slope(_f32(x)) == 0).any() and (slope(_f32(x)) == np.float32(ALPHA)).any()
    assert np.array_equal(got, ref)
