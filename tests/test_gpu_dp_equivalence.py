"""
Data-parallel identity THROUGH THE ENGINE (reference counterpart DLWP/model/models.py:369-374): two ranks that
share cuda:0 (gloo process group: host-staged all-reduce -- the RCCL call is the same `dist.all_reduce`) each train on their
shard of a global batch; a single process trains on the whole batch.  After 3 Adam steps the flat fp32 parameter buffers
agree to <= 1e-6 relative, in eager mode and with the hipGraph-replayed step (fwd+bwd graph, all-reduce, optimizer graph).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, C, BASE, B, STEPS = 8, 4, 4, 8, 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data():
    rng = np.random.default_rng(21)
    xs = rng.standard_normal((STEPS, B, 6, N, N, C)).astype(np.float32)
    ts = rng.standard_normal((STEPS, B, 6, N, N, C)).astype(np.float32)
    return xs, ts


def _train(rank, world, use_graphs, w0=None, dtype='float32', buckets=None):
    """the training loop both arrangements share; returns the final flat parameters (numpy)"""
    for p in (ROOT, os.path.join(ROOT, 'dlwp-cs_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from DLWP import parallel
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    backend.set_device('cuda:0')
    np.random.seed(7 + rank)                    # different initial weights per rank: compile() must broadcast rank 0's
    backend.set_compute_dtype(dtype)
    try:
        model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=BASE)
    finally:
        backend.set_compute_dtype('float32')
    model.use_graphs = use_graphs
    if buckets is not None:
        model.exchange_buckets = buckets
    model.compile(optimizer='adam', loss='mse')
    if w0 is not None:
        model.set_weights(w0)
    xs, ts = _data()
    lo, hi = parallel.shard_bounds(B, rank, world)
    dev = torch.device('cuda', 0)
    # static device buffers, refilled in place every step (the captured graphs read them directly)
    dx = [torch.empty((hi - lo, 6, N, N, C), dtype=torch.bfloat16 if dtype == 'bfloat16' else torch.float32, device=dev)]
    dt = [torch.empty((hi - lo, 6, N, N, C), dtype=torch.float32, device=dev)]
    model.static_batch_buffers = True
    w_init = model.get_weights()
    for s in range(STEPS):
        dx[0].copy_(torch.from_numpy(xs[s, lo:hi]))
        dt[0].copy_(torch.from_numpy(ts[s, lo:hi]))
        model.train_on_device_batch(dx, dt)
    torch.cuda.synchronize()
    if use_graphs:
        assert model._graphs, 'the step should have been captured'
        g = next(iter(model._graphs.values()))
        split = buckets == 2
        assert (g['update'] is not None) == (world > 1 or split)  # DP: all-reduce between the graphs
        assert (g['bwd_b'] is not None) == split                  # two buckets: the backward pass is two graphs
    if buckets == 2:
        assert model._did_split and model._plan_exchange() is not None
        sa, sb = model._exchange_slices()
        assert sa.numel() > 0 and sb.numel() > 0 and sa.numel() + sb.numel() == model._flat_grads.numel()
    return w_init, model._flat_params.detach().cpu().numpy().copy()


def _worker(rank, world, port, use_graphs, ret, dtype='float32', buckets=None):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    w_init, flat = _train(rank, world, use_graphs, dtype=dtype, buckets=buckets)
    ret[rank] = ([a.copy() for a in w_init], flat)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('buckets', [1, 2])
@pytest.mark.parametrize('use_graphs', [False, True])
def test_two_rank_shards_equal_one_rank_global_batch(use_graphs, buckets):
    """buckets = 2: the exchange in two buckets, the first one in flight during the encoder-side half of the backward pass"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), use_graphs, ret, 'float32', buckets), nprocs=world, join=True)
    (w0_a, flat_a), (w0_b, flat_b) = ret[0], ret[1]
    for a, b in zip(w0_a, w0_b):
        assert np.array_equal(a, b)                              # broadcast at compile()
    assert np.array_equal(flat_a, flat_b)                        # replicas stay bitwise identical
    _, flat_1 = _train(0, 1, use_graphs, w0=w0_a)
    scale = np.abs(flat_1).max()
    err = np.abs(flat_a - flat_1).max() / scale
    assert err <= 1e-6, err
    # and the training actually moved the parameters
    flat_0 = np.concatenate([np.pad(a.ravel(), (0, (-a.size) % 64)) for a in w0_a])
    assert np.abs(flat_1 - flat_0).max() > 1e-3


def test_two_rank_shards_equal_one_rank_global_batch_bf16():
    """the same identity in the bf16 mode (pre-masked gradients, batched weight gradients, two exchange buckets, graphs):
    per-sample arithmetic is identical, only the fp32 summation order of the weight gradients differs"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), True, ret, 'bfloat16', 2), nprocs=world, join=True)
    (w0_a, flat_a), (w0_b, flat_b) = ret[0], ret[1]
    assert np.array_equal(flat_a, flat_b)
    _, flat_1 = _train(0, 1, True, w0=w0_a, dtype='bfloat16')
    flat_0 = np.concatenate([np.pad(a.ravel(), (0, (-a.size) % 64)) for a in w0_a])
    d2, d1 = flat_a - flat_0, flat_1 - flat_0
    cos = float(np.dot(d2, d1) / (np.linalg.norm(d2) * np.linalg.norm(d1)))
    assert cos > 0.999, cos


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('use_graphs', [False, True])
def test_split_backward_pass_equals_the_single_pass(dtype, use_graphs):
    """DLWPCS_EXCHANGE_BUCKETS=2 at world size 1: forward + decoder-side backward | encoder-side backward | optimizer against
    the one-pass step -- the same gradients (fp32: same launches per layer, bitwise; bf16: the batched weight gradient is cut
    into two launches with their own partial-sum layout, so the fp32 sums differ in the last bits)"""
    from DLWP import ops
    w0, flat_one = _train(0, 1, use_graphs, dtype=dtype, buckets=1)
    ops._wb_plans.clear()
    _, flat_two = _train(0, 1, use_graphs, w0=w0, dtype=dtype, buckets=2)
    if dtype == 'bfloat16':
        # the decoder-side and the encoder-side layers each got a batched weight-gradient launch (and plan) of their own
        assert len(ops._wb_plans) == 2
    _, flat_one = _train(0, 1, use_graphs, w0=w0, dtype=dtype, buckets=1)
    flat_0 = np.concatenate([np.pad(a.ravel(), (0, (-a.size) % 64)) for a in w0])
    if dtype == 'float32':
        assert np.abs(flat_two - flat_one).max() <= 1e-6 * np.abs(flat_one).max()
    else:
        d2, d1 = flat_two - flat_0, flat_one - flat_0
        cos = float(np.dot(d2, d1) / (np.linalg.norm(d2) * np.linalg.norm(d1)))
        assert cos > 0.9995, cos
    assert np.abs(flat_one - flat_0).max() > 1e-3
