"""
CPU tests of the multi-GPU path (world_size 2, gloo): parameter broadcast, flat-gradient all-reduce and batch sharding
of DLWP.parallel, and the data-parallel identity the engine relies on -- averaging per-rank gradients of equal shards
equals the gradient of the global-batch mean loss (computed with the oracle, since the product has no CPU compute path).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'dlwp-cs_amd'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from DLWP import parallel
    from oracle import cs_oracle as orc
    out = {}
    assert parallel.world() == (rank, world)

    # 1. broadcast: replicas start identical
    flat = torch.full((1000,), float(rank + 1))
    parallel.broadcast_parameters(flat)
    out['bcast_ok'] = bool((flat == 1.0).all())

    # 2. sharding is a partition
    out['shard'] = parallel.shard_bounds(7)

    # 3. DP identity on a tiny U-Net with the oracle as compute engine
    params = orc.make_unet2_params(3, 3, base=4, seed=5)
    leaves = [v.requires_grad_(True) for prm in params for v in prm.values()]
    rng = np.random.default_rng(9)
    x = torch.tensor(rng.standard_normal((4, 6, 8, 8, 3)))
    t = torch.tensor(rng.standard_normal((4, 6, 8, 8, 3)))
    lo, hi = parallel.shard_bounds(4)
    loss = orc.mse_loss(orc.unet2_forward(x[lo:hi], params), t[lo:hi])
    loss.backward()
    flat_g = torch.cat([v.grad.reshape(-1) for v in leaves])
    scale = parallel.allreduce_gradients(flat_g)
    out['scale'] = scale
    if rank == 0:
        for v in leaves:
            v.grad = None
        full = orc.mse_loss(orc.unet2_forward(x, params), t)
        full.backward()
        ref = torch.cat([v.grad.reshape(-1) for v in leaves])
        out['dp_err'] = float((flat_g * scale - ref).abs().max() / ref.abs().max())

    # 4. Model.compile() broadcasts rank 0's initial weights (host-side graph + flat buffer only, no kernels)
    from DLWP.keras import backend
    backend.set_device('cpu')
    np.random.seed(100 + rank)                      # different initial weights per rank on purpose
    from DLWP.model.cs_unet import build_cs_model
    model = build_cs_model((6, 8, 8, 3), 3, 'unet2', base_filter_number=4)
    model.compile(optimizer='adam', loss='mse')
    w = model._flat_params.clone()
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    out['compile_bcast_ok'] = bool(torch.equal(gathered[0], gathered[1])) and model._world == world
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0]['bcast_ok'] and ret[1]['bcast_ok']
    assert ret[0]['shard'] == (0, 4) and ret[1]['shard'] == (4, 7)
    assert ret[0]['scale'] == 0.5 and ret[1]['scale'] == 0.5
    assert ret[0]['dp_err'] < 1e-12
    assert ret[0]['compile_bcast_ok'] and ret[1]['compile_bcast_ok']


def test_single_process_defaults():
    import sys
    from DLWP import parallel
    assert parallel.world() == (0, 1)
    g = torch.ones(4)
    assert parallel.allreduce_gradients(g) == 1.0 and torch.equal(g, torch.ones(4))
    assert parallel.shard_bounds(10) == (0, 10)
    assert parallel.shard_bounds(10, 1, 4) == (3, 6)
