"""
The data-parallel code path of bench.py (one process per GPU, flat-gradient all-reduce between the two hipGraphs, MAX-over-
ranks timing, one JSON line from rank 0) exercised end to end with world_size 2 on ONE GPU: both ranks share cuda:0 and
the process group is gloo (host-staged all-reduce) instead of RCCL.  Guards against rank-asymmetric collectives (a hang).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, DLWPCS_BENCH_BACKEND='gloo', DLWPCS_BENCH_SHARE_GPU='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4',
           '--warmup', '3', '--batch', '4', '--face', '16', '--base', '8', '--channels', '6', '--min-block-s', '0.05',
           '--blocks', '3']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['config']['global_batch'] == 8 and r['value'] > 0
    assert r['scaling'] == 'weak' and 'roofline' in r and 'cpu_baseline' not in r
    ex = r['exchange']
    assert ex['rccl_ranks'] == 2 and ex['backend'] == 'gloo' and ex['allreduce_us'] > 0 and ex['bytes'] > 0
    assert ex['buckets'] in (1, 2) and set(ex['bucket_trials_ms_per_step']) == {'1', '2'}
    # the line says which collective served and that the replicas were identical after the warm-up steps (gloo here: torch's)
    assert ex['replicas_identical_after_warmup'] is True and ex['collective'].startswith('torch.distributed.all_reduce')
    assert ex['native_comm_requested'] is True
    assert ex['exposed_us'] is not None and ex['ms_per_step_without_exchange'] > 0
    assert len(r['timing']['block_ms_per_step']) == 3 and r['ms_per_step'] > 0


def test_bench_plain_invocation_spawns_its_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a torchrun wrapper (how the driver calls it): bench.py starts the ranks itself"""
    env = dict(os.environ, DLWPCS_BENCH_BACKEND='gloo', DLWPCS_BENCH_SHARE_GPU='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '3', '--batch', '4',
           '--face', '16', '--base', '8', '--channels', '6', '--min-block-s', '0.05', '--blocks', '3', '--no-roofline']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['config']['global_batch'] == 8 and r['value'] > 0
    assert r['exchange']['rccl_ranks'] == 2
    assert r['device_self_check']['device_busy_fraction'] > 0


def test_rccl_executes_on_one_gpu():
    """RCCL itself (backend 'nccl'), world size 1: the flat gradient buffer goes through parallel.allreduce_gradients and
    the broadcast of the initial parameters -- the collectives of the data-parallel step -- in a process of their own."""
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch, torch.distributed as dist
from DLWP import parallel
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1, device_id=torch.device('cuda', 0))
assert dist.get_backend() == 'nccl'
g = torch.arange(673628, dtype=torch.float32, device='cuda')
ref = g.clone()
os.environ['DLWPCS_EXCHANGE_FORCE'] = '1'
# torch's all_reduce is the default exchange; the library-owned communicator (dlwpcs_comm_*) is opt-in ...
assert parallel.native_comm() is None and 'not requested' in parallel.native_comm_status()
parallel.allreduce_gradients(g)
torch.cuda.synchronize()
assert torch.equal(g, ref)
# ... created over the torch group once every rank asked for it and its test sum came back right; its all-reduce runs on the
# CURRENT stream ...
parallel.enable_native_comm(True)
assert parallel.native_comm() is not None and 'test sum verified' in parallel.native_comm_status()
scale = parallel.allreduce_gradients(g)
parallel.broadcast_parameters(g)
torch.cuda.synchronize()
assert scale == 1.0 and torch.equal(g, ref)
# ... so that a capture holds it as ONE kernel node on the capturing stream: no event / wait nodes of a side stream around it
gr = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(gr, capture_error_mode='thread_local'):
    parallel.allreduce_gradients(g)
import ctypes
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
n = ctypes.c_size_t(0)
assert hip.hipGraphGetNodes(ctypes.c_void_p(gr.raw_cuda_graph()), None, ctypes.byref(n)) == 0
nodes = (ctypes.c_void_p * n.value)()
hip.hipGraphGetNodes(ctypes.c_void_p(gr.raw_cuda_graph()), nodes, ctypes.byref(n))
kinds = []
for nd in nodes:
    t = ctypes.c_int(-1)
    hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
    kinds.append(t.value)
assert all(k == 0 for k in kinds), kinds        # kernel nodes only (none at all here: RCCL makes an in-place sum over ONE rank a no-op)
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
assert torch.equal(g, ref)
del gr
# torch's own all_reduce serves again when the request is withdrawn; the environment variable asks like the call does
parallel.enable_native_comm(False)
assert parallel.native_comm() is None
parallel.allreduce_gradients(g)
torch.cuda.synchronize()
assert torch.equal(g, ref)
parallel.enable_native_comm(None)
os.environ['DLWPCS_NATIVE_RCCL'] = '1'
c1 = parallel.native_comm()
assert c1 is not None
# a first call inside a capture of a fresh state creates nothing (the agreement reads flags back) and decides nothing
parallel.native_comm_release()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode='thread_local'):
    assert parallel.native_comm() is None
    parallel.allreduce_gradients(g)
gr.replay()
torch.cuda.synchronize()
assert torch.equal(g, ref)
del gr
assert parallel.native_comm() is not None
# the communicator belongs to ONE process group: destroy + init again -> the stale one is released, a fresh one is created
old_group = parallel._native['group']
dist.destroy_process_group()
assert parallel.native_comm() is None
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29534', rank=0, world_size=1, device_id=torch.device('cuda', 0))
c2 = parallel.native_comm()
assert c2 is not None and parallel._native['group'] is not old_group
parallel.allreduce_gradients(g)
torch.cuda.synchronize()
assert torch.equal(g, ref)
os.environ.pop('DLWPCS_NATIVE_RCCL')
os.environ.pop('DLWPCS_EXCHANGE_FORCE')
# the captured form of the step: all-reduce between two graph replays
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        dist.all_reduce(g)
torch.cuda.synchronize()
assert torch.equal(g, ref)
from DLWP import parallel as _par
_par.native_comm_release()
dist.destroy_process_group()
print('RCCL_OK')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'RCCL_OK' in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_bucketed_exchange_runs_over_rccl_on_one_gpu():
    """The two-bucket step with REAL RCCL collectives (backend 'nccl', one rank, DLWPCS_EXCHANGE_FORCE=1): the first bucket's async
    all-reduce is started between the two backward graphs, the second behind them, the optimizer graph waits for both handles.
    One rank sums to itself, so the parameters must equal the plain single-process run bit for bit (fp32) -- what is exercised is
    RCCL's stream against the hipGraph replays."""
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import numpy as np, torch, torch.distributed as dist
from DLWP.keras import backend
from DLWP.model.cs_unet import build_cs_model
torch.cuda.set_device(0)
backend.set_device('cuda:0')
dev = torch.device('cuda', 0)
rng = np.random.default_rng(2)
x = torch.tensor(rng.standard_normal((4, 6, 8, 8, 4)), dtype=torch.float32, device=dev)
t = torch.tensor(rng.standard_normal((4, 6, 8, 8, 4)), dtype=torch.float32, device=dev)
def train(buckets, w0=None):
    np.random.seed(3)
    m = build_cs_model((6, 8, 8, 4), 4, 'unet2', base_filter_number=4)
    m.exchange_buckets = buckets
    m.compile(optimizer='adam', loss='mse')
    if w0 is not None:
        m.set_weights(w0)
    w_init = m.get_weights()
    for _ in range(5):
        m.train_on_device_batch([x], [t])
    torch.cuda.synchronize()
    return w_init, m._flat_params.detach().cpu().numpy().copy(), m
w0, plain, _ = train(1)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29541', rank=0, world_size=1, device_id=dev)
os.environ['DLWPCS_EXCHANGE_FORCE'] = '1'
from DLWP import parallel as _par
_, one_t, _m = train(1, w0)                       # default exchange: torch.distributed.all_reduce
assert _par.native_comm() is None and np.array_equal(plain, one_t)
_par.enable_native_comm(True)                     # opt-in: the library-owned communicator
_, one, m1 = train(1, w0)
assert _par.native_comm() is not None
_, two, m2 = train(2, w0)
g = next(iter(m2._graphs.values()))
assert g['bwd_b'] is not None and g['update'] is not None and m2._did_split
assert np.array_equal(plain, one), np.abs(plain - one).max()
assert np.abs(plain - two).max() <= 1e-6 * np.abs(plain).max()
from DLWP import parallel as _par
_par.native_comm_release()
dist.destroy_process_group()
print('RCCL_BUCKETS_OK')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'RCCL_BUCKETS_OK' in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_data_parallel_step_is_one_graph_with_the_exchange_captured():
    """Round 4: the data-parallel step = the world-1 step with its last launch split at the exchange -- local reduction, all-reduce
    CAPTURED inside the step's hipGraph, ONE launch that applies the update (scale + Adam + gradient clear + packed operands + loss
    tail, dlwpcs_wgrad_batch_apply).  Exercised with real RCCL on one rank (DLWPCS_EXCHANGE_FORCE=1): one graph per step, and --
    one rank sums to itself -- parameters BITWISE equal to the plain world-1 run, in fp32 and in bf16 (where the fused reduction +
    Adam of the plain step and reduction | apply of this one must produce the same bits), also for the two-graph fallback
    (DLWPCS_DP_ONE_GRAPH=0) and across a learning-rate change."""
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import numpy as np, torch, torch.distributed as dist
from DLWP.keras import backend, mixed_precision
from DLWP.model.cs_unet import build_cs_model
torch.cuda.set_device(0)
backend.set_device('cuda:0')
dev = torch.device('cuda', 0)
rng = np.random.default_rng(2)
def data(N, C):
    return (torch.tensor(rng.standard_normal((4, 6, N, N, C)), dtype=torch.float32, device=dev),
            torch.tensor(rng.standard_normal((4, 6, N, N, C)), dtype=torch.float32, device=dev))
def train(policy, N, C, base, x, t, w0=None):
    np.random.seed(3)
    mixed_precision.set_policy(policy)
    try:
        m = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=base)
    finally:
        mixed_precision.set_policy('float32')
    m.compile(optimizer='adam', loss='mse', metrics=['mae'])
    if w0 is not None:
        m.set_weights(w0)
    w_init = m.get_weights()
    losses = []
    xx = x.to(torch.bfloat16) if policy != 'float32' else x
    for i in range(6):
        if i == 4:
            m.optimizer.lr = 3e-4
        losses.append(m.train_on_device_batch([xx], [t]).clone())
    torch.cuda.synchronize()
    return w_init, m._flat_params.detach().cpu().numpy().copy(), torch.stack(losses).cpu().numpy(), m
cases = [('float32', 8, 4, 4), ('mixed_bfloat16', 16, 8, 8)]
plain = {}
for pol, N, C, base in cases:
    x, t = data(N, C)
    plain[pol] = (x, t) + train(pol, N, C, base, x, t)[:3]
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29547', rank=0, world_size=1, device_id=dev)
os.environ['DLWPCS_EXCHANGE_FORCE'] = '1'
from DLWP import parallel as _par
for pol, N, C, base in cases:
    x, t, w0, ref, lref = plain[pol]
    for one_graph, native in (('1', True), ('1', False), ('0', True)):
        os.environ['DLWPCS_DP_ONE_GRAPH'] = one_graph
        _par.enable_native_comm(native)           # (the library's communicator on the compute stream | torch's all_reduce)
        _, got, lgot, m = train(pol, N, C, base, x, t, w0)
        assert (_par.native_comm() is not None) == native
        g = next(iter(m._graphs.values()))
        assert (g['update'] is None) == (one_graph == '1'), (pol, one_graph)
        assert np.array_equal(ref, got), (pol, one_graph, np.abs(ref - got).max())
        assert np.array_equal(lref, lgot), (pol, one_graph)
        if pol != 'float32':
            assert m._wb_done and len(m._wb_done) == 11           # the apply launch covered every layer
from DLWP import parallel as _par
_par.native_comm_release()
dist.destroy_process_group()
print('DP_ONE_GRAPH_OK')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'DP_ONE_GRAPH_OK' in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_roofline_times_the_launches_inside_the_replayed_graph():
    """bench.py's `roofline.avg_launch_us` comes from HIP events captured as external event-record nodes of the step graph
    (csrc/prof.cpp: the form the timed region runs, the one rocprofv3 sees), with the eager figure beside it."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '3', '--batch', '4', '--face', '16',
           '--base', '8', '--channels', '6', '--min-block-s', '0.05', '--blocks', '3', '--no-pmc', '--no-companion',
           '--no-configs', '--no-cpu-baseline', '--no-dp-form']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    rf = r['roofline']
    assert rf['launch_time_from'].startswith('HIP events as external event-record nodes'), rf['launch_time_from']
    assert rf['avg_launch_us'] > 0 and rf['avg_launch_us_eager'] > 0
    # same kernels, same work: the two forms of timing agree within a factor (tiny launches: the eager ones carry host gaps)
    assert 0.2 < rf['avg_launch_us'] / rf['avg_launch_us_eager'] < 5.0
    timed = [k for k, v in rf['per_kernel'].items() if 'avg_us_eager' in v]
    assert len(timed) >= 3, rf['per_kernel'].keys()
