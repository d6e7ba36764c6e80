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
    assert len(r['timing']['block_ms_per_step']) == 3 and r['ms_per_step'] > 0
