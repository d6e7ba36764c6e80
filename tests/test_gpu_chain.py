"""
GPU tests of the multi-layer persistent convolution chain (csrc/conv_chain.hip, dlwpcs_conv_chain_fwd): the consecutive fused
convolutions of a forward pass -- pad -> CubeSphereConv2D -> ReLU [-> AveragePooling3D as a second output], reference
Azure/train_cs.py:277-305 over DLWP/custom.py:921-1002,1198-1308 -- as ONE launch whose phases are separated by sample-group
barriers.  Every phase runs the code of the per-layer kernel on the same tiles in the same order, so the bar is BIT-IDENTITY with
the layer-by-layer path (which tests/test_gpu_parity.py, test_gpu_bf16.py and test_gpu_fullsize.py pin to the fp64 oracle):
inference and training, eager and hipGraph-replayed, batch sizes that give 8 / 4 / 2 / 1 sample groups, the bounded-spin abort path,
and two processes sharing the GPU with chains forced on (must terminate: a result or the reported timeout, never a hang).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def _build(N, cin, cout, base, chain):
    from DLWP.keras import Input, Model, backend
    from DLWP.model.cs_unet import CubeSphereNet
    backend.set_compute_dtype('bfloat16')
    try:
        np.random.seed(11)
        net = CubeSphereNet(base_filter_number=base, output_channels=cout)
        inp = Input(shape=(6, N, N, cin), name='main_input')
        model = Model(inputs=inp, outputs=net.unet2(inp))
    finally:
        backend.set_compute_dtype('float32')
    model.use_chain = chain
    model.compile(optimizer='adam', loss='mse', metrics=['mae'])
    return model


def _launch_tags(fn):
    """kernel tags of the library launches `fn` issues (the library's launch profiler)"""
    import ctypes
    from DLWP import _native as nat
    lib = nat.lib()
    lib.dlwpcs_prof_reset()
    lib.dlwpcs_prof_enable(1)
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        lib.dlwpcs_prof_enable(0)
    tags = []
    tag = ctypes.create_string_buffer(160)
    ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    for i in range(lib.dlwpcs_prof_count()):
        nat.check(lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), 'prof_get')
        tags.append(tag.value.decode())
    lib.dlwpcs_prof_reset()
    return tags


@pytest.mark.parametrize('B,N,cin,base', [(8, 48, 14, 32), (32, 48, 14, 32), (4, 24, 8, 16), (6, 16, 16, 8), (3, 16, 8, 8), (1, 24, 26, 32)])
def test_forward_chain_is_bit_identical_to_the_layer_by_layer_path(B, N, cin, base):
    """unet2 forward, bf16: runs of same-tiling 3x3 convolutions as chain launches (8 / 8 / 4 / 2 / 1 / 1 sample groups) against ten
    per-layer launches.  The first two cases are BASELINE config 3's geometry (chains of the 32- and the 64-channel tiling, a pooled
    second output inside a chain)."""
    from DLWP import ops
    rng = np.random.default_rng(B * 100 + N)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, cin)), dtype=torch.float32, device=_dev()).to(torch.bfloat16)
    ref_model = _build(N, cin, cin, base, chain=False)
    w0 = ref_model.get_weights()
    y_ref = ref_model.predict_on_device(x).float().cpu().numpy()
    model = _build(N, cin, cin, base, chain=True)
    model.set_weights(w0)
    tags = _launch_tags(lambda: model.predict_on_device(x))
    # (narrow test models: a pooled second output needs 32-channel output tiles, such layers keep their own launch)
    n_chain = tags.count('conv_chain_kernel')
    left = [t for t in tags if t.startswith('conv_mfma_ws_kernel<unsigned short, 3,')]
    assert n_chain >= 1 and n_chain + len(left) < 10, tags
    if (N, cin, base) == (48, 14, 32):
        assert n_chain == 1 and len(left) == 0, tags          # all ten 3x3 layers: one launch
    y = model.predict_on_device(x).float().cpu().numpy()
    ops.chain_check()
    assert np.isfinite(y_ref).all() and np.abs(y_ref).max() > 0
    assert np.array_equal(y, y_ref)
    # ... and again (the barrier words went back to zero), and with another batch through the same model
    y2 = model.predict_on_device(x).float().cpu().numpy()
    assert np.array_equal(y2, y_ref)
    ops.chain_check()


@pytest.mark.parametrize('graphs', [False, True])
def test_training_steps_with_the_forward_chain_are_bit_identical(graphs):
    """three Adam steps (bf16, pre-masked gradients, batched weight gradient, fused optimizer): parameters and losses bitwise equal
    with the forward pass as a chain and as ten launches, eager and hipGraph-replayed; the chain is one kernel node of the step graph"""
    B, N, cin, base = 8, 24, 8, 16
    rng = np.random.default_rng(7)
    x = torch.tensor(rng.standard_normal((B, 6, N, N, cin)), dtype=torch.float32, device=_dev()).to(torch.bfloat16)
    t = torch.tensor(rng.standard_normal((B, 6, N, N, cin)), dtype=torch.float32, device=_dev())
    out = []
    w0 = None
    for chain in (False, True):
        m = _build(N, cin, cin, base, chain)
        m.use_graphs = graphs
        if w0 is None:
            w0 = m.get_weights()
        else:
            m.set_weights(w0)
        losses = [m.train_on_device_batch([x], [t]).clone() for _ in range(4)]
        torch.cuda.synchronize()
        out.append((m._flat_params.detach().cpu().numpy().copy(), torch.stack(losses).cpu().numpy()))
    from DLWP import ops
    ops.chain_check()
    assert np.isfinite(out[0][1]).all()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


def test_rollout_with_chains_equals_the_layer_by_layer_rollout():
    """config-5 style: C96-like rollout state that stays on the device (here N = 48, 26 channels, padded state), 6 forward passes"""
    B, N, C = 8, 48, 26
    rng = np.random.default_rng(9)
    x = rng.standard_normal((B, 6, N, N, C)).astype(np.float32)
    series = []
    w0 = None
    for chain in (False, True):
        m = _build(N, C, C, 32, chain)
        if w0 is None:
            w0 = m.get_weights()
        else:
            m.set_weights(w0)
        out = np.empty((6 * 1, B, 6, N, N, C), dtype=np.float32)
        m.rollout_on_device(x, 6, 1, out)
        series.append(out)
    assert np.isfinite(series[0]).all()
    assert np.array_equal(series[0], series[1])


def test_chain_abort_path_is_reported_not_hung():
    """DLWPCS_CHAIN_SPIN=1: every barrier wait gives up at once -- the bounded-spin exit that a shared GPU would take.  The launch
    must END and the abort word must be reported (NativeError from chain_check, which also re-arms the barrier words)."""
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
sys.path.insert(0, %r)
import numpy as np, torch
sys.path.insert(0, os.path.join(%r, 'tests')); from test_gpu_chain import _build
from DLWP import ops, _native as nat
dev = torch.device('cuda', 0)
rng = np.random.default_rng(1)
x = torch.tensor(rng.standard_normal((8, 6, 24, 24, 8)), dtype=torch.float32, device=dev).to(torch.bfloat16)
m = _build(24, 8, 8, 16, True)
y = m.predict_on_device(x)
torch.cuda.synchronize()
try:
    ops.chain_check()
    print('NO_ABORT')            # (possible in principle: nobody had to wait longer than one poll)
except nat.NativeError as e:
    assert 'timed out' in str(e)
    print('ABORT_REPORTED')
''' % (ROOT, ROOT, ROOT)
    env = dict(os.environ, DLWPCS_CHAIN_SPIN='1')
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and ('ABORT_REPORTED' in out.stdout or 'NO_ABORT' in out.stdout), (out.stdout[-500:], out.stderr[-2000:])
    assert 'ABORT_REPORTED' in out.stdout, 'with a spin limit of one poll some barrier wait was expected to give up'


def test_two_processes_sharing_the_gpu_terminate_with_chains_forced_on():
    """Two independent processes run chained forward passes on the SAME GPU at the same time.  A persistent launch needs its 256
    workgroups co-resident; when two such launches interleave on the CUs neither may complete its barriers.  Every spin is bounded:
    each process must terminate, every pass either bit-identical to the layer-by-layer reference or reported as timed out."""
    code = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
sys.path.insert(0, %r)
import numpy as np, torch
sys.path.insert(0, os.path.join(%r, 'tests')); from test_gpu_chain import _build
from DLWP import ops, _native as nat
dev = torch.device('cuda', 0)
rng = np.random.default_rng(1)
x = torch.tensor(rng.standard_normal((8, 6, 48, 48, 14)), dtype=torch.float32, device=dev).to(torch.bfloat16)
ref = _build(48, 14, 14, 32, False)
w0 = ref.get_weights()
y_ref = ref.predict_on_device(x).float().cpu().numpy()
m = _build(48, 14, 14, 32, True)
m.set_weights(w0)
m.predict_on_device(x); torch.cuda.synchronize()
ok = bad = 0
t0 = time.time()
while time.time() - t0 < 6.0:
    ys = [m.predict_on_device(x, repack=False) for _ in range(20)]
    y = ys[-1].float().cpu().numpy()
    try:
        ops.chain_check()
        assert np.array_equal(y, y_ref)
        ok += 1
    except nat.NativeError:
        bad += 1
print('DONE ok=%%d timed_out=%%d' %% (ok, bad))
''' % (ROOT, ROOT, ROOT)
    env = dict(os.environ, DLWPCS_CHAIN_SPIN='40000')          # ~0.1 s per barrier wait at most
    procs = [subprocess.Popen([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for _ in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError('a process with chained launches did not terminate while sharing the GPU')
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and 'DONE' in so, (so[-300:], se[-2000:])
    print(' | '.join(o[0].strip().splitlines()[-1] for o in outs))


def test_barrier_free_flow_form_is_bit_identical():
    """DLWPCS_CHAIN_FLOW=1: the chain without barriers -- per-(phase, sample) completion counters, tiles dealt sample-major, table
    entries and dependency flags fetched one tile ahead.  Slower than everything else but the same bits."""
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'dlwp-cs_amd'))
sys.path.insert(0, %r)
import numpy as np, torch
sys.path.insert(0, os.path.join(%r, 'tests')); from test_gpu_chain import _build
from DLWP import ops
dev = torch.device('cuda', 0)
rng = np.random.default_rng(3)
for B, N, cin, base in ((8, 48, 14, 32), (6, 24, 8, 16)):
    x = torch.tensor(rng.standard_normal((B, 6, N, N, cin)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    ref = _build(N, cin, cin, base, False)
    y_ref = ref.predict_on_device(x).float().cpu().numpy()
    m = _build(N, cin, cin, base, True)
    m.set_weights(ref.get_weights())
    for _ in range(3):
        y = m.predict_on_device(x).float().cpu().numpy()
        ops.chain_check()
        assert np.array_equal(y, y_ref)
print('FLOW_OK')
''' % (ROOT, ROOT, ROOT)
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, DLWPCS_CHAIN_FLOW='1'))
    assert out.returncode == 0 and 'FLOW_OK' in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
