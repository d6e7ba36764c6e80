#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X-native DLWP-CS engine.

Metric (BASELINE.json): cubed-sphere samples/sec, forward + backward (+ Adam step), 6x48x48 U-Net `unet2` with 7
variables x 2 time steps = 14 input/output channels (BASELINE config 3: geometry AND dtype -- bf16 compute with fp32
master weights), batch 32 per GPU, synthetic data, random-init weights.  One "step" = one optimisation step (fwd + bwd +
gradient all-reduce when N > 1 + Adam) on one batch already resident in HBM.  One process per GPU; N > 1 is launched by
torch.distributed.run (RCCL).  At N = 1 the same workload is also measured in the exact-fp32 mode (the 1e-5 parity mode)
and reported under the key "f32".

Other workloads (same JSON contract, parity-test configurations of BASELINE.json timed by the driver's clock when asked):
  --workload encoder6 --channels 7 --dtype f32     BASELINE config 2 (6-layer encoder, 7 variables, fp32)
  --workload rollout                               BASELINE config 5 (unet2 C96, 26 channels, bf16, 40-step rollout = 20
                                                   forwards with the state in HBM; one GPU = one independent replica)

Timing: W untimed warm-up steps, then the K steps the contract names are timed between barrier + synchronize pairs; because
K steps of a ~1 ms step are a ~20 ms window, the run continues with timed BLOCKS (>= 5 blocks of >= 0.5 s each, every one
bracketed the same way, MAX over ranks) and `value` / `ms_per_step` are the MEDIAN block; the contract's K-step window is
reported beside it (`k_step_window_ms_per_step`).

Prints ONE JSON line on rank 0: value = whole-job samples/s, plus
  roofline     -- dominant convolution kernel: algorithmic FLOPs and bytes per launch / HIP-event time per launch (library
                  profiler, events on the launch stream: as event-record nodes INSIDE the replayed step graph for training
                  workloads -- `launch_time_from` -- with the eager figure beside it) against the roofline that bounds it;
                  `traffic`, `hbm_gbs` and
                  `mfma_busy` from rocprofv3 PMC passes of THIS command's workload collected live (child processes, one
                  counter group per pass, --kernel-trace only) -- or, when rocprofv3 is not usable, from the newest committed
                  profiles/rNN_*_pmc.json (source and its git blob hash are stated);
  cpu_baseline -- the CPU restatement of the reference path (oracle/, torch-CPU fp32, reference-structured) timed on the
                  host cores of this box at the same batch size (rank 0, N = 1 only).
"""
import argparse
import csv
import ctypes
import glob
import json
import os
import platform
import shutil
import subprocess
import sys
import tempfile
import time

# the host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing needs this (already exported on the boxes)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense (measured 2495)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz
N_SIMD = 1024                      # 256 CUs x 4 SIMDs
N_XCC = 8
PEAK_CLOCK_MHZ = 2400.0
# one rocprofv3 pass per group.  The guide's counter budget per block: TCC holds 4 (FETCH_SIZE costs 3, WRITE_SIZE 2: never both in one
# pass), SQ 8, GRBM 2, independent of each other -- WRITE_SIZE shares its pass with the SQ / GRBM counters (two passes per workload
# instead of three: a pass is ~25 s of process start-up under the profiler)
PMC_GROUPS = (('FETCH_SIZE',), ('WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'))
# peaks MEASURED on an MI355X box with tools/peaks.hip (profiles/r01_peaks.txt): register-resident MFMA chains, streaming
# read / write / copy kernels -- what a perfect kernel reaches on this hardware, next to the datasheet numbers above
MEASURED_PEAKS = {'bf16_mfma_tflops': 2460.0, 'fp32_mfma_tflops': 156.0, 'hbm_read_gbs': 6400.0, 'hbm_write_gbs': 5200.0,
                  'hbm_copy_gbs': 4800.0, 'source': 'tools/peaks.hip, profiles/r01_peaks.txt'}


def conv_plan(workload, c_in, c_out, base):
    """(N_rel, Cin, Cout, k) per conv layer; N_rel = face size divisor."""
    b = base
    enc = [(1, c_in, b, 3), (1, b, b, 3), (2, b, 2 * b, 3), (2, 2 * b, 2 * b, 3), (4, 2 * b, 4 * b, 3),
           (4, 4 * b, 2 * b, 3)]
    if workload == 'encoder6':
        return enc
    return enc + [(2, 4 * b, 2 * b, 3), (2, 2 * b, b, 3), (1, 2 * b, b, 3), (1, b, b, 3), (1, b, c_out, 1)]


# --workload unet2x2: the model every reference CS script trains (Azure/train_cs.py:99-104,391-430): 4 variables x 2 time steps,
# insolation as a fifth input field per time step, two constant fields, integration_steps = 2 with shared weights
X2_VARS, X2_ITS, X2_CONST, X2_STEPS = 4, 2, 2, 2


def x2_channels():
    """(main-input channels, CNN input channels, output channels) of the production model"""
    c_main = (X2_VARS + 1) * X2_ITS
    return c_main, c_main + X2_CONST, X2_VARS * X2_ITS


def flops_per_sample(workload, N, c_in, c_out, base):
    if workload in ('unet2x2', 'rollout_x2'):
        _, ci, co = x2_channels()
        return X2_STEPS * flops_per_sample('unet2', N, ci, co, base)
    wl = 'unet2' if workload == 'rollout' else workload
    return sum(2.0 * 6 * (N // r) ** 2 * k * k * ci * co for (r, ci, co, k) in conv_plan(wl, c_in, c_out, base))


def build_model(workload, N, c_in, c_out, base):
    if workload in ('unet2x2', 'rollout_x2'):
        from DLWP.model.cs_unet import build_cs_model
        c_main, _, co = x2_channels()
        return build_cs_model((6, N, N, c_main), co, 'unet2', base_filter_number=base, integration_steps=X2_STEPS,
                              io_time_steps=X2_ITS, insolation_shape=(X2_ITS, 6, N, N, 1), constants_shape=(6, N, N, X2_CONST))
    from DLWP.model.cs_unet import CubeSphereNet
    from DLWP.keras.layers import Input
    from DLWP.keras.models import Model
    net = CubeSphereNet(c_out, base, 'unet2')
    x = Input(shape=(6, N, N, c_in), name='main_input')
    y = net.encoder6(x) if workload == 'encoder6' else net.unet2(x)
    return Model(inputs=x, outputs=y)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def cpu_baseline(workload, N, c_in, c_out, base, batch, warmup=3, timed=10, budget_s=25.0):
    """Reference-structured CPU port (oracle/cs_oracle.py), torch-CPU fp32, host cores of this box: >= 3 warm-up + >= 10
    timed steps at the benchmark's own batch size (fewer only if the time budget runs out), MEDIAN step time."""
    from oracle import cs_oracle as orc
    cores = os.cpu_count() or 1
    train = workload != 'rollout'
    params = orc.make_unet2_params(c_in, c_out, base=base, seed=1, dtype=torch.float32)
    if workload == 'encoder6':
        params = params[:6]
    leaves = [v.requires_grad_(train) for prm in params for v in prm.values()]
    ms = [torch.zeros_like(v) for v in leaves]
    vs = [torch.zeros_like(v) for v in leaves]
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.standard_normal((batch, 6, N, N, c_in)), dtype=torch.float32)
    fwd = orc.encoder6_forward if workload == 'encoder6' else orc.unet2_forward
    with torch.no_grad():
        tshape = fwd(x[:1], params).shape[1:]
    tgt = torch.tensor(rng.standard_normal((batch,) + tuple(tshape)), dtype=torch.float32)

    def step(t):
        if not train:                       # rollout: one forward pass (a 40-step rollout = 20 of them)
            with torch.no_grad():
                fwd(x, params)
            return
        for v in leaves:
            v.grad = None
        loss = orc.mse_loss(fwd(x, params), tgt)
        loss.backward()
        with torch.no_grad():
            for p, m, v in zip(leaves, ms, vs):
                orc.adam_step(p, p.grad, m, v, t)
    # thread count: a fixed pool of 16 (DLWPCS_CPU_THREADS overrides).  Rounds 1-5 searched {all, 64, 32, 16} host cores with two steps
    # each -- half of this function's wall time -- and 16 or 32 threads won on every box seen (many-core hosts oversubscribe the small
    # per-face convolutions: 256 threads run this step 3-4x slower than 16)
    threads = max(1, min(cores, int(os.environ.get('DLWPCS_CPU_THREADS', '16'))))
    torch.set_num_threads(threads)
    for i in range(warmup):
        step(i + 2)
    times = []
    t_start = time.perf_counter()          # the budget bounds the TIMED steps (tuning and warm-up above are a few steps)
    while len(times) < timed and (len(times) < 3 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        step(len(times) + 2)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    unit = 'samples/s' if train else 'forward-samples/s'
    what = 'fwd+bwd+Adam' if train else 'one forward pass (a 40-step rollout is 20 of them)'
    return {'value': round(batch / med, 3), 'unit': unit, 'cores': threads, 'host_cores': cores, 'cpu_model': _cpu_model(),
            'torch': torch.__version__, 'kind': 'port', 'median_step_s': round(med, 4),
            'sample': '%d warm-up + %d timed steps of batch %d (median), %s, %s, torch-CPU fp32, %d threads (fixed pool; the '
                      'host has %d cores), reference-structured oracle (materialised halo padding, 6 '
                      'per-face conv2d per layer)' % (warmup, len(times), batch, workload, what, threads, cores)}


# ---------------------------------------------------------------------------------------------------------------------- #
# PMC counters (rocprofv3): live child passes, or the newest committed profile
# ---------------------------------------------------------------------------------------------------------------------- #

def _clean_kernel_name(n):
    import re
    return re.sub(r'\(.*\)$', '', n).replace('void ', '').replace('dlwpcs::', '')


PMC_SEPARATOR = 'lds_oob_probe_kernel'     # two launches in a row = boundary between two workloads of one profiler pass


def parse_pmc_dir(d, segments=1):
    """[per kernel name -> {counter: [launches, sum]}] x segments, from one rocprofv3 --pmc output dir.  A pass that profiles several
    workloads (pmc_child with --pmc-plan) separates them by a RUN of >= 2 launches of PMC_SEPARATOR (in dispatch order); segment i
    holds what ran behind the i-th run."""
    rows = []
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r.get('Dispatch_Id') or 0), _clean_kernel_name(r['Kernel_Name']), r['Counter_Name'], float(r['Counter_Value'])))
    rows.sort(key=lambda r: r[0])
    out = [{} for _ in range(segments)]
    seg, run, last_id = -1, 0, None         # (what runs in front of the first separator -- first-use uploads -- belongs to nobody)
    for did, k, cname, val in rows:
        if PMC_SEPARATOR in k:
            if did != last_id:                  # (one row per counter and dispatch)
                run += 1
                last_id = did
                if run == 2:
                    seg += 1
            continue
        if '__amd_rocclr_fillBuffer' in k and run:      # (the probe's own memset node sits between two separator launches)
            continue
        run = 0
        if 0 <= seg < segments:
            a = out[seg].setdefault(k, {}).setdefault(cname, [0, 0.0])
            a[0] += 1
            a[1] += val
    return out


def pmc_record(counters):
    """counter sums of ONE kernel -> per-launch record.  FETCH_SIZE x 2: MI355X_MICROARCH.md (HBM): on gfx950 rocprofv3
    reports exactly half of the bytes of wide coalesced streaming reads; WRITE_SIZE as reported (KB)."""
    rec = {}
    f, w = counters.get('FETCH_SIZE'), counters.get('WRITE_SIZE')
    if f and w and f[0] and w[0]:
        rec['fetch_kb_x2'] = round(2.0 * f[1] / f[0], 1)
        rec['fetch_kb_raw'] = round(f[1] / f[0], 1)             # as the counter reports it (the x 2 is the guide's gfx950 calibration
        rec['traffic_raw'] = int((f[1] / f[0] + w[1] / w[0]) * 1024)   # for wide streaming reads: an estimate, not a measurement)
        rec['write_kb'] = round(w[1] / w[0], 1)
        rec['traffic'] = int((2.0 * f[1] / f[0] + w[1] / w[0]) * 1024)
        rec['launches'] = f[0]
    mb, ga = counters.get('SQ_VALU_MFMA_BUSY_CYCLES'), counters.get('GRBM_GUI_ACTIVE')
    if mb and ga and ga[1] > 0:
        # MfmaUtil of rocprofv3's derived counters = SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs / (GRBM_GUI_ACTIVE x
        # SIMDs).  The CSV holds GRBM_GUI_ACTIVE SUMMED over the 8 XCCs (measured: 22.5 k counts per us of kernel time =
        # 8 x ~2.2 GHz, plus a fixed ~17 k cycles per XCC around every dispatch), the derived metric takes the max over them.
        rec['mfma_busy'] = round(mb[1] / (ga[1] / N_XCC * N_SIMD), 4)
        rec['mfma_busy_cycles'] = round(mb[1] / mb[0], 1)          # per launch, summed over the 1024 SIMDs
        rec['gui_active_cycles'] = round(ga[1] / ga[0] / N_XCC, 1)
    sb = counters.get('SQ_BUSY_CYCLES')
    if sb and ga and ga[1] > 0:
        rec['sq_busy_cycles'] = round(sb[1] / sb[0], 1)
    return rec


_PMC_CACHE = {}


def _pmc_key(args, dtype):
    return (args.workload, dtype, int(args.face), int(args.channels), int(args.batch), int(args.base), int(getattr(args, 'rollout_steps', 0)))


def collect_pmc_plan(plan, timeout_s=900, groups=None):
    """ONE rocprofv3 process per counter group for ALL workloads of `plan` ([(args, dtype)]; round 5 ran one process per group AND
    workload: ~25 s of start-up under the profiler each, 8 of them).  Fills _PMC_CACHE[key] = (records | None, source)."""
    plan = [(a, dt) for a, dt in plan if _pmc_key(a, dt) not in _PMC_CACHE]
    if not plan:
        return
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    reason = None
    if exe is None:
        reason = 'rocprofv3 not found'
    elif os.environ.get('ROCP_TOOL_LIBRARIES') or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        reason = 'already running under a profiler'
    if reason:
        for a, dt in plan:
            _PMC_CACHE[_pmc_key(a, dt)] = (None, reason)
        return
    spec = json.dumps([{'workload': a.workload, 'dtype': dt, 'batch': a.batch, 'face': a.face, 'channels': a.channels, 'base': a.base,
                        'rollout_steps': getattr(a, 'rollout_steps', 40)} for a, dt in plan])
    merged = [{} for _ in plan]
    tmp = tempfile.mkdtemp(prefix='dlwpcs_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    groups_in = groups
    groups = list(groups or PMC_GROUPS)
    errors = []
    try:
        i = 0
        while groups:
            group = groups.pop(0)
            d = os.path.join(tmp, 'pass%d' % i)
            i += 1
            cmd = [exe, '--kernel-trace', '--pmc'] + list(group) + ['-d', d, '-o', 'p', '--output-format', 'csv', '--',
                   sys.executable, os.path.abspath(__file__), '--pmc-child', '--pmc-plan', spec]
            err = None
            try:
                r = subprocess.run(cmd, env=env, cwd='/tmp', capture_output=True, text=True, timeout=timeout_s)
                if r.returncode != 0:
                    err = 'rocprofv3 pass (%s) failed (rc %d): %s' % (' '.join(group), r.returncode, (r.stderr or '')[-200:].replace('\n', ' '))
            except subprocess.TimeoutExpired:
                err = 'rocprofv3 pass (%s) timed out' % ' '.join(group)
            if err is not None:
                # a group of several hardware blocks that the profiler refuses is retried block by block; a failed pass costs its
                # own counters only (the kernels it would have covered are listed in roofline.pmc_missing)
                if len(group) > 1 and 'WRITE_SIZE' in group:
                    groups = [('WRITE_SIZE',), tuple(c for c in group if c != 'WRITE_SIZE')] + groups
                else:
                    errors.append(err)
                continue
            parsed = parse_pmc_dir(d, segments=2 * len(plan))[1::2]      # (even segments: the workloads' first-use steps)
            if not any(parsed):
                errors.append('rocprofv3 pass (%s): no counter rows behind a separator (rc 0); child said: %s'
                              % (' '.join(group), ((r.stderr or '') + (r.stdout or ''))[-400:].replace('\n', ' ')))
            for seg, per_kernel in zip(merged, parsed):
                for k, cs in per_kernel.items():
                    seg.setdefault(k, {}).update(cs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    src = 'live rocprofv3 --kernel-trace --pmc passes (%s), ONE process per counter group for %d workloads, 3 eager steps each%s' % (
        ' | '.join(' '.join(g) for g in (groups_in or PMC_GROUPS)), len(plan), ('; FAILED: ' + '; '.join(errors)) if errors else '')
    for (a, dt), seg in zip(plan, merged):
        recs = {k: pmc_record(v) for k, v in seg.items()}
        recs = {k: v for k, v in recs.items() if v}
        _PMC_CACHE[_pmc_key(a, dt)] = (recs or None, src if recs else ('; '.join(errors) or 'no counter records'))


def collect_pmc_live(args, dtype, timeout_s=900, groups=None):
    """({kernel: record}, source string) or (None, reason) for this workload: out of the run's shared passes (collect_pmc_plan,
    called by main() for every workload of the line), else a pass of its own."""
    key = _pmc_key(args, dtype)
    if key not in _PMC_CACHE:
        collect_pmc_plan([(args, dtype)], timeout_s=timeout_s, groups=groups)
    return _PMC_CACHE[key]


def newest_committed_pmc(workload, dtype):
    """Newest profiles/rNN_*_pmc.json holding this workload / dtype -> ({kernel: record}, 'path @ git blob')."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_*_pmc.json'))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        recs = doc.get('%s/%s' % (workload, dtype))
        if recs:
            best = (path, recs)
    if best is None:
        return None, 'no committed profiles/rNN_*_pmc.json for %s/%s' % (workload, dtype)
    path, recs = best
    try:
        blob = subprocess.run(['git', 'hash-object', path], capture_output=True, text=True, cwd=ROOT).stdout.strip()[:12]
    except OSError:
        blob = ''
    return recs, '%s (git blob %s)' % (os.path.relpath(path, ROOT), blob or 'n/a')


def pmc_child(args):
    """Body of one rocprofv3 pass: per workload of --pmc-plan a separator (two probe launches) and 3 eager steps (no graphs)."""
    import copy
    import gc
    from DLWP import _native as nat
    from DLWP.keras import backend
    backend.set_device('cuda:0')
    dev = torch.device('cuda', 0)
    nat.lds_oob_reads_zero(dev)            # (the engine's own one-time probe launch: before the first separator)
    plan = json.loads(args.pmc_plan) if args.pmc_plan else [{'workload': args.workload, 'dtype': args.dtype, 'batch': args.batch,
                                                             'face': args.face, 'channels': args.channels, 'base': args.base,
                                                             'rollout_steps': args.rollout_steps}]
    nz = torch.zeros(1, dtype=torch.int32, device=dev)
    for ent in plan:
        a = copy.copy(args)
        a.workload, a.dtype, a.batch, a.face, a.channels, a.base = (ent['workload'], ent['dtype'], ent['batch'], ent['face'],
                                                                    ent['channels'], ent['base'])
        a.rollout_steps = ent.get('rollout_steps', 40)
        state = prepare(a, a.dtype, rank=0)
        state['model'].use_graphs = False
        def separator():
            for _ in range(2):
                nat.check(nat.lib().dlwpcs_lds_oob_probe(nz.data_ptr(), torch.cuda.current_stream().cuda_stream), 'separator')
        # two segments per workload: [separator] first-use step (table uploads, workspace growth, packing: nobody's) [separator] 3 steps
        separator()
        run_step(state)
        torch.cuda.synchronize()
        separator()
        for _ in range(3):
            run_step(state)
        torch.cuda.synchronize()
        del state
        gc.collect()
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------- #
# the measured step
# ---------------------------------------------------------------------------------------------------------------------- #

def prepare(args, dtype, rank):
    from DLWP.keras import backend
    N, C, base, B = args.face, args.channels, args.base, args.batch
    backend.set_compute_dtype('bfloat16' if dtype == 'bf16' else 'float32')
    try:
        np.random.seed(1)
        model = build_model(args.workload, N, C, C, base)
    finally:
        backend.set_compute_dtype('float32')
    model.use_graphs = not args.no_graphs
    model.static_batch_buffers = True      # the batch lives in the same HBM tensors every step (inputs resident in HBM)
    adt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    rng = np.random.default_rng(1000 + rank)
    dev = backend.device()
    def rnd(*shape):
        return torch.tensor(rng.standard_normal(shape), dtype=torch.float32, device=dev)
    if args.workload == 'rollout_x2':
        # the production model's forecast as TimeSeriesEstimator.predict runs it (DLWP/model/extensions.py:252-308 in the reference):
        # the whole two-output model applied `seq` times on its own last output, insolation re-injected from an HBM-resident array
        c_main, _, co = x2_channels()
        dx = [rnd(B, 6, N, N, c_main).to(adt), rnd(B, X2_ITS, 6, N, N, 1).to(adt), rnd(B, 6, N, N, X2_CONST).to(adt)]
        seq = max(1, args.rollout_steps // (X2_ITS * X2_STEPS))
        rows = seq * X2_ITS * X2_STEPS + B + 4
        return {'model': model, 'dx': dx, 'dev': dev, 'train': False, 'x2roll': True, 'seq': seq, 'n_fwd': seq * X2_STEPS,
                'sol': torch.rand((rows, 6, N, N), dtype=torch.float32, device=dev), 'start': np.arange(B, dtype=np.int64) % 4}
    if args.workload == 'unet2x2':
        c_main, _, co = x2_channels()
        # [main_input, solar_1, constants] as the reference's generator yields them (Azure/train_cs.py:191-194,392-396)
        dx = [rnd(B, 6, N, N, c_main).to(adt), rnd(B, X2_ITS, 6, N, N, 1).to(adt), rnd(B, 6, N, N, X2_CONST).to(adt)]
        st = {'model': model, 'dx': dx, 'dev': dev, 'train': True}
        model.compile(optimizer='adam', loss='mse', loss_weights=[1. / X2_STEPS] * X2_STEPS, metrics=['mae'])
        st['dt'] = [rnd(B, 6, N, N, co) for _ in range(X2_STEPS)]
        return st
    dx = [rnd(B, 6, N, N, C).to(adt)]
    st = {'model': model, 'dx': dx, 'dev': dev, 'train': args.workload != 'rollout'}
    if st['train']:
        model.compile(optimizer='adam', loss='mse')
        with torch.no_grad():
            oshape = model.predict_on_device(dx[0][:1]).shape[1:]
        st['dt'] = [rnd(*((B,) + tuple(oshape)))]
    else:
        st['n_fwd'] = args.rollout_steps // 2          # time_dim = 2: one forward pass per two forecast steps
    return st


def run_step(st):
    if st['train']:
        st['model'].train_on_device_batch(st['dx'], st['dt'])
        return
    if st.get('x2roll'):
        st['last'] = st['model'].rollout_with_forcing(st['dx'], st['seq'], insolation=st['sol'], start_index=st['start'], io_time_steps=X2_ITS)
        return
    # one "step" = one 40-step rollout of the batch, state resident in HBM: the chain of forward passes the product's
    # predict_timeseries runs (Model.rollout_passes_on_device: one hipGraph replay per rollout when graphs are on; from the second
    # pass on the state travels with its 26 channels padded to 32 -- zero channels -- per pixel)
    st['last'] = st['model'].rollout_passes_on_device(st['dx'][0], st['n_fwd'])


def roofline_pass(st, steps=3):
    """Eager steps with the library's per-launch HIP-event profiler on; aggregate per kernel name."""
    from DLWP import _native as nat
    lib = nat.lib()
    model = st['model']
    use_graphs = model.use_graphs
    model.use_graphs = False
    lib.dlwpcs_prof_reset()
    lib.dlwpcs_prof_enable(1)
    for _ in range(steps):
        run_step(st)
    torch.cuda.synchronize()
    lib.dlwpcs_prof_enable(0)
    agg = {}
    tag = ctypes.create_string_buffer(160)
    ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    for i in range(lib.dlwpcs_prof_count()):
        nat.check(lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), 'prof_get')
        a = agg.setdefault(tag.value.decode(), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms.value
        a[2] += fl.value
        a[3] += by.value
    lib.dlwpcs_prof_reset()
    model.use_graphs = use_graphs
    return agg


def roofline_pass_graph(st, replays=8):
    """Per-kernel durations INSIDE the replayed step graph: the step is captured once more with the library's profiler on, which
    turns its two events per launch into external event-record nodes of the graph (csrc/prof.cpp); every replay records them
    again.  Returns {kernel: [launches, ms, flops, bytes]} summed over `replays` replays, or None when the form does not apply
    (inference, graphs off) or the runtime does not time such nodes -- the eager pass stands alone then."""
    import gc
    from DLWP import _native as nat
    lib = nat.lib()
    model = st['model']
    if not (st['train'] and model.use_graphs):
        return None
    agg = {}
    tag = ctypes.create_string_buffer(160)
    ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.dlwpcs_prof_reset()
    model._graphs.clear()
    model._seen_batch.clear()
    try:
        lib.dlwpcs_prof_enable(1)
        run_step(st)                                    # eager warm-up of the shape
        run_step(st)                                    # capture + first replay
        torch.cuda.synchronize()
        lib.dlwpcs_prof_enable(0)
        if not model._graphs:
            return None
        n = lib.dlwpcs_prof_count()
        for _ in range(replays):
            run_step(st)
            torch.cuda.synchronize()
            for i in range(n):
                # (records of a discarded trial capture were never recorded by a replay: their query fails -- skipped)
                if lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
                    continue
                name = tag.value.decode()
                if not name.endswith('@graph') or not (ms.value > 0.0):
                    continue
                a = agg.setdefault(name[:-len('@graph')], [0, 0.0, 0.0, 0.0])
                a[0] += 1
                a[1] += ms.value
                a[2] += fl.value
                a[3] += by.value
    except Exception as exc:                            # a runtime without timed external event nodes: keep the eager figures
        sys.stderr.write('roofline_pass_graph: %s: %s\n' % (type(exc).__name__, exc))
        agg = {}
    finally:
        lib.dlwpcs_prof_enable(0)
        torch.cuda.synchronize()
        model._graphs.clear()                           # the graph goes before the events its nodes record
        model._seen_batch.clear()
        gc.collect()
        lib.dlwpcs_prof_reset()
    return agg or None


def allreduce_probe(model, world, reps=20):
    """Duration of the step's one exchange -- all_reduce(SUM) of the flat fp32 gradient buffer -- on its own (MAX over ranks)."""
    if world <= 1 or model._flat_grads is None:
        return None
    buf = torch.zeros_like(model._flat_grads)
    for _ in range(3):
        torch.distributed.all_reduce(buf)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.distributed.all_reduce(buf)
    torch.cuda.synchronize()
    el = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=buf.device)
    torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    return float(el.item()) * 1e6


def replicas_in_sync(model, world):
    """N > 1: after the warm-up steps (eager, captured, replayed -- every one of them exchanged gradients) the replicas' parameters
    must be IDENTICAL on every rank: MAX and MIN over the ranks of two checksums of the flat parameter buffer agree.  A collective
    that summed nothing, summed a subset or ran as a fallback on some ranks fails here, before the timed region.  Every rank calls it."""
    if world <= 1 or model._flat_params is None:
        return None
    from DLWP import parallel as _par
    p = model._flat_params.double()
    cs = torch.stack([p.sum(), p.abs().sum(), (p * torch.arange(1, p.numel() + 1, device=p.device, dtype=torch.float64)).sum()])
    hi, lo = cs.clone(), cs.clone()
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    same = bool(torch.equal(hi, lo))
    native = _par.native_comm() is not None
    asked = bool(_par.NATIVE_COMM)
    info = {'replicas_identical_after_warmup': same,
            'collective': 'dlwpcs_allreduce_f32 (library-owned RCCL communicator, compute stream, a node of the step graph)'
            if native else 'torch.distributed.all_reduce (ProcessGroupNCCL stream)',
            'native_comm_requested': asked, 'native_comm_status': _par.native_comm_status()}
    if not same:
        raise RuntimeError('bench.py: the replicas diverged during warm-up (%s): the gradient exchange is broken; refusing to time it'
                           % info['collective'])
    if torch.distributed.get_backend() == 'nccl' and asked and not native:
        sys.stderr.write('bench.py: WARNING: the library-owned communicator was asked for and is NOT in use (%s); '
                         'torch.distributed.all_reduce serves the timed region\n' % info['native_comm_status'])
    return info


def time_without_exchange(st, timed, per_block):
    """Seconds per step of the SAME step form with the all-reduce calls turned into no-ops (the collective is captured inside
    the step graph, so the graphs are captured anew for this and once more afterwards)."""
    from DLWP import parallel as _par
    model = st['model']

    def recapture():
        model._graphs.clear()
        model._seen_batch.clear()
        for _ in range(4):
            run_step(st)
    _par.SKIP_EXCHANGE_FOR_TIMING = True
    try:
        recapture()
        timed(per_block)
        return float(np.median([timed(per_block) / per_block for _ in range(3)]))
    finally:
        _par.SKIP_EXCHANGE_FOR_TIMING = False
        recapture()


def graph_kernel_nodes(cuda_graph):
    """Kernel nodes of a captured hipGraph (hipGraphGetNodes + hipGraphNodeGetType through ctypes), None where torch does not
    hand out the raw graph."""
    try:
        raw = cuda_graph.raw_cuda_graph()
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
            return None
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
            return None
        kinds = {}
        for nd in nodes:
            t = ctypes.c_int(-1)
            hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
            kinds[t.value] = kinds.get(t.value, 0) + 1
        return {'kernel': kinds.get(0, 0), 'memcpy': kinds.get(1, 0), 'memset': kinds.get(2, 0),
                'other': sum(v for k, v in kinds.items() if k not in (0, 1, 2))}
    except Exception:
        return None


def dp_form_probe(args, dtype, plain_ms):
    """N = 1: the DATA-PARALLEL form of the step on this one GPU -- a process group of one rank over RCCL with
    DLWPCS_EXCHANGE_FORCE=1, so that the step takes exactly the launch sequence it takes at N > 1 (local reduction -> all-reduce,
    captured in the step graph -> ONE launch that applies the update) -- timed with the same blocks, next to the plain step."""
    import socket
    from DLWP import parallel as _par
    own_group = not torch.distributed.is_initialized()
    prev = os.environ.get('DLWPCS_EXCHANGE_FORCE')
    try:
        if own_group:
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            torch.distributed.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                                 device_id=torch.device('cuda', torch.cuda.current_device()))
        os.environ['DLWPCS_EXCHANGE_FORCE'] = '1'
        os.environ['DLWPCS_KEEP_GRAPH'] = '1'               # (the step graph's nodes are counted below)
        st = prepare(args, dtype, 0)
        model = st['model']

        def timed(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                run_step(st)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        for _ in range(5):
            run_step(st)
        est = timed(50) / 50
        per_block = max(50, int(np.ceil(0.3 / max(est, 1e-9))))
        ms = 1e3 * float(np.median([timed(per_block) / per_block for _ in range(3)]))
        g = next(iter(model._graphs.values())) if model._graphs else None
        graphs = 0 if g is None else 1 + (g.get('bwd_b') is not None) + (g.get('update') is not None)
        nodes = None
        if g is not None and graphs == 1:
            nodes = graph_kernel_nodes(g['fwd_bwd'])
        noex = 1e3 * time_without_exchange(st, timed, per_block)
        return {'ms_per_step': round(ms, 4), 'vs_plain_step': round(ms / plain_ms, 4), 'graphs_per_step': graphs,
                'launches': None if nodes is None else nodes['kernel'], 'graph_nodes': nodes,
                'exposed_us': round(1e3 * (ms - noex), 1), 'ms_per_step_without_exchange': round(noex, 4),
                'backend': torch.distributed.get_backend(), 'ranks': torch.distributed.get_world_size(),
                'collective': 'dlwpcs_allreduce_f32 (library-owned RCCL communicator, compute stream)' if _par.native_comm() is not None
                else 'torch.distributed.all_reduce (ProcessGroupNCCL stream)',
                'note': 'the step as it runs at N > 1 (reduction | all-reduce captured in the step graph | one launch: scale + Adam '
                        '+ gradient clear + packed operands + loss tail), here with a one-rank RCCL group: the collective itself '
                        'moves nothing, what is measured is the form of the step'}
    except Exception as exc:
        return {'error': '%s: %s' % (type(exc).__name__, exc)}
    finally:
        os.environ.pop('DLWPCS_KEEP_GRAPH', None)
        if prev is None:
            os.environ.pop('DLWPCS_EXCHANGE_FORCE', None)
        else:
            os.environ['DLWPCS_EXCHANGE_FORCE'] = prev
        if own_group and torch.distributed.is_initialized():
            _par.native_comm_release()
            torch.distributed.destroy_process_group()


def measure(args, dtype, rank, world, with_roofline, with_pmc, pmc_groups=None):
    """Build the workload in `dtype`, warm up, time the contract's K steps and then >= 5 blocks of >= 0.5 s (each bracketed
    by barrier + synchronize, MAX over ranks); optionally the per-launch roofline pass and the PMC passes.  Returns the
    result dict on rank 0, None elsewhere."""
    N, C, base, B = args.face, args.channels, args.base, args.batch
    st = prepare(args, dtype, rank)
    model, dev = st['model'], st['dev']

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            run_step(st)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    for _ in range(max(args.warmup, 3)):      # >= 3: eager warm-up, graph capture, first replay
        run_step(st)
    # N > 1: the exchange as ONE all-reduce behind the backward pass, or in two buckets with the first one in flight during the
    # encoder-side half of the backward pass (Model.exchange_buckets).  Which is faster depends on how RCCL's workgroups share
    # the CUs with the persistent kernels: both are timed here (untimed region, every rank takes the same MAX-over-ranks
    # decision), the faster one is what the timed region below runs; both numbers go into the JSON line.
    bucket_trials = None
    if world > 1 and st['train'] and os.environ.get('DLWPCS_EXCHANGE_BUCKETS') is None:
        bucket_trials = {}
        for nb in (1, 2):
            model.exchange_buckets = nb
            model._graphs.clear()
            model._seen_batch.clear()
            for _ in range(4):
                run_step(st)
            timed(20)
            bucket_trials[nb] = timed(100) / 100
        best = min(bucket_trials, key=bucket_trials.get)
        if best != model.exchange_buckets:
            model.exchange_buckets = best
            model._graphs.clear()
            model._seen_batch.clear()
            for _ in range(4):
                run_step(st)
    sync = replicas_in_sync(model, world) if st['train'] else None
    k_elapsed = timed(args.steps)             # the contract's window: exactly K steps
    est = k_elapsed / args.steps
    per_block = max(args.steps, int(np.ceil(args.min_block_s / max(est, 1e-9))))
    if world > 1:                             # every rank must run the same number of steps per block
        pb = torch.tensor([per_block], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(pb, op=torch.distributed.ReduceOp.MAX)
        per_block = int(pb.item())
    blocks = [timed(per_block) / per_block for _ in range(args.blocks)]
    step_s = float(np.median(blocks))
    # self-check that the timed work ran on the device: the same steps between two HIP events on the compute stream (device
    # clock) against the host clock, and the device the process is bound to
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(per_block):
        run_step(st)
    e1.record()
    barrier()
    wall = time.perf_counter() - t0
    self_check = {'device': torch.cuda.get_device_name(dev), 'arch': getattr(torch.cuda.get_device_properties(dev), 'gcnArchName', ''),
                  'steps': per_block, 'hip_event_ms_per_step': round(e0.elapsed_time(e1) / per_block, 4),
                  'host_wall_ms_per_step': round(1e3 * wall / per_block, 4),
                  'device_busy_fraction': round(min(1.0, e0.elapsed_time(e1) * 1e-3 / wall), 4)}
    try:
        self_check['smi_utilization_percent'] = int(torch.cuda.utilization(dev))
    except Exception as exc:                  # amdsmi not importable on the box: say so instead of reporting 0
        self_check['smi_utilization_percent'] = None
        self_check['smi_note'] = 'torch.cuda.utilization unavailable: %s' % type(exc).__name__
    ar_us = allreduce_probe(model, world) if st['train'] else None
    # how much of the exchange the step actually waits for: the same blocks with the all-reduce calls turned into no-ops
    # (DLWP.parallel.SKIP_EXCHANGE_FOR_TIMING: timing only -- the replicas diverge, so this runs after everything that is reported)
    noex_s = None
    if world > 1 and st['train']:
        noex_s = time_without_exchange(st, timed, per_block)
    # the roofline pass runs eager optimisation steps (gradient all-reduce included): EVERY rank takes part
    agg = roofline_pass(st) if with_roofline else None
    # (N = 1 only: at N > 1 the captured step holds a collective, and a capture that failed on one rank alone would leave the
    # others replaying it -- the eager pass above is the multi-rank form)
    agg_graph = roofline_pass_graph(st) if (with_roofline and world == 1) else None
    if rank != 0:
        return None
    fps = flops_per_sample(args.workload, N, C, C, base)
    prec = ('bf16 activations + bf16 MFMA, fp32 master weights / gradients / Adam' if dtype == 'bf16'
            else 'exact-fp32 MFMA')
    if st['train']:
        metric, unit = 'cubed-sphere samples/sec (fwd+bwd)', 'samples/s'
        what = '%s C%d: x (%d,6,%d,%d,%d) per GPU, %d out channels, base %d, MSE + Adam, fwd+bwd+update, %s' % (
            args.workload, N, B, N, N, C, C if args.workload == 'unet2' else 2 * base, base, prec)
        if args.workload == 'unet2x2':
            c_main, ci, co = x2_channels()
            what = ('unet2x2 C%d (the reference scripts\' production model, Azure/train_cs.py:99-104,391-430): integration_steps = %d with '
                    'shared weights, inputs main (%d,6,%d,%d,%d) + solar_1 (%d,%d,6,%d,%d,1) + constants (%d,6,%d,%d,%d) per GPU -> the CNN '
                    'sees %d channels, %d outputs of %d channels, base %d, loss_weights [1/%d]*%d, MSE + MAE + Adam, fwd+bwd+update, %s'
                    % (N, X2_STEPS, B, N, N, c_main, B, X2_ITS, N, N, B, N, N, X2_CONST, ci, X2_STEPS, co, base, X2_STEPS, X2_STEPS, prec))
        flops_step = 3 * fps * B
    else:
        metric, unit = 'cubed-sphere rollouts/sec (%d-step, inference)' % args.rollout_steps, 'rollouts/s'
        what = 'unet2 C%d: state (%d,6,%d,%d,%d) per GPU, %d-step rollout = %d forward passes with the state in HBM ' \
               '(independent replicas, no communication), %s' % (N, B, N, N, C, args.rollout_steps, st['n_fwd'], prec)
        flops_step = fps * B * st['n_fwd']
        if st.get('x2roll'):
            what = ('unet2x2 C%d (the reference scripts\' production model) forecast as TimeSeriesEstimator.predict runs it: batch %d, %d forecast '
                    'steps = %d applications of the two-output model (%d network passes) on its own last output, insolation re-injected '
                    'from an HBM-resident array, state never leaves HBM (Model.rollout_with_forcing), %s'
                    % (N, B, st['seq'] * X2_ITS * X2_STEPS, st['seq'], st['n_fwd'], prec))
            flops_step = fps * B * st['seq']
    value = B * world / step_s
    result = {
        'metric': metric, 'value': round(value, 2), 'unit': unit,
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * step_s, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
        'config': {'workload': what, 'global_batch': B * world, 'parallelism': 'dp%d' % world,
                   'hip_graphs': bool(model.use_graphs)},
        'timing': {'value_from': 'median of %d blocks of %d steps (each >= %.2f s, barrier + synchronize on both sides, '
                                 'max over ranks)' % (args.blocks, per_block, args.min_block_s),
                   'block_ms_per_step': [round(1e3 * b, 4) for b in blocks],
                   'k_step_window_ms_per_step': round(1e3 * k_elapsed / args.steps, 4)},
        'model_tflops': round(flops_step * world / step_s / 1e12, 3),
        'device_self_check': self_check,
    }
    if not st['train']:
        result['model_steps_per_s'] = round(B * world * st['n_fwd'] / step_s, 1)
        result['ms_per_forward'] = round(1e3 * step_s / st['n_fwd'], 4)
    if world > 1:
        result['exchange'] = {'allreduce_us': None if ar_us is None else round(ar_us, 1),
                              'bytes': int(model._flat_grads.numel() * 4) if st['train'] else 0,
                              'backend': torch.distributed.get_backend(), 'rccl_ranks': torch.distributed.get_world_size(),
                              **(sync or {}),
                              'buckets': 2 if (st['train'] and getattr(model, '_did_split', False)) else 1,
                              'bucket_trials_ms_per_step': None if not bucket_trials else
                              {str(k): round(1e3 * v, 4) for k, v in bucket_trials.items()},
                              'exposed_us': None if noex_s is None else round(1e6 * (step_s - noex_s), 1),
                              'ms_per_step_without_exchange': None if noex_s is None else round(1e3 * noex_s, 4),
                              'graphs_per_step': (lambda g: None if g is None else 1 + (g.get('bwd_b') is not None) +
                                                  (g.get('update') is not None))(next(iter(model._graphs.values()), None)),
                              'overlap': 'buckets = 2: the decoder-side gradients are summed over the ranks (RCCL, the process '
                                         'group\'s stream) while the encoder-side half of the backward pass and its weight '
                                         'gradients run; the second bucket and the update graph wait for both.  buckets = 1: '
                                         'reduction | ONE all-reduce captured inside the step graph | one launch that applies the '
                                         'update (graphs_per_step = 1).  Both were timed '
                                         '(bucket_trials_ms_per_step), the faster one ran. '
                                         'allreduce_us = ONE all-reduce of the whole buffer on an idle GPU; exposed_us = '
                                         'ms_per_step minus the same step with the all-reduce calls skipped'}
    if with_roofline and agg:
        # per kernel: the roofline that bounds it = the larger of (algorithmic flops / matrix peak of the instruction it
        # issues) and (algorithmic bytes / HBM peak); frac = that bound time / measured time
        # (the matrix peak is that of the instruction the STEP's dtype makes the kernel issue -- wgrad_batch_kernel carries bf16 and
        # fp32 segment bodies under one name -- except for instantiations that name their element type)
        def is_bf16_mfma(name):
            return '<float' not in name and (dtype == 'bf16' or '<unsigned short' in name)

        def bound_of(name, cnt, ms, fl, by):
            peak_f = PEAK_BF16_MFMA_TFLOPS if is_bf16_mfma(name) else PEAK_FP32_MFMA_TFLOPS
            t_f, t_b = fl / (peak_f * 1e12), by / (PEAK_HBM_GBS * 1e9)
            t = ms * 1e-3
            if t_f >= t_b:
                return {'bound': 'mfma', 'achieved': round(fl / t / 1e12, 3), 'peak': peak_f, 'unit': 'TFLOP/s',
                        'frac': round(t_f / t, 4)}
            return {'bound': 'hbm', 'achieved': round(by / t / 1e9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(t_b / t, 4)}
        # durations: inside the replayed graph where the runtime times external event-record nodes (the form the timed region
        # runs; this is what rocprofv3 sees under profiles/), else from the eager steps; the eager figure rides along
        agg_eager = agg
        if agg_graph:
            agg = dict(agg_eager)
            agg.update(agg_graph)
        name, (cnt, ms, fl, by) = max(agg.items(), key=lambda kv: kv[1][1])
        rf = bound_of(name, cnt, ms, fl, by)
        rf['launch_time_from'] = ('HIP events as external event-record nodes inside the replayed step graph (csrc/prof.cpp), '
                                  '%d replays' % cnt if (agg_graph and name in agg_graph) else
                                  'HIP events around the launches of eager steps (graph form unavailable)')
        if agg_graph and name in agg_graph and name in agg_eager:
            rf['avg_launch_us_eager'] = round(1e3 * agg_eager[name][1] / agg_eager[name][0], 2)
        mp = (MEASURED_PEAKS['bf16_mfma_tflops'] if is_bf16_mfma(name) else MEASURED_PEAKS['fp32_mfma_tflops']) \
            if rf['bound'] == 'mfma' else MEASURED_PEAKS['hbm_read_gbs']
        rf['measured_peaks'] = MEASURED_PEAKS
        rf['frac_vs_measured_peak'] = round(rf['achieved'] / mp, 4)
        rf.update({'kernel': name, 'launches': cnt, 'avg_launch_us': round(1e3 * ms / cnt, 2),
                   'algorithmic_gflop_per_launch': round(fl / cnt / 1e9, 3),
                   'algorithmic_mbytes_per_launch': round(by / cnt / 1e6, 3)})
        recs, src = (None, 'PMC passes disabled (--no-pmc)')
        if with_pmc:
            recs, src = collect_pmc_live(args, dtype, groups=pmc_groups)
            if recs is None:
                live_err = src
                recs, src = newest_committed_pmc(args.workload, dtype)
                src = '%s [live collection unavailable: %s]' % (src, live_err)
        if recs and getattr(args, 'pmc_out', None):
            try:
                doc = json.load(open(args.pmc_out)) if os.path.exists(args.pmc_out) else {}
            except ValueError:
                doc = {}
            doc['%s/%s' % (args.workload, dtype)] = {k: v for k, v in recs.items()
                                                     if any(x in k for x in ('conv_mfma', 'wgrad', 'pw_', 'pad_', 'src_pair',
                                                                             'avgpool', 'mse_', 'adam', 'pack_batch', 'wb_',
                                                                             'state_repack', 'batch_gather'))}
            doc['_source'] = src
            with open(args.pmc_out, 'w') as f:
                json.dump(doc, f, indent=1, sort_keys=True)
        # profiler tag -> PMC record: the tag IS the kernel's name as rocprofv3 prints it (tests/test_abi.py pins every tag to the
        # library's symbol table); "(mode)" suffixes of a tag name a mode of the same kernel
        def pmc_of(k):
            import re
            if not recs:
                return None
            base = re.sub(r'\(.*\)$', '', k)
            return recs.get(k) or recs.get(base) or next((v for kk, v in recs.items() if kk.replace(' ', '') == base.replace(' ', '')), None)
        rec = pmc_of(name)
        # (a kernel of the step without a counter record must not pass silently)
        rf['pmc_missing'] = sorted(k for k in agg if pmc_of(k) is None) if (with_pmc and recs) else None
        if rf['pmc_missing']:
            sys.stderr.write('bench.py: ERROR: no PMC record for %s in: %s\n' % (rf['pmc_missing'], src))
        if rec is None:
            rf.update({'traffic': None, 'hbm_gbs': None, 'mfma_busy': None})
            rf['pmc_error'] = 'NO PMC RECORD for the dominant kernel %r in: %s' % (name, src)
            sys.stderr.write('bench.py: ERROR: %s\n' % rf['pmc_error'])
        else:
            rf['traffic'] = rec.get('traffic')
            rf['traffic_raw_counters'] = rec.get('traffic_raw')
            rf['traffic_note'] = 'traffic = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: rocprofv3 on gfx950 reports half of the ' \
                                 'bytes of wide streaming reads); traffic_raw_counters = FETCH_SIZE + WRITE_SIZE as reported' 
            rf['traffic_vs_algorithmic'] = (round(rec['traffic'] / (by / cnt), 3) if rec.get('traffic') and by else None)
            rf['hbm_gbs'] = round(rec['traffic'] / (ms / cnt * 1e-3) / 1e9, 1) if rec.get('traffic') else None
            rf['hbm_frac_of_peak'] = round(rf['hbm_gbs'] / PEAK_HBM_GBS, 4) if rf['hbm_gbs'] else None
            # two denominators: the profiler's own active window (GRBM_GUI_ACTIVE, includes ~8 us of dispatch overhead per
            # launch under the profiler) and this kernel's HIP-event launch time at the 2.4 GHz peak clock
            rf['mfma_busy'] = rec.get('mfma_busy')
            if rec.get('mfma_busy_cycles'):
                rf['mfma_busy_vs_launch_time_at_peak_clock'] = round(
                    rec['mfma_busy_cycles'] / (N_SIMD * (1e3 * ms / cnt) * PEAK_CLOCK_MHZ), 4)
        rf['pmc_source'] = src
        tot_ms = sum(v[1] for v in agg.values())
        tot_fl = sum(v[2] for v in agg.values())
        rf['all_mfma_kernels_tflops'] = round(tot_fl / (tot_ms * 1e-3) / 1e12, 3)
        rf['per_kernel'] = {}
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            bk = bound_of(k, *v)
            pk = {'launches': v[0], 'avg_us': round(1e3 * v[1] / v[0], 2),
                  'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2), 'bound': bk['bound'], 'frac': bk['frac']}
            r2 = pmc_of(k)
            if r2:
                if r2.get('traffic'):
                    pk['traffic'] = r2['traffic']
                if r2.get('mfma_busy') is not None:
                    pk['mfma_busy'] = r2['mfma_busy']
            if agg_graph and k in agg_graph and k in agg_eager:
                pk['avg_us_eager'] = round(1e3 * agg_eager[k][1] / agg_eager[k][0], 2)
            rf['per_kernel'][k] = pk
        result['roofline'] = rf
    del model, st
    torch.cuda.empty_cache()
    return result


_REAL_STDOUT = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='unet2', choices=['unet2', 'encoder6', 'rollout', 'unet2x2', 'rollout_x2'])
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU per step')
    ap.add_argument('--face', type=int, default=None, help='cube face size (48; rollout: 96)')
    ap.add_argument('--channels', type=int, default=None,
                    help='input (= output) channels: 14 = 7 variables x 2 time steps (rollout: 26 = 13 x 2)')
    ap.add_argument('--base', type=int, default=32)
    ap.add_argument('--rollout-steps', type=int, default=40, help='forecast steps of --workload rollout (2 per forward)')
    ap.add_argument('--dtype', default='bf16', choices=['f32', 'bf16'],
                    help='activation dtype of the headline number.  bf16 (default; BASELINE config 3 names bf16 compute, '
                         'the reference trains under TF AMP): bf16 activations + bf16 MFMA, fp32 master weights / '
                         'gradients / Adam.  f32: exact-fp32 MFMA everywhere (the 1e-5 parity mode).')
    ap.add_argument('--blocks', type=int, default=5, help='timed blocks behind the headline value (median)')
    ap.add_argument('--min-block-s', type=float, default=0.5, help='minimum duration of a timed block')
    ap.add_argument('--no-companion', action='store_true',
                    help='skip the second measurement in the other dtype (N = 1 only) reported under "f32" / "bf16"')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true',
                    help='N = 1, unet2: skip the extra measurements of BASELINE configs 2 (encoder6, fp32) and 5 (rollout, bf16)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-dp-form', action='store_true',
                    help='N = 1: skip the measurement of the data-parallel form of the step (one-rank RCCL group)')
    ap.add_argument('--no-pmc', action='store_true', help='no rocprofv3 child passes (traffic / mfma_busy stay null)')
    ap.add_argument('--no-graphs', action='store_true')
    ap.add_argument('--pmc-out', default=None,
                    help='also write the per-kernel PMC records of this run to FILE (JSON; tools/make_profiles.py commits it as '
                         'profiles/rNN_*_pmc.json, the fallback source when rocprofv3 cannot run)')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--pmc-plan', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.face is None:
        args.face = 96 if args.workload == 'rollout' else 48
    if args.channels is None:
        args.channels = 26 if args.workload == 'rollout' else (x2_channels()[1] if args.workload in ('unet2x2', 'rollout_x2') else 14)

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C stdio when
    # a communicator is created, flushed at process exit): file descriptor 1 is pointed at stderr for the life of the process and
    # the JSON line goes to the ORIGINAL stdout at the very end.
    global _REAL_STDOUT
    if _REAL_STDOUT is None and not args.pmc_child:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('DLWPCS_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # invoked plainly (`python bench.py --gpus N ...`): start the N ranks ourselves, one process per GPU, exactly as
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` would; rank 0 of the
        # children prints the JSON line on our stdout
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, stdout=_REAL_STDOUT if _REAL_STDOUT is not None else None).returncode)
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write('bench.py: no HIP device visible; the engine has no CPU path\n')
        sys.exit(2)
    if args.pmc_child:
        torch.cuda.set_device(0)
        pmc_child(args)
        return
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL over xGMI.  DLWPCS_BENCH_BACKEND=gloo + DLWPCS_BENCH_SHARE_GPU=1 exist only to exercise this exact code path
        # on a single-GPU box (all ranks on cuda:0, host-staged all-reduce); never used for a reported number.
        be = os.environ.get('DLWPCS_BENCH_BACKEND', 'nccl')
        if be == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            torch.distributed.init_process_group(be)

    from DLWP.keras import backend
    backend.set_device('cuda:%d' % local_rank)
    single = world == 1
    # the other workloads of the default line: BASELINE configs 2 and 5 and the reference scripts' production model
    import copy
    cfg_runs = []
    if single and not args.no_configs and args.workload == 'unet2':
        for key, wl, ch, face, dt in (('cfg2_encoder6_f32', 'encoder6', 7, 48, 'f32'), ('cfg5_rollout_bf16', 'rollout', 26, 96, 'bf16'),
                                      ('production_unet2x2_bf16', 'unet2x2', x2_channels()[1], 48, 'bf16'),
                                      ('production_rollout_bf16', 'rollout_x2', x2_channels()[1], 48, 'bf16')):
            a2 = copy.copy(args)
            a2.workload, a2.channels, a2.face, a2.dtype = wl, ch, face, dt
            a2.blocks, a2.min_block_s, a2.steps, a2.warmup, a2.pmc_out = 3, 0.3, (20 if 'rollout' in wl else 100), 5, None
            cfg_runs.append((key, a2, dt))
    if single and not args.no_pmc and not args.no_roofline:
        # every counter pass of this line up front: ONE rocprofv3 process per counter group runs all workloads (round 5: one per group
        # and workload -- eight process start-ups under the profiler, a third of bench.py's wall time)
        plan = [(args, args.dtype)]
        if not args.no_companion:
            plan.append((args, 'f32' if args.dtype == 'bf16' else 'bf16'))
        plan += [(a2, dt) for _, a2, dt in cfg_runs]
        collect_pmc_plan(plan)
    if world > 1:
        # the exchange this run ASKS for: the library-owned RCCL communicator on the compute stream (opt-in in the product, see
        # DLWP/parallel.py; DLWPCS_BENCH_NATIVE_RCCL=0: torch.distributed.all_reduce).  What actually served the timed region is
        # written into the line (exchange.collective) and checked against the request there.
        from DLWP import parallel as _par
        _par.enable_native_comm(os.environ.get('DLWPCS_BENCH_NATIVE_RCCL', '1') == '1')
    result = measure(args, args.dtype, rank, world, with_roofline=not args.no_roofline,
                     with_pmc=single and not args.no_pmc)
    if world > 1:
        torch.distributed.barrier()
    if single and not args.no_dp_form and 'rollout' not in args.workload:
        result['dp_form'] = dp_form_probe(args, args.dtype, result['ms_per_step'])
    if single and not args.no_companion:
        other = 'f32' if args.dtype == 'bf16' else 'bf16'
        comp = measure(args, other, rank, world, with_roofline=not args.no_roofline, with_pmc=not args.no_pmc)
        keep = ('value', 'unit', 'ms_per_step', 'dtype', 'model_tflops', 'timing', 'roofline')
        result[other] = {k: comp[k] for k in keep if k in comp}
    if cfg_runs:
        # BASELINE configs 2 and 5 and the production model, timed by the same clock in the same run (their own lines: --workload ...)
        result['configs'] = {}
        for key, a2, dt in cfg_runs:
            r2 = measure(a2, dt, rank, world, with_roofline=not args.no_roofline, with_pmc=not args.no_pmc)
            ent = {k: r2[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'model_tflops', 'config') if k in r2}
            for k in ('ms_per_forward', 'model_steps_per_s'):
                if k in r2:
                    ent[k] = r2[k]
            rf2 = r2.get('roofline')
            if rf2:
                ent['roofline'] = {k: rf2.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_vs_measured_peak',
                                                           'avg_launch_us', 'launches', 'algorithmic_gflop_per_launch',
                                                           'algorithmic_mbytes_per_launch', 'traffic', 'traffic_raw_counters',
                                                           'traffic_vs_algorithmic', 'hbm_gbs', 'mfma_busy', 'pmc_source', 'pmc_error', 'pmc_missing')
                                   if k in rf2}
            result['configs'][key] = ent
    if rank == 0 and single and not args.no_cpu_baseline and args.workload not in ('unet2x2', 'rollout_x2'):     # (the port times single-input networks)
        result['cpu_baseline'] = cpu_baseline(args.workload, args.face, args.channels, args.channels, args.base, args.batch)
    if rank == 0:
        line = json.dumps(result) + '\n'
        if _REAL_STDOUT is not None:
            os.write(_REAL_STDOUT, line.encode())
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    if world > 1:
        from DLWP import parallel as _par
        torch.distributed.barrier()
        _par.native_comm_release()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
