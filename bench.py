#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X-native DLWP-CS engine.

Metric (BASELINE.json): cubed-sphere samples/sec, forward + backward (+ Adam step), 6x48x48 U-Net `unet2` with 7
variables x 2 time steps = 14 input/output channels (BASELINE config 3: geometry AND dtype -- bf16 compute with fp32
master weights), batch 32 per GPU, synthetic data, random-init weights.  At N = 1 the same workload is also measured in
the exact-fp32 mode (the 1e-5 parity mode) and reported under the key "f32".  One "step" = one optimisation step (fwd + bwd + gradient all-reduce when N > 1 + Adam) on one batch
already resident in HBM.  One process per GPU; N > 1 is launched by torch.distributed.run (RCCL).

Prints ONE JSON line on rank 0 (see the task contract): value = whole-job samples/s, plus
  roofline     -- dominant convolution kernel: algorithmic FLOPs and bytes per launch / HIP-event time per launch against
                  the roofline that bounds it (matrix peak of the instruction it issues, or HBM)
  cpu_baseline -- the CPU restatement of the reference path (oracle/, torch-CPU fp32, reference-structured) timed on the
                  host cores of this box on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
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
BF16_MFMA_KERNELS = ('conv_mfma_ws_kernel<unsigned short', 'wgrad_bf16_kernel')
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz
FLOP_PER_SAMPLE_FWD = {            # BASELINE.md section 2 (2*6*N^2*k^2*Cin*Cout summed over the conv layers)
    'unet2': None, 'encoder6': None}


def conv_plan(workload, c_in, c_out, base):
    """(N_rel, Cin, Cout, k) per conv layer; N_rel = face size divisor."""
    b = base
    enc = [(1, c_in, b, 3), (1, b, b, 3), (2, b, 2 * b, 3), (2, 2 * b, 2 * b, 3), (4, 2 * b, 4 * b, 3),
           (4, 4 * b, 2 * b, 3)]
    if workload == 'encoder6':
        return enc
    return enc + [(2, 4 * b, 2 * b, 3), (2, 2 * b, b, 3), (1, 2 * b, b, 3), (1, b, b, 3), (1, b, c_out, 1)]


def flops_per_sample(workload, N, c_in, c_out, base):
    return sum(2.0 * 6 * (N // r) ** 2 * k * k * ci * co for (r, ci, co, k) in conv_plan(workload, c_in, c_out, base))


def build_model(workload, N, c_in, c_out, base):
    from DLWP.model.cs_unet import CubeSphereNet
    from DLWP.keras.layers import Input
    from DLWP.keras.models import Model
    net = CubeSphereNet(c_out, base, 'unet2')
    x = Input(shape=(6, N, N, c_in), name='main_input')
    y = net.unet2(x) if workload == 'unet2' else net.encoder6(x)
    return Model(inputs=x, outputs=y)


def cpu_baseline(workload, N, c_in, c_out, base, budget_s=20.0, batch=4):
    """Reference-structured CPU port (oracle/cs_oracle.py), torch-CPU fp32, all host cores; fwd + bwd + Adam."""
    from oracle import cs_oracle as orc
    cores = os.cpu_count() or 1
    params = orc.make_unet2_params(c_in, c_out, base=base, seed=1, dtype=torch.float32)
    if workload == 'encoder6':
        params = params[:6]
    leaves = [v.requires_grad_(True) for prm in params for v in prm.values()]
    ms = [torch.zeros_like(v) for v in leaves]
    vs = [torch.zeros_like(v) for v in leaves]
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.standard_normal((batch, 6, N, N, c_in)), dtype=torch.float32)
    fwd = orc.unet2_forward if workload == 'unet2' else orc.encoder6_forward
    with torch.no_grad():
        tshape = fwd(x[:1], params).shape[1:]
    tgt = torch.tensor(rng.standard_normal((batch,) + tuple(tshape)), dtype=torch.float32)

    def step(t):
        for v in leaves:
            v.grad = None
        loss = orc.mse_loss(fwd(x, params), tgt)
        loss.backward()
        with torch.no_grad():
            for p, m, v in zip(leaves, ms, vs):
                orc.adam_step(p, p.grad, m, v, t)
    # pick the thread count that runs this workload fastest on this host (many-core boxes oversubscribe small convs)
    best = None
    for thr in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16)}, reverse=True):
        torch.set_num_threads(thr)
        step(1)                              # warm-up at this thread count
        t0 = time.perf_counter()
        step(1)
        el = time.perf_counter() - t0
        if best is None or el < best[0]:
            best = (el, thr)
    threads = best[1]
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    iters = 0
    while True:
        step(iters + 2)
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= 50:
            break
    return {'value': round(batch * iters / el, 3), 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'sample': '%d steps of batch %d, %s fwd+bwd+Adam, torch-CPU fp32, reference-structured oracle '
                      '(materialised halo padding, 6 per-face conv2d per layer)' % (iters, batch, workload)}


PMC_PROFILE = os.path.join(ROOT, 'profiles', 'r01_bench_unet2_b32_hbm_pmc.txt')


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC summary (tools/make_profiles.py), or (None, reason)."""
    try:
        for line in open(PMC_PROFILE):
            if line.startswith(kernel + ' ') or line.startswith(kernel[:84] + ' '):
                cols = line[84:].split()
                if len(cols) >= 4:
                    return int((float(cols[2]) + float(cols[3])) * 1024), os.path.relpath(PMC_PROFILE, ROOT)
    except (OSError, ValueError):
        pass
    return None, 'no PMC record for this kernel in profiles/'


def roofline_pass(model, dx, dt, steps=3):
    """Eager steps with the library's per-launch HIP-event profiler on; aggregate per kernel name."""
    from DLWP import _native as nat
    lib = nat.lib()
    use_graphs = model.use_graphs
    model.use_graphs = False
    lib.dlwpcs_prof_reset()
    lib.dlwpcs_prof_enable(1)
    for _ in range(steps):
        model.train_on_device_batch(dx, dt)
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


def measure(args, dtype, rank, world, local_rank, with_roofline):
    """Build the model in `dtype`, warm up, time exactly args.steps steps (barrier + synchronize on both sides, MAX over
    ranks), optionally run the per-launch roofline pass.  Returns the result dict on rank 0, None elsewhere."""
    from DLWP.keras import backend
    N, C, base, B = args.face, args.channels, args.base, args.batch
    backend.set_compute_dtype('bfloat16' if dtype == 'bf16' else 'float32')
    np.random.seed(1)
    model = build_model(args.workload, N, C, C, base)
    backend.set_compute_dtype('float32')
    model.use_graphs = not args.no_graphs
    model.static_batch_buffers = True      # the batch lives in the same HBM tensors every step (inputs resident in HBM)
    model.compile(optimizer='adam', loss='mse')
    adt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    rng = np.random.default_rng(1000 + rank)
    dev = backend.device()
    dx = [torch.tensor(rng.standard_normal((B, 6, N, N, C)), dtype=torch.float32, device=dev).to(adt)]
    with torch.no_grad():
        oshape = model.predict_on_device(dx[0][:1]).shape[1:]
    dt = [torch.tensor(rng.standard_normal((B,) + tuple(oshape)), dtype=torch.float32, device=dev)]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):      # >= 3: eager warm-up, graph capture, first replay
        model.train_on_device_batch(dx, dt)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_on_device_batch(dx, dt)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # the roofline pass runs eager optimisation steps (gradient all-reduce included): EVERY rank takes part
    agg = roofline_pass(model, dx, dt) if with_roofline else None
    if rank != 0:
        return None
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world * args.steps / elapsed
    fps = flops_per_sample(args.workload, N, C, C, base)
    prec = ('bf16 activations + bf16 MFMA, fp32 master weights / gradients / Adam' if dtype == 'bf16'
            else 'exact-fp32 MFMA')
    result = {
        'metric': 'cubed-sphere samples/sec (fwd+bwd)', 'value': round(value, 2), 'unit': 'samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
        'config': {'workload': '%s C%d: x (%d,6,%d,%d,%d) per GPU, %d out channels, base %d, MSE + Adam, '
                               'fwd+bwd+update, %s' % (args.workload, N, B, N, N, C, C, base, prec),
                   'global_batch': B * world, 'parallelism': 'dp%d' % world,
                   'hip_graphs': bool(model.use_graphs)},
        'model_tflops': round(3 * fps * value / 1e12, 3),
    }
    if with_roofline:
        if agg:
            # per kernel: the roofline that bounds it = the larger of (algorithmic flops / matrix peak of the instruction
            # it issues) and (algorithmic bytes / HBM peak); frac = that bound time / measured time
            def bound_of(name, cnt, ms, fl, by):
                peak_f = PEAK_BF16_MFMA_TFLOPS if name.startswith(BF16_MFMA_KERNELS) else PEAK_FP32_MFMA_TFLOPS
                t_f, t_b = fl / (peak_f * 1e12), by / (PEAK_HBM_GBS * 1e9)
                t = ms * 1e-3
                if t_f >= t_b:
                    return {'bound': 'mfma', 'achieved': round(fl / t / 1e12, 3), 'peak': peak_f, 'unit': 'TFLOP/s',
                            'frac': round(t_f / t, 4)}
                return {'bound': 'hbm', 'achieved': round(by / t / 1e9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(t_b / t, 4)}
            name, (cnt, ms, fl, by) = max(agg.items(), key=lambda kv: kv[1][1])
            rf = bound_of(name, cnt, ms, fl, by)
            # HBM-side bytes need PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, cannot run inside
            # this process): they are collected with this same command and committed per kernel in profiles/; `traffic` is
            # that measurement for the dominant kernel (FETCH_SIZE x 2 per the gfx950 calibration + WRITE_SIZE), bytes/launch
            traffic, tsrc = pmc_traffic(name)
            rf.update({'traffic': traffic, 'traffic_source': tsrc, 'kernel': name, 'launches': cnt,
                       'avg_launch_us': round(1e3 * ms / cnt, 2),
                       'algorithmic_gflop_per_launch': round(fl / cnt / 1e9, 3),
                       'algorithmic_mbytes_per_launch': round(by / cnt / 1e6, 3)})
            tot_ms = sum(v[1] for v in agg.values())
            tot_fl = sum(v[2] for v in agg.values())
            rf['all_mfma_kernels_tflops'] = round(tot_fl / (tot_ms * 1e-3) / 1e12, 3)
            rf['per_kernel'] = {}
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                bk = bound_of(k, *v)
                rf['per_kernel'][k] = {'launches': v[0], 'avg_us': round(1e3 * v[1] / v[0], 2),
                                       'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2), 'bound': bk['bound'],
                                       'frac': bk['frac']}
            result['roofline'] = rf
    del model
    torch.cuda.empty_cache()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='unet2', choices=['unet2', 'encoder6'])
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU per step')
    ap.add_argument('--face', type=int, default=48)
    ap.add_argument('--channels', type=int, default=14, help='input (= output) channels: 7 variables x 2 time steps')
    ap.add_argument('--base', type=int, default=32)
    ap.add_argument('--dtype', default='bf16', choices=['f32', 'bf16'],
                    help='activation dtype of the headline number.  bf16 (default; BASELINE config 3 names bf16 compute, '
                         'the reference trains under TF AMP): bf16 activations + bf16 MFMA, fp32 master weights / '
                         'gradients / Adam.  f32: exact-fp32 MFMA everywhere (the 1e-5 parity mode).')
    ap.add_argument('--no-companion', action='store_true',
                    help='skip the second measurement in the other dtype (N = 1 only) reported under "f32" / "bf16"')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-graphs', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('DLWPCS_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            sys.stderr.write('bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`\n'
                             % (args.gpus, args.gpus))
            sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write('bench.py: no HIP device visible; the engine has no CPU path\n')
        sys.exit(2)
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
    result = measure(args, args.dtype, rank, world, local_rank, with_roofline=not args.no_roofline)
    if world > 1:
        torch.distributed.barrier()
    if world == 1 and not args.no_companion:
        other = 'f32' if args.dtype == 'bf16' else 'bf16'
        comp = measure(args, other, rank, world, local_rank, with_roofline=not args.no_roofline)
        keep = ('value', 'unit', 'ms_per_step', 'dtype', 'model_tflops', 'roofline')
        result[other] = {k: comp[k] for k in keep if k in comp}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(args.workload, args.face, args.channels, args.channels, args.base)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
