#!/usr/bin/env python3
"""
BASELINE config 5 (inference): `unet2`, base 32, C96 cube (6 x 96 x 96), 13 variables x 2 time steps = 26 channels in and
out, bf16, `predict_timeseries(time_steps=40)` with time_dim = 2 -> 20 sequential forward passes per sample with the state
kept in HBM (DLWP/model/models.py: rollout_on_device).  Prints rollouts/s, model-steps/s and the forward TFLOP/s on ONE GPU
(config 5 runs 8 independent replicas: no communication, so the 8-GPU number is 8x this one).
usage: tools/bench_rollout.py [--batch 32] [--face 96] [--channels 26] [--steps 40] [--dtype bf16|f32] [--repeat 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from bench import build_model, flops_per_sample   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--face', type=int, default=96)
    ap.add_argument('--channels', type=int, default=26)
    ap.add_argument('--base', type=int, default=32)
    ap.add_argument('--steps', type=int, default=40, help='forecast time steps (2 per forward pass)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--repeat', type=int, default=5)
    a = ap.parse_args()
    from DLWP.keras import backend
    backend.set_device('cuda:0')
    backend.set_compute_dtype('bfloat16' if a.dtype == 'bf16' else 'float32')
    np.random.seed(1)
    model = build_model('unet2', a.face, a.channels, a.channels, a.base)
    backend.set_compute_dtype('float32')
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    state0 = torch.randn(a.batch, 6, a.face, a.face, a.channels, device='cuda:0').to(dt)
    n_fwd = a.steps // 2

    def rollout():
        state = state0
        with torch.no_grad():
            for i in range(n_fwd):
                state = model.predict_on_device(state, repack=(i == 0))       # as Model.rollout_on_device does
        return state

    rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        out = rollout()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.repeat
    assert torch.isfinite(out.float()).all()
    fps = flops_per_sample('unet2', a.face, a.channels, a.channels, a.base)
    print(json.dumps({'metric': 'cubed-sphere rollouts/sec (40-step, inference)', 'value': round(a.batch / el, 2),
                      'unit': 'rollouts/s', 'n_gpus': 1, 'model_steps_per_s': round(a.batch * n_fwd / el, 1),
                      'ms_per_forward': round(1e3 * el / n_fwd, 3), 'dtype': a.dtype,
                      'forward_tflops': round(fps * a.batch * n_fwd / el / 1e12, 2),
                      'config': {'workload': 'unet2 C%d, %d channels, batch %d, %d forwards per rollout, state in HBM'
                                             % (a.face, a.channels, a.batch, n_fwd)}}))


if __name__ == '__main__':
    main()
