#!/usr/bin/env python3
"""Time the 1x1 head (32 -> 14, bf16, batch 32, N = 48) forward / data-gradient launches in isolation."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dlwp-cs_amd'))
import torch
from DLWP import ops
from DLWP._native import ACT_NONE

dev = torch.device('cuda', 0)
B, N, C, Co = 32, 48, 32, int(os.environ.get('COUT', '14'))
x = torch.randn(B, 6, N, N, C, device=dev).to(torch.bfloat16).requires_grad_(True)
w = [torch.randn(1, 1, C, Co, device=dev).requires_grad_(True) for _ in range(2)]
b = [torch.randn(Co, device=dev).requires_grad_(True) for _ in range(2)]
gy = torch.randn(B, 6, N, N, Co, device=dev).to(torch.bfloat16)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


with torch.no_grad():
    t_f = timed(lambda: ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=1, halo=False, act=ACT_NONE))
print('forward only: %.1f us per call' % t_f)


def fb():
    y = ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=1, halo=False, act=ACT_NONE)
    y.backward(gy)


t_fb = timed(fb)
print('forward + backward (dgrad + wgrad + reduce): %.1f us per call' % t_fb)
