#!/usr/bin/env python3
"""Would two half-batches on two streams hide the per-launch bubbles?    In steady state the bf16 convolution
kernels are bandwidth-bound; every launch has ~9 us of ramp-up / ramp-down in which HBM idles.  Two INDEPENDENT training steps of
batch B/2 (two models, two captured step graphs) replayed on two streams at the same time overlap one's bubbles with the other's
streaming.  Compare: one step of batch B | two steps of batch B/2 back to back on one stream | the same two on two streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np      # noqa: E402
import torch            # noqa: E402

from DLWP.keras import backend                     # noqa: E402
from DLWP.model.cs_unet import build_cs_model      # noqa: E402

dev = torch.device('cuda', 0)
backend.set_device('cuda:0')
B = int(os.environ.get('B', '32'))


def make(batch):
    backend.set_compute_dtype('bfloat16')
    try:
        np.random.seed(1)
        m = build_cs_model((6, 48, 48, 14), 14, 'unet2', base_filter_number=32)
    finally:
        backend.set_compute_dtype('float32')
    m.static_batch_buffers = True
    m.compile(optimizer='adam', loss='mse')
    x = [torch.randn(batch, 6, 48, 48, 14, device=dev).to(torch.bfloat16)]
    t = [torch.randn(batch, 6, 48, 48, 14, device=dev)]
    for _ in range(4):
        m.train_on_device_batch(x, t)
    torch.cuda.synchronize()
    g = next(iter(m._graphs.values()))
    return m, x, t, g['fwd_bwd']


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = make(B)
ha, hb = make(B // 2), make(B // 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print('one step of batch %d:                 %.4f ms' % (B, timed(lambda: full[3].replay())))


def seq():
    ha[3].replay()
    hb[3].replay()


def par():
    with torch.cuda.stream(sa):
        ha[3].replay()
    with torch.cuda.stream(sb):
        hb[3].replay()


print('two steps of batch %d, one stream:    %.4f ms' % (B // 2, timed(seq)))
print('two steps of batch %d, two streams:   %.4f ms' % (B // 2, timed(par)))
