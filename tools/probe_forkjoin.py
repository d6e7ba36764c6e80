#!/usr/bin/env python3
"""What does a stream fork / join cost inside a replayed hipGraph?  (ProcessGroupNCCL runs a captured all-reduce on its own stream:
current stream -> event -> RCCL stream -> event -> current stream.)  Two chains of 32 small kernels: plain, and with a fork / join to
a second stream (carrying one tiny kernel, or nothing) in the middle."""
import time
import torch

dev = torch.device('cuda', 0)
x = torch.zeros(1 << 16, device=dev)
side = torch.cuda.Stream()


def chain(kind):
    for i in range(32):
        x.add_(1.0)
        if i == 15 and kind != 'plain':
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                if kind == 'fork+kernel':
                    x.mul_(1.0)
            cur.wait_stream(side)


for kind in ('plain', 'fork+kernel', 'fork only'):
    chain(kind)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(kind)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    torch.cuda.synchronize()
    print('%-12s %.1f us per replay' % (kind, (time.perf_counter() - t0) / 500 * 1e6))
