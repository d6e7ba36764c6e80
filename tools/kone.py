#!/usr/bin/env python3
"""run ONE fused conv forward shape a few times (for rocprofv3 --pmc): N CIN COUT B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import torch
from DLWP import ops
N, cin, cout, B = [int(v) for v in sys.argv[1:5]]
dev = torch.device('cuda', 0)
x = torch.randn(B, 6, N, N, cin, device=dev)
w = [torch.randn(3, 3, cin, cout, device=dev) / (9 * cin) ** .5 for _ in range(2)]
b = [torch.zeros(cout, device=dev) for _ in range(2)]
for _ in range(4):
    y = ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
torch.cuda.synchronize()
