#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP import ops
dev = torch.device('cuda', 0)
B, N, C0, Cout = int(os.environ.get('B', 32)), 48, int(os.environ.get('CIN', 32)), 32
x = torch.randn(B, 6, N, N, C0, device=dev)
w = [torch.randn(3, 3, C0, Cout, device=dev) / 17 for _ in range(2)]
b = [torch.zeros(Cout, device=dev) for _ in range(2)]
dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
for it in range(3):
    if it == 2: os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
    y = ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
torch.cuda.synchronize()
tall = dbg.cpu().numpy().reshape(256, 64)
which = os.environ.get('WHICH', 'cons')
t = tall[:, :32] if which == 'cons' else tall[:, 32:]
nz = (t > 0).sum(axis=1); k = nz.min()
d = np.diff(t[:, :k], axis=1).astype(np.float64)
print('marks', nz.min(), nz.max())
for i in range(min(k - 1, 40)):
    print('  %2d median %8.0f  p10 %8.0f  p90 %8.0f' % (i, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
