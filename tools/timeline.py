#!/usr/bin/env python3
"""Dump the in-kernel s_memtime timeline of the persistent conv kernel (library built with -DDLWPCS_TIMELINE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np
import torch
from DLWP import _native as nat, ops

dev = torch.device('cuda', 0)
B, N, C0, Cout = int(os.environ.get('B', 32)), 48, 32, 32
x = torch.randn(B, 6, N, N, C0, device=dev)
w = [torch.randn(3, 3, C0, Cout, device=dev) / 17 for _ in range(2)]
b = [torch.zeros(Cout, device=dev) for _ in range(2)]
nblocks = 512
dbg = torch.zeros(nblocks * 64, dtype=torch.int64, device=dev)
for it in range(3):
    if it == 2:
        os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
    y = ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(nblocks, 64)
nz = (t > 0).sum(axis=1)
print('marks per block: min %d max %d' % (nz.min(), nz.max()))
t0 = t[:, 0].min()
# marks: start, after-prologue, then per tile: [c0: fetch-issued, compute-issued, commit-done] [last: fetch-issued, compute-issued, commit+epilogue-done] ..., end
k = nz.min()
d = np.diff(t[:, :k], axis=1).astype(np.float64)
labels = ['prologue'] + ['c0 fetch+tbl', 'c0 compute', 'c0 commit', 'c0 barrier+mid chunks+last fetch', 'last compute', 'last commit+epilogue'] * 8
for i in range(k - 1):
    print('  %2d %-34s median %8.0f  p10 %8.0f  p90 %8.0f' % (i, labels[i] if i < len(labels) else '', np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print('lifetime median', np.median(t[np.arange(nblocks), nz - 1] - t[:, 0]), ' kernel span', (t.max() - t0))
