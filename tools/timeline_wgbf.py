#!/usr/bin/env python3
"""s_memtime phase timeline of wgrad_bf16_kernel (library built with -DDLWPCS_TIMELINE). env: CIN COUT N B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP import ops
dev = torch.device('cuda', 0)
E = os.environ.get
B, N, C0, Cout = int(E('B', 32)), int(E('N', 48)), int(E('CIN', 32)), int(E('COUT', 32))
x = torch.randn(B, 6, N, N, C0, device=dev).to(torch.bfloat16)
w = [(torch.randn(3, 3, C0, Cout, device=dev) / 17).requires_grad_(True) for _ in range(2)]
b = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(2)]
gy = torch.randn(B, 6, N, N, Cout, device=dev).to(torch.bfloat16)
dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
for it in range(3):
    y = ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
    if it == 2: os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
    y.backward(gy)
torch.cuda.synchronize()
tall = dbg.cpu().numpy().reshape(256, 64)
for which, t in (('consumer', tall[:, :32]), ('producer', tall[:, 32:])):
    nz = (t > 0).sum(axis=1)
    rows = nz >= 4
    if rows.sum() == 0:
        print(which, 'no marks'); continue
    k = nz[rows].min()
    tt = t[rows][:, :k]
    d = np.diff(tt, axis=1).astype(np.float64)
    print('%s marks %d..%d (%d workers) total median %.0f cycles' % (which, nz[rows].min(), nz[rows].max(), rows.sum(), np.median(tt[:, -1] - tt[:, 0])))
    print('   ' + ' '.join('%6.0f' % np.median(d[:, i]) for i in range(min(k - 1, 31))))
