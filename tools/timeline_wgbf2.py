#!/usr/bin/env python3
"""Start-up timeline of wgrad_bf16_kernel (library built with -DDLWPCS_TIMELINE): s_memtime marks of consumer wave 0 and
producer thread 0 of every worker, each relative to the worker's own first consumer mark (the XCDs' counters are not
synchronised).  HOT=1 runs the kernel twice back to back (only a small reduction kernel in between) and reports the second
launch: its code is still in the instruction caches.  env: CIN COUT N B HOT"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP import ops
dev = torch.device('cuda', 0)
E = os.environ.get
B, N, C0, Cout = int(E('B', 32)), int(E('N', 48)), int(E('CIN', 32)), int(E('COUT', 32))
hot = int(E('HOT', 0))
x = torch.randn(B, 6, N, N, C0, device=dev).to(torch.bfloat16)
w = [(torch.randn(3, 3, C0, Cout, device=dev) / 17).requires_grad_(True) for _ in range(2)]
b = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(2)]
gy = torch.randn(B, 6, N, N, Cout, device=dev).to(torch.bfloat16)
dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
conv = lambda: ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
for it in range(2):
    conv().backward(gy)
y1, y2 = conv(), conv()
torch.cuda.synchronize()
if hot:
    y1.backward(gy)
os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
y2.backward(gy)
torch.cuda.synchronize()
tall = dbg.cpu().numpy().reshape(256, 64).astype(np.float64)
live = tall[:, 0] > 0
t0 = tall[live, 0:1]
for which, t in (('consumer', tall[live, :32]), ('producer', tall[live, 32:])):
    k = int((t > 0).sum(axis=1).min())
    rel = t[:, :k] - t0
    print('%s: %d workers, %d marks; median time of each mark, cycles after the worker\'s consumer wave 0 started' % (which, live.sum(), k))
    print('   ' + ' '.join('%6.0f' % np.median(rel[:, i]) for i in range(min(k, 16))))
