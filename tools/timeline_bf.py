#!/usr/bin/env python3
"""s_memtime phase timeline of the wave-specialised conv kernel (library built with -DDLWPCS_TIMELINE).
env: DTYPE=f32|bf16  CIN COUT N B  MODE=fwd|dgrad"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP import ops
dev = torch.device('cuda', 0)
E = os.environ.get
B, N, C0, Cout = int(E('B', 32)), int(E('N', 48)), int(E('CIN', 32)), int(E('COUT', 32))
dt = torch.bfloat16 if E('DTYPE', 'bf16') == 'bf16' else torch.float32
mode = E('MODE', 'fwd')
x = torch.randn(B, 6, N, N, C0, device=dev).to(dt).requires_grad_(mode == 'dgrad')
w = [torch.randn(3, 3, C0, Cout, device=dev) / 17 for _ in range(2)]
b = [torch.zeros(Cout, device=dev) for _ in range(2)]
gy = torch.randn(B, 6, N, N, Cout, device=dev).to(dt)
dbg = torch.zeros(256 * 64 + 8, dtype=torch.int64, device=dev)
dbg[256 * 64] = int(E('SKIP_STORE', 0))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    y = ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
    if mode == 'dgrad':
        if it == 2: os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
        ev0.record()
        y.backward(gy)
        ev1.record()
    elif it == 1:
        os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
torch.cuda.synchronize()
tall = dbg.cpu().numpy()[:256 * 64].reshape(256, 64)
for which, t in (('consumer', tall[:, :32]), ('producer', tall[:, 32:])):
    nz = (t > 0).sum(axis=1); k = nz.min()
    if k < 2:
        print(which, 'no marks'); continue
    d = np.diff(t[:, :k], axis=1).astype(np.float64)
    print('%s marks %d..%d  total median %.0f cycles' % (which, nz.min(), nz.max(), np.median(t[:, k - 1] - t[:, 0])))
    print('   ' + ' '.join('%6.0f' % np.median(d[:, i]) for i in range(min(k - 1, 31))))
# absolute picture: offsets from the earliest mark of any workgroup (~ kernel start), in cycles
t0 = min(tall[:, 0][tall[:, 0] > 0].min(), tall[:, 32][tall[:, 32] > 0].min())
for which, t in (('consumer', tall[:, :32]), ('producer', tall[:, 32:])):
    nz = (t > 0).sum(axis=1)
    live = nz > 1
    first = (t[live, 0] - t0).astype(np.float64)
    second = (t[live, 1] - t0).astype(np.float64)
    last = np.array([t[i, nz[i] - 1] - t0 for i in np.where(live)[0]], dtype=np.float64)
    print('%s: first mark at %.0f / %.0f / %.0f (min/median/max), second mark %.0f / %.0f / %.0f, last mark %.0f / %.0f / %.0f, '
          'marks per workgroup %d..%d' % (which, first.min(), np.median(first), first.max(), second.min(), np.median(second),
                                         second.max(), last.min(), np.median(last), last.max(), nz[live].min(), nz[live].max()))
print('kernel (HIP events around the backward call, dgrad mode only): %.1f us' % (1e3 * ev0.elapsed_time(ev1)) if mode == 'dgrad' else '')
