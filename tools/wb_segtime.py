#!/usr/bin/env python3
"""
Per-segment timing of the batched weight-gradient kernel (development; library built with -DDLWPCS_WB_SEGTIME):
  DLWPCS_LIB_TAG=seg DLWPCS_EXTRA_CFLAGS=-DDLWPCS_WB_SEGTIME python dlwp-cs_amd/build.py
  DLWPCS_LIB_TAG=seg python tools/wb_segtime.py [--bf16] [--mask]
Fits, per layer, ticks(segment) = a + b * items by least squares and prints b relative to the plan's cost model -- the plan
cuts the work list into equal-cost chains, so a layer whose items cost x % more than modelled makes the launch x % longer.
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('', 'dlwp-cs_amd', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, d))
import ctypes
import numpy as np
import torch
import test_gpu_wgrad_batch as T
import test_wgrad_batch_plan as TP
from DLWP import ops, _native as nat

f32 = '--bf16' not in sys.argv
rng = np.random.default_rng(0)
B = 32
lays = [T.Layer(rng, B, *c, f32=f32) for c in T.UNET2]
ent = []
for l in lays:
    e = l.entry()
    if '--mask' in sys.argv and l.cfg[6] == 3:
        d = nat.ConvDesc.from_buffer_copy(l.d)
        d.act, d.alpha, d.vmax = nat.ACT_LEAKY_CLIP, 0.1, 10.0
        e = (d, e[1], e[2], e[3], e[4], e[5], torch.randn_like(e[3]))
    ent.append(e)
for _ in range(3):
    ops.wgrad_batch(ent)
torch.cuda.synchronize()
arr = (nat.WgradItem * len(ent))()
for it, e in zip(arr, ent):
    it.d = e[0]
    it.dw_eq, it.dw_pol, it.db_eq, it.db_pol = 1, 1, 1, 1
    if len(e) > 6:
        it.y = 1
P = TP._plan(arr)
nseg = P['n_segs']
acc = np.zeros(nseg)
R = 10
for _ in range(R):
    dbg = torch.zeros(nseg + 8, dtype=torch.int64, device='cuda')
    os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
    ops.wgrad_batch(ent)
    torch.cuda.synchronize()
    acc += dbg[:nseg].cpu().numpy()
acc /= R
segs = P['segs']
st = P['seg_start']
chain = np.array([acc[st[w]:st[w + 1]].sum() for w in range(256)])
print('chains: min %.0f  mean %.0f  max %.0f ticks (10 ns)' % (chain.min(), chain.mean(), chain.max()))
worst = np.argsort(chain)[-5:]
for w in worst:
    print('  worker %3d: %6.0f ticks:' % (w, chain[w]), [(int(s[0]), int(s[5] - s[4])) for s in segs[st[w]:st[w + 1]]])
print('layer  segs  items   ticks/item (fit)  fixed   (N, C0, C1, up, Cout, k, halo)')
base = None
for l in range(len(lays)):
    m = segs[:, 0] == l
    n = (segs[m, 5] - segs[m, 4]).astype(np.float64)
    t = acc[m]
    A = np.stack([np.ones_like(n), n], 1)
    (a, b), *_ = np.linalg.lstsq(A, t, rcond=None)
    print('%5d %5d %6d   %10.2f   %8.1f   %s' % (l, m.sum(), n.sum(), b, a, T.UNET2[l]))
