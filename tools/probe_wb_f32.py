"""time dlwpcs_wgrad_batch on single fp32 layers of unet2 (B = 32): which variant costs what"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'dlwp-cs_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import torch
import test_gpu_wgrad_batch as T
from DLWP import ops, _native as nat

rng = np.random.default_rng(0)
f32 = '--bf16' not in sys.argv
B = 32
sets = [[T.UNET2[1]], [T.UNET2[0]], [T.UNET2[3]], [T.UNET2[10]], T.UNET2[1:-1]] if "--few" in sys.argv else [[c] for c in T.UNET2] + [T.UNET2, T.UNET2[1:-1], T.UNET2[:-1], T.UNET2[1:]]
if '--all' in sys.argv:
    sets = [T.UNET2]
if '--mix' in sys.argv:
    U = T.UNET2
    sets = [U[1:-1], U[1:-1] + [U[1]], U[1:-1] + [U[0]], U[1:-1] + [U[10]], [U[0], U[10]], [U[1], U[9]], [U[0], U[1]], [U[1], U[10]],
            [U[1], U[3]], [U[1], U[3], U[5]]]
for cfgs in sets:
    lays = [T.Layer(rng, B, *c, f32=f32) for c in cfgs]
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
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.wgrad_batch(ent)
    b.record()
    torch.cuda.synchronize()
    print('%-60s %8.1f us' % (str(cfgs[0]) if len(cfgs) == 1 else '%d layers ..%s' % (len(cfgs), cfgs[-1]), a.elapsed_time(b) / 20 * 1e3))
