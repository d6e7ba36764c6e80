#!/usr/bin/env python3
"""
s_memtime accounting of the batched weight-gradient kernel (development; library built with -DDLWPCS_WB_TIMING):
  DLWPCS_LIB_TAG=wbt DLWPCS_EXTRA_CFLAGS=-DDLWPCS_WB_TIMING python dlwp-cs_amd/build.py
  DLWPCS_LIB_TAG=wbt python tools/wb_timing.py [--batch 32] [--layers 3]
Prints, per worker class, cycles per item spent by one producer thread (issuing loads / waiting for them + LDS writes /
barrier) and one consumer thread (barrier / MFMA phase), and the epilogues.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from DLWP import _native as nat   # noqa: E402
from DLWP import ops              # noqa: E402
from wb_bench import UNET2        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--layers', default='')
    ap.add_argument('--mask', default='', help='layers (of the full list) whose item carries y: act\' applied on load')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    B = args.batch
    sel = [int(v) for v in args.layers.split(',')] if args.layers else range(len(UNET2))
    entries = []
    for i in sel:
        N, C0, C1, up0, Cout, k, halo = UNET2[i]
        n0 = N // 2 if up0 else N
        x0 = torch.randn(B, 6, n0, n0, C0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, 6, N, N, C1, device=dev).to(torch.bfloat16) if C1 else None
        No = N if halo else N - k + 1
        dz = torch.randn(B, 6, No, No, Cout, device=dev).to(torch.bfloat16)
        cin = C0 + C1
        g = (torch.zeros(k, k, cin, Cout, device=dev), torch.zeros(k, k, cin, Cout, device=dev), None,
             torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), None)
        d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=halo, up0=up0, flip_north_pole=1, act=0, alpha=0.,
                         vmax=0., dtype=nat.BF16, flags=0, c0_valid=0)
        e = (d, x0, x1, dz, nat.halo_tables(N, 1, dev)[0] if halo else None, g)
        if str(i) in args.mask.split(','):
            d.act, d.alpha, d.vmax = nat.ACT_LEAKY_CLIP, 0.1, 10.0
            e = e + (torch.randn_like(dz.float()).mul_(3).to(torch.bfloat16),)
        entries.append(e)
    for _ in range(2):
        ops.wgrad_batch(entries)
    torch.cuda.synchronize()
    dbg = torch.zeros(256 * 2 * 8, dtype=torch.int64, device=dev)
    os.environ['DLWPCS_DBG_PTR'] = str(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.wgrad_batch(entries)
    e1.record()
    torch.cuda.synchronize()
    del os.environ['DLWPCS_DBG_PTR']
    t = dbg.cpu().numpy().reshape(256, 2, 8).astype(np.float64)
    us = 1e3 * e0.elapsed_time(e1)
    print('launch + reduce: %.1f us' % us)
    items = t[:, 0, 4]
    live = items > 0
    tot_p = t[:, 0, :4].sum(axis=1)
    tot_c = t[:, 1, :3].sum(axis=1)
    print('s_memtime ticks of the slowest worker / wall us: %.0f MHz-equivalent' % (tot_c.max() / us))
    print('workers %d, items per worker %.1f (min %d max %d)' % (live.sum(), items[live].mean(), items[live].min(), items[live].max()))
    print('producer thread, cycles per worker: total median %.0f max %.0f' % (np.median(tot_p[live]), tot_p[live].max()))
    print('   per item: issue %.0f  wait+LDS %.0f  barrier %.0f | epilogue per worker %.0f' % (
        np.median(t[live, 0, 0] / items[live]), np.median(t[live, 0, 1] / items[live]), np.median(t[live, 0, 2] / items[live]),
        np.median(t[live, 0, 3])))
    print('consumer thread, cycles per worker: total median %.0f max %.0f' % (np.median(tot_c[live]), tot_c[live].max()))
    print('   per item: barrier %.0f  mma %.0f | epilogue per worker %.0f' % (
        np.median(t[live, 1, 0] / items[live]), np.median(t[live, 1, 1] / items[live]), np.median(t[live, 1, 2])))
    q = np.percentile(tot_c[live], [5, 25, 50, 75, 95, 100])
    print('   consumer total percentiles 5/25/50/75/95/100: ' + ' '.join('%.0f' % v for v in q))


if __name__ == '__main__':
    main()
