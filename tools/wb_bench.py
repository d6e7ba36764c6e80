#!/usr/bin/env python3
"""
Micro-benchmark of the batched weight gradient (dlwpcs_wgrad_batch) on the `unet2` layer list (BASELINE config 3 geometry):
HIP-event time of the persistent launch and of the reduction, against the eleven per-layer launches it replaces.
Usage: python tools/wb_bench.py [--batch 32] [--reps 20] [--per-layer]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import torch   # noqa: E402

from DLWP import _native as nat   # noqa: E402
from DLWP import ops              # noqa: E402

UNET2 = [(48, 14, 0, 0, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (24, 32, 0, 0, 64, 3, 1), (24, 64, 0, 0, 64, 3, 1),
         (12, 64, 0, 0, 128, 3, 1), (12, 128, 0, 0, 64, 3, 1), (24, 64, 64, 1, 64, 3, 1), (24, 64, 0, 0, 32, 3, 1),
         (48, 32, 32, 1, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (48, 32, 0, 0, 14, 1, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--layers', default='')
    ap.add_argument('--spec', default='', help="layer list instead of unet2's: 'N,C0,C1,up0,Cout,k,halo;...'")
    ap.add_argument('--mask', default='', help='layers (of the full list) whose item carries y: act\' applied on load')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    lib = nat.lib()
    B = args.batch
    global UNET2
    if args.spec:
        UNET2 = [tuple(int(v) for v in item.split(',')) for item in args.spec.split(';')]
    sel = [int(v) for v in args.layers.split(',')] if args.layers else range(len(UNET2))
    entries = []
    keep = []
    for i in sel:
        N, C0, C1, up0, Cout, k, halo = UNET2[i]
        n0 = N // 2 if up0 else N
        x0 = torch.randn(B, 6, n0, n0, C0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, 6, N, N, C1, device=dev).to(torch.bfloat16) if C1 else None
        No = N if halo else N - k + 1
        dz = torch.randn(B, 6, No, No, Cout, device=dev).to(torch.bfloat16)
        cin = C0 + C1
        g = [torch.zeros(k, k, cin, Cout, device=dev), torch.zeros(k, k, cin, Cout, device=dev), None,
             torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), None]
        d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=halo, up0=up0, flip_north_pole=1, act=0, alpha=0.,
                         vmax=0., dtype=nat.BF16, flags=0, c0_valid=0)
        table = nat.halo_tables(N, 1, dev)[0] if halo else None
        e = (d, x0, x1, dz, table, tuple(g))
        if str(i) in args.mask.split(','):
            d.act, d.alpha, d.vmax = nat.ACT_LEAKY_CLIP, 0.1, 10.0
            e = e + (torch.randn_like(dz.float()).mul_(3).to(torch.bfloat16),)
        entries.append(e)
        keep.append((x0, x1, dz, g))
    ops.wgrad_batch(entries)
    torch.cuda.synchronize()
    lib.dlwpcs_prof_reset()
    lib.dlwpcs_prof_enable(1)
    for _ in range(args.reps):
        ops.wgrad_batch(entries)
    torch.cuda.synchronize()
    lib.dlwpcs_prof_enable(0)
    tag = ctypes.create_string_buffer(160)
    ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    acc = {}
    for i in range(lib.dlwpcs_prof_count()):
        lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
        a = acc.setdefault(tag.value.decode(), [0.0, 0, fl.value, by.value])
        a[0] += ms.value
        a[1] += 1
    for name, (t, n, f, b) in acc.items():
        us = 1e3 * t / n
        print('%-24s %8.1f us  %7.1f TFLOP/s  %6.2f TB/s (algorithmic)' % (name, us, f / us / 1e6, b / us / 1e6))
    # wall time of the pair, back to back
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ops.wgrad_batch(entries)
    e1.record()
    torch.cuda.synchronize()
    print('wall per call: %.1f us' % (1e3 * e0.elapsed_time(e1) / args.reps))


if __name__ == '__main__':
    main()
