#!/usr/bin/env python3
"""
Kernel micro-benchmark: time the fused-conv C-ABI entry points layer by layer (HIP-event profiler of the library).
Usage: python tools/kbench.py [--layers unet2] [--batch 32] [--reps 10] [--only fwd,dgrad,wgrad]
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

# (name, N, C0, C1, up0, Cout, k, halo, act)
UNET2 = [
    ('conv_2d_1', 48, 14, 0, 0, 32, 3, 1, 1),
    ('conv_2d_1_2', 48, 32, 0, 0, 32, 3, 1, 1),
    ('conv_2d_2', 24, 32, 0, 0, 64, 3, 1, 1),
    ('conv_2d_2_2', 24, 64, 0, 0, 64, 3, 1, 1),
    ('conv_2d_5_2', 12, 64, 0, 0, 128, 3, 1, 1),
    ('conv_2d_5', 12, 128, 0, 0, 64, 3, 1, 1),
    ('conv_2d_6_2', 24, 64, 64, 1, 64, 3, 1, 1),
    ('conv_2d_6', 24, 64, 0, 0, 32, 3, 1, 1),
    ('conv_2d_7', 48, 32, 32, 1, 32, 3, 1, 1),
    ('conv_2d_7_2', 48, 32, 0, 0, 32, 3, 1, 1),
    ('conv_2d_8', 48, 32, 0, 0, 14, 1, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='fwd,dgrad,wgrad')
    ap.add_argument('--layer', default='')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    lib = nat.lib()
    B = args.batch
    only = args.only.split(',')
    rows = []
    for (name, N, C0, C1, up0, Cout, k, halo, act) in UNET2:
        if args.layer and args.layer != name:
            continue
        n0 = N // 2 if up0 else N
        adt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
        src0 = torch.randn(B, 6, n0, n0, C0, device=dev).to(adt).requires_grad_(True)
        src1 = torch.randn(B, 6, N, N, C1, device=dev).to(adt).requires_grad_(True) if C1 else None
        cin = C0 + C1
        w = [(torch.randn(k, k, cin, Cout, device=dev) / (k * k * cin) ** 0.5).requires_grad_(True) for _ in range(2)]
        b = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(2)]
        No = N if halo else N - k + 1
        gy = torch.randn(B, 6, No, No, Cout, device=dev).to(adt)

        def run():
            y = ops.cs_conv(src0, w[0], w[1], None, b[0], b[1], None, src1=src1, ksize=k, halo=bool(halo),
                            up0=bool(up0), act=nat.ACT_LEAKY_CLIP if act else nat.ACT_NONE, alpha=0.1, vmax=10.)
            y.backward(gy)
        run()
        torch.cuda.synchronize()
        lib.dlwpcs_prof_reset()
        lib.dlwpcs_prof_enable(1)
        for _ in range(args.reps):
            run()
        torch.cuda.synchronize()
        lib.dlwpcs_prof_enable(0)
        tag = ctypes.create_string_buffer(160)
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        per = {}
        n = lib.dlwpcs_prof_count()
        for i in range(n):
            lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
            name = tag.value.decode()
            kind = 'wgrad' if 'wgrad' in name else ('fwd' if i % 3 == 0 else 'dgrad')
            a = per.setdefault(kind, [tag.value.decode(), 0.0, 0.0, 0])
            a[1] += ms.value
            a[2] += fl.value
            a[3] += 1
        lib.dlwpcs_prof_reset()
        for kind in ('fwd', 'dgrad', 'wgrad'):
            if kind in per and kind in only:
                t, msum, fsum, cnt = per[kind]
                rows.append((name, kind, 1e3 * msum / cnt, fsum / (msum * 1e-3) / 1e12, t))
    tot = sum(r[2] for r in rows)
    for r in rows:
        print('%-12s %-6s %8.1f us %7.2f TF  %s' % r)
    for kind in ('fwd', 'dgrad', 'wgrad'):
        print('sum %-6s %.1f us' % (kind, sum(r[2] for r in rows if r[1] == kind)))
    print('total %.1f us' % tot)


if __name__ == '__main__':
    main()
