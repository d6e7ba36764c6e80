#!/usr/bin/env python3
"""
Micro-benchmark of the bf16 data gradient (dlwpcs_conv_bwd_data_masked) on the `unet2` layers that have one, with and without the
pre-masked store epilogue (m0), HIP-event time per kernel from the library's profiler.
Usage: python tools/dgrad_bench.py [--batch 32] [--reps 20]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import torch   # noqa: E402

from DLWP import _native as nat   # noqa: E402

# (name, N, C0, C1, up0, Cout)
LAYERS = [('L2', 48, 32, 0, 0, 32), ('L3', 24, 32, 0, 0, 64), ('L4', 24, 64, 0, 0, 64), ('L5', 12, 64, 0, 0, 128),
          ('L6', 12, 128, 0, 0, 64), ('L7', 24, 64, 64, 1, 64), ('L8', 24, 64, 0, 0, 32), ('L9', 48, 32, 32, 1, 32),
          ('L10', 48, 32, 0, 0, 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='', help='comma-separated layer names')
    ap.add_argument('--gather', type=int, default=0, help='1: data gradient in gather form (CONV_DGRAD_GATHER)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    lib = nat.lib()
    B = args.batch
    for (name, N, C0, C1, up0, Cout) in LAYERS:
        if args.only and name not in args.only.split(','):
            continue
        n0 = N // 2 if up0 else N
        d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=3, halo=1, up0=up0, flip_north_pole=1, act=0, alpha=0., vmax=0.,
                         dtype=nat.BF16, flags=nat.CONV_DGRAD_GATHER if args.gather else 0, c0_valid=0)
        cin = C0 + C1
        dz = torch.randn(B, 6, N, N, Cout, device=dev).to(torch.bfloat16)
        w = [torch.randn(3, 3, cin, Cout, device=dev) / (9 * cin) ** 0.5 for _ in range(2)]
        s0 = torch.randn(B, 6, n0, n0, C0, device=dev).to(torch.bfloat16)
        s1 = torch.randn(B, 6, N, N, C1, device=dev).to(torch.bfloat16) if C1 else None
        d0, d1 = torch.empty_like(s0), (torch.empty_like(s1) if C1 else None)
        inv = nat.halo_tables(N, 1, dev)[1]
        nbytes = lib.dlwpcs_conv_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = []
        for masked in (False, True):
            if masked and up0:
                out.append('   -   ')
                continue

            def run():
                nat.check(lib.dlwpcs_conv_bwd_data_masked(ctypes.byref(d), nat.ptr(dz), nat.ptr(w[0]), nat.ptr(w[1]), None,
                                                          nat.ptr(d0), nat.ptr(d1), nat.ptr(s0) if masked else None, None, 0.1,
                                                          10.0, nat.ptr(inv), nat.ptr(ws), ws.numel(), nat.stream_ptr()), 'bwd')
            run()
            torch.cuda.synchronize()
            lib.dlwpcs_prof_reset()
            lib.dlwpcs_prof_enable(1)
            for _ in range(args.reps):
                run()
            torch.cuda.synchronize()
            lib.dlwpcs_prof_enable(0)
            tag = ctypes.create_string_buffer(200)
            ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            per = {}
            for i in range(lib.dlwpcs_prof_count()):
                lib.dlwpcs_prof_get(i, tag, 200, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
                a = per.setdefault(tag.value.decode(), [0.0, 0])
                a[0] += ms.value
                a[1] += 1
            lib.dlwpcs_prof_reset()
            out.append(' | '.join('%s %.1f us' % (k.split('<')[0][:20] + ('<' + k.split('<')[1][-40:] if '<' in k else ''),
                                                  1e3 * v[0] / v[1]) for k, v in per.items()))
        print('%-4s plain : %s\n     masked: %s' % (name, out[0], out[1]))


if __name__ == '__main__':
    main()
