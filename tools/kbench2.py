#!/usr/bin/env python3
"""ad-hoc: fwd conv TF vs Cin at N=48, Cout=32 (sector-efficiency experiment)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import torch
from DLWP import _native as nat, ops
dev = torch.device('cuda', 0); lib = nat.lib()
B = int(os.environ.get('B', 32))
for (N, cin, cout) in [(48, 16, 32), (48, 32, 32), (48, 64, 32), (48, 128, 32), (24, 64, 64), (24, 128, 64)]:
    x = torch.randn(B, 6, N, N, cin, device=dev)
    w = [torch.randn(3, 3, cin, cout, device=dev) / (9 * cin) ** .5 for _ in range(2)]
    b = [torch.zeros(cout, device=dev) for _ in range(2)]
    f = lambda: ops.cs_conv(x, w[0], w[1], None, b[0], b[1], None, ksize=3, halo=True, act=1, alpha=0.1, vmax=10.)
    f(); torch.cuda.synchronize()
    lib.dlwpcs_prof_reset(); lib.dlwpcs_prof_enable(1)
    for _ in range(5): f()
    torch.cuda.synchronize(); lib.dlwpcs_prof_enable(0)
    tag = ctypes.create_string_buffer(160); ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    tot = 0; flops = 0
    for i in range(lib.dlwpcs_prof_count()):
        lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)); tot += ms.value; flops += fl.value
    print('N=%d Cin=%3d Cout=%d  %7.1f us  %6.2f TF  %s' % (N, cin, cout, 1e3 * tot / 5, flops / tot / 1e9, tag.value.decode()))
