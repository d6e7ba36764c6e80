#!/usr/bin/env python3
"""
Phase sums of the batched weight gradient (side build: DLWPCS_LIB_TAG=tl, -DDLWPCS_WB_TL=1): s_memtime differences accumulated in
registers by the first producer and the first consumer wave of every workgroup, one store per segment.
usage: DLWPCS_LIB_TAG=tl python tools/wb_timeline.py [--layers 1]
producer buckets: issue (set-up + load issue since the last barrier) | wait_data | lds_write (+ bias sums) | barrier | epilogue
consumer buckets: barrier (wait at B_k) | compute (fragment reads + MFMAs) | epilogue (reduction, partial sums)
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np   # noqa: E402
import torch         # noqa: E402
from DLWP import _native as nat   # noqa: E402
from DLWP import ops              # noqa: E402
import wb_bench                   # noqa: E402

TLW, TLM = 256, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='')
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    nat.lib()
    lib = ctypes.CDLL(nat.LIB_PATH)
    B = a.batch
    sel = [int(v) for v in a.layers.split(',')] if a.layers else range(len(wb_bench.UNET2))
    entries = []
    for i in sel:
        N, C0, C1, up0, Cout, k, halo = wb_bench.UNET2[i]
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
        entries.append((d, x0, x1, dz, table, tuple(g)))
    for _ in range(3):
        ops.wgrad_batch(entries)
    torch.cuda.synchronize()
    words = TLW * 2 * TLM
    buf = (ctypes.c_longlong * words)()
    lib.dlwpcs_wb_timeline.restype = ctypes.c_int
    lib.dlwpcs_wb_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = lib.dlwpcs_wb_timeline(buf, words)
    assert rc > 0, rc
    arr = np.frombuffer(buf, dtype=np.int64).reshape(TLW, 2, TLM).astype(np.float64)
    prod, cons = arr[:, 0], arr[:, 1]
    names_p = ['issue', 'wait_data', 'lds_write', 'barrier', 'epilogue']
    names_c = ['barrier', 'compute', 'epilogue', 'first_frag']
    print('phase sums per workgroup (s_memtime ticks of its first producer / first consumer wave, the last launch), mean over the 256 '
          'workgroups [min .. max]:')
    tot_p = prod[:, :5].sum(axis=1)
    tot_c = cons[:, :4].sum(axis=1)
    print('  producer total %8.0f [%8.0f .. %8.0f]' % (tot_p.mean(), tot_p.min(), tot_p.max()))
    for i, nm in enumerate(names_p):
        v = prod[:, i]
        print('    %-10s %8.0f  (%4.1f %%)  [%8.0f .. %8.0f]' % (nm, v.mean(), 100 * v.mean() / tot_p.mean(), v.min(), v.max()))
    print('  consumer total %8.0f [%8.0f .. %8.0f]' % (tot_c.mean(), tot_c.min(), tot_c.max()))
    for i, nm in enumerate(names_c):
        v = cons[:, i]
        print('    %-10s %8.0f  (%4.1f %%)  [%8.0f .. %8.0f]' % (nm, v.mean(), 100 * v.mean() / tot_c.mean(), v.min(), v.max()))


if __name__ == '__main__':
    main()
