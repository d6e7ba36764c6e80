#!/usr/bin/env python3
"""
How well is the batched weight gradient's plan balanced?  Side build DLWPCS_LIB_TAG=tl (-DDLWPCS_WB_TL=1): per worker the measured
time of its first consumer wave (s_memtime ticks) beside the plan's segments (layer, items), and a least-squares fit of
ticks = sum over segments (a[layer] * items + b) -- the per-item cost per layer the plan SHOULD have used, relative to the mean.
usage: DLWPCS_LIB_TAG=tl python tools/wb_balance.py
"""
import ctypes
import os
import re
import struct
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
    dev = torch.device('cuda', 0)
    nat.lib()
    lib = ctypes.CDLL(nat.LIB_PATH)
    B = 32
    entries = []
    n = len(wb_bench.UNET2)
    arr = (nat.WgradItem * n)()
    for it, (N, C0, C1, up0, Cout, k, halo) in zip(arr, wb_bench.UNET2):
        n0 = N // 2 if up0 else N
        x0 = torch.randn(B, 6, n0, n0, C0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, 6, N, N, C1, device=dev).to(torch.bfloat16) if C1 else None
        No = N if halo else N - k + 1
        dz = torch.randn(B, 6, No, No, Cout, device=dev).to(torch.bfloat16)
        g = [torch.zeros(k, k, C0 + C1, Cout, device=dev), torch.zeros(k, k, C0 + C1, Cout, device=dev), None,
             torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), None]
        d = nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=halo, up0=up0, flip_north_pole=1, act=0, alpha=0.,
                         vmax=0., dtype=nat.BF16, flags=0, c0_valid=0)
        table = nat.halo_tables(N, 1, dev)[0] if halo else None
        entries.append((d, x0, x1, dz, table, tuple(g)))
        it.d = d
        it.dw_eq = it.dw_pol = it.db_eq = it.db_pol = 64
    for _ in range(3):
        ops.wgrad_batch(entries)
    torch.cuda.synchronize()
    words = TLW * 2 * TLM
    buf = (ctypes.c_longlong * words)()
    lib.dlwpcs_wb_timeline.restype = ctypes.c_int
    lib.dlwpcs_wb_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.dlwpcs_wb_timeline(buf, words) > 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(TLW, 2, TLM).astype(np.float64)
    tot = a[:, 1, :4].sum(axis=1)                  # consumer wave: the whole chain of the worker
    wait = a[:, 1, 0]
    # the plan
    L = nat.lib()
    pb, wb = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.dlwpcs_wgrad_batch_sizes(arr, n, ctypes.byref(pb), ctypes.byref(wb)) == 0
    host = (ctypes.c_char * pb.value)()
    assert L.dlwpcs_wgrad_batch_plan(arr, n, host, pb.value) == 0
    pbuf = bytes(host)
    magic, n_layers, n_segments, n_workers, n_groups, lds, off_layers, off_segs, off_groups, total = struct.unpack_from('10I', pbuf, 0)
    src = open(os.path.join(ROOT, 'include', 'dlwpcs.h')).read()
    MAXL = int(re.search(r'#define\s+DLWPCS_WGRAD_BATCH_MAX\s+(\d+)', src).group(1))
    seg_start = struct.unpack_from('257I', pbuf, 48 + 4 * (MAXL + 1))
    segs = [struct.unpack_from('6iIi', pbuf, off_segs + 32 * s) for s in range(n_segments)]
    # remap: worker w of the plan runs on workgroup blockIdx with xcd_remap(blockIdx) == w; the timeline is indexed by blockIdx
    def xcd_remap(bid, nblk=256):
        xcd, idx = bid % 8, bid // 8
        q, r = nblk // 8, nblk % 8
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        return base + idx
    A = np.zeros((n_workers, n_layers * 3 + 1))
    desc = []
    for bid in range(n_workers):
        w = xcd_remap(bid)
        parts = []
        for s in range(seg_start[w], seg_start[w + 1]):
            l, cls, cit, cot, t0, t1, slot, bias = segs[s]
            A[bid, l * 3 + min(cls, 1) * 1 + (1 if cls == 2 else 0)] += t1 - t0 if False else 0
            A[bid, l * 3 + cls] += t1 - t0
            A[bid, -1] += 1
            parts.append('L%d.c%d x%d' % (l, cls, t1 - t0))
        desc.append(' + '.join(parts))
    print('consumer-wave ticks per worker: mean %.0f  min %.0f  max %.0f  (max / mean %.3f); waiting at item barriers: mean %.0f'
          % (tot.mean(), tot.min(), tot.max(), tot.max() / tot.mean(), wait.mean()))
    order = np.argsort(tot)
    print('shortest:')
    for b in order[:6]:
        print('  wg %3d %8.0f  %s' % (b, tot[b], desc[b]))
    print('longest:')
    for b in order[-10:]:
        print('  wg %3d %8.0f  %s' % (b, tot[b], desc[b]))
    # fit per (layer) per-item cost + per-segment constant
    Al = np.zeros((n_workers, n_layers + 1))
    for l in range(n_layers):
        Al[:, l] = A[:, l * 3:(l + 1) * 3].sum(axis=1)
    Al[:, -1] = A[:, -1]
    coef, *_ = np.linalg.lstsq(Al, tot, rcond=None)
    res = tot - Al @ coef
    print('fit: ticks per item by layer (and per segment %.0f), residual rms %.0f (%.1f %% of the mean):' % (coef[-1], np.sqrt((res ** 2).mean()),
                                                                                                         100 * np.sqrt((res ** 2).mean()) / tot.mean()))
    for l in range(n_layers):
        print('  layer %2d %s: %8.1f ticks / item  (items %d)' % (l, wb_bench.UNET2[l], coef[l], int(Al[:, l].sum())))
    hist = np.histogram(tot / tot.mean(), bins=[0, 0.7, 0.8, 0.9, 0.95, 1.0, 1.03, 1.06, 1.1, 2])[0]
    print('histogram of worker time / mean, bins [0, .7, .8, .9, .95, 1, 1.03, 1.06, 1.1, 2]:', hist.tolist())


if __name__ == '__main__':
    main()
