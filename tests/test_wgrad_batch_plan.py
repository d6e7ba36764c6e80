"""
CPU tests of the host side of the batched weight gradient (dlwpcs_wgrad_batch_plan, csrc/wgrad_batch.hip): the plan is plain
host arithmetic -- every work item of every (layer, channel-tile group, face class) must be covered exactly once by the
segments, every segment must own a distinct partial-sum slot, the chains of the workers must be balanced.  No device work.
"""
import ctypes
import struct

import numpy as np
import pytest

from DLWP import _native as nat

WB_MAX = nat.WGRAD_BATCH_MAX

# (N, C0, C1, up0, Cout, k, halo): the eleven convolutions of the DLWP-CS `unet2` (Azure/train_cs.py:277-305), 14 channels
UNET2 = [(48, 14, 0, 0, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (24, 32, 0, 0, 64, 3, 1), (24, 64, 0, 0, 64, 3, 1),
         (12, 64, 0, 0, 128, 3, 1), (12, 128, 0, 0, 64, 3, 1), (24, 64, 64, 1, 64, 3, 1), (24, 64, 0, 0, 32, 3, 1),
         (48, 32, 32, 1, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (48, 32, 0, 0, 14, 1, 0)]


def _desc(B, N, C0, C1, up0, Cout, k, halo, dtype=nat.BF16, c0_valid=0):
    return nat.ConvDesc(B=B, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=halo, up0=up0, flip_north_pole=1, act=0, alpha=0.,
                        vmax=0., dtype=dtype, flags=0, c0_valid=c0_valid)


def _items(B, layers, bias=True):
    arr = (nat.WgradItem * len(layers))()
    for it, lay in zip(arr, layers):
        it.d = _desc(B, *lay)
        # only NULL-ness of the gradient pointers enters the plan
        it.dw_eq, it.dw_pol = 1, 1
        if bias:
            it.db_eq, it.db_pol = 1, 1
    return arr


def _plan(arr):
    lib = nat.lib()
    pb, wb = ctypes.c_size_t(), ctypes.c_size_t()
    nat.check(lib.dlwpcs_wgrad_batch_sizes(arr, len(arr), ctypes.byref(pb), ctypes.byref(wb)), 'sizes')
    host = (ctypes.c_char * pb.value)()
    nat.check(lib.dlwpcs_wgrad_batch_plan(arr, len(arr), host, pb.value), 'plan')
    raw = bytes(host)
    magic, n_layers, n_segs, n_workers, n_groups, lds, off_l, off_s, off_g, total = struct.unpack_from('<10I', raw, 0)
    ws_floats, = struct.unpack_from('<Q', raw, 40)
    red_first = struct.unpack_from('<%dI' % (WB_MAX + 1), raw, 48)
    seg_start = struct.unpack_from('<257I', raw, 48 + 4 * (WB_MAX + 1))
    layers = np.frombuffer(raw, dtype=np.int32, count=36 * n_layers, offset=off_l).reshape(n_layers, 36)
    segs = np.frombuffer(raw, dtype=np.int32, count=8 * n_segs, offset=off_s).reshape(n_segs, 8)
    groups = np.frombuffer(raw, dtype=np.int32, count=4 * n_groups, offset=off_g).reshape(n_groups, 4)
    return dict(magic=magic, n_layers=n_layers, n_segs=n_segs, n_workers=n_workers, lds=lds, total=total, ws_floats=ws_floats,
                ws_bytes=wb.value, red_first=red_first, seg_start=seg_start, layers=layers, segs=segs, groups=groups)


# WbLayer field positions (csrc/wgrad_batch.hip)
L_B, L_NO, L_PIX, L_NBANDS, L_CT, L_NT, L_NCIT, L_NCOT, L_GBASE, L_SLOT = 0, 2, 12, 13, 21, 22, 23, 24, 28, 30


@pytest.mark.parametrize('B', [1, 3, 32])
def test_plan_covers_every_item_once(B):
    arr = _items(B, UNET2)
    P = _plan(arr)
    assert P['magic'] == 0x57424c31 and P['n_layers'] == len(UNET2)
    assert P['lds'] <= 160 * 1024
    segs, layers = P['segs'], P['layers']
    cover = {}
    for s in segs:
        layer, cls, cit, cot, t0, t1 = [int(v) for v in s[:6]]
        assert t1 > t0
        cover.setdefault((layer, cit, cot, cls), []).append((t0, t1))
    n_groups = 0
    for l, L in enumerate(layers):
        for cit in range(L[L_NCIT]):
            for cot in range(L[L_NCOT]):
                for cls in range(3):
                    n_groups += 1
                    want = int(L[L_B]) * (4 if cls == 0 else 1) * int(L[L_NBANDS])
                    rs = sorted(cover[(l, cit, cot, cls)])
                    assert rs[0][0] == 0 and rs[-1][1] == want
                    for a, b in zip(rs, rs[1:]):
                        assert a[1] == b[0]
                    g = P['groups'][int(L[L_GBASE]) + (cit * int(L[L_NCOT]) + cot) * 3 + cls]
                    assert int(g[2]) == len(rs) and int(g[1]) == int(L[L_SLOT])
    assert n_groups == len(cover)
    # distinct, in-range partial-sum slots; the segments of a group are consecutive in memory
    slots = segs[:, 6].astype(np.int64)
    sizes = np.array([int(layers[int(s[0])][L_SLOT]) for s in segs], dtype=np.int64)
    order = np.argsort(slots)
    assert np.all(slots[order][1:] >= (slots + sizes)[order][:-1])
    assert int((slots + sizes).max()) <= P['ws_floats'] and P['ws_floats'] * 4 <= P['ws_bytes']
    # worker chains: monotone, cover all segments
    st = P['seg_start']
    assert st[0] == 0 and all(a <= b for a, b in zip(st, st[1:])) and st[256] == P['n_segs']


def test_plan_is_balanced_and_small():
    P = _plan(_items(32, UNET2))
    segs, layers = P['segs'], P['layers']
    # pixels streamed per worker as a proxy of its time: within 35 % of the mean (the cost model also weighs channels)
    st = P['seg_start']
    load = np.zeros(256)
    for w in range(256):
        for s in segs[st[w]:st[w + 1]]:
            L = layers[int(s[0])]
            load[w] += (int(s[5]) - int(s[4])) * int(L[L_PIX]) * (int(L[L_CT]) + int(L[L_NT]))
    assert load.min() > 0
    assert load.max() / load.mean() < 1.6
    # far fewer partial sums than one full-size set per CU and layer (153 MB in the per-layer scheme)
    assert P['ws_floats'] * 4 < 40e6
    assert P['n_segs'] < 256 + 3 * sum(int(L[L_NCIT]) * int(L[L_NCOT]) for L in layers) + 8


def test_unsupported_layers_are_reported():
    lib = nat.lib()
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 24, 32, 0, 0, 64, 3, 1))) == 1
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 24, 32, 0, 0, 64, 3, 1, dtype=nat.F32))) == 1   # (round 4)
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 48, 14, 0, 0, 32, 3, 1, dtype=nat.F32))) == 1   # 8-B loads
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 24, 14, 16, 0, 32, 3, 1, dtype=nat.F32))) == 0  # a vector would straddle
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 24, 7, 0, 0, 64, 3, 1, dtype=nat.F32))) == 0
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 24, 7, 0, 0, 64, 3, 1))) == 0          # odd channel count
    assert lib.dlwpcs_wgrad_batch_supported(ctypes.byref(_desc(4, 24, 8, 0, 0, 64, 3, 1, c0_valid=7))) == 1
    arr = _items(2, [(24, 7, 0, 0, 64, 3, 1)])
    pb, wb = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.dlwpcs_wgrad_batch_sizes(arr, 1, ctypes.byref(pb), ctypes.byref(wb)) == -2
    assert b'wgrad_batch' in lib.dlwpcs_last_error()
    assert lib.dlwpcs_wgrad_batch_sizes(arr, 0, ctypes.byref(pb), ctypes.byref(wb)) == -1
