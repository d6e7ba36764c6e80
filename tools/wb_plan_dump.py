#!/usr/bin/env python3
"""Host-only: the plan of the batched weight gradient for the unet2 layer list -- cost share, items and modelled HBM bytes per layer.
Usage: python tools/wb_plan_dump.py [--batch 32] [--dtype bf16|f32]"""
import argparse, ctypes, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
from DLWP import _native as nat
UNET2 = [(48, 14, 0, 0, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (24, 32, 0, 0, 64, 3, 1), (24, 64, 0, 0, 64, 3, 1),
         (12, 64, 0, 0, 128, 3, 1), (12, 128, 0, 0, 64, 3, 1), (24, 64, 64, 1, 64, 3, 1), (24, 64, 0, 0, 32, 3, 1),
         (48, 32, 32, 1, 32, 3, 1), (48, 32, 0, 0, 32, 3, 1), (48, 32, 0, 0, 14, 1, 0)]
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
lib = nat.lib()
n = len(UNET2)
arr = (nat.WgradItem * n)()
es = 2 if a.dtype == 'bf16' else 4
for it, (N, C0, C1, up0, Cout, k, halo) in zip(arr, UNET2):
    it.d = nat.ConvDesc(B=a.batch, N=N, C0=C0, C1=C1, Cout=Cout, ksize=k, halo=halo, up0=up0, flip_north_pole=1, act=0, alpha=0.,
                        vmax=0., dtype=nat.BF16 if a.dtype == 'bf16' else nat.F32, flags=0, c0_valid=0)
    it.dw_eq = it.dw_pol = it.db_eq = it.db_pol = 64     # non-null markers (host plan only looks at null-ness)
pb, wb = ctypes.c_size_t(), ctypes.c_size_t()
assert lib.dlwpcs_wgrad_batch_sizes(arr, n, ctypes.byref(pb), ctypes.byref(wb)) == 0, lib.dlwpcs_last_error()
host = (ctypes.c_char * pb.value)()
assert lib.dlwpcs_wgrad_batch_plan(arr, n, host, pb.value) == 0, lib.dlwpcs_last_error()
buf = bytes(host)
magic, n_layers, n_segments, n_workers, n_groups, lds, off_layers, off_segs, off_groups, total = struct.unpack_from('10I', buf, 0)
print('segments %d workers %d lds %d ws %.1f MB' % (n_segments, n_workers, lds, wb.value / 1e6))
hdr_fixed = 40 + 8
MAXL = (len(buf) and None)
# seg_start lives at the end of the header: find it by size: header = 40 + 8 + 4*(MAXL+1) + 4*257
import re
src = open(os.path.join(ROOT, 'include', 'dlwpcs.h')).read()
MAXL = int(re.search(r'#define\s+DLWPCS_WGRAD_BATCH_MAX\s+(\d+)', src).group(1))
seg_start = struct.unpack_from('257I', buf, 48 + 4 * (MAXL + 1))
layers = [struct.unpack_from('16i5I11i2f2i', buf, off_layers + 144 * l) for l in range(n_layers)]
segs = [struct.unpack_from('6iIi', buf, off_segs + 32 * s) for s in range(n_segments)]
per = {}
for w in range(n_workers):
    for s in range(seg_start[w], seg_start[w + 1]):
        l, cls, cit, cot, t0, t1, slot, bias = segs[s]
        p = per.setdefault(l, [0, 0, set()]); p[0] += t1 - t0; p[1] += 1; p[2].add(w)
tot_alg = tot_act = 0
for l in range(n_layers):
    L = layers[l]
    B, Nin, No, C0, C1, Cin, Cout, up0, halo, KS, W2, rows, pix, nbands, pix_cap, variant = L[:16]
    CT, NT, ncit, ncot = L[21:25]
    items, nseg, ws = per[l]
    # modelled HBM bytes: per item X tile rows*W2*cin_grp (upsampled source: stored resolution rows) + dz pix*cout_grp
    x_alg = 6 * ((Nin // 2) ** 2 * C0 if up0 else Nin * Nin * C0) * es + 6 * Nin * Nin * C1 * es
    dz_alg = 6 * No * No * Cout * es
    tile = rows * W2
    x_act = 6 * nbands * (tile * C1 + (tile * C0 / (2.0 if up0 else 1.0) if up0 else tile * C0)) * es * ncot
    if up0:   # stored-resolution rows touched: ceil((rows)/2)+1 rows of W2/2+1 cells
        x_act = 6 * nbands * (tile * C1 + ((rows // 2 + 1) * (W2 // 2 + 1)) * C0) * es * ncot
    dz_act = dz_alg * ncit
    tot_alg += (x_alg + dz_alg) * B; tot_act += (x_act + dz_act) * B
    print('layer %2d N=%2d %3d+%3d->%3d v%2d CT%d NT%d pix %3d rows %2d nbands %2d | items %5d segs %3d workers %3d | alg %6.1f MB  modelled %6.1f MB (x %.2f)'
          % (l, Nin, C0, C1, Cout, variant, CT, NT, pix, rows, nbands, items, nseg, len(ws), (x_alg + dz_alg) * B / 1e6,
             (x_act + dz_act) * B / 1e6, (x_act + dz_act) / (x_alg + dz_alg)))
print('total algorithmic %.1f MB, modelled reads %.1f MB, + partial sums %.1f MB' % (tot_alg / 1e6, tot_act / 1e6, wb.value / 1e6))
