"""
Gather form of the cubed-sphere data gradient: the host-side PLAN, built numerically from the halo table, plus its statistics.

Forward (DLWP/custom.py:921-1002 behind :1198-1308, p = 1, 3x3):  xpad[f][i][j] = x[T[f, i, j]],
    z[f][y][x] = sum_{ty,tx} xpad[f][y + ty][x + tx] . W_v(f)[ty][tx]          v(f) = 0 (faces 0-3), 1 (face 4), 2 (face 5)
(variant 2 = the polar kernel with its rows reversed when flip_north_pole, or the independent north-pole kernel).  Its adjoint:
    dX[c] = sum over padded cells (f, i, j) with T[f, i, j] == c, taps (ty, tx) with 0 <= i - ty, j - tx < N of
            dz[f][i - ty][j - tx] . W_v(f)[ty][tx]^T
The engine's data-gradient kernel computes this on the padded grid today and a second launch folds the halo ring back
(pad_ring_fix / src_pair / avgpool2_bwd_masked).  In GATHER form every cell c sums its own terms:
  * MAIN terms: a plain 3x3 correlation of dz, gathered through a data-gradient halo table Tdg (the forward's table where the
    neighbour's term is exactly the one the correlation forms -- same kernel variant, same tap -- and -1 = zero elsewhere) with
    the flipped kernel of c's own face: what the forward kernel's MODE_HALO loader + MFMA loop do as they are;
  * EDGE terms: what is left, on border cells only: (source cell q, kernel variant v, tap) triples, grouped by `wid` = (v, tap) so
    that one MFMA pass serves 32 border cells with one weight fragment and a per-lane gathered dz row.
This file builds both from T, checks the decomposition against autograd of the oracle (fp64) and prints the pass statistics the
kernel design in DESIGN.md 4.8 rests on.  Test infrastructure / design tool: imports the oracle, is not imported by the product.
(The SHIPPED plan -- dlwpcs_dgrad_gather_plan, csrc/halo_table.cpp, pinned by tests/test_dgrad_gather_plan.py -- states the same
decomposition differently: the kernel keeps the forward's full halo in its LDS tile, so a border cell carries a mask of the crossing
taps to CANCEL and, per edge term, the position of its source inside the cell's own 3 x 3 window instead of a flat source index;
`-1` cells of Tdg here == the halo positions whose taps are cancelled there.)
"""
import os
import sys
from collections import Counter, defaultdict

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import cs_oracle  # noqa: E402


def variant(f):
    return 0 if f < 4 else (1 if f == 4 else 2)


def true_terms(N):
    """cell c (flat) -> Counter of (q_flat, v, ty, tx): the adjoint of halo gather + per-face 3x3 correlation."""
    T = cs_oracle.halo_table(N, 1)
    M = N + 2
    terms = defaultdict(Counter)
    for f in range(6):
        v = variant(f)
        for i in range(M):
            for j in range(M):
                c = int(T[f, i, j])
                for ty in range(3):
                    for tx in range(3):
                        y, x = i - ty, j - tx
                        if 0 <= y < N and 0 <= x < N:
                            terms[c][((f * N + y) * N + x, v, ty, tx)] += 1
    return T, terms


def build_plan(N):
    """
    Returns (Tdg, edge): Tdg (6, N+2, N+2) int32 = the data-gradient halo table (-1 = zero cell), edge = {c: [(q, v, ty, tx), ...]}
    for the border cells that have terms the main correlation does not form.
    The main correlation of output c = (f0, y, x) reads dzpad[f0][y + a][x + b] (a, b in 0..2) with the weight of the TRUE tap
    (ty, tx) = (2 - a, 2 - b) of variant v(f0).
    """
    T, terms = true_terms(N)
    M = N + 2
    Tdg = np.full((6, M, M), -1, dtype=np.int32)
    for f in range(6):
        Tdg[f, 1:N + 1, 1:N + 1] = T[f, 1:N + 1, 1:N + 1]
    # a halo cell keeps its forward source iff EVERY output that reads it finds the resulting term among its true terms
    for f0 in range(6):
        v0 = variant(f0)
        for i in range(M):
            for j in range(M):
                if 1 <= i <= N and 1 <= j <= N:
                    continue
                q = int(T[f0, i, j])
                ok = True
                for a in range(3):
                    for b in range(3):
                        y, x = i - a, j - b
                        if 0 <= y < N and 0 <= x < N:
                            c = (f0 * N + y) * N + x
                            if terms[c][(q, v0, 2 - a, 2 - b)] < 1:
                                ok = False
                if ok:
                    Tdg[f0, i, j] = q
    # main terms per cell, then the multiset difference
    edge = {}
    for f0 in range(6):
        v0 = variant(f0)
        for y in range(N):
            for x in range(N):
                c = (f0 * N + y) * N + x
                main = Counter()
                for a in range(3):
                    for b in range(3):
                        q = int(Tdg[f0, y + a, x + b])
                        if q >= 0:
                            main[(q, v0, 2 - a, 2 - b)] += 1
                extra = main - terms[c]
                assert not extra, ('main term not a true term', c, extra)
                rest = terms[c] - main
                if rest:
                    assert y in (0, N - 1) or x in (0, N - 1), 'edge terms on an interior cell'
                    edge[c] = sorted(rest.elements())
    return Tdg, edge


def apply_plan(N, Tdg, edge, dz, W):
    """dz (6, N, N, Co), W (3, 3, 3, Ci, Co) [variant][ty][tx] -> dX (6, N, N, Ci) by the gather form (numpy, fp64)."""
    Co = dz.shape[-1]
    Ci = W.shape[3]
    flat = dz.reshape(6 * N * N, Co)
    dX = np.zeros((6 * N * N, Ci))
    for f0 in range(6):
        v0 = variant(f0)
        pad = np.where(Tdg[f0][..., None] >= 0, flat[np.maximum(Tdg[f0], 0)], 0.0)      # (M, M, Co)
        for a in range(3):
            for b in range(3):
                dX[f0 * N * N:(f0 + 1) * N * N] += (pad[a:a + N, b:b + N] @ W[v0, 2 - a, 2 - b].T).reshape(N * N, Ci)
    for c, lst in edge.items():
        for (q, v, ty, tx) in lst:
            dX[c] += flat[q] @ W[v, ty, tx].T
    return dX.reshape(6, N, N, Ci)


def reference_dgrad(N, dz, W):
    """autograd of the oracle's padding + per-face correlation (fp64)."""
    import torch
    Ci = W.shape[3]
    x = torch.zeros(1, 6, N, N, Ci, dtype=torch.float64, requires_grad=True)
    xp = cs_oracle.cs_pad(x, 1)
    outs = []
    for f in range(6):
        w = torch.as_tensor(W[variant(f)])
        outs.append(cs_oracle.conv2d_tf(xp[:, f], w))
    z = torch.stack(outs, dim=1)
    z.backward(torch.as_tensor(dz)[None])
    return x.grad[0].numpy()


def stats(N, Tdg, edge, rows_per_wave=2, band_rows=8):
    T = cs_oracle.halo_table(N, 1)
    M = N + 2
    kept = int(((Tdg >= 0).sum() - 6 * N * N))
    print('N = %d: halo cells kept by the data-gradient table: %d of %d' % (N, kept, 6 * (M * M - N * N)))
    for f in range(6):
        cells = [c for c in edge if c // (N * N) == f]
        wids = Counter()
        per_cell = Counter()
        dup = 0
        for c in cells:
            per_cell[len(edge[c])] += 1
            seen = Counter((v, ty, tx) for (_, v, ty, tx) in edge[c])
            dup += sum(1 for k in seen.values() if k > 1)
            wids.update(seen.keys())
        halo_kept = int((Tdg[f] >= 0).sum() - N * N)
        print('  face %d: %3d border cells with edge terms, %2d distinct (variant, tap) passes, terms per cell %s, '
              'cells with a repeated pass %d, halo cells kept %d' % (f, len(cells), len(wids), dict(sorted(per_cell.items())), dup,
                                                                       halo_kept))
    # passes per (face, band, wave): a wave owns `rows_per_wave` rows of a band of `band_rows` rows
    worst = 0
    tot_pass = 0
    n_units = 0
    for f in range(6):
        for y0 in range(0, N, rows_per_wave):
            cells = [c for c in edge if c // (N * N) == f and y0 <= (c // N) % N < y0 + rows_per_wave]
            if not cells:
                n_units += 1
                continue
            wids = set()
            for c in cells:
                wids.update((v, ty, tx) for (_, v, ty, tx) in edge[c])
            ntile = (len(cells) + 31) // 32
            worst = max(worst, len(wids) * ntile)
            tot_pass += len(wids) * ntile
            n_units += 1
    print('  per wave (%d rows): mean %.2f, worst %d (edge M tile x pass) MFMA groups; the main loop has 9 taps x %d M tiles'
          % (rows_per_wave, tot_pass / n_units, worst, rows_per_wave * N // 32))


def main():
    rng = np.random.default_rng(0)
    for N in (4, 6, 12):
        Tdg, edge = build_plan(N)
        Ci, Co = 3, 5
        W = rng.standard_normal((3, 3, 3, Ci, Co))
        dz = rng.standard_normal((6, N, N, Co))
        got = apply_plan(N, Tdg, edge, dz, W)
        ref = reference_dgrad(N, dz, W)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print('N = %2d: gather form vs autograd of the oracle: max rel err %.2e' % (N, err))
        assert err < 1e-12
    for N in (12, 24, 48, 96):
        Tdg, edge = build_plan(N)
        stats(N, Tdg, edge)


if __name__ == '__main__':
    main()
