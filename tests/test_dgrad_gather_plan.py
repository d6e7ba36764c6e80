"""
Gather form of the data gradient: the host plan of the C ABI (dlwpcs_dgrad_gather_plan, csrc/halo_table.cpp) against the adjoint of the
oracle's padding + per-face correlation (DLWP/custom.py:1198-1308 + :921-1002), in fp64 on the CPU.  No GPU needed: the plan is
integer work; applying it with numpy reproduces autograd to rounding.
"""
import numpy as np
import pytest
import torch

from oracle import cs_oracle


@pytest.fixture(scope='module')
def native():
    from DLWP import _native as nat
    return nat


def _variant(f):
    return 0 if f < 4 else (1 if f == 4 else 2)


def _decode(plan, N):
    M, nb = N + 2, 4 * N - 4
    o = 6 * N * N * 4
    hdr = plan[o:o + 8]
    assert hdr[0] == 0x44474731 and hdr[1] == N and hdr[5] == nb and hdr[6] == 8
    tab = plan[hdr[2]:hdr[2] + 6 * M * M].reshape(6, M, M)
    cells = plan[hdr[3]:hdr[3] + 6 * nb * 8].reshape(6, nb, 8)
    wid = plan[hdr[4]:hdr[4] + 36].reshape(6, 2, 3)
    return plan[:o].reshape(6 * N * N, 4), tab, cells, wid


def _apply(N, tab, cells, wid, dz, W):
    """what the kernel does: correlation of the halo-padded dz with the face's own flipped kernel, minus the wrong crossing taps,
    plus the neighbours' terms read out of the same halo cells"""
    Co, Ci = dz.shape[-1], W.shape[3]
    flat = dz.reshape(6 * N * N, Co)
    dX = np.zeros((6, N, N, Ci))
    for f0 in range(6):
        v0 = _variant(f0)
        pad = flat[tab[f0]]                                                          # (M, M, Co): the forward's gather
        for a in range(3):
            for b in range(3):
                dX[f0] += pad[a:a + N, b:b + N] @ W[v0, 2 - a, 2 - b].T
        for y in range(N):
            for x in range(N):
                if y not in (0, N - 1) and x not in (0, N - 1):
                    continue
                o = x if y == 0 else (N + x if y == N - 1 else 2 * N + 2 * (y - 1) + (1 if x else 0))
                rec = cells[f0, o]
                for tap in range(9):
                    if (rec[6] >> tap) & 1:
                        a, b = tap // 3, tap % 3
                        assert not (1 <= y + a <= N and 1 <= x + b <= N)           # only halo cells are ever wrong
                        dX[f0, y, x] -= pad[y + a, x + b] @ W[v0, 2 - a, 2 - b].T
                trip = wid[f0, 1 if y == N - 1 else 0]
                for s in range(6):
                    pos = rec[s]
                    if pos < 0:
                        continue
                    w = trip[s % 3]
                    assert w >= 0
                    v, a, b = w // 9, (w % 9) // 3, w % 3
                    dX[f0, y, x] += pad[y + pos // 3, x + pos % 3] @ W[v, 2 - a, 2 - b].T
    return dX


def _autograd(N, dz, W):
    Ci = W.shape[3]
    x = torch.zeros(1, 6, N, N, Ci, dtype=torch.float64, requires_grad=True)
    xp = cs_oracle.cs_pad(x, 1)
    z = torch.stack([cs_oracle.conv2d_tf(xp[:, f], torch.as_tensor(W[_variant(f)])) for f in range(6)], dim=1)
    z.backward(torch.as_tensor(dz)[None])
    return x.grad[0].numpy()


@pytest.mark.parametrize('N', [8, 12, 24])
def test_gather_plan_is_the_adjoint_of_padding_and_correlation(native, N):
    plan = native.dgrad_gather_plan_host(N)
    assert plan is not None and plan.size == native.lib().dlwpcs_dgrad_gather_plan_ints(N)
    inv, tab, cells, wid = _decode(plan, N)
    np.testing.assert_array_equal(inv, native.halo_inverse_table_host(N, 1))       # same head: serves every inv_table_dev user
    np.testing.assert_array_equal(tab, cs_oracle.halo_table(N, 1))                   # the forward's own table
    # what the round-4 design rests on: equatorial-equatorial edges need no correction (cells between the edge rows of faces 0-3
    # have no record), every other border cell cancels its crossing taps and adds <= 6 terms
    mid = cells[:4, 2 * N:]
    assert (mid[:, :, :6] == -1).all() and (mid[:, :, 6] == 0).all()
    assert (cells[4:, :, 6] != 0).all() and (cells[:, :, 7] == 0).all()
    rng = np.random.default_rng(N)
    Ci, Co = 3, 5
    W = rng.standard_normal((3, 3, 3, Ci, Co))
    dz = rng.standard_normal((6, N, N, Co))
    got = _apply(N, tab, cells, wid, dz, W)
    ref = _autograd(N, dz, W)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_gather_plan_refuses_small_faces(native):
    assert native.lib().dlwpcs_dgrad_gather_plan_ints(4) == 0
    assert native.dgrad_gather_plan_host(6) is None
