"""
CPU tests: the oracle (oracle/cs_oracle.py) against the golden vectors produced by the reference's own layer code
(tests/golden/gen_golden.py).  Integer-exact for the halo gather, <=1e-12 rel (fp64) for the convolution.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('N,p', [(4, 1), (8, 1), (8, 2), (8, 3), (12, 1), (24, 1), (48, 1), (96, 1)])
def test_halo_table_matches_reference(golden_dir, N, p):
    g = _load(golden_dir, 'g1_halo_tables.npz')
    assert np.array_equal(orc.halo_table(N, p), g['table_N%d_p%d' % (N, p)])


def test_halo_table_worked_example():
    # out[0][0][0] = in[4][3][0], out[4][0][0] = in[2][0][3]  (N=4, p=1)
    T = orc.halo_table(4, 1)
    assert T[0, 0, 0] == (4 * 4 + 3) * 4 + 0
    assert T[4, 0, 0] == (2 * 4 + 0) * 4 + 3


@pytest.mark.parametrize('p', [1, 2, 3])
def test_halo_fanout_at_most_five(p):
    # every source cell is read at most 5 times (identity copy included) -> backward is a <=5-term inverse gather
    T = orc.halo_table(8, p)
    counts = np.bincount(T.reshape(-1), minlength=6 * 64)
    assert counts.min() >= 1 and counts.max() <= 5


@pytest.mark.parametrize('p', [1, 2])
def test_padding_matches_reference(golden_dir, p):
    g = _load(golden_dir, 'g2_padding.npz')
    x = g['x']
    assert np.array_equal(orc.cs_pad(x, p, 'channels_last'), g['cl_p%d' % p])
    xcf = np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3))
    assert np.array_equal(orc.cs_pad(xcf, p, 'channels_first'), g['cf_p%d' % p])


def test_conv_cases_match_reference(golden_dir):
    g = _load(golden_dir, 'g3_conv.npz')
    x = torch.tensor(g['x'])
    w = {k: torch.tensor(g['w_' + k]) for k in ('eq', 'pol', 'np')}
    b = {k: torch.tensor(g['b_' + k]) for k in ('eq', 'pol', 'np')}
    for name in g['case_names']:
        name = str(name)
        parts = name.split('_')
        flip, indep = parts[1] == 'flip1', parts[2] == 'indep1'
        df = 'channels_last' if parts[3] == 'cl' else 'channels_first'
        use_bias, dil, stride, padding = parts[4] == 'bias1', int(parts[5][3:]), int(parts[6][1:]), parts[7]
        xin = x if df == 'channels_last' else x.permute(0, 4, 1, 2, 3)
        y = orc.cs_conv2d(xin, w['eq'], w['pol'], w['np'],
                          b['eq'] if use_bias else None, b['pol'] if use_bias else None,
                          b['np'] if use_bias else None,
                          strides=(stride, stride), padding=padding, dilation=(dil, dil), data_format=df,
                          flip_north_pole=flip, independent_north_pole=indep).numpy()
        ref = g[name]
        assert y.shape == ref.shape, name
        assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max(), name


def test_cfg1_matches_reference(golden_dir):
    g = _load(golden_dir, 'cfg1.npz')
    x = torch.tensor(g['x'], dtype=torch.float64)
    xp = orc.cs_pad(x, 1, 'channels_last')
    y = orc.cs_conv2d(xp, torch.tensor(g['w_eq'], dtype=torch.float64), torch.tensor(g['w_pol'], dtype=torch.float64),
                      None, torch.tensor(g['b_eq'], dtype=torch.float64), torch.tensor(g['b_pol'], dtype=torch.float64))
    assert np.abs(y.numpy() - g['y']).max() <= 1e-12 * np.abs(g['y']).max()


def test_unet2_tiny_matches_reference_layers(golden_dir):
    g = _load(golden_dir, 'g4_unet2_tiny.npz')
    params = orc.make_unet2_params(3, 3, base=4, seed=1)
    y = orc.unet2_forward(torch.tensor(g['x']), params).numpy()
    assert np.abs(y - g['y']).max() <= 1e-12 * np.abs(g['y']).max()


def test_flip_conv_flip_equals_row_reversed_kernel():
    # property used by the HIP kernels: face 5 = conv with the kernel's rows reversed (stride 1)
    rng = np.random.default_rng(5)
    x = torch.tensor(rng.standard_normal((1, 9, 9, 3)))
    w = torch.tensor(rng.standard_normal((3, 3, 3, 2)))
    a = torch.flip(orc.conv2d_tf(torch.flip(x, dims=(1,)), w), dims=(1,))
    b = orc.conv2d_tf(x, torch.flip(w, dims=(0,)))
    assert torch.allclose(a, b, atol=1e-12)


def test_oracle_unet2_equals_the_reference_model_function(golden_dir):
    """oracle.unet2_forward against the reference's own `unet2` function executed over its own layers
    (/root/reference/Azure/train_cs.py:277-305, tests/golden/gen_golden_wirings.py)."""
    import torch
    g = np.load(os.path.join(golden_dir, 'g9_wirings.npz'))
    order = ['conv_2d_1', 'conv_2d_1_2', 'conv_2d_2', 'conv_2d_2_2', 'conv_2d_5_2', 'conv_2d_5', 'conv_2d_6_2', 'conv_2d_6',
             'conv_2d_7', 'conv_2d_7_2', 'conv_2d_8']
    assert sorted(order) == sorted(str(n) for n in g['unet2/layers'])
    params = [{k: torch.tensor(g['unet2/%s/%s' % (n, k)], dtype=torch.float64)
               for k in ('equatorial_kernel', 'polar_kernel', 'equatorial_bias', 'polar_bias')} for n in order]
    y = orc.unet2_forward(torch.tensor(g['unet2/x'], dtype=torch.float64), params).numpy()
    assert np.abs(y - g['unet2/y']).max() <= 1e-12 * np.abs(g['unet2/y']).max()
