"""
Batch feed.  CPU: the host path of DLWP.model.generators.ArrayDataGenerator against batches produced by the
reference's own class (tests/golden/g5_generators.npz, generator script next to it).  GPU: the HBM-resident path (one
gather kernel per tensor, through the C ABI) must reproduce the host path bit for bit (fp32) / to the rounding (bf16).
"""
import os

import numpy as np
import pytest


class _Meta(object):
    is_convolutional, is_recurrent, impute = True, False, False


CASES = {
    'a': dict(rank=3, batch_size=3, input_time_steps=2, output_time_steps=2, channels_last=True),
    'b': dict(rank=3, batch_size=4, input_slice=slice(0, 3), output_slice=slice(1, 4), input_time_steps=2,
              output_time_steps=2, sequence=2, interval=2, channels_last=True, drop_remainder=True, _const=True),
    'c': dict(rank=3, batch_size=5, input_time_steps=1, output_time_steps=1, channels_last=False, _const=True),
}


def _make(g, name, **extra):
    from DLWP.model.generators import ArrayDataGenerator
    kw = dict(CASES[name])
    const = g['constants'] if kw.pop('_const', False) else None
    kw.update(extra)
    return ArrayDataGenerator(_Meta(), g['array'], insolation_array=g['insolation'], constants=const, **kw)


@pytest.mark.parametrize('name', sorted(CASES))
def test_host_generator_matches_reference_batches(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'g5_generators.npz'))
    gen = _make(g, name)
    assert len(gen) == int(g['%s_len' % name])
    for prop in ('shape', 'convolution_shape', 'output_convolution_shape', 'insolation_shape', 'shape_2d', 'output_shape',
                 'dense_shape', 'output_dense_shape'):
        assert tuple(getattr(gen, prop)) == tuple(int(v) for v in g['%s_%s' % (name, prop)]), prop
    assert gen.n_features == int(g['%s_n_features' % name])
    p, t = gen[1]
    p = p if isinstance(p, list) else [p]
    t = t if isinstance(t, list) else [t]
    assert len(p) == int(g['%s_np' % name]) and len(t) == int(g['%s_nt' % name])
    for i, a in enumerate(p):
        assert np.array_equal(a, g['%s_p%d' % (name, i)]), ('p', i)
    for i, a in enumerate(t):
        assert np.array_equal(a, g['%s_t%d' % (name, i)]), ('t', i)


def test_nan_samples_are_dropped_and_index_lists_work(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_generators.npz'))
    from DLWP.model.generators import ArrayDataGenerator
    arr = g['array'].copy()
    arr[5, 1, 2, 1, 1] = np.nan
    gen = ArrayDataGenerator(_Meta(), arr, rank=3, batch_size=8, input_slice=[0, 2], output_slice=[3], channels_last=True)
    p, t = gen.generate(np.arange(8))
    assert p.shape == (8, 6, 4, 4, 2) and t.shape == (8, 6, 4, 4, 1)      # variable 1 (the NaN) is not selected
    gen = ArrayDataGenerator(_Meta(), arr, rank=3, batch_size=8, channels_last=True)
    p, t = gen.generate(np.arange(8))
    assert p.shape[0] == 6 and t.shape[0] == 6                               # samples 4 (target) and 5 (input) dropped
    assert not np.isnan(p).any() and not np.isnan(t).any()


def test_tf_data_generator_names(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_generators.npz'))
    from DLWP.model.generators import tf_data_generator
    gen = _make(g, 'b')
    ds = tf_data_generator(gen, batch_size=4, input_names=['main_input', 'solar_1', 'constants'],
                           output_names=['output', 'output_1'])
    x, y = next(iter(ds))
    assert sorted(x) == ['constants', 'main_input', 'solar_1'] and sorted(y) == ['output', 'output_1']
    assert len(ds) == len(gen)
    with pytest.raises(ValueError):
        tf_data_generator(gen, input_names=['only_one'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_device_generator_equals_host_generator(golden_dir, name, dtype):
    import torch
    g = np.load(os.path.join(golden_dir, 'g5_generators.npz'))
    host = _make(g, name)
    dev = _make(g, name, device='cuda:0', dtype=dtype)
    for index in (0, len(host) - 1):
        ph, th = host[index]
        pd, td = dev[index]
        ph, pd = (ph if isinstance(ph, list) else [ph]), (pd if isinstance(pd, list) else [pd])
        th, td = (th if isinstance(th, list) else [th]), (td if isinstance(td, list) else [td])
        assert len(ph) == len(pd) and len(th) == len(td)
        for a, b in zip(ph, pd):
            assert b.is_cuda and b.dtype == (torch.bfloat16 if dtype == 'bfloat16' else torch.float32)
            ref = torch.tensor(a).to(b.dtype)                   # host batch rounded the same way (identity for fp32)
            assert tuple(b.shape) == a.shape and torch.equal(b.cpu(), ref)
        for a, b in zip(th, td):
            assert b.dtype == torch.float32 and torch.equal(b.cpu(), torch.tensor(a))


@pytest.mark.gpu
def test_fit_generator_from_device_batches(golden_dir):
    """End to end: DLWPFunctional.fit_generator consuming HBM-resident batches (no host copies per step)."""
    from DLWP.keras import Input, Model
    from DLWP.model import DLWPFunctional
    from DLWP.model.cs_unet import CubeSphereNet
    from DLWP.model.generators import ArrayDataGenerator
    rng = np.random.default_rng(3)
    np.random.seed(3)                  # shuffle order + weight init
    t_axis = np.linspace(0, 6, 20)[:, None, None, None, None]
    arr = (np.sin(t_axis + rng.random((1, 3, 6, 8, 8)) * 6) + 0.05 * rng.standard_normal((20, 3, 6, 8, 8))).astype(np.float32)
    dlwp = DLWPFunctional(is_convolutional=True, time_dim=2)
    gen = ArrayDataGenerator(dlwp, arr, rank=3, batch_size=4, input_time_steps=2, output_time_steps=2,
                             channels_last=True, shuffle=True, device=True)
    inp = Input(shape=gen.convolution_shape, name='main_input')
    net = CubeSphereNet(base_filter_number=4, output_channels=gen.output_convolution_shape[-1])
    dlwp.build_model(Model(inputs=inp, outputs=net.unet2(inp)), loss='mse', optimizer='adam')
    dlwp.fit_generator(gen, epochs=3, verbose=0)          # returns None, like the reference (models.py:398-406)
    losses = dlwp.model.history.history['loss']
    assert len(losses) == 3 and np.isfinite(losses).all() and losses[-1] < losses[0]


def test_insolation_matches_reference(golden_dir):
    import pandas as pd
    from DLWP.util import insolation
    g = np.load(os.path.join(golden_dir, 'g6_insolation.npz'))
    dates = pd.to_datetime(list(g['dates']))
    for got, ref in ((insolation(dates, g['lat1'], g['lon1']), g['sol_1d']),
                     (insolation(dates, g['lat2'], g['lon2'], S=1361.), g['sol_2d']),
                     (insolation(dates, g['lat1'], g['lon1'], daily=True), g['sol_daily'])):
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError):
        insolation(dates, g['lat1'], g['lon2'])
