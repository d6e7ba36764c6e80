"""
CPU tests of the host-side mirror of the reference API (no device work): layer signatures / configs / error behaviour
(b), graph building and the fusion plan, DLWPFunctional bookkeeping (8 a6), callbacks, persistence.
"""
import os
import pickle
import sys

import numpy as np
import pytest

from oracle import cs_oracle as orc


@pytest.fixture(autouse=True)
def _cpu_device():
    from DLWP.keras import backend
    backend.set_device('cpu')
    yield


def test_padding_constructor_matches_reference(golden_dir):
    from DLWP.custom import CubeSpherePadding2D
    g = np.load(os.path.join(golden_dir, 'g2_padding.npz'))
    # the reference's own default `padding=(1, 1)` raises in the keras base class; so does ours
    assert int(g['default_raises']) == 1
    with pytest.raises(ValueError, match='`padding` should have 3 elements'):
        CubeSpherePadding2D()
    lay = CubeSpherePadding2D(2, data_format='channels_last')
    assert np.array_equal(np.array(lay.padding), g['padding_attr_p2'])
    assert lay.padding == ((0, 0), (2, 2), (2, 2))
    assert CubeSpherePadding2D(1).data_format == 'channels_first'          # reference default
    cfg = lay.get_config()
    assert cfg['padding'] == ((0, 0), (2, 2), (2, 2)) and cfg['data_format'] == 'channels_last'
    again = CubeSpherePadding2D.from_config(cfg)
    assert again.padding == lay.padding
    assert lay.compute_output_shape((None, 6, 8, 8, 3)) == (None, 6, 12, 12, 3)
    assert CubeSpherePadding2D(1).compute_output_shape((None, 3, 6, 8, 8)) == (None, 3, 6, 10, 10)


def test_conv_config_keys_and_weights(golden_dir):
    from DLWP.custom import CubeSphereConv2D
    g = np.load(os.path.join(golden_dir, 'g3_conv.npz'))
    lay = CubeSphereConv2D(4, 3, data_format='channels_last', independent_north_pole=True, flip_north_pole=False)
    assert sorted(lay.get_config().keys()) == sorted(str(k) for k in g['config_keys']) + [] or \
        set(str(k) for k in g['config_keys']) <= set(lay.get_config().keys())
    for k in ('filters', 'kernel_size', 'strides', 'padding', 'data_format', 'dilation_rate', 'activation', 'use_bias',
              'flip_north_pole', 'independent_north_pole', 'kernel_initializer', 'bias_initializer',
              'kernel_regularizer', 'bias_regularizer', 'activity_regularizer', 'kernel_constraint',
              'bias_constraint'):
        assert k in lay.get_config()
    lay.build((None, 6, 10, 10, 3))
    names = [n.split('/')[1] for n in lay._weight_names]
    assert names == ['equatorial_kernel:0', 'polar_kernel:0', 'north_pole_kernel:0', 'equatorial_bias:0',
                     'polar_bias:0', 'north_pole_bias:0']
    assert tuple(lay.equatorial_kernel.shape) == (3, 3, 3, 4)
    assert lay.compute_output_shape((None, 6, 10, 10, 3)) == (None, 6, 8, 8, 4)
    assert CubeSphereConv2D(5, 3, strides=2, padding='same').compute_output_shape((None, 2, 6, 9, 9)) == (None, 5, 6, 5, 5)
    # defaults of the reference signature
    d = CubeSphereConv2D(8, 3)
    assert (d.strides, d.padding, d.data_format, d.dilation_rate, d.use_bias, d.flip_north_pole,
            d.independent_north_pole) == ((1, 1), 'valid', 'channels_first', (1, 1), True, True, False)
    with pytest.raises(ValueError, match='channel dimension'):
        CubeSphereConv2D(4, 3, data_format='channels_last').build((None, 6, 10, 10, None))
    # glorot-uniform bound
    limit = np.sqrt(6.0 / (9 * 3 + 9 * 4))
    assert np.abs(lay.get_weights()[0]).max() <= limit + 1e-7
    assert np.all(lay.get_weights()[3] == 0)
    clone = CubeSphereConv2D.from_config(lay.get_config())
    assert clone.get_config() == dict(lay.get_config(), name=clone.name) or clone.filters == 4


def _tiny_model(steps=1, solar=False, constants=False):
    from DLWP.model.cs_unet import build_cs_model
    return build_cs_model((6, 8, 8, 4), 2 if solar else 4, 'unet2', base_filter_number=4, integration_steps=steps,
                          io_time_steps=2, insolation_shape=(2, 6, 8, 8, 1) if solar else None,
                          constants_shape=(6, 8, 8, 2) if constants else None)


def test_unet2_graph_and_fusion_plan():
    model = _tiny_model()
    plan = orc.unet2_channel_plan(4, 4, 4)
    assert model.n_fused == 10                                   # every pad -> conv3x3 -> relu is one launch
    kinds = [s[0] for s in model._plan]
    # the 1x1 head; both pools feed a skip connection too -> their backward is fused with the skip gradient's add
    assert kinds.count('fused_conv') == 10 and kinds.count('layer') == 1 and kinds.count('pool_skip') == 2
    n_params = sum(2 * (k * k * ci * co + co) for ci, co, k in plan)
    assert model.count_params() == n_params
    assert model.outputs[0].shape == (None, 6, 8, 8, 4)
    fused = [s for s in model._plan if s[0] == 'fused_conv']
    assert [bool(s[5]) for s in fused] == [False] * 6 + [True, False, True, False]        # upsample folded twice
    assert [s[4] is not None for s in fused] == [False] * 6 + [True, False, True, False]  # concat folded twice


def test_reference_param_count_unet2_base32():
    from DLWP.model.cs_unet import build_cs_model
    m = build_cs_model((6, 48, 48, 14), 14, 'unet2', base_filter_number=32)
    assert m.count_params() == 673628
    m7 = build_cs_model((6, 48, 48, 7), 7, 'unet2', base_filter_number=32)
    assert m7.count_params() == 669134


def test_multistep_model_with_solar_and_constants():
    model = _tiny_model(steps=2, solar=True, constants=True)
    assert [t.layer.name for t in model.inputs] == ['main_input', 'solar_1', 'constants']
    assert len(model.outputs) == 2 and model.outputs[1].shape == (None, 6, 8, 8, 2)
    # shared layers: parameters counted once
    convs = [l for l in model.layers if type(l).__name__ == 'CubeSphereConv2D']
    assert len(convs) == 11
    cfg = model.get_config()
    from DLWP.keras.models import Model
    clone = Model.from_config(cfg)
    assert clone.count_params() == model.count_params() and clone.n_fused == model.n_fused


@pytest.mark.parametrize('name', ['basic', 'unet', 'unet2', 'unet3', 'unet4'])
def test_all_reference_wirings_build(name):
    from DLWP.model.cs_unet import build_cs_model
    m = build_cs_model((6, 16, 16, 3), 3, name, base_filter_number=4)
    assert m.outputs[0].shape == (None, 6, 16, 16, 3)
    assert m.n_fused >= 6


class _StubModel(object):
    """stands in for the compiled network: out = in + 1 (single output) or two outputs."""

    def __init__(self, n_out):
        self.outputs = [None] * n_out
        self.compiled_with = None

    def compile(self, **kw):
        self.compiled_with = kw

    def predict(self, x, **kw):
        if len(self.outputs) == 1:
            return x + 1
        return [x + 1, x + 2]


@pytest.mark.parametrize('n_out,time_dim,keep', [(1, 1, False), (1, 2, False), (1, 2, True), (2, 2, False), (2, 2, True)])
def test_predict_timeseries_bookkeeping(n_out, time_dim, keep):
    from DLWP.model import DLWPFunctional
    dlwp = DLWPFunctional(is_convolutional=True, time_dim=time_dim)
    dlwp.build_model(_StubModel(n_out), loss='mse')
    assert dlwp._n_steps == n_out and dlwp.model.compiled_with == {'loss': 'mse'}
    x = np.random.default_rng(0).standard_normal((3, 4, 6, 5, 5)).astype(np.float32)      # channels_first (B,C,6,N,N)
    stub = dlwp.model
    ref = orc.predict_timeseries_ref(stub.predict, x, 5, n_steps=n_out, time_dim=time_dim, keep_time_dim=keep)
    out = dlwp.predict_timeseries(x, 5, keep_time_dim=keep)
    assert out.shape == ref.shape and np.array_equal(out, ref)
    assert out.dtype == np.float32


def _g7_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g7_rollout.npz'))
    return g, [str(n) for n in g['names']]


def test_predict_timeseries_pinned_to_reference(golden_dir):
    """a6: DLWPFunctional.predict_timeseries against outputs of the reference's OWN method body
    (/root/reference/DLWP/model/models.py:418-460, executed by tests/golden/gen_golden_rollout.py), both layouts,
    n_steps x time_dim x keep_time_dim x time_steps; bit-exact (pure bookkeeping around a +1 model)."""
    from DLWP.model import DLWPFunctional
    g, names = _g7_cases(golden_dir)
    assert len(names) == 32
    for name in names:
        layout, n, t, k, s = name.split('_')
        n_out, time_dim, keep, steps = int(n[1:]), int(t[1:]), bool(int(k[1:])), int(s[1:])
        dlwp = DLWPFunctional(is_convolutional=True, time_dim=time_dim)
        dlwp.build_model(_StubModel(n_out), loss='mse')
        x = g[name + '_x']
        x0 = x.copy()
        out = dlwp.predict_timeseries(x, steps, keep_time_dim=keep)
        ref = g[name + '_y']
        assert out.shape == ref.shape and out.dtype == ref.dtype == np.float32, name
        assert np.array_equal(out, ref), name
        assert np.array_equal(x, x0), name                     # the caller's array is not modified (reference copies)
        # the oracle restatement used by the remaining tests agrees with the reference too
        ref2 = orc.predict_timeseries_ref(dlwp.model.predict, x, steps, n_steps=n_out, time_dim=time_dim, keep_time_dim=keep)
        assert np.array_equal(ref2, ref), name
    errs = dict(zip([str(a) for a in g['err_labels']], [str(a) for a in g['err_types']]))
    assert errs == {'list_input': 'NotImplementedError', 'zero_steps': 'ValueError'}


def test_callbacks_pinned_to_reference(golden_dir):
    """N3: EarlyStoppingMin / SaveWeightsOnEpoch traces of the reference's own on_epoch_end bodies
    (/root/reference/DLWP/custom.py:113-191, executed by tests/golden/gen_golden_rollout.py) over scripted loss sequences."""
    sys.path.insert(0, golden_dir)
    try:
        import gen_golden_rollout as gg
    finally:
        sys.path.remove(golden_dir)
    from DLWP.custom import EarlyStoppingMin, SaveWeightsOnEpoch
    g = np.load(os.path.join(golden_dir, 'g8_callbacks.npz'))
    for i, (kw, losses) in enumerate(gg.ES_CASES):
        assert np.array_equal(g['es%d_losses' % i], np.array(losses))
        cb = EarlyStoppingMin(**kw)
        model = gg.FakeModel()
        cb.set_model(model)
        cb.on_train_begin()
        trace = []
        for epoch, loss in enumerate(losses):
            model.w = [np.array([float(epoch)])]
            cb.on_epoch_end(epoch, {kw['monitor']: loss})
            trace.append([float(model.stop_training), float(cb.wait), float(cb.best), float(cb.stopped_epoch),
                          float(model.w[0][0])])
            if model.stop_training:
                break
        assert np.array_equal(np.array(trace), g['es%d_trace' % i]), (i, trace, g['es%d_trace' % i])
    assert str(g['es_bad_min_epochs']) == 'ValueError'
    j = 0
    while 'sw%d_saved' % j in g:
        interval = int(g['sw%d_interval' % j])
        cb = SaveWeightsOnEpoch('w.h5', interval=None if interval < 0 else interval)
        model = gg.FakeModel()
        model.fail_on = set(int(v) for v in g['sw%d_fail_on' % j])
        cb.set_model(model)
        raised = []
        for epoch in range(6):
            try:
                cb.on_epoch_end(epoch)
                raised.append(0)
            except OSError:
                raised.append(1)
        assert [str(a) for a in g['sw%d_saved' % j]] == model.saved, j
        assert np.array_equal(np.array(raised), g['sw%d_raised' % j]), j
        j += 1
    assert j == 4


def test_dlwpfunctional_errors_and_attributes():
    from DLWP.model import DLWPFunctional
    with pytest.raises(ValueError, match="'time_dim' must be >= 1"):
        DLWPFunctional(time_dim=0)
    dlwp = DLWPFunctional()
    for attr, val in (('is_convolutional', True), ('is_recurrent', False), ('time_dim', 1), ('impute', False),
                      ('scaler', None), ('_n_steps', 1), ('gpus', 1), ('FHW_DIMS', True)):
        assert getattr(dlwp, attr) == val
    with pytest.raises(TypeError, match="'gpus' argument must be an int"):
        dlwp.build_model(_StubModel(1), gpus=1.0)
    dlwp.build_model(_StubModel(1))
    with pytest.raises(NotImplementedError):
        dlwp.predict_timeseries([np.zeros((1, 2))], 3)
    with pytest.raises(ValueError, match='time_steps must be an int > 0'):
        dlwp.predict_timeseries(np.zeros((1, 2, 6, 4, 4), np.float32), 0)
    X, y = np.zeros(3), np.ones(3)
    assert dlwp.scaler_transform(X) is X and dlwp.scaler_transform(X, y) == (X, y)


def test_callbacks_behaviour():
    from DLWP.custom import EarlyStoppingMin, GeneratorEpochEnd, SaveWeightsOnEpoch

    class M(object):
        stop_training = False
        saved = []

        def get_weights(self):
            return ['w']

        def set_weights(self, w):
            self.restored = w

        def save_weights(self, path, save_format=None):
            self.saved.append(path)
    m = M()
    es = EarlyStoppingMin(min_epochs=2, max_epochs=6, monitor='loss', patience=1, restore_best_weights=True)
    es.set_model(m)
    es.on_train_begin()
    es.on_epoch_end(0, {'loss': 5.0})          # ignored: below min_epochs, best not tracked
    assert es.best == np.inf
    es.on_epoch_end(2, {'loss': 3.0})
    assert es.best == 3.0 and not m.stop_training
    es.on_epoch_end(3, {'loss': 4.0})          # patience 1 -> stop and restore
    assert m.stop_training and m.restored == ['w']
    with pytest.raises(ValueError):
        EarlyStoppingMin(min_epochs=-1)
    sv = SaveWeightsOnEpoch('/tmp/x.tmp', interval=2)
    sv.set_model(m)
    sv.on_epoch_end(1)
    sv.on_epoch_end(2)
    assert m.saved == ['/tmp/x.tmp', '/tmp/x.tmp.2']

    class Gen(object):
        n = 0

        def on_epoch_end(self):
            self.n += 1
    gen = Gen()
    GeneratorEpochEnd(gen).on_epoch_end(0)
    assert gen.n == 1


def test_save_load_roundtrip(tmp_path):
    from DLWP.model import DLWPFunctional
    from DLWP.util import is_channels_last, load_model, save_model
    model = _tiny_model()
    dlwp = DLWPFunctional(time_dim=2)
    dlwp.build_model(model, loss='mse', optimizer='adam', metrics=['mae'])
    base = str(tmp_path / 'm')

    class H(object):
        history = {'loss': [1.0, 0.5]}
    save_model(dlwp, base, history=H())
    for ext in ('.keras', '.pkl', '.history'):
        assert os.path.exists(base + ext)
    loaded, hist = load_model(base, history=True)
    assert hist == {'loss': [1.0, 0.5]} and loaded.time_dim == 2 and loaded._n_steps == 1
    for a, b in zip(model.get_weights(), loaded.model.get_weights()):
        assert np.array_equal(a, b)
    assert is_channels_last(loaded)
    w = str(tmp_path / 'w.h5')
    model.save_weights(w, save_format='h5')
    ws = [a + 1 for a in model.get_weights()]
    model.set_weights(ws)
    model.load_weights(w)
    assert np.array_equal(model.get_weights()[0], loaded.model.get_weights()[0])
    with pytest.raises(ValueError):
        model.set_weights(ws[:-1])


def test_model_log_names_match_keras():
    m1 = _tiny_model()
    m1.compile(loss='mse', optimizer='adam', metrics=['mae'])
    assert m1._metric_names() == ['loss', 'mean_absolute_error']
    m2 = _tiny_model(steps=2, solar=True)
    m2.compile(loss='mse', loss_weights=[0.5, 0.5], optimizer='adam', metrics=['mae'])
    assert m2._metric_names() == ['loss', 'output_loss', 'output_1_loss', 'output_mean_absolute_error',
                                  'output_1_mean_absolute_error']
    with pytest.raises(ValueError):
        m2.compile(loss='mse', loss_weights=[1.0])
    with pytest.raises(NotImplementedError):
        m1.compile(loss='mae')


def test_mixed_precision_policy_and_exports():
    """the counterpart of the reference's AMP switch (Azure/train_cs.py:429) and the DLWP.model exports"""
    from DLWP.keras import Input, Model, backend, mixed_precision
    from DLWP.keras.optimizers import Adam
    from DLWP.model import ArrayDataGenerator, DLWPFunctional, tf_data_generator   # noqa: F401
    from DLWP.model.cs_unet import CubeSphereNet
    assert mixed_precision.global_policy().name == 'float32' and backend.compute_dtype() == 'float32'
    opt = Adam()
    try:
        assert mixed_precision.enable_mixed_precision_graph_rewrite(opt) is opt
        pol = mixed_precision.global_policy()
        assert (pol.name, pol.compute_dtype, pol.variable_dtype) == ('mixed_bfloat16', 'bfloat16', 'float32')
        inp = Input(shape=(6, 8, 8, 3))
        model = Model(inputs=inp, outputs=CubeSphereNet(base_filter_number=4, output_channels=3).unet2(inp))
        assert model.compute_dtype == 'bfloat16'
        import torch
        assert all(w.dtype == torch.float32 for lay in model.layers for w in getattr(lay, 'weights', []))
    finally:
        mixed_precision.disable_mixed_precision_graph_rewrite()
    assert backend.compute_dtype() == 'float32'
    with pytest.raises(ValueError):
        mixed_precision.set_policy('mixed_float16')
    inp = Input(shape=(6, 8, 8, 3))
    assert Model(inputs=inp, outputs=CubeSphereNet(base_filter_number=4, output_channels=3).unet2(inp)).compute_dtype == 'float32'


def test_mixed_precision_rewrite_after_model_construction():
    """The reference order: Model(...) at Azure/train_cs.py:411, enable_mixed_precision_graph_rewrite(Adam()) at :429,
    compile at :430 -- the optimizer carries the switch into compile()."""
    from DLWP.keras import backend, mixed_precision
    from DLWP.keras.optimizers import Adam
    model = _tiny_model()
    assert model.compute_dtype == 'float32'
    try:
        opt = mixed_precision.enable_mixed_precision_graph_rewrite(Adam())
    finally:
        mixed_precision.disable_mixed_precision_graph_rewrite()
    assert backend.compute_dtype() == 'float32'
    model.compile(optimizer=opt, loss='mse')
    assert model.compute_dtype == 'bfloat16'
    other = _tiny_model()
    other.compile(optimizer=Adam(), loss='mse')
    assert other.compute_dtype == 'float32'


def test_saved_model_keeps_its_compute_dtype(tmp_path):
    from DLWP.keras import Input, Model, backend
    from DLWP.keras.models import clone_model, load_model
    from DLWP.model.cs_unet import CubeSphereNet
    backend.set_compute_dtype('bfloat16')
    try:
        inp = Input(shape=(6, 8, 8, 3))
        model = Model(inputs=inp, outputs=CubeSphereNet(base_filter_number=4, output_channels=3).unet2(inp))
    finally:
        backend.set_compute_dtype('float32')
    path = str(tmp_path / 'm.keras')
    model.save(path)
    assert load_model(path, compile=False).compute_dtype == 'bfloat16'
    assert clone_model(model).compute_dtype == 'bfloat16'


# ---------------------------------------------------------------------------------------------------------------------
# N2: forecast metadata on the cubed sphere (reference DLWP/verify.py:291-325), pinned to the reference function (g11)
# ---------------------------------------------------------------------------------------------------------------------

class _MetaDs(object):
    """what the function reads of an xarray.Dataset: `.dims` {name: size}, `ds[name]` -> coordinate values"""
    def __init__(self, coords):
        self._c = dict(coords)
        self.dims = {k: len(v) for k, v in self._c.items()}

    def __getitem__(self, k):
        return self._c[k]


def _verify_meta(level):
    c = {'sample': np.arange('2001-01-01T00', '2001-01-01T18', 6, dtype='datetime64[h]').astype('datetime64[ns]'),
         'face': np.arange(6), 'height': np.arange(2), 'width': np.arange(2)}
    if level:
        c['variable'] = np.array(['z', 't', 'u'])
        c['level'] = np.array([500.0, 850.0])
    else:
        c['varlev'] = np.array(['z/500', 't/850', 'u/500', 'tcwv/0', 'z/1000'])
    return _MetaDs(c)


def test_add_metadata_to_forecast_cs_matches_reference(golden_dir):
    from DLWP.verify import add_metadata_to_forecast_cs
    g = np.load(os.path.join(golden_dir, 'g11_verify.npz'))
    f_hour = g['f_hour']
    assert len(g['cases']) == 8
    for key in g['cases']:
        key = str(key)
        level, cl, td = int(key[3]), int(key[7]), int(key[11])
        r = add_metadata_to_forecast_cs(g[key + '_in'], f_hour, _verify_meta(level), f_hour_timedelta_type=bool(td),
                                        channels_last=bool(cl))
        assert tuple(r.dims) == tuple(str(d) for d in g[key + '_dims'])
        assert r.values.shape == g[key + '_values'].shape and np.array_equal(r.values, g[key + '_values'])
        for d in r.dims:
            c = np.asarray(r.coords[d])
            if key + '_coord_%s_dtype' % d in g:
                assert str(c.dtype) == str(g[key + '_coord_%s_dtype' % d])
                c = c.astype(np.int64)
            assert np.array_equal(c, g[key + '_coord_' + d]), (key, d)
        assert r.name == 'forecast'
        # the record slices like the reference's DataArray
        one = r.isel(f_hour=1, time=0)
        assert one.dims == r.dims[2:] and np.array_equal(one.values, g[key + '_values'][1, 0])
    with pytest.raises(ValueError):
        add_metadata_to_forecast_cs(g['lev0_cl1_td0_in'], f_hour[:-1], _verify_meta(0), channels_last=True)
