#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the build container (needs /root/reference); never on the GPU box.

It imports the reference's own `DLWP/custom.py` against an in-memory stub of the ~15 TensorFlow/Keras symbols that
module touches, executes the reference `CubeSpherePadding2D.call` and `CubeSphereConv2D.call`
bodies verbatim (DLWP/custom.py:921-1002, :1082-1308) on seeded inputs and writes the numeric results to
`tests/golden/*.npz`.  Only numbers are stored -- no reference source text.

The arithmetic primitive inside the stub (`K.conv2d`) is torch-CPU float64 `conv2d` (TensorFlow 2.1 is not installable
here); everything else (slicing, reversal, transposition, concatenation, weight-group selection, north-pole flip) is the
reference's own code.

Usage:  python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'


# ------------------------------------------------------------------------------------------------------------------ #
# TensorFlow stub
# ------------------------------------------------------------------------------------------------------------------ #

def _install_tf_stub():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    tf = mod('tensorflow')
    tf.transpose = lambda x, perm: np.transpose(x, perm)

    compat = mod('tensorflow.compat')
    v1 = mod('tensorflow.compat.v1')
    keras_v1 = mod('tensorflow.compat.v1.keras')
    backend = mod('tensorflow.compat.v1.keras.backend')
    tf.compat = compat
    compat.v1 = v1
    v1.keras = keras_v1
    keras_v1.backend = backend

    def _conv2d(x, kernel, strides=(1, 1), padding='valid', data_format=None, dilation_rate=(1, 1)):
        xt = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64)
        if data_format == 'channels_last':
            xt = xt.permute(0, 3, 1, 2)
        w = torch.as_tensor(np.ascontiguousarray(kernel), dtype=torch.float64).permute(3, 2, 0, 1)
        if padding == 'same':
            pads = []
            for n, k, s, d in ((xt.shape[3], w.shape[3], strides[1], dilation_rate[1]),
                               (xt.shape[2], w.shape[2], strides[0], dilation_rate[0])):
                out = -(-n // s)
                total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
                pads += [total // 2, total - total // 2]
            xt = F.pad(xt, pads)
        y = F.conv2d(xt, w, None, stride=tuple(strides), dilation=tuple(dilation_rate))
        if data_format == 'channels_last':
            y = y.permute(0, 2, 3, 1)
        return y.numpy()

    def _bias_add(x, b, data_format=None):
        b = np.asarray(b)
        if data_format == 'channels_first':
            return x + b.reshape((1, -1) + (1,) * (x.ndim - 2))
        return x + b

    def _reverse(x, axes):
        if isinstance(axes, int):
            axes = [axes]
        return np.flip(x, axis=tuple(axes))

    backend.concatenate = lambda xs, axis=-1: np.concatenate(xs, axis=axis)
    backend.expand_dims = lambda x, axis=-1: np.expand_dims(x, axis)
    backend.reverse = _reverse
    backend.conv2d = _conv2d
    backend.bias_add = _bias_add
    backend.backend = lambda: 'tensorflow'
    backend.cast_to_floatx = lambda x: np.asarray(x, dtype=np.float32)

    keras = mod('tensorflow.keras')
    tf.keras = keras
    callbacks = mod('tensorflow.keras.callbacks')
    callbacks.Callback = type('Callback', (), {})
    callbacks.EarlyStopping = type('EarlyStopping', (callbacks.Callback,), {})
    layers = mod('tensorflow.keras.layers')

    def _norm_df(v):
        if v is None:
            return 'channels_last'
        v = v.lower()
        if v not in ('channels_first', 'channels_last'):
            raise ValueError(v)
        return v

    class Layer(object):
        def __init__(self, name=None, **kwargs):
            self.name = name
            self.built = False

        def __call__(self, *a, **k):
            return self.call(*a, **k)

        def get_config(self):
            return {'name': self.name}

        def add_weight(self, shape=None, initializer=None, name=None, **kw):
            raise RuntimeError('stub: set weights directly')

    class ZeroPadding3D(Layer):
        def __init__(self, padding=(1, 1, 1), data_format=None, **kwargs):
            super().__init__(**kwargs)
            self.data_format = _norm_df(data_format)
            if isinstance(padding, int):
                self.padding = ((padding, padding),) * 3
            elif hasattr(padding, '__len__'):
                if len(padding) != 3:
                    raise ValueError('`padding` should have 3 elements. Found: ' + str(padding))
                self.padding = tuple((q, q) if isinstance(q, int) else tuple(q) for q in padding)
            else:
                raise ValueError(padding)

    layers.Layer = Layer
    layers.ZeroPadding3D = ZeroPadding3D
    layers.ZeroPadding2D = type('ZeroPadding2D', (Layer,), {})
    layers.LocallyConnected2D = type('LocallyConnected2D', (Layer,), {})
    layers.Lambda = type('Lambda', (Layer,), {})
    losses = mod('tensorflow.keras.losses')
    losses.mean_absolute_error = None
    losses.mean_squared_error = None

    mod('tensorflow.python')
    mod('tensorflow.python.keras')
    utils = mod('tensorflow.python.keras.utils')
    conv_utils = mod('tensorflow.python.keras.utils.conv_utils')
    utils.conv_utils = conv_utils

    def normalize_tuple(value, n, name):
        if isinstance(value, int):
            return (value,) * n
        t = tuple(int(v) for v in value)
        if len(t) != n:
            raise ValueError(name)
        return t

    def conv_output_length(input_length, filter_size, padding, stride, dilation=1):
        if input_length is None:
            return None
        dk = filter_size + (filter_size - 1) * (dilation - 1)
        if padding == 'same':
            out = input_length
        else:
            out = input_length - dk + 1
        return (out + stride - 1) // stride

    conv_utils.normalize_tuple = normalize_tuple
    conv_utils.normalize_padding = lambda v: v.lower()
    conv_utils.normalize_data_format = _norm_df
    conv_utils.conv_output_length = conv_output_length
    mod('tensorflow.python.keras.engine')
    base_layer = mod('tensorflow.python.keras.engine.base_layer')
    base_layer.InputSpec = lambda **kw: kw

    for nm in ('activations', 'initializers', 'regularizers', 'constraints'):
        m = mod('tensorflow.keras.' + nm)
        m.get = (lambda x: None if x in (None, 'linear') else x)
        m.serialize = (lambda x: x)
        setattr(keras, nm, m)


def _load_reference_custom():
    _install_tf_stub()
    sys.path.insert(0, REFERENCE)
    import DLWP.custom as ref      # noqa
    assert ref.__file__.startswith(REFERENCE)
    return ref


# ------------------------------------------------------------------------------------------------------------------ #
# Fixtures
# ------------------------------------------------------------------------------------------------------------------ #

def main():
    ref = _load_reference_custom()
    out = {}

    # G1: gather tables, produced by pushing arange through the reference layer in BOTH data formats
    tables = {}
    for (N, p) in [(4, 1), (8, 1), (8, 2), (8, 3), (12, 1), (24, 1), (48, 1), (96, 1)]:
        idx = np.arange(6 * N * N, dtype=np.float64).reshape(1, 6, N, N, 1)
        cl = ref.CubeSpherePadding2D(p, data_format='channels_last')(idx)
        cf = ref.CubeSpherePadding2D(p, data_format='channels_first')(idx.transpose(0, 4, 1, 2, 3))
        assert cl.shape == (1, 6, N + 2 * p, N + 2 * p, 1)
        assert np.array_equal(cl[0, ..., 0], cf[0, 0]), 'reference data formats disagree'
        tables['table_N%d_p%d' % (N, p)] = cl[0, ..., 0].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, 'g1_halo_tables.npz'), **tables)

    # G2: padding of random fp32 data
    rng = np.random.default_rng(100)
    g2 = {}
    x = rng.standard_normal((2, 6, 8, 8, 3)).astype(np.float32)
    g2['x'] = x
    for p in (1, 2):
        g2['cl_p%d' % p] = ref.CubeSpherePadding2D(p, data_format='channels_last')(x)
        g2['cf_p%d' % p] = ref.CubeSpherePadding2D(p, data_format='channels_first')(
            np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3)))
    # default-constructor behaviour: padding=(1,1) must raise in the Keras base class
    try:
        ref.CubeSpherePadding2D()
        g2['default_raises'] = np.array(0)
    except ValueError:
        g2['default_raises'] = np.array(1)
    lay = ref.CubeSpherePadding2D(2, data_format='channels_last')
    g2['padding_attr_p2'] = np.array(lay.padding)
    np.savez_compressed(os.path.join(HERE, 'g2_padding.npz'), **g2)

    # G3: convolution cases
    g3 = {}
    rng = np.random.default_rng(200)
    x = rng.standard_normal((2, 6, 10, 10, 3))
    g3['x'] = x
    wk = {k: rng.standard_normal((3, 3, 3, 4)) * 0.3 for k in ('eq', 'pol', 'np')}
    bk = {k: rng.standard_normal((4,)) for k in ('eq', 'pol', 'np')}
    for k in wk:
        g3['w_' + k] = wk[k]
        g3['b_' + k] = bk[k]
    cases = []
    for flip in (True, False):
        for indep in (True, False):
            for df in ('channels_last', 'channels_first'):
                for use_bias in (True, False):
                    for dil in (1, 2):
                        cases.append(dict(flip=flip, indep=indep, df=df, use_bias=use_bias, dil=dil, stride=1,
                                          padding='valid'))
    # a few off-hot-path options (strides / same padding), channels_last only
    cases.append(dict(flip=True, indep=False, df='channels_last', use_bias=True, dil=1, stride=2, padding='valid'))
    cases.append(dict(flip=True, indep=False, df='channels_last', use_bias=True, dil=1, stride=1, padding='same'))
    cases.append(dict(flip=True, indep=False, df='channels_last', use_bias=True, dil=1, stride=2, padding='same'))
    names = []
    for c in cases:
        lay = ref.CubeSphereConv2D(4, 3, strides=c['stride'], padding=c['padding'], data_format=c['df'],
                                   dilation_rate=c['dil'], use_bias=c['use_bias'], flip_north_pole=c['flip'],
                                   independent_north_pole=c['indep'])
        lay.equatorial_kernel, lay.polar_kernel, lay.north_pole_kernel = wk['eq'], wk['pol'], wk['np']
        lay.equatorial_bias, lay.polar_bias, lay.north_pole_bias = bk['eq'], bk['pol'], bk['np']
        xin = x if c['df'] == 'channels_last' else np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3))
        y = lay(xin)
        name = 'y_flip%d_indep%d_%s_bias%d_dil%d_s%d_%s' % (c['flip'], c['indep'], 'cl' if c['df'].endswith('last') else 'cf',
                                                           c['use_bias'], c['dil'], c['stride'], c['padding'])
        assert tuple(y.shape) == tuple(lay.compute_output_shape(xin.shape)), (y.shape, name)
        g3[name] = y
        names.append(name)
    g3['case_names'] = np.array(names)
    # get_config keys of the reference layer (DLWP/custom.py:1033-1052)
    g3['config_keys'] = np.array(sorted(k for k in lay.get_config().keys()))
    np.savez_compressed(os.path.join(HERE, 'g3_conv.npz'), **g3)

    # CFG1: BASELINE config 1 -- pad(1) + 3x3 conv 4->4 on (1,6,48,48,4); x seed 0, weights seed 1
    rng0 = np.random.default_rng(0)
    rng1 = np.random.default_rng(1)
    x = rng0.standard_normal((1, 6, 48, 48, 4)).astype(np.float32)
    limit = np.sqrt(6.0 / (9 * 4 + 9 * 4))
    w_eq = rng1.uniform(-limit, limit, size=(3, 3, 4, 4)).astype(np.float32)
    w_pol = rng1.uniform(-limit, limit, size=(3, 3, 4, 4)).astype(np.float32)
    b_eq = rng1.normal(0, 0.1, size=(4,)).astype(np.float32)
    b_pol = rng1.normal(0, 0.1, size=(4,)).astype(np.float32)
    lay = ref.CubeSphereConv2D(4, 3, padding='valid', data_format='channels_last')
    lay.equatorial_kernel, lay.polar_kernel = w_eq.astype(np.float64), w_pol.astype(np.float64)
    lay.equatorial_bias, lay.polar_bias = b_eq.astype(np.float64), b_pol.astype(np.float64)
    xp = ref.CubeSpherePadding2D(1, data_format='channels_last')(x)
    y = lay(xp.astype(np.float64))
    assert y.shape == (1, 6, 48, 48, 4)
    np.savez_compressed(os.path.join(HERE, 'cfg1.npz'), x=x, w_eq=w_eq, w_pol=w_pol, b_eq=b_eq, b_pol=b_pol,
                        y=y)

    # G4: tiny unet2 forward through the REFERENCE pad/conv layers (N=8, base=4, C=3), fp64.
    # Stock Keras ops between them (ReLU(0.1,10), 2x2 mean pool, nearest upsample, concat) follow SURVEY App. C.
    sys.path.insert(0, os.path.join(HERE, '..', '..'))
    from oracle import cs_oracle as orc
    params = orc.make_unet2_params(3, 3, base=4, seed=1)
    rng = np.random.default_rng(300)
    x = rng.standard_normal((2, 6, 8, 8, 3))
    pad = ref.CubeSpherePadding2D(1, data_format='channels_last')

    def conv(xin, prm, k):
        lay = ref.CubeSphereConv2D(prm['equatorial_kernel'].shape[-1], k, padding='valid',
                                   data_format='channels_last')
        lay.equatorial_kernel, lay.polar_kernel = prm['equatorial_kernel'].numpy(), prm['polar_kernel'].numpy()
        lay.equatorial_bias, lay.polar_bias = prm['equatorial_bias'].numpy(), prm['polar_bias'].numpy()
        return lay(xin)

    def relu(v):
        return np.where(v >= 0, np.minimum(v, 10.0), 0.1 * v)

    def pool(v):
        B, Fc, H, W, C = v.shape
        return v.reshape(B, Fc, H // 2, 2, W // 2, 2, C).mean(axis=(3, 5))

    def up(v):
        return v.repeat(2, axis=2).repeat(2, axis=3)

    x0 = relu(conv(pad(x), params[0], 3))
    x0 = relu(conv(pad(x0), params[1], 3))
    x1 = pool(x0)
    x1 = relu(conv(pad(x1), params[2], 3))
    x1 = relu(conv(pad(x1), params[3], 3))
    x2 = pool(x1)
    x2 = relu(conv(pad(x2), params[4], 3))
    x2 = relu(conv(pad(x2), params[5], 3))
    xx = np.concatenate([up(x2), x1], axis=-1)
    xx = relu(conv(pad(xx), params[6], 3))
    xx = relu(conv(pad(xx), params[7], 3))
    xx = np.concatenate([up(xx), x0], axis=-1)
    xx = relu(conv(pad(xx), params[8], 3))
    xx = relu(conv(pad(xx), params[9], 3))
    y = conv(xx, params[10], 1)
    np.savez_compressed(os.path.join(HERE, 'g4_unet2_tiny.npz'), x=x, y=y, x0=x0, x1=x1)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
