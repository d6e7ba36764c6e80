#!/usr/bin/env python3
"""
Golden vectors for the U-Net wirings of the reference training script: the model functions `basic`, `unet`,
`unet2`, `unet3`, `unet4` are CUT OUT of /root/reference/Azure/train_cs.py (lines 233-388) at generation time and executed
verbatim; the names they use resolve to
  * cube_padding_1 / conv_2d_*  -- the REFERENCE's own CubeSpherePadding2D / CubeSphereConv2D (DLWP/custom.py, imported under
    the TF stub of gen_golden.py; torch-CPU fp64 conv2d is the arithmetic primitive), filter counts as train_cs.py:208-228;
  * relu / pooling_2 / up_sampling_2 / concatenate -- the stock Keras ops restated in numpy (SURVEY App. C:
    ReLU(negative_slope=0.1, max_value=10), AveragePooling3D((1,2,2)), UpSampling3D((1,2,2)), concatenate).
Stored per wiring: the input, every layer's weights (created with a seeded generator at first use) and the output.
Runs ONLY in the build container.  Output: tests/golden/g9_wirings.npz
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden   # noqa: E402

REF_SCRIPT = '/root/reference/Azure/train_cs.py'
N, C, BASE = 8, 3, 4
CONVS = ['conv_2d_1', 'conv_2d_1_2', 'conv_2d_1_3', 'conv_2d_2', 'conv_2d_2_2', 'conv_2d_2_3', 'conv_2d_3', 'conv_2d_3_2',
         'conv_2d_4', 'conv_2d_4_2', 'conv_2d_5', 'conv_2d_5_2', 'conv_2d_5_3', 'conv_2d_6', 'conv_2d_6_2', 'conv_2d_6_3',
         'conv_2d_7', 'conv_2d_7_2', 'conv_2d_7_3', 'conv_2d_8']


def filters(name, skip):
    """train_cs.py:208-228"""
    b = BASE
    table = {'conv_2d_1': b, 'conv_2d_1_2': b, 'conv_2d_1_3': b, 'conv_2d_2': 2 * b, 'conv_2d_2_2': 2 * b, 'conv_2d_2_3': 2 * b,
             'conv_2d_3': 4 * b, 'conv_2d_3_2': 4 * b, 'conv_2d_4': 4 * b if skip else 8 * b, 'conv_2d_4_2': 8 * b,
             'conv_2d_5': 2 * b if skip else 4 * b, 'conv_2d_5_2': 4 * b, 'conv_2d_5_3': 4 * b,
             'conv_2d_6': b if skip else 2 * b, 'conv_2d_6_2': 2 * b, 'conv_2d_6_3': 2 * b, 'conv_2d_7': b, 'conv_2d_7_2': b,
             'conv_2d_7_3': b, 'conv_2d_8': C}
    return table[name]


def main():
    ref = gen_golden._load_reference_custom()
    src = open(REF_SCRIPT).read()
    store = {}
    for wiring in ('basic', 'unet', 'unet2', 'unet3', 'unet4'):
        body = re.search(r'^def %s\(x\):\n.*?\n    return x\n' % wiring, src, re.S | re.M).group(0)
        skip = 'unet' in wiring
        rng = np.random.default_rng(900 + len(wiring) + sum(map(ord, wiring)))
        weights = {}

        class Conv(object):
            def __init__(self, name):
                self.name, self.lay = name, None

            def __call__(self, x):
                if self.lay is None:
                    f, k = filters(self.name, skip), (1 if self.name == 'conv_2d_8' else 3)
                    cin = x.shape[-1]
                    lim = np.sqrt(6.0 / (k * k * (cin + f)))
                    self.lay = ref.CubeSphereConv2D(f, k, padding='valid', data_format='channels_last', dilation_rate=1,
                                                    activation='linear', independent_north_pole=False, flip_north_pole=True)
                    w = {'equatorial_kernel': rng.uniform(-lim, lim, (k, k, cin, f)),
                         'polar_kernel': rng.uniform(-lim, lim, (k, k, cin, f)),
                         'equatorial_bias': rng.normal(0, 0.1, (f,)), 'polar_bias': rng.normal(0, 0.1, (f,))}
                    for kname, v in w.items():
                        setattr(self.lay, kname, v)
                        weights['%s/%s' % (self.name, kname)] = v.astype(np.float32)
                    # the engine holds fp32 weights: the reference run uses the same rounded values
                    for kname in w:
                        setattr(self.lay, kname, weights['%s/%s' % (self.name, kname)].astype(np.float64))
                return self.lay(x)

        def pool(v):
            B, Fc, H, W, Cc = v.shape
            return v.reshape(B, Fc, H // 2, 2, W // 2, 2, Cc).mean(axis=(3, 5))

        ns = {'cube_padding_1': ref.CubeSpherePadding2D(1, data_format='channels_last'),
              'relu': lambda v: np.where(v >= 0, np.minimum(v, 10.0), 0.1 * v),
              'pooling_2': pool, 'up_sampling_2': lambda v: v.repeat(2, axis=2).repeat(2, axis=3),
              'concatenate': lambda xs, axis=-1: np.concatenate(xs, axis=axis)}
        ns.update({n: Conv(n) for n in CONVS})
        exec(compile(body, 'train_cs.py:' + wiring, 'exec'), ns)
        n = 16 if wiring == 'unet4' else N     # three poolings: 16 -> 2 (the engine's halo tables need faces of >= 2 cells)
        x = np.random.default_rng(950 + len(wiring)).standard_normal((2, 6, n, n, C)).astype(np.float32)
        y = ns[wiring](x.astype(np.float64))
        assert y.shape == (2, 6, n, n, C), y.shape
        store[wiring + '/x'] = x
        store[wiring + '/y'] = y
        store[wiring + '/layers'] = np.array(sorted({k.split('/')[0] for k in weights}))
        for k, v in weights.items():
            store['%s/%s' % (wiring, k)] = v
        print(wiring, len(weights) // 4, 'conv layers', float(np.abs(y).max()))
    np.savez_compressed(os.path.join(HERE, 'g9_wirings.npz'), **store)


if __name__ == '__main__':
    main()
