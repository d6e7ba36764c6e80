#!/usr/bin/env python3
"""
Golden vectors for the batch feed: runs the reference's own `ArrayDataGenerator`
(/root/reference/DLWP/model/generators.py:636-1011) verbatim on small synthetic arrays and stores the batches it
produces.  Only numbers are committed (tests/golden/g5_generators.npz).  Needs /root/reference, so it runs in the build
container only; tensorflow / xarray are replaced by empty stub modules (ArrayDataGenerator touches neither beyond the
`Sequence` base class), and `numpy.int` (removed in numpy 2, used at generators.py:874-876) is aliased to `int`.
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g5_generators.npz')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    np.int = int
    tf = _stub('tensorflow')
    keras = _stub('tensorflow.keras')
    utils = _stub('tensorflow.keras.utils', Sequence=object)
    tf.keras, keras.utils = keras, utils
    _stub('xarray')
    # import the generators module without executing DLWP/model/__init__ (which pulls in keras models)
    pkg = types.ModuleType('DLWP'); pkg.__path__ = [os.path.join(REF, 'DLWP')]; sys.modules['DLWP'] = pkg
    mpkg = types.ModuleType('DLWP.model'); mpkg.__path__ = [os.path.join(REF, 'DLWP', 'model')]
    sys.modules['DLWP.model'] = mpkg
    _stub('sklearn'); _stub('sklearn.preprocessing'); _stub('sklearn.impute')
    import importlib.util
    uspec = importlib.util.spec_from_file_location('DLWP.util', os.path.join(REF, 'DLWP', 'util.py'))
    try:
        util = importlib.util.module_from_spec(uspec); sys.modules['DLWP.util'] = util; uspec.loader.exec_module(util)
    except Exception:
        # util.py imports keras at module level; only three pure-python helpers are needed by the generator
        import re
        src = open(os.path.join(REF, 'DLWP', 'util.py')).read()
        util = types.ModuleType('DLWP.util'); sys.modules['DLWP.util'] = util
        ns = {'np': np, 'numpy': np}
        for fn in ('delete_nan_samples', 'insolation', 'to_bool'):
            m = re.search(r'^def %s\(.*?(?=^def |\Z)' % fn, src, re.S | re.M)
            exec(compile(m.group(0), 'util.py:' + fn, 'exec'), ns)
            setattr(util, fn, ns[fn])
    gspec = importlib.util.spec_from_file_location('DLWP.model.generators', os.path.join(REF, 'DLWP', 'model', 'generators.py'))
    gen = importlib.util.module_from_spec(gspec); sys.modules['DLWP.model.generators'] = gen
    gspec.loader.exec_module(gen)

    class M(object):
        is_convolutional, is_recurrent, impute = True, False, False

    rng = np.random.default_rng(5)
    T, V, N = 24, 4, 4
    array = rng.standard_normal((T, V, 6, N, N)).astype(np.float32)
    insol = rng.random((T, 6, N, N)).astype(np.float32)
    const = rng.standard_normal((2, 6, N, N)).astype(np.float32)
    out = {'array': array, 'insolation': insol, 'constants': const}
    cases = {
        'a': dict(rank=3, batch_size=3, input_time_steps=2, output_time_steps=2, insolation_array=insol, channels_last=True),
        'b': dict(rank=3, batch_size=4, input_slice=slice(0, 3), output_slice=slice(1, 4), input_time_steps=2, output_time_steps=2,
                  sequence=2, interval=2, insolation_array=insol, constants=const, channels_last=True, drop_remainder=True),
        'c': dict(rank=3, batch_size=5, input_time_steps=1, output_time_steps=1, insolation_array=insol, constants=const,
                  channels_last=False),
    }
    for name, kw in cases.items():
        g = gen.ArrayDataGenerator(M(), array, **kw)
        out['%s_len' % name] = np.array(len(g))
        for prop in ('shape', 'convolution_shape', 'output_convolution_shape', 'insolation_shape', 'shape_2d',
                     'output_shape', 'dense_shape', 'output_dense_shape'):
            out['%s_%s' % (name, prop)] = np.array(getattr(g, prop))
        out['%s_n_features' % name] = np.array(g.n_features)
        p, t = g[1]
        p = p if isinstance(p, list) else [p]
        t = t if isinstance(t, list) else [t]
        out['%s_np' % name], out['%s_nt' % name] = np.array(len(p)), np.array(len(t))
        for i, a in enumerate(p):
            out['%s_p%d' % (name, i)] = np.ascontiguousarray(a)
        for i, a in enumerate(t):
            out['%s_t%d' % (name, i)] = np.ascontiguousarray(a)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: v.shape for k, v in out.items() if k.endswith(('p0', 't0', 'p1', 'p2'))})


if __name__ == '__main__':
    main()
