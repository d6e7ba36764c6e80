#!/usr/bin/env python3
"""
Golden vectors for N2: the reference's `TimeSeriesEstimator.__init__` and `.predict` (/root/reference/DLWP/model/extensions.py
:23-481) executed VERBATIM -- the class source is cut out of the reference file at generation time -- for the configuration
the DLWP-CS scripts run: a `DLWPFunctional` sequence model (`_n_steps` = 2 outputs) with insolation re-injection and constants,
channels_last cubed-sphere data (extensions.py:263-310 + the output assembly :384-436).

What stands in for the third-party pieces (xarray / TensorFlow are not installable here):
  * `xr.DataArray`: a ~40-line record with exactly the members that code path touches (`values`, one coordinate attribute per
    dimension, `isel`, `dims`, `coords`); coordinates are numpy datetime64 / timedelta64 arrays with a `.values` attribute;
  * the generator: an object carrying the attributes the estimator reads (`ds.sample / lat / lon / dims / variables / coords`,
    `_input_sel`, shapes, `constants`, ...) whose `generate()` returns the batches of the ENGINE's ArrayDataGenerator -- which is
    itself pinned bit-exactly to the reference generator (g5);
  * the model: a stub `DLWPFunctional` whose `predict` is the known function of tests/test_estimator.py::_StubNet;
  * `insolation`: the reference's own function, cut out of /root/reference/DLWP/util.py.
Stored: inputs, the forecast values, its dims and the f_hour / time coordinates, for steps in {3, 8, 11} x keep_time_dim.
Output: tests/golden/g10_estimator.npz.  Runs ONLY in the build container.
"""
import os
import re
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'dlwp-cs_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
REF = '/root/reference'


class Coord(np.ndarray):
    """numpy array with the `.values` attribute of an xarray coordinate; indexing keeps the type (0-d included)"""
    def __new__(cls, a):
        return np.asarray(a).view(cls)

    @property
    def values(self):
        return np.asarray(self)

    def __getitem__(self, k):
        r = np.ndarray.__getitem__(self, k)
        return r if isinstance(r, Coord) else Coord(r)


class DataArray(object):
    def __init__(self, data, coords=None, dims=None, name=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims)
        self.coords = {d: Coord(np.asarray(c)) for d, c in zip(self.dims, coords)}
        self.name = name

    def __getattr__(self, k):
        c = self.__dict__.get('coords', {})
        if k in c:
            return c[k]
        raise AttributeError(k)

    def isel(self, **kw):
        idx = [slice(None)] * self.values.ndim
        coords = dict(self.coords)
        for k, v in kw.items():
            idx[self.dims.index(k)] = v
            coords[k] = coords[k][v]
        return DataArray(self.values[tuple(idx)], [coords[d] for d in self.dims], self.dims, self.name)


class _XR(object):
    DataArray = DataArray


def main():
    import test_estimator as te
    from DLWP.keras import backend
    backend.set_device('cpu')
    from DLWP.model import DLWPFunctional as EngineFunctional
    src = open(os.path.join(REF, 'DLWP', 'model', 'extensions.py')).read()
    cls_src = re.search(r'^class TimeSeriesEstimator\(object\):.*?(?=^class |\Z)', src, re.S | re.M).group(0)
    usrc = open(os.path.join(REF, 'DLWP', 'util.py')).read()
    uns = {'np': np, 'pd': pd}
    for fn in ('day_of_year', 'insolation'):
        fsrc = re.search(r'^def %s\(.*?(?=^def |\Z)' % fn, usrc, re.S | re.M).group(0)
        exec(compile(fsrc, 'util.py:' + fn, 'exec'), uns)

    class DLWPNeuralNet(object):
        pass

    class DLWPTorchNN(object):
        pass

    class DLWPFunctional(object):
        def __init__(self, net, n_steps, time_dim):
            self.net, self._n_steps, self.time_dim = net, n_steps, time_dim

        def predict(self, p, **kw):
            return self.net.predict(p, **kw)

    class DataGenerator(object):
        pass

    class SeriesDataGenerator(object):
        pass

    class ArrayDataGenerator(object):
        pass

    ns = {'np': np, 'pd': pd, 'xr': _XR, 'warnings': warnings, 'insolation': uns['insolation'],
          'DLWPNeuralNet': DLWPNeuralNet, 'DLWPFunctional': DLWPFunctional, 'DLWPTorchNN': DLWPTorchNN,
          'DataGenerator': DataGenerator, 'SeriesDataGenerator': SeriesDataGenerator, 'ArrayDataGenerator': ArrayDataGenerator}
    exec(compile(cls_src, 'extensions.py:TimeSeriesEstimator', 'exec'), ns)
    RefEstimator = ns['TimeSeriesEstimator']

    # data: the arrays of tests/test_estimator.py, but with the REAL insolation of a lat/lon grid and 6-hourly times so that
    # the reference recomputes exactly what the generator holds
    N, V, ITS, K, T = te.N, te.V, te.ITS, te.K, te.T
    rng = np.random.default_rng(41)
    arr = rng.standard_normal((T, V, 6, N, N)).astype(np.float32)
    const = rng.standard_normal((K, 6, N, N)).astype(np.float32)
    lat = rng.uniform(-90, 90, (6, N, N))
    lon = rng.uniform(0, 360, (6, N, N))
    times = pd.date_range('2015-03-01', periods=T + 40, freq='6h').values
    sol_all = uns['insolation'](times, lat, lon).astype(np.float32)
    n_out = 2
    eng_dlwp = EngineFunctional(is_convolutional=True, time_dim=ITS)
    eng_dlwp._n_steps = n_out
    # interval = 1: g10_estimator.npz (the scripts' configuration); interval = 2: the same with strided samples -- in this branch
    # of the reference the interval only reaches the f_hour coordinate (extensions.py:263-310 never reads it)
    for interval, out_name in ((1, 'g10_estimator.npz'), (2, 'g10_estimator_interval2.npz')):
        _iv = interval
        from DLWP.model.generators import ArrayDataGenerator as EngGen
        gen = EngGen(eng_dlwp, arr, rank=3, batch_size=4, input_time_steps=ITS, output_time_steps=ITS, sequence=n_out,
                     insolation_array=sol_all[:T], constants=const, channels_last=True, interval=interval)

        class DS(object):
            dims = ('sample', 'varlev', 'x0', 'x1', 'x2')
            variables = {'predictors': None}
            sample = Coord(times[:T])
            lat_ = Coord(lat)
            lon_ = Coord(lon)
            coords = {'varlev': Coord(np.arange(V))}

            def __getitem__(self, k):
                return {'sample': self.sample}[k]
        DS.lat = DS.lat_
        DS.lon = DS.lon_

        class FakeGen(SeriesDataGenerator):
            ds = DS()
            _add_insolation = True
            _input_sel, _output_sel = {}, {}
            _input_time_steps = _output_time_steps = ITS
            _interval = _iv
            rank = 3
            channels_last = True
            _keep_time_axis = False
            constants = const
            convolution_shape = tuple(gen.convolution_shape)
            output_convolution_shape = tuple(gen.output_convolution_shape)
            shape = tuple(gen.shape)
            _n_sample = gen._n_sample

            def generate(self, samples, scale_and_impute=True):
                return gen.generate(samples)

        net = te._StubNet(n_out)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            est = RefEstimator(DLWPFunctional(net, n_out, ITS), FakeGen())
        store = {'array': arr, 'constants': const, 'lat': lat, 'lon': lon, 'times': times.astype('datetime64[ns]').astype(np.int64),
                 'insolation': sol_all, 'samples': np.array([0, 3, 5, 12])}
        names = []
        for steps in (3, 8, 11):
            for keep in (False, True):
                da = est.predict(steps, samples=list(store['samples']), keep_time_dim=keep)
                name = 's%d_k%d' % (steps, int(keep))
                names.append(name)
                store[name + '_values'] = np.asarray(da.values, dtype=np.float32)
                store[name + '_dims'] = np.array(da.dims)
                store[name + '_f_hour'] = np.asarray(da.coords['f_hour'].values, dtype=np.float64)
                store[name + '_time'] = np.asarray(da.coords['time'].values).astype('datetime64[ns]').astype(np.int64)
                store[name + '_varlev'] = np.asarray(da.coords['varlev'].values)
                print(name, da.dims, da.values.shape, da.coords['f_hour'].values[:4])
        store['names'] = np.array(names)
        np.savez_compressed(os.path.join(HERE, out_name), **store)


if __name__ == '__main__':
    main()
