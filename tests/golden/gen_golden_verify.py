#!/usr/bin/env python3
"""
Golden vectors for `add_metadata_to_forecast_cs` (/root/reference/DLWP/verify.py:291-325): the reference function body, cut
out of the reference file at generation time, executed under the DataArray stand-in of gen_golden_estimator.py (xarray is not
installable here).  Stored per case: the output values, its dimension names and every coordinate (datetime / timedelta
coordinates as int64 + their dtype string).  Cases: with / without a separate `level` dimension x channels_last x
f_hour_timedelta_type.  Output: tests/golden/g11_verify.npz.  Runs ONLY in the build container.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = '/root/reference'
from gen_golden_estimator import Coord, DataArray, _XR   # noqa: E402


class MetaDs(object):
    """the members of an xarray.Dataset the function touches: `.dims` {name: size} and `ds[name]` -> coordinate"""
    def __init__(self, coords):
        self._c = {k: Coord(np.asarray(v)) for k, v in coords.items()}
        self.dims = {k: len(v) for k, v in self._c.items()}

    def __getitem__(self, k):
        return self._c[k]


def meta(level):
    c = {'sample': np.arange('2001-01-01T00', '2001-01-01T18', 6, dtype='datetime64[h]').astype('datetime64[ns]'),
         'face': np.arange(6), 'height': np.arange(2), 'width': np.arange(2)}
    if level:
        c['variable'] = np.array(['z', 't', 'u'])
        c['level'] = np.array([500.0, 850.0])
    else:
        c['varlev'] = np.array(['z/500', 't/850', 'u/500', 'tcwv/0', 'z/1000'])
    return MetaDs(c)


def main():
    src = open(os.path.join(REF, 'DLWP', 'verify.py')).read()
    fn_src = re.search(r'^def add_metadata_to_forecast_cs\(.*?(?=^def |\Z)', src, re.S | re.M).group(0)
    ns = {'np': np, 'xr': _XR}
    exec(compile(fn_src, 'verify.py:add_metadata_to_forecast_cs', 'exec'), ns)
    fn = ns['add_metadata_to_forecast_cs']
    out = {}
    names = []
    rng = np.random.default_rng(0)
    f_hour = np.arange(6, 6 * 5 + 1, 6)
    for level in (False, True):
        m = meta(level)
        nv = 6 if level else 5
        for cl in (False, True):
            shape = (len(f_hour), 3, 6, 2, 2, nv) if cl else (len(f_hour), 3, nv, 6, 2, 2)
            x = rng.standard_normal(shape).astype(np.float32)
            for td in (False, True):
                key = 'lev%d_cl%d_td%d' % (level, cl, td)
                names.append(key)
                r = fn(x.copy(), f_hour, m, f_hour_timedelta_type=td, channels_last=cl)
                out[key + '_in'] = x
                out[key + '_values'] = r.values
                out[key + '_dims'] = np.array(r.dims)
                for d in r.dims:
                    c = np.asarray(r.coords[d])
                    if c.dtype.kind in 'mM':
                        out[key + '_coord_%s_dtype' % d] = np.array(str(c.dtype))
                        c = c.astype(np.int64)
                    out[key + '_coord_' + d] = c
    out['cases'] = np.array(names)
    out['f_hour'] = f_hour
    np.savez_compressed(os.path.join(HERE, 'g11_verify.npz'), **out)
    print('wrote g11_verify.npz:', len(names), 'cases')


if __name__ == '__main__':
    main()
