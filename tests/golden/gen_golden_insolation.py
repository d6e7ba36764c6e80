#!/usr/bin/env python3
"""
Golden vectors for DLWP.util.insolation (the solar input of the cubed-sphere models): executes the reference's
own `day_of_year` / `insolation` (/root/reference/DLWP/util.py:300-364; the two function bodies are exec'd out of the file
because the module imports keras at import time) on a few dates and 1-d / 2-d coordinate grids and stores inputs + outputs in
tests/golden/g6_insolation.npz.  Build container only.
"""
import os
import re

import numpy as np
import pandas as pd

REF = '/root/reference/DLWP/util.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g6_insolation.npz')


def main():
    src = open(REF).read()
    ns = {'np': np, 'pd': pd}
    for fn in ('day_of_year', 'insolation'):
        m = re.search(r'^def %s\(.*?(?=^def |\Z)' % fn, src, re.S | re.M)
        exec(compile(m.group(0), 'util.py:' + fn, 'exec'), ns)
    dates = pd.to_datetime(['1995-01-01 00:00', '1995-03-21 06:00', '2003-06-21 12:00', '2012-02-29 18:00', '2016-12-31 21:00'])
    lat1, lon1 = np.linspace(-90, 90, 7), np.linspace(0, 330, 12)
    rng = np.random.default_rng(0)
    lat2 = rng.uniform(-90, 90, (6, 4, 4)); lon2 = rng.uniform(0, 360, (6, 4, 4))       # cubed-sphere style (face, h, w)
    out = {'dates': np.array([str(d) for d in dates]), 'lat1': lat1, 'lon1': lon1, 'lat2': lat2, 'lon2': lon2,
           'sol_1d': ns['insolation'](dates, lat1, lon1), 'sol_2d': ns['insolation'](dates, lat2, lon2, S=1361.),
           'sol_daily': ns['insolation'](dates, lat1, lon1, daily=True)}
    np.savez_compressed(OUT, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
