"""
Labelling a cubed-sphere forecast array (behaviour of reference DLWP/verify.py:291-325, pinned by tests/golden/g11_verify.npz:
dimension names and order, coordinate values, the split of the channel axis into variable x level).

The engine has no xarray: the labelled result is the `Forecast` record of DLWP.model.extensions (values, dimension names, one
coordinate array per dimension, `isel`).  `meta_ds` is anything with a `dims` mapping {name: size} and `meta_ds[name]` ->
coordinate values -- an xarray.Dataset qualifies.
"""
import numpy as np

from .model.extensions import Forecast

_SPATIAL = ('face', 'height', 'width')


def _data_dims(split_levels, channels_last):
    """Names of the axes behind `f_hour`, in storage order; the sample axis is labelled 'time' in the result."""
    channel = ('variable', 'level') if split_levels else ('varlev',)
    return ('sample',) + (_SPATIAL + channel if channels_last else channel + _SPATIAL)


def add_metadata_to_forecast_cs(forecast, f_hour, meta_ds, f_hour_timedelta_type=False, channels_last=False):
    """
    Label `forecast` (forecast hour, initialisation time, then channels and the face / height / width axes in the order
    `channels_last` says) with the coordinates of `meta_ds`.

    When `meta_ds` has a 'level' dimension the single channel axis is unfolded into ('variable', 'level') with the sizes
    `meta_ds.dims` gives; otherwise it is 'varlev'.  `f_hour_timedelta_type` turns the forecast-hour coordinate into
    numpy timedelta64[h].  Raises ValueError when `f_hour` and the first axis disagree, or a coordinate does not fit its axis.
    """
    values = np.asarray(getattr(forecast, 'values', forecast))
    lead = np.asarray(f_hour)
    if lead.shape[0] != values.shape[0]:
        raise ValueError("'f_hour' coordinate must have same size as the first axis of 'forecast'")
    if f_hour_timedelta_type:
        lead = lead.astype('timedelta64[h]')
    split = 'level' in meta_ds.dims
    source = _data_dims(split, channels_last)
    if split:
        values = values.reshape((lead.shape[0],) + tuple(int(meta_ds.dims[d]) for d in source))
    names = ('f_hour',) + tuple('time' if d == 'sample' else d for d in source)
    coords = {'f_hour': lead}
    for name, src in zip(names[1:], source):
        c = meta_ds[src]
        coords[name] = np.asarray(getattr(c, 'values', c))
    if values.ndim != len(names):
        raise ValueError('forecast has %d axes, expected %d: %s' % (values.ndim, len(names), ', '.join(names)))
    for axis, name in enumerate(names):
        if coords[name].shape[0] != values.shape[axis]:
            raise ValueError('axis %d (%s) has %d entries but its coordinate has %d'
                             % (axis, name, values.shape[axis], coords[name].shape[0]))
    return Forecast(values, list(names), coords, name='forecast')
