"""
Forecast metadata on the cubed sphere (reference DLWP/verify.py:291-325, the function the DLWP-CS evaluation scripts call on
the output of `TimeSeriesEstimator.predict` / `DLWPFunctional.predict_timeseries`).

xarray is not part of this engine: the result is the small `Forecast` record of DLWP.model.extensions (values + named
dimensions + one coordinate array per dimension, `isel`), laid out exactly like the reference's `xarray.DataArray` -- same
dimension names and order, same coordinate values, same reshape of the variable / level axes.  `meta_ds` may be an
xarray.Dataset (when xarray is installed) or any object with a `dims` mapping {name: size} and `meta_ds[name]` -> coordinate
values.
"""
import numpy as np

from .model.extensions import Forecast


def _coord(meta_ds, name):
    c = meta_ds[name]
    return np.asarray(getattr(c, 'values', c))


def add_metadata_to_forecast_cs(forecast, f_hour, meta_ds, f_hour_timedelta_type=False, channels_last=False):
    """
    Add metadata to a forecast based on the initialization times and coordinates in meta_ds, which is on a cubed sphere.

    :param forecast: ndarray: (forecast_hour, time, variable, height, width, face)
    :param f_hour: iterable: forecast hour coordinate values
    :param meta_ds: Dataset-like: contains metadata for time, variable, height, width, and face
    :param f_hour_timedelta_type: bool: if True, converts f_hour dimension into a timedelta type. May not always be
        compatible with netCDF applications.
    :param channels_last: bool: if True, assumes varlev or variable/level are last dimensions
    :return: Forecast: array with metadata (dims 'f_hour', 'time', then the data dimensions)
    """
    forecast = np.asarray(getattr(forecast, 'values', forecast))
    nf = len(f_hour)
    if f_hour_timedelta_type:
        f_hour = np.array(f_hour).astype('timedelta64[h]')
    if nf != forecast.shape[0]:
        raise ValueError("'f_hour' coordinate must have same size as the first axis of 'forecast'")
    if 'level' in meta_ds.dims:
        if channels_last:
            dims_order = ['sample', 'face', 'height', 'width', 'variable', 'level']
        else:
            dims_order = ['sample', 'variable', 'level', 'face', 'height', 'width']
        forecast = forecast.reshape([nf] + [meta_ds.dims[d] for d in dims_order])
    else:
        if channels_last:
            dims_order = ['sample', 'face', 'height', 'width', 'varlev']
        else:
            dims_order = ['sample', 'varlev', 'face', 'height', 'width']
    dims = ['f_hour'] + ['time' if d == 'sample' else d for d in dims_order]
    coords = [np.asarray(f_hour)] + [_coord(meta_ds, d) for d in dims_order]
    for d, c, n in zip(dims, coords, forecast.shape):
        if c.shape[0] != n:
            raise ValueError("conflicting sizes for dimension %r: length %d on the data but length %d on the coordinate"
                             % (d, n, c.shape[0]))
    return Forecast(forecast, dims, dict(zip(dims, coords)), name='forecast')
