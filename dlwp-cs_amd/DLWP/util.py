#
# Utilities of the MI355X-native DLWP-CS engine (subset of the reference's DLWP/util.py that the hot path touches).
#

"""
Model persistence and small helpers.  `save_model` / `load_model` keep the reference's file triple
(`<name>.keras` model, `<name>.pkl` wrapper, `<name>.history`; reference DLWP/util.py:127-193); the `.keras` file is
written in the engine's native format (pickled config + numpy weights) because HDF5 needs h5py + TensorFlow.
"""

import importlib
import pickle
import re
from copy import copy

import numpy as np


def make_keras_picklable():
    """No-op: DLWP.keras models are plain Python objects (the reference patched keras.Model, util.py:28-80)."""
    return None


def get_from_class(module_name, class_name):
    """Return `class_name` from module `module_name` (reference DLWP/util.py:83-94)."""
    mod = importlib.import_module(module_name)
    return getattr(mod, class_name)


def get_classes(module_name):
    """dict name -> class for every class of a module (reference DLWP/util.py:97-110)."""
    mod = importlib.import_module(module_name)
    return {k: getattr(mod, k) for k in dir(mod) if isinstance(getattr(mod, k), type)}


def get_methods(module_name):
    """dict name -> callable for every callable of a module (reference DLWP/util.py:113-124)."""
    mod = importlib.import_module(module_name)
    return {k: getattr(mod, k) for k in dir(mod) if callable(getattr(mod, k))}


def save_model(model, file_name, history=None):
    """
    Save a DLWP wrapper (an object with a `model` attribute): `<file_name>.keras` (model + weights + optimizer state),
    `<file_name>.pkl` (the wrapper without the model) and, if given, `<file_name>.history`.
    """
    net = model.base_model if hasattr(model, 'base_model') else model.model
    net.save('%s.keras' % file_name)
    model_copy = copy(model)
    model_copy.model = None
    if hasattr(model, 'base_model'):
        model_copy.base_model = None
    with open('%s.pkl' % file_name, 'wb') as f:
        pickle.dump(model_copy, f, protocol=pickle.HIGHEST_PROTOCOL)
    if history is not None:
        with open('%s.history' % file_name, 'wb') as f:
            pickle.dump(history.history, f, protocol=pickle.HIGHEST_PROTOCOL)


def load_model(file_name, history=False, custom_objects=None, gpus=1):
    """
    Load a model saved with `save_model`.  Every class of DLWP.custom is available to the loader automatically.

    :return: model [, history dict]
    """
    from .keras import models as keras_models
    with open('%s.pkl' % file_name, 'rb') as f:
        model = pickle.load(f)
    custom_objects = dict(custom_objects or {})
    custom_objects.update(get_classes('DLWP.custom'))
    loaded = keras_models.load_model('%s.keras' % file_name, custom_objects=custom_objects, compile=True)
    model.base_model = loaded
    model.model = loaded
    if gpus > 1:
        model.gpus = gpus
    if history:
        with open('%s.history' % file_name, 'rb') as f:
            h = pickle.load(f)
        return model, h
    return model


def to_bool(v):
    """Parse a command-line style boolean (reference DLWP/util.py:385-401)."""
    if isinstance(v, bool):
        return v
    if str(v).lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if str(v).lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise ValueError('Boolean value expected.')


def remove_chars(s):
    """Strip characters with unintended effects on file paths (reference DLWP/util.py:425-431)."""
    return ''.join(re.split('[$/\\\\]', s))


def is_channels_last(model):
    """True if the first layer that has a `data_format` uses channels_last (reference DLWP/util.py:434-444)."""
    for layer in model.model.layers:
        if hasattr(layer, 'data_format'):
            return layer.data_format == 'channels_last'
    return False


def insolation(dates, lat, lon, S=1., daily=False):
    """
    Approximate top-of-atmosphere solar insolation, the `solar` input channel of the DLWP-CS models (reference
    DLWP/util.py:306-364; pinned to the reference by tests/golden/g6_insolation.npz).

    :param dates: 1-d sequence of datetimes / Timestamps / datetime64
    :param lat, lon: both 1-d (a regular grid is formed) or both N-d of equal shape (e.g. cubed-sphere (6, N, N)); degrees,
        lon in 0-360
    :param S: solar constant scaling
    :param daily: True -> daily maximum (local noon) instead of the instantaneous value
    :return: float32 array (date, *grid)
    """
    import pandas as pd
    lat, lon = np.asarray(lat), np.asarray(lon)
    if lat.ndim != lon.ndim:
        raise ValueError("'lat' and 'lon' must either both be 1d or both be 2d'")
    if lat.ndim >= 2 and lat.shape != lon.shape:
        raise ValueError("shape mismatch between lat (%s) and lon (%s)" % (lat.shape, lon.shape))
    if lat.ndim == 1:
        lon, lat = np.meshgrid(lon, lat)
    # fractional day of the year (leap days ignored), float32 like the reference: the hour angle below inherits its rounding
    stamps = pd.DatetimeIndex(pd.to_datetime(list(dates)))
    start = pd.DatetimeIndex([pd.Timestamp(d.year, 1, 1) for d in stamps])
    day = ((stamps - start).total_seconds() / 86400.).values.astype(np.float32)
    day = day.reshape((-1,) + (1,) * lat.ndim)
    lon32 = lon.astype(np.float32)
    if daily:
        day = 0.5 + np.round(day)
        lon32 = np.zeros_like(lon32)
    # orbital constants of 1995
    obliquity, ecc, perihelion = np.deg2rad(23.4441), 0.016715, np.deg2rad(282.7)
    mean_lon = ecc * (1. + np.sqrt(1 - ecc ** 2.)) * np.sin(perihelion) + 2. * np.pi * (day - 80.5) / 365.
    true_lon = mean_lon + 2. * ecc * np.sin(mean_lon - perihelion)
    decl = np.arcsin(np.sin(obliquity) * np.sin(true_lon))
    hour = 2 * np.pi * (day + lon32 / 360.)
    dist = (1. - ecc ** 2.) / (1. + ecc * np.cos(true_lon - perihelion))
    phi = np.deg2rad(lat)[None, ...]
    sol = S * (np.sin(phi) * np.sin(decl) - np.cos(phi) * np.cos(decl) * np.cos(hour)) * dist ** -2.
    return np.maximum(sol, 0.).astype(np.float32)
