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
