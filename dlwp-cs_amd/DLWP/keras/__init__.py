"""
Minimal, TensorFlow-free functional-API shim: exactly the subset of `tensorflow.keras` that the DLWP-CS scripts use
(reference Azure/train_cs.py:186-455, Tutorials/3): Input, Model, ReLU, AveragePooling3D, UpSampling3D, Concatenate /
concatenate, Reshape, Permute, Adam, callbacks.  Symbolic graph building is pure host Python; execution dispatches to the
HIP kernels of libdlwpcs.so through DLWP.ops.
"""
from . import backend, callbacks, layers, mixed_precision, models, optimizers   # noqa: F401
from .layers import Input   # noqa: F401
from .models import Model   # noqa: F401
