"""Engine-wide settings: compute device, float type, seeding."""
import os

import numpy as np
import torch

_device = None
_rng = None


def device():
    """The HIP device of this process (one process per GPU: LOCAL_RANK), or 'cpu' for host-only graph building."""
    global _device
    if _device is None:
        if torch.cuda.is_available():
            idx = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(idx)
            _device = torch.device('cuda', idx)
        else:
            _device = torch.device('cpu')
    return _device


def set_device(dev):
    global _device
    _device = torch.device(dev)


def floatx():
    return 'float32'


# Compute dtype of the ACTIVATIONS ('float32' | 'bfloat16'); parameters, gradients and optimizer state are always fp32.
# 'bfloat16' is the engine's counterpart of the reference's mixed-precision graph rewrite (Azure/train_cs.py:429).
_compute_dtype = 'float32'


def set_compute_dtype(name):
    global _compute_dtype
    name = {'bf16': 'bfloat16', 'f32': 'float32', 'mixed_bfloat16': 'bfloat16'}.get(name, name)
    if name not in ('float32', 'bfloat16'):
        raise ValueError('compute dtype must be "float32" or "bfloat16", got %r' % (name,))
    _compute_dtype = name


def compute_dtype():
    return _compute_dtype


def torch_dtype(name=None):
    return torch.bfloat16 if (name or _compute_dtype) == 'bfloat16' else torch.float32


def image_data_format():
    return 'channels_last'


def set_seed(seed):
    """Seed the initializer stream (the reference seeds numpy + TF, Azure/train_cs.py:62-64)."""
    global _rng
    _rng = np.random.RandomState(seed)


def rng():
    # default: follow numpy's global generator so that `np.random.seed(s)` in user scripts makes runs repeatable
    return _rng if _rng is not None else np.random
