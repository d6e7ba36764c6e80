"""
Symbolic tensors, the Layer protocol and initializers of the Keras-compatible shim.

The Layer protocol is the drop-in boundary of the reference's plugin interface: `__init__(**config)`,
`build(input_shape)`, `call(inputs)`, `compute_output_shape(input_shape)`, `get_config()`, `get_weights()/set_weights()`,
instances callable on symbolic or concrete tensors and reusable (shared weights, reference Azure/train_cs.py:190).
"""
import math
import re
import threading

import numpy as np
import torch

from . import backend

_uid_lock = threading.Lock()
_uids = {}


def unique_name(prefix):
    with _uid_lock:
        n = _uids.get(prefix, 0)
        _uids[prefix] = n + 1
    return prefix if n == 0 else '%s_%d' % (prefix, n)


def reset_uids():
    with _uid_lock:
        _uids.clear()


def _snake(name):
    """keras' to_snake_case: CubeSphereConv2D -> cube_sphere_conv2d, ReLU -> re_lu."""
    intermediate = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', name)
    insecure = re.sub('([a-z])([A-Z])', r'\1_\2', intermediate).lower()
    if insecure[0] != '_':
        return insecure
    return 'private' + insecure


class KTensor(object):
    """Symbolic tensor: static shape (batch = None) + the node (layer application) that produced it."""
    __slots__ = ('shape', 'layer', 'node_inputs', 'index', 'name', 'uid')
    _counter = [0]

    def __init__(self, shape, layer=None, node_inputs=(), index=0, name=None):
        self.shape = tuple(shape)
        self.layer = layer
        self.node_inputs = tuple(node_inputs)
        self.index = index
        self.name = name
        KTensor._counter[0] += 1
        self.uid = KTensor._counter[0]

    def __repr__(self):
        return '<KTensor %s shape=%s>' % (self.name, self.shape)


def is_symbolic(x):
    if isinstance(x, KTensor):
        return True
    if isinstance(x, (list, tuple)):
        return any(is_symbolic(v) for v in x)
    return False


# ------------------------------------------------------------------------------------------------------------------ #
# initializers / activations (string or callable, as `keras.initializers.get`)
# ------------------------------------------------------------------------------------------------------------------ #

def _fans(shape):
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    receptive = int(np.prod(shape[:-2]))
    return shape[-2] * receptive, shape[-1] * receptive


def glorot_uniform(shape):
    fan_in, fan_out = _fans(shape)
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return backend.rng().uniform(-limit, limit, size=shape).astype(np.float32)


def glorot_normal(shape):
    fan_in, fan_out = _fans(shape)
    return (backend.rng().standard_normal(size=shape) * math.sqrt(2.0 / (fan_in + fan_out))).astype(np.float32)


def he_uniform(shape):
    fan_in, _ = _fans(shape)
    limit = math.sqrt(6.0 / fan_in)
    return backend.rng().uniform(-limit, limit, size=shape).astype(np.float32)


def he_normal(shape):
    fan_in, _ = _fans(shape)
    return (backend.rng().standard_normal(size=shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)


_INITIALIZERS = {
    'glorot_uniform': glorot_uniform, 'glorot_normal': glorot_normal, 'he_uniform': he_uniform, 'he_normal': he_normal,
    'zeros': lambda shape: np.zeros(shape, dtype=np.float32),
    'ones': lambda shape: np.ones(shape, dtype=np.float32),
    'random_normal': lambda shape: (backend.rng().standard_normal(size=shape) * 0.05).astype(np.float32),
    'random_uniform': lambda shape: backend.rng().uniform(-0.05, 0.05, size=shape).astype(np.float32),
}


def get_initializer(spec):
    if spec is None:
        return None
    if callable(spec):
        return spec
    if isinstance(spec, dict):      # keras-serialised {'class_name': 'GlorotUniform', 'config': {...}}
        spec = _snake(spec.get('class_name', ''))
    key = str(spec).lower()
    if key not in _INITIALIZERS:
        raise ValueError('Unknown initializer: %r' % (spec,))
    fn = _INITIALIZERS[key]
    fn.__dict__.setdefault('_dlwp_name', key)
    return fn


def serialize_initializer(fn):
    if fn is None:
        return None
    return getattr(fn, '_dlwp_name', getattr(fn, '__name__', str(fn)))


def get_activation(spec):
    """`keras.activations.get`: None / 'linear' -> None (identity); 'relu' -> leaky-clip kernel with slope 0."""
    if spec is None or spec == 'linear':
        return None
    if callable(spec):
        return spec
    if spec == 'relu':
        from .. import ops

        def relu(x):
            return ops.leaky_clip_relu(x, 0.0, None)
        relu._dlwp_name = 'relu'
        return relu
    raise ValueError('Unknown activation: %r (the engine provides linear and relu; use the ReLU layer for '
                     'negative_slope / max_value)' % (spec,))


def serialize_activation(fn):
    if fn is None:
        return 'linear'
    return getattr(fn, '_dlwp_name', getattr(fn, '__name__', str(fn)))


def _passthrough_get(spec):
    """activity regularizers: accepted and stored for config round-trips; None is the only value acted upon (weight regularizers
    and constraints are served: DLWP.keras.regularizers / constraints)."""
    if spec is None:
        return None
    raise NotImplementedError('activity regularizers are not part of the DLWP-CS hot path (the reference scripts always pass '
                              'None); got %r' % (spec,))


# ------------------------------------------------------------------------------------------------------------------ #
# Layer
# ------------------------------------------------------------------------------------------------------------------ #

class Layer(object):
    def __init__(self, name=None, trainable=True, dtype=None, **kwargs):
        if kwargs:
            # keras accepts input_shape / batch_input_shape etc.; they do not affect the functional path
            allowed = {'input_shape', 'batch_input_shape', 'batch_size', 'weights', 'input_dtype'}
            unknown = set(kwargs) - allowed
            if unknown:
                raise TypeError('Keyword argument not understood: %s' % sorted(unknown)[0])
        self.name = name or unique_name(_snake(type(self).__name__))
        self.trainable = trainable
        self.dtype = dtype or backend.floatx()
        self.built = False
        self._weights = []          # list of torch tensors (views into the model's flat buffer once compiled)
        self._weight_names = []
        self._weight_rules = []     # per weight: (regularizer | None, constraint | None), see DLWP.keras.Model._weight_rules
        self.input_spec = None

    # -- weights ----------------------------------------------------------------------------------------------------
    def add_weight(self, shape=None, initializer=None, name=None, regularizer=None, constraint=None, trainable=True,
                   **kwargs):
        init = get_initializer(initializer) or _INITIALIZERS['zeros']
        value = np.asarray(init(tuple(int(s) for s in shape)), dtype=np.float32)
        t = torch.from_numpy(np.ascontiguousarray(value)).to(backend.device())
        t.requires_grad_(bool(trainable and self.trainable))
        self._weights.append(t)
        self._weight_names.append('%s/%s:0' % (self.name, name))
        self._weight_rules.append((regularizer, constraint))
        return t

    @property
    def weights(self):
        return list(self._weights)

    @property
    def trainable_weights(self):
        return [w for w in self._weights if w.requires_grad]

    def _weight_attr_names(self):
        """attribute names that alias entries of self._weights (refreshed after a rebind)."""
        return []

    def _rebind(self, new_tensors):
        """Swap the weight tensors (used by Model.compile to move them into one flat buffer)."""
        old = self._weights
        self._weights = list(new_tensors)
        for attr in self._weight_attr_names():
            cur = getattr(self, attr, None)
            for i, o in enumerate(old):
                if cur is o:
                    setattr(self, attr, self._weights[i])

    def get_weights(self):
        return [w.detach().cpu().numpy().copy() for w in self._weights]

    def set_weights(self, weights):
        if len(weights) != len(self._weights):
            raise ValueError('You called `set_weights(weights)` on layer "%s" with a weight list of length %d, but the '
                             'layer was expecting %d weights.' % (self.name, len(weights), len(self._weights)))
        with torch.no_grad():
            for w, v in zip(self._weights, weights):
                v = np.asarray(v, dtype=np.float32)
                if tuple(v.shape) != tuple(w.shape):
                    raise ValueError('Layer weight shape %s not compatible with provided weight shape %s'
                                     % (tuple(w.shape), tuple(v.shape)))
                w.copy_(torch.from_numpy(np.ascontiguousarray(v)))

    def count_params(self):
        return int(sum(w.numel() for w in self._weights))

    # -- protocol ---------------------------------------------------------------------------------------------------
    def build(self, input_shape):
        self.built = True

    def call(self, inputs, **kwargs):
        return inputs

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        return {'name': self.name, 'trainable': self.trainable, 'dtype': self.dtype}

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def __call__(self, inputs, *args, **kwargs):
        if is_symbolic(inputs):
            ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
            shapes = [t.shape for t in ins]
            shape_arg = shapes if isinstance(inputs, (list, tuple)) else shapes[0]
            if not self.built:
                self.build(shape_arg)
                self.built = True
            out_shape = self.compute_output_shape(shape_arg)
            return KTensor(out_shape, layer=self, node_inputs=ins, name=self.name)
        # concrete tensors: eager execution on the HIP device (numpy arrays are uploaded)
        if isinstance(inputs, np.ndarray):
            inputs = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).to(backend.device())
        elif isinstance(inputs, (list, tuple)):
            inputs = [torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(backend.device())
                      if isinstance(t, np.ndarray) else t for t in inputs]
        if not self.built:
            if isinstance(inputs, (list, tuple)):
                self.build([tuple(t.shape) for t in inputs])
            else:
                self.build(tuple(inputs.shape))
            self.built = True
        return self.call(inputs, *args, **kwargs)
