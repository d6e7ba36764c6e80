"""
The Keras stock layers the DLWP-CS U-Net wires between the custom layers (reference Azure/train_cs.py:191-199,
:391-409): Input, ReLU(negative_slope, max_value), AveragePooling3D((1,2,2)), UpSampling3D((1,2,2)), Concatenate /
concatenate, Reshape, Permute.  Semantics follow SURVEY.md Appendix C.
"""
import numpy as np
import torch

from .. import ops
from .engine import KTensor, Layer, unique_name


class InputLayer(Layer):
    def __init__(self, input_shape=None, name=None, **kwargs):
        super().__init__(name=name or unique_name('input'))
        self.batch_input_shape = (None,) + tuple(input_shape)
        self.built = True

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'batch_input_shape': self.batch_input_shape})
        return cfg

    @classmethod
    def from_config(cls, config):
        return cls(input_shape=tuple(config['batch_input_shape'][1:]), name=config.get('name'))


def Input(shape=None, name=None, batch_shape=None, dtype=None, **kwargs):
    """keras.layers.Input: a symbolic tensor of shape (None,) + shape."""
    if batch_shape is not None:
        shape = tuple(batch_shape[1:])
    if shape is None:
        raise ValueError('Please provide to Input either a `shape` or a `batch_shape` argument.')
    layer = InputLayer(input_shape=tuple(shape), name=name)
    return KTensor(layer.batch_input_shape, layer=layer, node_inputs=(), name=layer.name)


class ReLU(Layer):
    """keras.layers.ReLU(max_value=None, negative_slope=0, threshold=0)."""

    def __init__(self, max_value=None, negative_slope=0., threshold=0., **kwargs):
        super().__init__(**kwargs)
        if max_value is not None and max_value < 0.:
            raise ValueError('max_value of Relu layer cannot be negative value: ' + str(max_value))
        if negative_slope < 0.:
            raise ValueError('negative_slope of Relu layer cannot be negative value: ' + str(negative_slope))
        if threshold != 0.:
            raise NotImplementedError('ReLU(threshold != 0) is not part of the DLWP-CS hot path')
        self.max_value = None if max_value is None else float(max_value)
        self.negative_slope = float(negative_slope)
        self.threshold = float(threshold)

    def call(self, inputs):
        return ops.leaky_clip_relu(inputs, self.negative_slope, self.max_value)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'max_value': self.max_value, 'negative_slope': self.negative_slope, 'threshold': self.threshold})
        return cfg


def _norm3(v, name):
    if isinstance(v, int):
        return (v,) * 3
    v = tuple(int(q) for q in v)
    if len(v) != 3:
        raise ValueError('The `%s` argument must be a tuple of 3 integers. Received: %s' % (name, v))
    return v


class _FacePlane3D(Layer):
    """shared argument handling of the (1,2,2) pooling / upsampling layers on (B, 6, H, W, C) tensors."""

    def _check(self, size, data_format, what):
        if data_format not in (None, 'channels_last', 'channels_first'):
            raise ValueError('The `data_format` argument must be one of "channels_first", "channels_last". Received: %r'
                             % (data_format,))
        if tuple(size) != (1, 2, 2):
            raise NotImplementedError('%s: only size (1, 2, 2) -- per-face 2x2 -- is built (reference '
                                      'Azure/train_cs.py:197-198); got %r' % (what, tuple(size)))
        # channels_first (B, C, 6, H, W): a layer called on its own transposes in and out; inside a uniformly channels_first
        # DLWP.keras.Model the whole graph runs channels_last between ONE transpose at the inputs and one at the outputs
        self.data_format = data_format or 'channels_last'

    def _plane_shape(self, s, f):
        if self.data_format == 'channels_first':
            return (s[0], s[1], s[2], None if s[3] is None else f(s[3]), None if s[4] is None else f(s[4]))
        return (s[0], s[1], None if s[2] is None else f(s[2]), None if s[3] is None else f(s[3]), s[4])

    def _run(self, op, inputs):
        if self.data_format == 'channels_first':
            return ops.channels_last_to_first(op(ops.channels_first_to_last(inputs)))
        return op(inputs)


class AveragePooling3D(_FacePlane3D):
    def __init__(self, pool_size=(2, 2, 2), strides=None, padding='valid', data_format=None, **kwargs):
        super().__init__(**kwargs)
        self.pool_size = _norm3(pool_size, 'pool_size')
        self.strides = self.pool_size if strides is None else _norm3(strides, 'strides')
        self.padding = padding
        if self.strides != self.pool_size or padding != 'valid':
            raise NotImplementedError('AveragePooling3D: strides == pool_size and padding="valid" only')
        self._check(self.pool_size, data_format, 'AveragePooling3D')

    def compute_output_shape(self, s):
        return self._plane_shape(s, lambda n: n // 2)

    def call(self, inputs):
        return self._run(ops.avgpool2, inputs)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'pool_size': self.pool_size, 'strides': self.strides, 'padding': self.padding,
                    'data_format': self.data_format})
        return cfg


class UpSampling3D(_FacePlane3D):
    def __init__(self, size=(2, 2, 2), data_format=None, **kwargs):
        super().__init__(**kwargs)
        self.size = _norm3(size, 'size')
        self._check(self.size, data_format, 'UpSampling3D')

    def compute_output_shape(self, s):
        return self._plane_shape(s, lambda n: n * 2)

    def call(self, inputs):
        return self._run(ops.upsample2, inputs)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'size': self.size, 'data_format': self.data_format})
        return cfg


class Concatenate(Layer):
    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def _axis(self, ndim):
        return self.axis if self.axis >= 0 else ndim + self.axis

    def compute_output_shape(self, shapes):
        if not isinstance(shapes, (list, tuple)) or not isinstance(shapes[0], (list, tuple)):
            raise ValueError('A `Concatenate` layer should be called on a list of inputs')
        ax = self._axis(len(shapes[0]))
        out = list(shapes[0])
        for s in shapes[1:]:
            for i, (a, b) in enumerate(zip(out, s)):
                if i != ax and a is not None and b is not None and a != b:
                    raise ValueError('A `Concatenate` layer requires inputs with matching shapes except for the concat '
                                     'axis. Got inputs shapes: %s' % (list(shapes),))
            out[ax] = None if (out[ax] is None or s[ax] is None) else out[ax] + s[ax]
        return tuple(out)

    def call(self, inputs):
        ax = self._axis(inputs[0].dim())
        if ax == inputs[0].dim() - 1:
            return ops.concat_channels(list(inputs))
        return torch.cat(list(inputs), dim=ax)      # non-channel concat: pure data movement, not on the hot path

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'axis': self.axis})
        return cfg


def concatenate(inputs, axis=-1, **kwargs):
    return Concatenate(axis=axis, **kwargs)(inputs)


class Reshape(Layer):
    """keras.layers.Reshape(target_shape): batch axis untouched, one -1 allowed (Azure/train_cs.py:402,404)."""

    def __init__(self, target_shape, **kwargs):
        super().__init__(**kwargs)
        self.target_shape = tuple(int(v) for v in target_shape)

    def _resolve(self, in_shape):
        known = int(np.prod([v for v in in_shape]))
        tgt = list(self.target_shape)
        if tgt.count(-1) > 1:
            raise ValueError('Can only specify one unknown dimension.')
        if -1 in tgt:
            rest = int(np.prod([v for v in tgt if v != -1]))
            if rest == 0 or known % rest:
                raise ValueError('total size of new array must be unchanged')
            tgt[tgt.index(-1)] = known // rest
        elif int(np.prod(tgt)) != known:
            raise ValueError('total size of new array must be unchanged')
        return tuple(tgt)

    def compute_output_shape(self, s):
        return (s[0],) + self._resolve(s[1:])

    def call(self, inputs):
        return inputs.reshape((inputs.shape[0],) + self._resolve(tuple(inputs.shape[1:])))

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'target_shape': self.target_shape})
        return cfg


class Permute(Layer):
    """keras.layers.Permute(dims): 1-based permutation of the non-batch axes (Azure/train_cs.py:403)."""

    def __init__(self, dims, **kwargs):
        super().__init__(**kwargs)
        self.dims = tuple(int(d) for d in dims)
        if sorted(self.dims) != list(range(1, len(self.dims) + 1)):
            raise ValueError('Invalid permutation `dims` for Permute Layer: %s. The set of indices in `dims` must be '
                             'consecutive and start from 1.' % (self.dims,))

    def compute_output_shape(self, s):
        return (s[0],) + tuple(s[d] for d in self.dims)

    def call(self, inputs):
        return inputs.permute((0,) + self.dims).contiguous()

    def get_config(self):
        cfg = super().get_config()
        cfg.update({'dims': self.dims})
        return cfg
