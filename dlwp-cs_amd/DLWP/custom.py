#
# MI355X-native counterparts of the reference's custom Keras layers for the cubed sphere.
#

"""
Custom layer classes of the DLWP-CS hot path, same names / constructor signatures / configs / weight order as the
reference module `DLWP/custom.py`, dispatching to hand-written HIP kernels (libdlwpcs.so) instead of TensorFlow:

    CubeSpherePadding2D   reference DLWP/custom.py:1057-1308
    CubeSphereConv2D      reference DLWP/custom.py:755-1054

plus the training callbacks the reference scripts pass to `fit` (reference DLWP/custom.py:33-208), re-hosted on the
TF-free shim in `DLWP.keras`.
"""

import numpy as np

from . import ops
from ._native import ACT_LEAKY_CLIP, ACT_NONE
from .keras import constraints, engine, regularizers
from .keras.callbacks import Callback, EarlyStopping
from .keras.engine import Layer


# ==================================================================================================================== #
# helpers mirroring keras' conv_utils
# ==================================================================================================================== #

def _normalize_tuple(value, n, name):
    if isinstance(value, int):
        return (value,) * n
    try:
        value_tuple = tuple(value)
    except TypeError:
        raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) + ' integers. Received: ' +
                         str(value))
    if len(value_tuple) != n:
        raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) + ' integers. Received: ' +
                         str(value))
    for single_value in value_tuple:
        try:
            int(single_value)
        except (ValueError, TypeError):
            raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) + ' integers. Received: ' +
                             str(value) + ' including element ' + str(single_value) + ' of type' + ' ' +
                             str(type(single_value)))
    return tuple(int(v) for v in value_tuple)


def _normalize_data_format(value):
    if value is None:
        value = 'channels_last'
    data_format = value.lower()
    if data_format not in {'channels_first', 'channels_last'}:
        raise ValueError('The `data_format` argument must be one of "channels_first", "channels_last". Received: ' +
                         str(value))
    return data_format


def _normalize_padding(value):
    if isinstance(value, (list, tuple)):
        return value
    padding = value.lower()
    if padding not in {'valid', 'same', 'causal'}:
        raise ValueError('The `padding` argument must be a list/tuple or one of "valid", "same" (or "causal", only for '
                         '`Conv1D). Received: ' + str(padding))
    return padding


def _conv_output_length(input_length, filter_size, padding, stride, dilation=1):
    if input_length is None:
        return None
    dilated_filter_size = filter_size + (filter_size - 1) * (dilation - 1)
    if padding == 'same':
        output_length = input_length
    else:
        output_length = input_length - dilated_filter_size + 1
    return (output_length + stride - 1) // stride


# ==================================================================================================================== #
# Cubed-sphere layers
# ==================================================================================================================== #

class CubeSpherePadding2D(Layer):
    """
    Padding layer for 2D data on a cubed sphere: fills a halo of width `padding` around each of the 6 faces with the
    data of the neighbouring faces (rotated / reversed as the cube geometry requires).

    - input  (batch, channels, 6, height, width) for channels_first, (batch, 6, height, width, channels) for
      channels_last; faces 0-3 are equatorial, 4 and 5 the south and north polar faces
    - output has height and width increased by 2 * padding

    Same constructor, attributes and config as the reference layer (DLWP/custom.py:1072-1080), which derives from
    keras' ZeroPadding3D: `padding` must be an int or a 3-sequence (the reference's default `(1, 1)` raises exactly as
    it does there), and the face axis padding is forced to (0, 0).

    On the device the whole layer is ONE gather through a precomputed table (kernel `pad_fwd_kernel`), and when it is
    followed by a CubeSphereConv2D inside a `DLWP.keras.Model` it is not executed at all: the halo is resolved inside
    the convolution's load.
    """

    def __init__(self, padding=(1, 1), data_format='channels_first', **kwargs):
        data_format = _normalize_data_format(data_format)
        super(CubeSpherePadding2D, self).__init__(**kwargs)
        # --- keras.layers.ZeroPadding3D argument handling ---
        if isinstance(padding, int):
            padding = ((padding, padding), (padding, padding), (padding, padding))
        elif hasattr(padding, '__len__'):
            if len(padding) != 3:
                raise ValueError('`padding` should have 3 elements. Found: ' + str(padding))
            dim1_padding = _normalize_tuple(padding[0], 2, '1st entry of padding')
            dim2_padding = _normalize_tuple(padding[1], 2, '2nd entry of padding')
            dim3_padding = _normalize_tuple(padding[2], 2, '3rd entry of padding')
            padding = (dim1_padding, dim2_padding, dim3_padding)
        else:
            raise ValueError('`padding` should be either an int, a tuple of 3 ints (symmetric_dim1_pad, '
                             'symmetric_dim2_pad, symmetric_dim3_pad), or a tuple of 3 tuples of 2 ints ((left_dim1_pad, '
                             'right_dim1_pad), (left_dim2_pad, right_dim2_pad), (left_dim3_pad, right_dim2_pad)). '
                             'Found: ' + str(padding))
        self.data_format = data_format
        self.padding = ((0, 0),) + tuple(padding[1:])

    def compute_output_shape(self, input_shape):
        input_shape = tuple(input_shape)
        add = [q[0] + q[1] for q in self.padding]
        if self.data_format == 'channels_first':
            dims = [None if input_shape[2 + i] is None else input_shape[2 + i] + add[i] for i in range(3)]
            return (input_shape[0], input_shape[1]) + tuple(dims)
        dims = [None if input_shape[1 + i] is None else input_shape[1 + i] + add[i] for i in range(3)]
        return (input_shape[0],) + tuple(dims) + (input_shape[4],)

    def call(self, inputs):
        p = self.padding[1][0]
        if self.data_format == 'channels_first':
            x = ops.channels_first_to_last(inputs)
            return ops.channels_last_to_first(ops.cs_pad(x, p))
        return ops.cs_pad(inputs, p)

    def get_config(self):
        config = {'padding': self.padding, 'data_format': self.data_format}
        base_config = super(CubeSpherePadding2D, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))


class CubeSphereConv2D(Layer):
    """
    2D convolutional layer for data on a cubed sphere.

    - input (batch, channels, 6, height, width) for channels_first, (batch, 6, height, width, channels) for
      channels_last; the last two faces (4, 5) are the polar faces
    - learns one kernel + bias for the four equatorial faces and one for the polar faces; optionally a third set for
      the north pole (`independent_north_pole`); with `flip_north_pole` the north-pole face is convolved in the
      south pole's orientation (rows reversed before and after)
    - should be preceded by CubeSpherePadding2D, otherwise faces are not connected

    Same constructor signature, attributes, `get_config()` keys and weight creation order as the reference
    (DLWP/custom.py:824-919,1032-1054): equatorial_kernel, polar_kernel, [north_pole_kernel], equatorial_bias,
    polar_bias, [north_pole_bias], kernels in HWIO layout.

    Device path: the hot configuration (kernel 3 or 1, stride 1, dilation 1, 'valid') runs on the MFMA implicit-GEMM
    kernel `conv_mfma_kernel`; the pole weight sharing is a per-face choice of packed weight variant, the north-pole
    flip a row-reversed kernel variant (no data is moved).  Other options run on the generic direct kernels.
    """

    def __init__(self,
                 filters,
                 kernel_size,
                 strides=1,
                 padding='valid',
                 data_format='channels_first',
                 dilation_rate=1,
                 activation=None,
                 use_bias=True,
                 flip_north_pole=True,
                 independent_north_pole=False,
                 kernel_initializer='glorot_uniform',
                 bias_initializer='zeros',
                 kernel_regularizer=None,
                 bias_regularizer=None,
                 activity_regularizer=None,
                 kernel_constraint=None,
                 bias_constraint=None,
                 **kwargs):
        super(CubeSphereConv2D, self).__init__(**kwargs)
        self.filters = filters
        self.kernel_size = _normalize_tuple(kernel_size, 2, 'kernel_size')
        self.strides = _normalize_tuple(strides, 2, 'strides')
        self.padding = _normalize_padding(padding)
        self.data_format = _normalize_data_format(data_format)
        self.dilation_rate = _normalize_tuple(dilation_rate, 2, 'dilation_rate')
        self.activation = engine.get_activation(activation)
        self.use_bias = use_bias
        self.flip_north_pole = flip_north_pole
        self.independent_north_pole = independent_north_pole
        self.kernel_initializer = engine.get_initializer(kernel_initializer)
        self.bias_initializer = engine.get_initializer(bias_initializer)
        self.kernel_regularizer = regularizers.get(kernel_regularizer)
        self.bias_regularizer = regularizers.get(bias_regularizer)
        self.activity_regularizer = engine._passthrough_get(activity_regularizer)
        self.kernel_constraint = constraints.get(kernel_constraint)
        self.bias_constraint = constraints.get(bias_constraint)
        self.rank = 3

        self.equatorial_kernel = None
        self.equatorial_bias = None
        self.polar_kernel = None
        self.polar_bias = None
        self.north_pole_kernel = None
        self.north_pole_bias = None

    def _weight_attr_names(self):
        return ['equatorial_kernel', 'polar_kernel', 'north_pole_kernel', 'equatorial_bias', 'polar_bias',
                'north_pole_bias']

    def build(self, input_shape):
        if self.data_format == 'channels_first':
            channel_axis = 1
        else:
            channel_axis = -1
        if input_shape[channel_axis] is None:
            raise ValueError('The channel dimension of the inputs should be defined. Found `None`.')
        input_dim = int(input_shape[channel_axis])
        kernel_shape = self.kernel_size + (input_dim, self.filters)

        self.equatorial_kernel = self.add_weight(shape=kernel_shape, initializer=self.kernel_initializer,
                                                 name='equatorial_kernel', regularizer=self.kernel_regularizer,
                                                 constraint=self.kernel_constraint)
        self.polar_kernel = self.add_weight(shape=kernel_shape, initializer=self.kernel_initializer,
                                            name='polar_kernel', regularizer=self.kernel_regularizer,
                                            constraint=self.kernel_constraint)
        if self.independent_north_pole:
            self.north_pole_kernel = self.add_weight(shape=kernel_shape, initializer=self.kernel_initializer,
                                                     name='north_pole_kernel', regularizer=self.kernel_regularizer,
                                                     constraint=self.kernel_constraint)
        if self.use_bias:
            self.equatorial_bias = self.add_weight(shape=(self.filters,), initializer=self.bias_initializer,
                                                   name='equatorial_bias', regularizer=self.bias_regularizer,
                                                   constraint=self.bias_constraint)
            self.polar_bias = self.add_weight(shape=(self.filters,), initializer=self.bias_initializer,
                                              name='polar_bias', regularizer=self.bias_regularizer,
                                              constraint=self.bias_constraint)
            if self.independent_north_pole:
                self.north_pole_bias = self.add_weight(shape=(self.filters,), initializer=self.bias_initializer,
                                                       name='north_pole_bias', regularizer=self.bias_regularizer,
                                                       constraint=self.bias_constraint)
        self.input_dim = input_dim
        self.built = True

    # ------------------------------------------------------------------------------------------------------------ #
    def _is_mfma_config(self):
        k = self.kernel_size
        return (k[0] == k[1] and k[0] in (1, 3) and self.strides == (1, 1) and self.dilation_rate == (1, 1)
                and self.padding == 'valid')

    def can_fuse_halo(self, pad_width):
        """True if a preceding CubeSpherePadding2D(pad_width) can be folded into this layer's load."""
        return (self._is_mfma_config() and self.kernel_size[0] == 3 and pad_width == 1 and self.activation is None)

    def fused_call(self, src0, src1=None, up0=False, halo=True, act=ACT_NONE, alpha=0.0, vmax=0.0, premask0=None,
                   premask1=None, dy_premasked=False, defer_ring0=False, want_pool=False, out_padded=False):
        """pad -> conv (-> ReLU) with optional upsample/concat on the input side, as one kernel (channels_last).
        premask0 / premask1 / dy_premasked: the pre-masked gradient convention of the training step (ops._CSConv)."""
        return ops.cs_conv(src0, self.equatorial_kernel, self.polar_kernel, self.north_pole_kernel,
                           self.equatorial_bias, self.polar_bias, self.north_pole_bias, src1=src1,
                           ksize=self.kernel_size[0], halo=halo, up0=up0, flip_north_pole=self.flip_north_pole,
                           act=act, alpha=alpha, vmax=vmax, premask0=premask0, premask1=premask1,
                           dy_premasked=dy_premasked, defer_ring0=defer_ring0, want_pool=want_pool, out_padded=out_padded)

    def call(self, inputs, channels_last_io=False, **kwargs):
        """channels_last_io (DLWP.keras.Model running a channels_first graph channels_last inside): the tensor arrives and
        leaves channels_last whatever the layer's data_format says."""
        channels_first = self.data_format == 'channels_first' and not channels_last_io
        x = ops.channels_first_to_last(inputs) if channels_first else inputs
        if self._is_mfma_config():
            outputs = ops.cs_conv(x, self.equatorial_kernel, self.polar_kernel, self.north_pole_kernel,
                                  self.equatorial_bias, self.polar_bias, self.north_pole_bias,
                                  ksize=self.kernel_size[0], halo=False, flip_north_pole=self.flip_north_pole)
        else:
            outputs = ops.cs_gconv(x, self.equatorial_kernel, self.polar_kernel, self.north_pole_kernel,
                                   self.equatorial_bias, self.polar_bias, self.north_pole_bias,
                                   strides=self.strides, padding=self.padding, dilation=self.dilation_rate,
                                   flip_north_pole=self.flip_north_pole)
        if channels_first:
            outputs = ops.channels_last_to_first(outputs)
        if self.activation is not None:
            return self.activation(outputs)
        return outputs

    def compute_output_shape(self, input_shape):
        if self.data_format == 'channels_last':
            # batch, face, height, width, ...
            space = input_shape[2:4]
        else:
            # batch, channels, face, height, width
            space = input_shape[-2:]
        new_space = []
        for i in range(len(space)):
            new_space.append(_conv_output_length(space[i], self.kernel_size[i], padding=self.padding,
                                                 stride=self.strides[i], dilation=self.dilation_rate[i]))
        if self.data_format == 'channels_last':
            return (input_shape[0], 6) + tuple(new_space) + (self.filters,)
        return (input_shape[0], self.filters, 6) + tuple(new_space)

    def get_config(self):
        config = {
            'filters': self.filters,
            'kernel_size': self.kernel_size,
            'strides': self.strides,
            'padding': self.padding,
            'data_format': self.data_format,
            'dilation_rate': self.dilation_rate,
            'activation': engine.serialize_activation(self.activation),
            'use_bias': self.use_bias,
            'flip_north_pole': self.flip_north_pole,
            'independent_north_pole': self.independent_north_pole,
            'kernel_initializer': engine.serialize_initializer(self.kernel_initializer),
            'bias_initializer': engine.serialize_initializer(self.bias_initializer),
            'kernel_regularizer': regularizers.serialize(self.kernel_regularizer),
            'bias_regularizer': regularizers.serialize(self.bias_regularizer),
            'activity_regularizer': self.activity_regularizer,
            'kernel_constraint': constraints.serialize(self.kernel_constraint),
            'bias_constraint': constraints.serialize(self.bias_constraint)
        }
        base_config = super(CubeSphereConv2D, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))


# ==================================================================================================================== #
# Training callbacks used by the cubed-sphere scripts (behaviour of reference DLWP/custom.py:33-208)
# ==================================================================================================================== #

def _adam_effective_lr(optimizer, beta_1, beta_2, with_bias_correction):
    it = float(optimizer.iterations)
    lr = float(optimizer.lr) / (1. + float(getattr(optimizer, 'decay', 0.)) * it)
    if with_bias_correction:
        t = it + 1.
        lr = lr * np.sqrt(1. - np.power(beta_2, t)) / (1. - np.power(beta_1, t))
    return lr


class AdamLearningRateTracker(Callback):
    """Prints the bias-corrected Adam step size at the end of each epoch (reference DLWP/custom.py:33-46)."""

    def on_epoch_end(self, epoch, logs=None, beta_1=0.9, beta_2=0.999, ):
        print(' - LR: {:.6f}'.format(_adam_effective_lr(self.model.optimizer, beta_1, beta_2, True)))


class SGDLearningRateTracker(Callback):
    """Prints the decayed learning rate at the end of each epoch (reference DLWP/custom.py:49-60)."""

    def on_epoch_end(self, epoch, logs=None):
        print(' - LR: {:.6f}'.format(_adam_effective_lr(self.model.optimizer, 0., 0., False)))


class BatchHistory(Callback):
    """Collects the per-batch logs, one dict of lists per epoch (reference DLWP/custom.py:63-81)."""

    def on_train_begin(self, logs=None):
        self.history, self.epoch = [], 0

    def on_epoch_begin(self, epoch, logs=None):
        self.history.append(dict())

    def on_batch_end(self, batch, logs=None):
        for key, value in (logs or {}).items():
            self.history[self.epoch].setdefault(key, []).append(value)

    def on_epoch_end(self, epoch, logs=None):
        self.epoch += 1


class RunHistory(Callback):
    """keras `History` that additionally forwards every epoch value to `run.log(key, value)` (an AzureML `Run`;
    reference DLWP/custom.py:84-105)."""

    def __init__(self, run):
        super(RunHistory, self).__init__()
        self.run = run
        self.epoch, self.history = [], {}

    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for key, value in (logs or {}).items():
            self.history.setdefault(key, []).append(value)
            self.run.log(key, value)


class RNNResetStates(Callback):
    """Resets recurrent states when an epoch begins (reference DLWP/custom.py:108-110); no-op for the CNN."""

    def on_epoch_begin(self, epoch, logs=None):
        self.model.reset_states()


class EarlyStoppingMin(EarlyStopping):
    """
    EarlyStopping that ignores the first `min_epochs` epochs entirely and stops unconditionally once `max_epochs` is
    reached, restoring the best weights in both cases when asked to (reference DLWP/custom.py:113-165).
    """

    def __init__(self, min_epochs=0, max_epochs=None, **kwargs):
        super(EarlyStoppingMin, self).__init__(**kwargs)
        if not isinstance(min_epochs, int) or min_epochs < 0:
            raise ValueError('min_epochs must be an integer >= 0')
        self.min_epochs = int(min_epochs)
        self.max_epochs = None if max_epochs is None else int(max_epochs)

    def _stop(self, epoch, message):
        self.stopped_epoch = epoch
        self.model.stop_training = True
        if self.restore_best_weights:
            if self.verbose > 0:
                print(message)
            self.model.set_weights(self.best_weights)

    def on_epoch_end(self, epoch, logs=None):
        if epoch < self.min_epochs:
            return
        current = self.get_monitor_value(logs)
        if current is None:
            return
        improved = self.monitor_op(current - self.min_delta, self.best)
        if improved:
            self.best, self.wait = current, 0
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self._stop(epoch, 'Restoring model weights from the end of the best epoch')
        if self.max_epochs is not None and epoch >= self.max_epochs:
            self._stop(epoch, 'Maximum epochs reached; restoring model weights from the end of the best epoch')
        if self.verbose > 1:
            print('EarlyStoppingMin: %d epochs since last minimum' % self.wait)


class SaveWeightsOnEpoch(Callback):
    """
    Writes the weights to `weights_file` after every epoch (I/O errors are swallowed so a locked file cannot kill a
    run); on every `interval`-th epoch the snapshot goes to `weights_file.<epoch>` instead
    (reference DLWP/custom.py:168-191).
    """

    def __init__(self, weights_file, interval=None):
        super(SaveWeightsOnEpoch, self).__init__()
        self.weights_file = str(weights_file)
        if interval is not None:
            assert isinstance(interval, int) and interval > 0, "'interval' must be an integer > 0"
        self.interval = interval

    def on_epoch_end(self, epoch, logs=None):
        if self.interval is not None and epoch % self.interval == 0:
            self.model.save_weights('%s.%s' % (self.weights_file, epoch), save_format='h5')
            return
        try:
            self.model.save_weights(self.weights_file, save_format='h5')
        except OSError:
            pass


class GeneratorEpochEnd(Callback):
    """Runs `generator.on_epoch_end()` (re-shuffle) after each epoch when the data are fed through a dataset wrapper
    that hides the generator from the training loop (reference DLWP/custom.py:194-208)."""

    def __init__(self, generator):
        super(GeneratorEpochEnd, self).__init__()
        self.generator = generator

    def on_epoch_end(self, epoch, logs=None):
        self.generator.on_epoch_end()
