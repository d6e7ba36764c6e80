#
# Cubed-sphere CNN wirings of the DLWP-CS training script, as a library.
#

"""
The network definitions that the reference keeps inside its training script (Azure/train_cs.py:186-421; identical to
Tutorial 3): layer objects created once and shared between integration steps, the `basic` / `unet` / `unet2` / `unet3`
/ `unet4` wirings, and the multi-step `complete_model` with solar / constant re-injection.  Written against the
`DLWP.keras` shim so that the result runs on the MI355X engine.  The filter table follows train_cs.py:208-228.
"""

from ..custom import CubeSphereConv2D, CubeSpherePadding2D
from ..keras.layers import (AveragePooling3D, Concatenate, Input, Permute, ReLU, Reshape, UpSampling3D, concatenate)
from ..keras.models import Model


class CubeSphereNet(object):
    """Holds the shared layer instances of the DLWP-CS CNN family (train_cs.py:196-228)."""

    def __init__(self, output_channels, base_filter_number=32, cnn_model_name='unet2', independent_north_pole=False):
        self.cnn_model_name = cnn_model_name
        if not hasattr(self, cnn_model_name) or cnn_model_name.startswith('_'):
            raise ValueError('unknown cnn_model_name %r' % (cnn_model_name,))
        b = int(base_filter_number)
        skip = 'unet' in cnn_model_name.lower()
        self.pad = CubeSpherePadding2D(1, data_format='channels_last')
        self.pool = AveragePooling3D((1, 2, 2), data_format='channels_last')
        self.up = UpSampling3D((1, 2, 2), data_format='channels_last')
        self.relu = ReLU(negative_slope=0.1, max_value=10.)
        kw = dict(dilation_rate=1, padding='valid', activation='linear', data_format='channels_last',
                  independent_north_pole=independent_north_pole, flip_north_pole=not independent_north_pole)

        def conv(filters, k=3, **extra):
            return CubeSphereConv2D(filters, k, **dict(kw, **extra))
        # same names and filter counts as train_cs.py:209-228
        self.conv_2d_1, self.conv_2d_1_2, self.conv_2d_1_3 = conv(b), conv(b), conv(b)
        self.conv_2d_2, self.conv_2d_2_2, self.conv_2d_2_3 = conv(2 * b), conv(2 * b), conv(2 * b)
        self.conv_2d_3, self.conv_2d_3_2 = conv(4 * b), conv(4 * b)
        self.conv_2d_4 = conv(4 * b if skip else 8 * b)
        self.conv_2d_4_2 = conv(8 * b)
        self.conv_2d_5 = conv(2 * b if skip else 4 * b)
        self.conv_2d_5_2, self.conv_2d_5_3 = conv(4 * b), conv(4 * b)
        self.conv_2d_6 = conv(b if skip else 2 * b)
        self.conv_2d_6_2, self.conv_2d_6_3 = conv(2 * b), conv(2 * b)
        self.conv_2d_7, self.conv_2d_7_2, self.conv_2d_7_3 = conv(b), conv(b), conv(b)
        self.conv_2d_8 = conv(output_channels, 1, name='output')

    def _block(self, x, *convs):
        for c in convs:
            x = self.relu(c(self.pad(x)))
        return x

    # ---- wirings (train_cs.py:233-388) ---------------------------------------------------------------------------
    def basic(self, x):
        x = self.pool(self._block(x, self.conv_2d_1))
        x = self.pool(self._block(x, self.conv_2d_2))
        x = self.up(self._block(x, self.conv_2d_3))
        x = self.up(self._block(x, self.conv_2d_6))
        x = self._block(x, self.conv_2d_7, self.conv_2d_7_2)
        return self.conv_2d_8(x)

    def unet(self, x):
        x0 = self._block(x, self.conv_2d_1)
        x1 = self._block(self.pool(x0), self.conv_2d_2)
        x2 = self.up(self._block(self.pool(x1), self.conv_2d_3))
        x = concatenate([x2, x1], axis=-1)
        x = self.up(self._block(x, self.conv_2d_6))
        x = concatenate([x, x0], axis=-1)
        x = self._block(x, self.conv_2d_7, self.conv_2d_7_2)
        return self.conv_2d_8(x)

    def unet2(self, x):
        x0 = self._block(x, self.conv_2d_1, self.conv_2d_1_2)
        x1 = self._block(self.pool(x0), self.conv_2d_2, self.conv_2d_2_2)
        x2 = self.up(self._block(self.pool(x1), self.conv_2d_5_2, self.conv_2d_5))
        x = concatenate([x2, x1], axis=-1)
        x = self.up(self._block(x, self.conv_2d_6_2, self.conv_2d_6))
        x = concatenate([x, x0], axis=-1)
        x = self._block(x, self.conv_2d_7, self.conv_2d_7_2)
        return self.conv_2d_8(x)

    def unet3(self, x):
        x0 = self._block(x, self.conv_2d_1, self.conv_2d_1_2, self.conv_2d_1_3)
        x1 = self._block(self.pool(x0), self.conv_2d_2, self.conv_2d_2_2, self.conv_2d_2_3)
        x2 = self.up(self._block(self.pool(x1), self.conv_2d_5_3, self.conv_2d_5_2, self.conv_2d_5))
        x = concatenate([x2, x1], axis=-1)
        x = self.up(self._block(x, self.conv_2d_6_3, self.conv_2d_6_2, self.conv_2d_6))
        x = concatenate([x, x0], axis=-1)
        x = self._block(x, self.conv_2d_7, self.conv_2d_7_2, self.conv_2d_7_3)
        return self.conv_2d_8(x)

    def unet4(self, x):
        x0 = self._block(x, self.conv_2d_1, self.conv_2d_1_2)
        x1 = self._block(self.pool(x0), self.conv_2d_2, self.conv_2d_2_2)
        x2 = self._block(self.pool(x1), self.conv_2d_3_2, self.conv_2d_3)
        x3 = self.up(self._block(self.pool(x2), self.conv_2d_4_2, self.conv_2d_4))
        x = concatenate([x3, x2], axis=-1)
        x = self.up(self._block(x, self.conv_2d_5_2, self.conv_2d_5))
        x = concatenate([x, x1], axis=-1)
        x = self.up(self._block(x, self.conv_2d_6_2, self.conv_2d_6))
        x = concatenate([x, x0], axis=-1)
        x = self._block(x, self.conv_2d_7, self.conv_2d_7_2)
        return self.conv_2d_8(x)

    def encoder6(self, x):
        """BASELINE config 2: the first six convolutions of unet2 (train_cs.py:278-291)."""
        x0 = self._block(x, self.conv_2d_1, self.conv_2d_1_2)
        x1 = self._block(self.pool(x0), self.conv_2d_2, self.conv_2d_2_2)
        return self._block(self.pool(x1), self.conv_2d_5_2, self.conv_2d_5)

    def __call__(self, x):
        return getattr(self, self.cnn_model_name)(x)


def build_cs_model(convolution_shape, output_channels, cnn_model_name='unet2', base_filter_number=32,
                   integration_steps=1, io_time_steps=2, insolation_shape=None, constants_shape=None,
                   independent_north_pole=False):
    """
    Build the (multi-step) cubed-sphere model of train_cs.py:391-421.

    :param convolution_shape: (6, N, N, C_in) of the main input (generator.convolution_shape, channels_last)
    :param output_channels: C_out of each step's output
    :param integration_steps: number of times the CNN is applied (one output per step; weights shared)
    :param io_time_steps: input/output time steps folded into the channel axis (time-major)
    :param insolation_shape: (T, 6, N, N, 1) of each `solar_<step>` input, or None
    :param constants_shape: (6, N, N, K) of the `constants` input, or None
    :return: DLWP.keras.Model with inputs [main_input, solar_1.., constants] and `integration_steps` outputs
    """
    cs = tuple(convolution_shape)
    net = CubeSphereNet(output_channels, base_filter_number, cnn_model_name, independent_north_pole)
    main_input = Input(shape=cs, name='main_input')
    input_solar = integration_steps > 1 and insolation_shape is not None
    has_constants = constants_shape is not None
    solar_inputs = [Input(shape=tuple(insolation_shape), name='solar_%d' % d) for d in range(1, integration_steps)] \
        if input_solar else []
    constant_input = Input(shape=tuple(constants_shape), name='constants') if has_constants else None

    xi = main_input
    if has_constants:
        xi = Concatenate(axis=-1)([xi, constant_input])
    outputs = [net(xi)]
    for step in range(1, integration_steps):
        xo = outputs[step - 1]
        if input_solar:
            xo = Reshape(cs[:-1] + (io_time_steps, -1))(xo)
            xo = Concatenate(axis=-1)([xo, Permute((2, 3, 4, 1, 5))(solar_inputs[step - 1])])
            xo = Reshape(cs)(xo)
        if has_constants:
            xo = Concatenate(axis=-1)([xo, constant_input])
        outputs.append(net(xo))

    if not input_solar and not has_constants:
        inputs = main_input
    else:
        inputs = [main_input] + solar_inputs + ([constant_input] if has_constants else [])
    model = Model(inputs=inputs, outputs=outputs if integration_steps > 1 else outputs[0])
    model.cs_net = net
    return model
