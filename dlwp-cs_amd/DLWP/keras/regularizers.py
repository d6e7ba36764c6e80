"""
Weight regularizers (keras.regularizers restated for the engine; reference use: CubeSphereConv2D(kernel_regularizer=, bias_regularizer=),
DLWP/custom.py:837-842 -> add_weight(regularizer=...), :898-914).  L1L2: penalty = l1 * sum|w| + l2 * sum w^2, added to the training
loss; its gradient is added to the weight's gradient by dlwpcs_l1l2_regularize before the optimizer step (DLWP.keras.Model).
"""


class Regularizer(object):
    def get_config(self):
        return {}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class L1L2(Regularizer):
    def __init__(self, l1=0., l2=0.):
        self.l1 = float(l1)
        self.l2 = float(l2)
        if self.l1 < 0 or self.l2 < 0:
            raise ValueError('L1L2: factors must be >= 0, got l1=%r l2=%r' % (l1, l2))

    def get_config(self):
        return {'l1': self.l1, 'l2': self.l2}


def l1(l=0.01):
    return L1L2(l1=l)


def l2(l=0.01):
    return L1L2(l2=l)


def l1_l2(l1=0.01, l2=0.01):
    return L1L2(l1=l1, l2=l2)


def serialize(reg):
    if reg is None:
        return None
    return {'class_name': type(reg).__name__, 'config': reg.get_config()}


def get(spec):
    """None | Regularizer | 'l1' / 'l2' / 'l1_l2' | {'class_name': 'L1L2', 'config': {...}} (keras.regularizers.get)"""
    if spec is None or isinstance(spec, Regularizer):
        return spec
    if isinstance(spec, str):
        table = {'l1': l1, 'l2': l2, 'l1_l2': l1_l2}
        if spec.lower() in table:
            return table[spec.lower()]()
        raise ValueError('Unknown regularizer: %r' % (spec,))
    if isinstance(spec, dict):
        name = spec.get('class_name')
        if name == 'L1L2':
            return L1L2(**spec.get('config', {}))
        raise ValueError('Unknown regularizer: %r' % (name,))
    raise ValueError('Could not interpret regularizer identifier: %r' % (spec,))
