"""
Weight constraints (keras.constraints restated for the engine; reference use: CubeSphereConv2D(kernel_constraint=, bias_constraint=),
DLWP/custom.py:837-842 -> add_weight(constraint=...), :898-914): applied to the weight in place after every optimizer step by
dlwpcs_weight_constraint.  Norms run over `axis`; the engine serves axes that are the weight's LEADING dimensions (the keras default
axis=0, and axis=[0, 1, 2] for a (k, k, Cin, Cout) convolution kernel: one norm per output filter).
"""
from .. import _native as nat


class Constraint(object):
    kind = 0
    axis = 0

    def params(self):
        return (0.0, 0.0, 1.0)

    def get_config(self):
        return {}

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def rows_cols(self, shape):
        """the weight as a (rows, cols) matrix with the norm over the rows"""
        axes = [self.axis] if isinstance(self.axis, int) else list(self.axis)
        nd = len(shape)
        axes = sorted(a % nd for a in axes)
        if axes != list(range(len(axes))):
            raise NotImplementedError('%s(axis=%r) on a weight of shape %r: the engine reduces over leading axes only'
                                      % (type(self).__name__, self.axis, tuple(shape)))
        rows = 1
        for a in axes:
            rows *= int(shape[a])
        total = 1
        for d in shape:
            total *= int(d)
        return rows, total // rows

    def apply(self, w):
        """w: fp32 device tensor, constrained in place"""
        rows, cols = (1, w.numel()) if self.kind == nat.CONSTRAINT_NON_NEG else self.rows_cols(tuple(w.shape))
        a, b, rate = self.params()
        nat.check(nat.lib().dlwpcs_weight_constraint(nat.ptr(w), rows, cols, self.kind, a, b, rate, nat.stream_ptr()),
                  'dlwpcs_weight_constraint')


class MaxNorm(Constraint):
    kind = nat.CONSTRAINT_MAX_NORM

    def __init__(self, max_value=2, axis=0):
        self.max_value = float(max_value)
        self.axis = axis

    def params(self):
        return (self.max_value, 0.0, 1.0)

    def get_config(self):
        return {'max_value': self.max_value, 'axis': self.axis}


class NonNeg(Constraint):
    kind = nat.CONSTRAINT_NON_NEG


class UnitNorm(Constraint):
    kind = nat.CONSTRAINT_UNIT_NORM

    def __init__(self, axis=0):
        self.axis = axis

    def get_config(self):
        return {'axis': self.axis}


class MinMaxNorm(Constraint):
    kind = nat.CONSTRAINT_MIN_MAX_NORM

    def __init__(self, min_value=0.0, max_value=1.0, rate=1.0, axis=0):
        self.min_value = float(min_value)
        self.max_value = float(max_value)
        self.rate = float(rate)
        self.axis = axis

    def params(self):
        return (self.min_value, self.max_value, self.rate)

    def get_config(self):
        return {'min_value': self.min_value, 'max_value': self.max_value, 'rate': self.rate, 'axis': self.axis}


max_norm, non_neg, unit_norm, min_max_norm = MaxNorm, NonNeg, UnitNorm, MinMaxNorm
_BY_NAME = {'MaxNorm': MaxNorm, 'max_norm': MaxNorm, 'NonNeg': NonNeg, 'non_neg': NonNeg, 'UnitNorm': UnitNorm,
            'unit_norm': UnitNorm, 'MinMaxNorm': MinMaxNorm, 'min_max_norm': MinMaxNorm}


def serialize(con):
    if con is None:
        return None
    return {'class_name': type(con).__name__, 'config': con.get_config()}


def get(spec):
    """None | Constraint | name | {'class_name': ..., 'config': {...}} (keras.constraints.get)"""
    if spec is None or isinstance(spec, Constraint):
        return spec
    if isinstance(spec, str):
        if spec in _BY_NAME:
            return _BY_NAME[spec]()
        raise ValueError('Unknown constraint: %r' % (spec,))
    if isinstance(spec, dict):
        name = spec.get('class_name')
        if name in _BY_NAME:
            return _BY_NAME[name](**spec.get('config', {}))
        raise ValueError('Unknown constraint: %r' % (name,))
    raise ValueError('Could not interpret constraint identifier: %r' % (spec,))
