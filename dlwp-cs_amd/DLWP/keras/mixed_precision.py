"""
Mixed precision switch.  The reference trains under TensorFlow's automatic mixed-precision graph rewrite
(`tf.train.experimental.enable_mixed_precision_graph_rewrite(Adam())`, Azure/train_cs.py:109-110,429: fp16 convolutions,
fp32 master weights, dynamic loss scaling).  The engine's counterpart is bfloat16 activations with fp32 master weights,
fp32 accumulation and fp32 gradients of the parameters -- bf16 keeps the fp32 exponent range, so no loss scaling.

    from DLWP.keras import mixed_precision
    opt = mixed_precision.enable_mixed_precision_graph_rewrite(Adam())     # drop-in for the reference call
    # or, keras-2.4 style:
    mixed_precision.set_policy('mixed_bfloat16')

The policy is read when a Model is constructed (`Model.compute_dtype`); an optimizer returned by
`enable_mixed_precision_graph_rewrite` also switches the model it is compiled into.
"""
from . import backend


class Policy(object):
    def __init__(self, name):
        if name not in ('float32', 'mixed_bfloat16'):
            raise ValueError('policy must be "float32" or "mixed_bfloat16" (got %r); float16 is not built' % (name,))
        self.name = name
        self.compute_dtype = 'bfloat16' if name == 'mixed_bfloat16' else 'float32'
        self.variable_dtype = 'float32'


def set_policy(policy):
    name = policy.name if isinstance(policy, Policy) else policy
    backend.set_compute_dtype(Policy(name).compute_dtype)


def global_policy():
    return Policy('mixed_bfloat16' if backend.compute_dtype() == 'bfloat16' else 'float32')


def enable_mixed_precision_graph_rewrite(opt, loss_scale='dynamic'):
    """Reference Azure/train_cs.py:429.  Returns the same optimizer object (no loss scaling is needed for bfloat16), tagged:
    like TF's rewrite, the switch travels WITH the optimizer -- `Model.compile(optimizer=opt)` puts a model that was built
    before this call (the reference builds its Model at train_cs.py:411, calls this at :429, compiles at :430) into the
    bfloat16 mode as well."""
    set_policy('mixed_bfloat16')
    try:
        opt._mixed_precision = 'bfloat16'
    except AttributeError:
        pass
    return opt


def disable_mixed_precision_graph_rewrite():
    set_policy('float32')


experimental = type('experimental', (), {'Policy': Policy, 'set_policy': staticmethod(set_policy),
                                          'global_policy': staticmethod(global_policy)})
