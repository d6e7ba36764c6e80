"""Optimizers of the shim.  Adam follows TF 2.1 keras defaults (reference Azure/train_cs.py:429: `Adam()`)."""
import torch

from .. import ops


class Optimizer(object):
    pass


class Adam(Optimizer):
    """
    keras.optimizers.Adam(learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False).
    State (m, v, step) lives in flat fp32 device buffers next to the model's flat parameter buffer; one HIP kernel
    (`adam_kernel`) updates every parameter of the model per step.
    """

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, lr=None, decay=0.,
                 name='Adam', **kwargs):
        if amsgrad:
            raise NotImplementedError('Adam(amsgrad=True) is not built')
        if lr is not None:
            learning_rate = lr
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.decay = float(decay)
        if self.decay != 0.:
            raise NotImplementedError('Adam(decay != 0) is not built')
        self.name = name
        self._m = self._v = self._step = None
        # {lr, beta1, beta2, eps, grad_scale} on the device: the Adam launch reads them from HBM, so that a step replayed
        # from a hipGraph sees `optimizer.lr = x` / a learning-rate schedule (kernel arguments are frozen at capture)
        self._hyper = None
        self._hyper_host = None

    @property
    def lr(self):
        return self.learning_rate

    @lr.setter
    def lr(self, value):
        self.learning_rate = float(value)

    def _hyper_values(self, grad_scale):
        return (float(self.learning_rate), float(self.beta_1), float(self.beta_2), float(self.epsilon), float(grad_scale))

    def sync_hyper(self, grad_scale=1.0):
        """Upload the hyper-parameters if they changed since the last upload (a 20-byte copy on the current stream; never
        inside a graph capture: callers sync before capturing / replaying)."""
        vals = self._hyper_values(grad_scale)
        if self._hyper is None or self._hyper_host != vals:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('optimizer hyper-parameters changed during graph capture')
            host = torch.tensor(vals, dtype=torch.float32)
            if self._hyper is None:
                self._hyper = host.to(self._m.device)
            else:
                self._hyper.copy_(host)
            self._hyper_host = vals

    @property
    def iterations(self):
        return 0 if self._step is None else int(self._step[0].item())

    def _ensure_state(self, flat_params):
        if self._m is None or self._m.numel() != flat_params.numel() or self._m.device != flat_params.device:
            self._m = torch.zeros_like(flat_params)
            self._v = torch.zeros_like(flat_params)
            self._step = torch.zeros(2, dtype=torch.int32, device=flat_params.device)    # {t - 1, ticket}
            self._hyper = self._hyper_host = None

    def apply(self, flat_params, flat_grads, grad_scale=1.0, zero_grads=False):
        self._ensure_state(flat_params)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper(grad_scale)
        elif self._hyper is None or self._hyper_host[4] != float(grad_scale):
            raise RuntimeError('call optimizer.sync_hyper(grad_scale) before capturing the update')
        ops.adam_step_dev(flat_params, flat_grads, self._m, self._v, self._step, self._hyper, zero_grads=zero_grads)

    def get_config(self):
        return {'name': self.name, 'learning_rate': self.learning_rate, 'beta_1': self.beta_1, 'beta_2': self.beta_2,
                'epsilon': self.epsilon, 'amsgrad': False, 'decay': self.decay}

    def state_dict(self):
        if self._m is None:
            return None
        return {'m': self._m.cpu().numpy(), 'v': self._v.cpu().numpy(), 'step': int(self._step[0].item())}

    def load_state_dict(self, state, flat_params):
        if state is None:
            return
        self._ensure_state(flat_params)
        self._m.copy_(torch.from_numpy(state['m']))
        self._v.copy_(torch.from_numpy(state['v']))
        self._step.zero_()
        self._step[0] = int(state['step'])


def get(identifier):
    if isinstance(identifier, Optimizer):
        return identifier
    if isinstance(identifier, str) and identifier.lower() == 'adam':
        return Adam()
    if isinstance(identifier, dict) and identifier.get('name', '').lower() == 'adam':
        cfg = dict(identifier)
        cfg.pop('amsgrad', None)
        return Adam(**cfg)
    raise ValueError('Could not interpret optimizer identifier: %r (the DLWP-CS engine provides Adam)' % (identifier,))
