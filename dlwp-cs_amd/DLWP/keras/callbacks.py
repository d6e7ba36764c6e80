"""Callback protocol of the shim (subset of keras.callbacks used by the reference scripts and DLWP.custom)."""
import warnings

import numpy as np


class Callback(object):
    def __init__(self):
        self.validation_data = None
        self.model = None
        self.params = {}

    def set_params(self, params):
        self.params = params

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass

    # TF2 spellings
    def on_train_batch_begin(self, batch, logs=None):
        self.on_batch_begin(batch, logs)

    def on_train_batch_end(self, batch, logs=None):
        self.on_batch_end(batch, logs)


def _overrides(cb, name):
    return getattr(type(cb), name, None) is not getattr(Callback, name)


class CallbackList(object):
    def __init__(self, callbacks, model, params):
        self.callbacks = list(callbacks or [])
        for cb in self.callbacks:
            cb.set_model(model)
            cb.set_params(params)
        # batch-level logs force a device->host read every step; only pay for it when someone listens
        self.wants_batch_logs = any(_overrides(cb, 'on_batch_end') or _overrides(cb, 'on_train_batch_end') or
                                    _overrides(cb, 'on_batch_begin') or _overrides(cb, 'on_train_batch_begin')
                                    for cb in self.callbacks)

    def call(self, hook, *args):
        for cb in self.callbacks:
            getattr(cb, hook)(*args)


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping (TF 2.1 semantics)."""

    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto', baseline=None,
                 restore_best_weights=False):
        super(EarlyStopping, self).__init__()
        self.monitor = monitor
        self.patience = patience
        self.verbose = verbose
        self.baseline = baseline
        self.min_delta = abs(min_delta)
        self.wait = 0
        self.stopped_epoch = 0
        self.restore_best_weights = restore_best_weights
        self.best_weights = None
        if mode not in ['auto', 'min', 'max']:
            warnings.warn('EarlyStopping mode %s is unknown, fallback to auto mode.' % mode)
            mode = 'auto'
        if mode == 'min':
            self.monitor_op = np.less
        elif mode == 'max':
            self.monitor_op = np.greater
        else:
            self.monitor_op = np.greater if 'acc' in self.monitor else np.less
        if self.monitor_op == np.greater:
            self.min_delta *= 1
        else:
            self.min_delta *= -1
        self.best = np.inf if self.monitor_op == np.less else -np.inf

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        if self.baseline is not None:
            self.best = self.baseline
        else:
            self.best = np.inf if self.monitor_op == np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        current = self.get_monitor_value(logs)
        if current is None:
            return
        if self.monitor_op(current - self.min_delta, self.best):
            self.best = current
            self.wait = 0
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
                if self.restore_best_weights:
                    if self.verbose > 0:
                        print('Restoring model weights from the end of the best epoch.')
                    self.model.set_weights(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose > 0:
            print('Epoch %05d: early stopping' % (self.stopped_epoch + 1))

    def get_monitor_value(self, logs):
        logs = logs or {}
        monitor_value = logs.get(self.monitor)
        if monitor_value is None:
            warnings.warn('Early stopping conditioned on metric `%s` which is not available. Available metrics are: %s'
                          % (self.monitor, ','.join(list(logs.keys()))), RuntimeWarning)
        return monitor_value


class TensorBoard(Callback):
    """Accepted for script compatibility (reference Azure/train_cs.py:444 constructs one and never passes it on)."""

    def __init__(self, log_dir='logs', update_freq='epoch', **kwargs):
        super(TensorBoard, self).__init__()
        self.log_dir = log_dir
        self.update_freq = update_freq
