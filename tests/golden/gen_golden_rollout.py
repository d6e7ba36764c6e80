#!/usr/bin/env python3
"""
Golden vectors for the host bookkeeping of SURVEY 8 a6 / N3.  Runs ONLY in the build container (needs /root/reference).

g7_rollout.npz   -- `DLWPFunctional.predict_timeseries` (/root/reference/DLWP/model/models.py:418-460) executed VERBATIM:
                    the method's source is cut out of the reference file at generation time, compiled, and bound to a stand-in
                    object whose `predict` is a known "model" (state + 1 per model step, or a list [state+1, state+2] for a
                    two-output sequence model).  Cases: n_steps in {1,2} x time_dim in {1,2} x keep_time_dim in {F,T} x
                    time_steps in {3,4} in BOTH layouts (channels_first (B,C,6,N,N) and channels_last (B,6,N,N,C)); the
                    reference splits the FIRST feature axis by time_dim in either layout (its TODO at models.py:455).
g8_callbacks.npz -- `EarlyStoppingMin.on_epoch_end` and `SaveWeightsOnEpoch.on_epoch_end` (/root/reference/DLWP/custom.py:113-191)
                    executed verbatim (class bodies cut out of the reference file) over scripted loss sequences; the Keras
                    base class `EarlyStopping` is third-party (TF 2.1, not under /root/reference) and is restated in this
                    script from its documented behaviour (monitor_op / min_delta sign / on_train_begin reset /
                    get_monitor_value).  Stored: per-epoch (stop_training, wait, best, stopped_epoch) traces, the epoch whose
                    weights the model holds at the end, and the save_weights call log.

Only numbers / file-name strings are committed -- no reference source text.
"""
import os
import re
import sys
import warnings

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _cut(src, pattern):
    m = re.search(pattern, src, re.S | re.M)
    if m is None:
        raise RuntimeError('pattern not found: %s' % pattern)
    return m.group(0)


# ------------------------------------------------------------------------------------------------------------------ #
# g7: predict_timeseries
# ------------------------------------------------------------------------------------------------------------------ #

def rollout_cases():
    out = []
    for layout in ('cf', 'cl'):
        for n_steps in (1, 2):
            for time_dim in (1, 2):
                for keep in (False, True):
                    for time_steps in (3, 4):
                        out.append((layout, n_steps, time_dim, keep, time_steps))
    return out


def gen_rollout():
    src = open(os.path.join(REF, 'DLWP', 'model', 'models.py')).read()
    cls = _cut(src, r'^class DLWPFunctional\(object\):.*?(?=^class |\Z)')
    body = _cut(cls, r'^    def predict_timeseries\(self.*?(?=^    def |\Z)')
    import textwrap
    ns = {'np': np}
    exec(compile(textwrap.dedent(body), 'models.py:predict_timeseries', 'exec'), ns)
    fn = ns['predict_timeseries']

    class Stub(object):
        is_recurrent = False

        def __init__(self, n_steps, time_dim):
            self._n_steps, self.time_dim = n_steps, time_dim

        def predict(self, p, **kwargs):
            if self._n_steps == 1:
                return p + 1.0
            return [p + float(k + 1) for k in range(self._n_steps)]

    rng = np.random.default_rng(7)
    B, C, N = 2, 4, 3
    store = {}
    names = []
    for layout, n_steps, time_dim, keep, time_steps in rollout_cases():
        shape = (B, C, 6, N, N) if layout == 'cf' else (B, 6, N, N, C)
        x = rng.standard_normal(shape).astype(np.float32)
        y = fn(Stub(n_steps, time_dim), x, time_steps, keep_time_dim=keep)
        name = '%s_n%d_t%d_k%d_s%d' % (layout, n_steps, time_dim, int(keep), time_steps)
        names.append(name)
        store[name + '_x'] = x
        store[name + '_y'] = np.ascontiguousarray(y)
    # error behaviour
    errs = {}
    for label, args in (('list_input', ([np.zeros((1, 2))], 2)), ('zero_steps', (np.zeros((1, 2, 6, 2, 2), np.float32), 0))):
        try:
            fn(Stub(1, 1), *args)
            errs[label] = 'none'
        except Exception as e:       # noqa: BLE001
            errs[label] = type(e).__name__
    store['names'] = np.array(names)
    store['err_labels'] = np.array(sorted(errs))
    store['err_types'] = np.array([errs[k] for k in sorted(errs)])
    np.savez_compressed(os.path.join(HERE, 'g7_rollout.npz'), **store)
    print('g7: %d cases, errors %s' % (len(names), errs))


# ------------------------------------------------------------------------------------------------------------------ #
# g8: callbacks
# ------------------------------------------------------------------------------------------------------------------ #

class Callback(object):
    """keras.callbacks.Callback: the attributes the reference callbacks touch."""

    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping (TF 2.1) restated from its documented behaviour -- third-party base class."""

    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto', baseline=None,
                 restore_best_weights=False):
        super(EarlyStopping, self).__init__()
        self.monitor, self.patience, self.verbose, self.baseline = monitor, patience, verbose, baseline
        self.min_delta = abs(min_delta)
        self.wait = 0
        self.stopped_epoch = 0
        self.restore_best_weights = restore_best_weights
        self.best_weights = None
        if mode not in ('auto', 'min', 'max'):
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

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        if self.baseline is not None:
            self.best = self.baseline
        else:
            self.best = np.inf if self.monitor_op == np.less else -np.inf

    def get_monitor_value(self, logs):
        logs = logs or {}
        value = logs.get(self.monitor)
        if value is None:
            warnings.warn('Early stopping conditioned on metric `%s` which is not available.' % self.monitor)
        return value


class FakeModel(object):
    def __init__(self):
        self.stop_training = False
        self.w = [np.array([-1.0])]
        self.saved = []
        self.fail_on = set()

    def get_weights(self):
        return [a.copy() for a in self.w]

    def set_weights(self, w):
        self.w = [np.array(a) for a in w]

    def save_weights(self, path, save_format=None):
        if len(self.saved) in self.fail_on:
            self.saved.append('OSError:' + path)
            raise OSError('locked')
        self.saved.append('%s|%s' % (path, save_format))


ES_CASES = [
    # (kwargs, loss sequence)
    (dict(min_epochs=0, max_epochs=None, monitor='val_loss', patience=2, restore_best_weights=True),
     [1.0, 0.8, 0.9, 0.85, 0.95, 0.7]),
    (dict(min_epochs=3, max_epochs=None, monitor='val_loss', patience=1, restore_best_weights=True),
     [1.0, 1.1, 1.2, 0.9, 0.95, 0.8]),
    (dict(min_epochs=0, max_epochs=3, monitor='val_loss', patience=10, restore_best_weights=True),
     [1.0, 0.5, 0.6, 0.7, 0.4, 0.3]),
    (dict(min_epochs=1, max_epochs=4, monitor='loss', patience=2, min_delta=0.05, restore_best_weights=False),
     [1.0, 0.99, 0.97, 0.90, 0.89, 0.88]),
    (dict(min_epochs=0, max_epochs=None, monitor='val_loss', patience=1, mode='max', restore_best_weights=True),
     [0.1, 0.3, 0.2, 0.4]),
    (dict(min_epochs=2, max_epochs=2, monitor='val_loss', patience=3, restore_best_weights=True),
     [0.5, 0.4, 0.6, 0.3]),
]


def gen_callbacks():
    src = open(os.path.join(REF, 'DLWP', 'custom.py')).read()
    ns = {'np': np, 'Callback': Callback, 'EarlyStopping': EarlyStopping}
    for cname in ('EarlyStoppingMin', 'SaveWeightsOnEpoch'):
        code = _cut(src, r'^class %s\(.*?(?=^class |^# =====|\Z)' % cname)
        exec(compile(code, 'custom.py:' + cname, 'exec'), ns)
    ESM, SWE = ns['EarlyStoppingMin'], ns['SaveWeightsOnEpoch']
    store = {}
    for i, (kw, losses) in enumerate(ES_CASES):
        cb = ESM(**kw)
        model = FakeModel()
        cb.set_model(model)
        cb.on_train_begin()
        trace = []
        for epoch, loss in enumerate(losses):
            model.w = [np.array([float(epoch)])]                 # the weights "are" the epoch that produced them
            cb.on_epoch_end(epoch, {kw['monitor']: loss})
            trace.append([float(model.stop_training), float(cb.wait), float(cb.best), float(cb.stopped_epoch),
                          float(model.w[0][0])])
            if model.stop_training:
                break
        store['es%d_trace' % i] = np.array(trace)
        store['es%d_losses' % i] = np.array(losses)
    # constructor validation
    try:
        ESM(min_epochs=-1)
        store['es_bad_min_epochs'] = np.array('none')
    except Exception as e:           # noqa: BLE001
        store['es_bad_min_epochs'] = np.array(type(e).__name__)
    for j, (interval, fail_on) in enumerate(((None, ()), (2, ()), (3, (1,)), (None, (0, 2)))):
        cb = SWE('w.h5', interval=interval)
        model = FakeModel()
        model.fail_on = set(fail_on)
        cb.set_model(model)
        raised = []
        for epoch in range(6):
            try:
                cb.on_epoch_end(epoch)
                raised.append(0)
            except OSError:
                raised.append(1)
        store['sw%d_saved' % j] = np.array(model.saved)
        store['sw%d_raised' % j] = np.array(raised)
        store['sw%d_interval' % j] = np.array(-1 if interval is None else interval)
        store['sw%d_fail_on' % j] = np.array(sorted(fail_on), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'g8_callbacks.npz'), **store)
    print('g8:', {k: v.shape for k, v in store.items() if k.endswith('trace')})


# ------------------------------------------------------------------------------------------------------------------ #
# ref_wrapper.pkl / ref_wrapper.history: what the reference's DLWP.util.save_model pickles next to the keras file
# ------------------------------------------------------------------------------------------------------------------ #

def gen_wrapper():
    """`save_model` (/root/reference/DLWP/util.py:127-155) pickles a copy of the DLWPFunctional wrapper with `model` and
    `base_model` set to None, and `history.history`.  The reference class (DLWP/model/models.py:320-351, cut out of the file and
    executed in a module object named DLWP.model.models) is instantiated and pickled exactly that way; the engine's
    DLWP.util.load_model must resurrect it through its own class of the same import path."""
    import pickle
    import types
    from copy import copy
    src = open(os.path.join(REF, 'DLWP', 'model', 'models.py')).read()
    cls_src = _cut(src, r'^class DLWPFunctional\(object\):.*?(?=^class |\Z)')
    saved = {k: sys.modules.get(k) for k in ('DLWP', 'DLWP.model', 'DLWP.model.models')}
    try:
        for name in ('DLWP', 'DLWP.model', 'DLWP.model.models'):
            sys.modules[name] = types.ModuleType(name)
        mod = sys.modules['DLWP.model.models']
        mod.np = np
        exec(compile(cls_src, 'models.py:DLWPFunctional', 'exec'), mod.__dict__)
        obj = mod.DLWPFunctional(is_convolutional=True, is_recurrent=False, time_dim=2)
        obj._n_steps = 2
        obj.model = object()             # stands for the keras model; save_model drops it
        obj.base_model = obj.model
        model_copy = copy(obj)
        model_copy.model = None
        model_copy.base_model = None
        with open(os.path.join(HERE, 'ref_wrapper.pkl'), 'wb') as f:
            pickle.dump(model_copy, f, protocol=pickle.HIGHEST_PROTOCOL)
        with open(os.path.join(HERE, 'ref_wrapper.history'), 'wb') as f:
            pickle.dump({'loss': [1.5, 0.75], 'val_loss': [1.25, 0.875]}, f, protocol=pickle.HIGHEST_PROTOCOL)
        print('ref_wrapper.pkl:', sorted(model_copy.__dict__))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


if __name__ == '__main__':
    gen_rollout()
    gen_callbacks()
    gen_wrapper()
