"""
Batch feed of the cubed-sphere models: the engine's counterpart of the reference's
`ArrayDataGenerator` + `tf_data_generator` (DLWP/model/generators.py:636-1011, 1014-1074).

Same constructor, shape properties and batch semantics as the reference class (golden vectors produced by the reference
class itself: tests/golden/g5_generators.npz):

  * predictors: `input_time_steps` consecutive states (stride `interval`) of the selected variables, flattened
    TIME-MAJOR into the channel axis, channel = t * (V_in + add_insolation) + v, the insolation of each step as the last
    channel of that step; in sequence mode the inputs are `[predictors, insolation of step 1.., constants]` and the
    targets a list of `sequence` consecutive forecast windows (DLWP/model/generators.py:878-957);
  * `channels_last` moves the channel axis last, solar inputs keep their time axis (`:968-982`).

Two execution paths behind the same API:

  * host (default): numpy views/fancy indexing exactly like the reference (any model type: dense, recurrent, conv);
  * `device=...` (MI355X): the data array, the insolation array and the constants are uploaded to HBM ONCE (a 40-year
    6-hourly C48 ERA5 set of 7 variables is 22 GB of the 288 GB) and `generate()` returns device tensors assembled by one
    gather kernel per tensor (`dlwpcs_batch_gather`: fancy-index + time-major channel packing + channels_last transpose +
    optional bf16 rounding in one pass) -- no host work, no PCIe traffic per batch.  `DLWPFunctional.fit_generator`
    accepts either kind.
"""
import numpy as np

from ..util import to_bool


def delete_nan_samples(predictors, targets, large_fill_value=False, threshold=None):
    """Drop every sample (row of axis 0) that contains a NaN in predictors or targets (reference DLWP/util.py:239-269)."""
    if threshold is not None and not (0 <= threshold <= 1):
        raise ValueError("'threshold' must be between 0 and 1")
    if large_fill_value:
        predictors[np.abs(predictors) >= 1.e20] = np.nan
        targets[np.abs(targets) >= 1.e20] = np.nan
    p2 = predictors.reshape((predictors.shape[0], -1))
    t2 = targets.reshape((targets.shape[0], -1))
    if threshold is None:
        bad = np.isnan(p2).any(axis=1) | np.isnan(t2).any(axis=1)
    else:
        bad = (np.isnan(p2).mean(axis=1) >= threshold) | (np.isnan(t2).mean(axis=1) >= threshold)
    return predictors[~bad], targets[~bad]


def _slice_indices(sel, n):
    """Variable selection (slice or index list) -> explicit index array."""
    if sel is None:
        sel = slice(None)
    if isinstance(sel, slice):
        return np.arange(n)[sel]
    return np.asarray(sel, dtype=np.int64).reshape(-1)


class ArrayDataGenerator(object):
    """
    Produces batches from a single array of data ordered (time, variable, *space) by manipulating index windows
    (reference DLWP/model/generators.py:636-1011).  See the module docstring for the `device` path.

    Deviation from the reference, on purpose: `insolation_array=None` is accepted (the reference dereferences it
    unconditionally at `:719-721`, so it can only be used with insolation).
    """

    def __init__(self, model, array, rank=2, batch_size=32, input_slice=None, output_slice=None,
                 input_time_steps=1, output_time_steps=1, sequence=None, interval=1,
                 shuffle=False, remove_nan=True, insolation_array=None, constants=None, channels_last=False,
                 drop_remainder=False, device=None, dtype=None):
        """
        :param model: DLWP model instance (metadata only: is_convolutional, is_recurrent, impute)
        :param array: ndarray (time, variable, *space)
        :param rank: number of spatial dimensions (3 for cubed-sphere data: face, height, width)
        :param device: None for host numpy batches, or a torch device / True to keep the data in HBM and assemble
            batches with the gather kernel (convolutional, non-recurrent models)
        :param dtype: 'float32' | 'bfloat16' | None: dtype of the device PREDICTORS (None: the engine's compute dtype when
            the generator is created); targets are always float32 (the loss is computed in fp32)
        (all other parameters: see the reference class)
        """
        for name, v in (('rank', rank), ('input_time_steps', input_time_steps), ('output_time_steps', output_time_steps),
                        ('batch_size', batch_size), ('interval', interval)):
            assert int(v) > 0, '%s must be positive' % name
        if sequence is not None:
            assert int(sequence) > 0
        self.array = array
        self._batch_size = int(batch_size)
        self._shuffle = shuffle
        self._remove_nan = remove_nan
        self._is_convolutional = model.is_convolutional
        self._keep_time_axis = model.is_recurrent
        self._impute_missing = model.impute
        self._indices = []
        self._sequence = sequence
        n_out = output_time_steps * (sequence if sequence is not None else 1)
        self._n_sample = array.shape[0] - interval * (input_time_steps + n_out) + 1
        self.rank = int(rank)
        self._input_slice = input_slice or slice(None)
        self._output_slice = output_slice or slice(None)
        self._input_vars = _slice_indices(input_slice, array.shape[1])
        self._output_vars = _slice_indices(output_slice, array.shape[1])
        self._input_size, self._output_size = len(self._input_vars), len(self._output_vars)
        self._input_time_steps = int(input_time_steps)
        self._output_time_steps = int(output_time_steps)
        self._interval = int(interval)
        self.drop_remainder = to_bool(drop_remainder)
        self.on_epoch_end()

        self.insolation_array = insolation_array
        self._add_insolation = 1 if insolation_array is not None else 0
        if insolation_array is not None:
            assert insolation_array.shape[-self.rank:] == self.shape[-self.rank:], \
                "spatial dimensions of insolation must be the same as input data; got %s and %s" % \
                (insolation_array.shape[-self.rank:], self.shape[-self.rank:])
        self.constants = constants
        if constants is not None:
            assert constants.shape[-self.rank:] == self.shape[-self.rank:], \
                "spatial dimensions of constants must be the same as input data; got %s and %s" % \
                (constants.shape[-self.rank:], self.shape[-self.rank:])

        self.channels_last = to_bool(channels_last)
        self._time_transpose = (0, 1,) + tuple(range(3, 3 + self.rank)) + (2,)
        self._transpose = self._time_transpose if self._keep_time_axis else \
            (0,) + tuple(range(2, 2 + self.rank)) + (1,)

        self.device = None
        self._dev = None
        if device is not None and device is not False:
            self._to_device(device, dtype)

    # ------------------------------------------------------------------------------------------------------------- #
    # shapes (reference :738-865)
    # ------------------------------------------------------------------------------------------------------------- #
    @property
    def shape(self):
        """(time_step, varlev, *space) of the input data; excludes insolation"""
        return (self._input_time_steps, self._input_size) + tuple(self.array.shape[2:])

    @property
    def n_features(self):
        return int(np.prod(self.shape)) + int(np.prod(self.shape[-self.rank:])) * self._input_time_steps \
            * self._add_insolation

    @property
    def dense_shape(self):
        if self._keep_time_axis:
            return (self.shape[0],) + (self.n_features // self.shape[0],)
        return (self.n_features,)

    def _conv_shape(self, time_steps, n_var, extra):
        if self._keep_time_axis:
            result = (time_steps, n_var + extra) + self.shape[-self.rank:]
            order = self._time_transpose
        else:
            result = (time_steps * (n_var + extra),) + self.shape[-self.rank:]
            order = self._transpose
        if self.channels_last:
            return tuple(result[s - 1] for s in order[1:])
        return result

    @property
    def convolution_shape(self):
        """shape of the predictors expected by the conv layers: (channels, *space), or (*space, channels) if
        channels_last (recurrent models: with the leading time axis); includes insolation"""
        return self._conv_shape(self._input_time_steps, int(np.prod(self.shape[1:-self.rank])), self._add_insolation)

    @property
    def shape_2d(self):
        keep, self._keep_time_axis = self._keep_time_axis, False
        try:
            return tuple(self.convolution_shape)
        finally:
            self._keep_time_axis = keep

    @property
    def output_shape(self):
        return (self._output_time_steps, self._output_size) + tuple(self.array.shape[2:])

    @property
    def output_n_features(self):
        return int(np.prod(self.output_shape))

    @property
    def output_dense_shape(self):
        if self._keep_time_axis:
            return (self.output_shape[0],) + (self.output_n_features // self.output_shape[0],)
        return (self.output_n_features,)

    @property
    def output_convolution_shape(self):
        return self._conv_shape(self._output_time_steps, int(np.prod(self.output_shape[1:-self.rank])), 0)

    @property
    def output_shape_2d(self):
        keep, self._keep_time_axis = self._keep_time_axis, False
        try:
            return tuple(self.output_convolution_shape)
        finally:
            self._keep_time_axis = keep

    @property
    def insolation_shape(self):
        """shape of the solar inputs of steps 1.. of an input sequence; always carries the time-step axis"""
        if self.channels_last:
            return tuple((self._input_time_steps,) + self.convolution_shape[:self.rank]) + (1,)
        return tuple((self._input_time_steps, 1) + self.convolution_shape[-self.rank:])

    def on_epoch_end(self):
        self._indices = np.arange(self._n_sample)
        if self._shuffle:
            np.random.shuffle(self._indices)

    def __len__(self):
        if self.drop_remainder:
            return int(np.floor(self._n_sample / self._batch_size))
        return int(np.ceil(self._n_sample / self._batch_size))

    def __getitem__(self, index):
        if int(index) < 0:
            index = len(self) + index
        if index > len(self):
            raise IndexError
        a, b = index * self._batch_size, (index + 1) * self._batch_size
        if self._dev is not None and len(self._indices):
            # the epoch's sample order lives on the device too (ONE upload per epoch, not one per batch): the batch's gathers are
            # enqueued without the host touching the link
            cached = self._dev.get('order')
            if cached is None or cached[0] is not self._indices:
                t = self._dev['torch'].from_numpy(np.ascontiguousarray(self._indices, dtype=np.int32))
                t = t.pin_memory().to(self.device, non_blocking=True) if self.device.type == 'cuda' else t.to(self.device)
                cached = self._dev['order'] = (self._indices, t)
            return self._generate_device(np.asarray(self._indices[a:b], dtype=np.int64), cached[1][a:b])
        return self.generate(self._indices[a:b])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    # ------------------------------------------------------------------------------------------------------------- #
    # batch assembly
    # ------------------------------------------------------------------------------------------------------------- #
    def generate(self, samples):
        samples = np.arange(self._n_sample, dtype=np.int64) if len(samples) == 0 else np.asarray(samples, dtype=np.int64)
        if self._dev is not None:
            return self._generate_device(samples)
        return self._generate_host(samples)

    def _windows(self):
        """(t_off, n_steps) of the predictor window and of every target window, in units of array rows."""
        its, ots, iv = self._input_time_steps, self._output_time_steps, self._interval
        seq = self._sequence if self._sequence is not None else 1
        return [(iv * (its + ots * s), ots) for s in range(seq)]

    def _generate_host(self, samples):
        n = len(samples)
        its, iv = self._input_time_steps, self._interval
        arr, space = self.array, tuple(self.array.shape[2:])
        vin, vout = self._input_vars, self._output_vars

        def window(source, var, t_off, steps):                 # (n, steps, len(var), *space)
            return np.stack([np.asarray(source[samples + t_off + k * iv])[:, var] for k in range(steps)], axis=1)

        p = window(arr, vin, 0, its)
        solar = []
        if self._add_insolation:
            seq = self._sequence if self._sequence is not None else 1
            for s in range(seq):
                solar.append(np.stack([np.asarray(self.insolation_array[samples + iv * (its * s + k)])
                                       for k in range(its)], axis=1)[:, :, np.newaxis])        # (n, its, 1, *space)
            p = np.concatenate([p, solar[0]], axis=2)
        targets = [window(arr, vout, t_off, steps) for t_off, steps in self._windows()]
        if self._remove_nan:
            keep = ~np.isnan(p.reshape(n, -1)).any(axis=1)
            for t in targets:
                keep &= ~np.isnan(t.reshape(n, -1)).any(axis=1)
            if not keep.all():
                p, targets = p[keep], [t[keep] for t in targets]
                solar = [s[keep] for s in solar]
                n = int(keep.sum())

        def shape_like(x, conv_shape, dense_shape):
            if self._is_convolutional:
                return x.reshape((n,) + conv_shape)
            if self._keep_time_axis:
                return x.reshape((n,) + dense_shape)
            return x.reshape((n, -1))
        cl, self.channels_last = self.channels_last, False      # shapes below are channels_first; transposed at the end
        try:
            p = shape_like(p, tuple(self.convolution_shape), self.dense_shape)
            targets = [shape_like(t, tuple(self.output_convolution_shape), self.output_dense_shape) for t in targets]
        finally:
            self.channels_last = cl
        if self._sequence is not None and self._add_insolation:
            p = [p] + solar[1:]
        if self.constants is not None:
            c = np.repeat(np.expand_dims(self.constants, axis=0), n, axis=0)
            if self._keep_time_axis:
                c = np.expand_dims(c, 1)
            p = (p if isinstance(p, list) else [p]) + [c]
        if self.channels_last:
            def tr(x):
                return x.transpose(self._transpose if x.ndim == len(self._transpose) else self._time_transpose)
            p = [tr(x) for x in p] if isinstance(p, list) else tr(p)
            targets = [tr(t) for t in targets]
        return p, (targets if self._sequence is not None else targets[0])

    # ------------------------------------------------------------------------------------------------------------- #
    # device path
    # ------------------------------------------------------------------------------------------------------------- #
    def _to_device(self, device, dtype):
        import torch
        from ..keras import backend
        from .. import ops
        if not self._is_convolutional or self._keep_time_axis:
            raise NotImplementedError('device-resident batches serve convolutional, non-recurrent models')
        dev = backend.device() if device is True else torch.device(device)
        self.device = dev
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)   # noqa: E731
        d = {'array': up(self.array), 'vin': torch.from_numpy(self._input_vars.astype(np.int32)).to(dev),
             'vout': torch.from_numpy(self._output_vars.astype(np.int32)).to(dev),
             'zero': torch.zeros(1, dtype=torch.int32, device=dev), 'ops': ops, 'torch': torch,
             'pdtype': backend.torch_dtype(dtype)}
        if self._remove_nan and bool(torch.isnan(d['array']).any().item()):
            raise NotImplementedError('remove_nan with NaNs present: use the host path (device=None)')
        if self._add_insolation:
            d['sol'] = up(self.insolation_array).unsqueeze(1)                      # (T, 1, *space)
        if self.constants is not None:
            c = up(self.constants)                                                  # (Cc, *space)
            if self.channels_last:
                c = c.permute(tuple(range(1, 1 + self.rank)) + (0,)).contiguous()
            d['const'] = c.to(d['pdtype'])
        self._dev = d

    def _generate_device(self, samples, smp_dev=None):
        d = self._dev
        torch, ops = d['torch'], d['ops']
        n = len(samples)
        its, iv = self._input_time_steps, self._interval
        space = tuple(self.array.shape[2:])
        # the gather kernels trust their indices: check every row the batch reads on the host (the host path raises
        # IndexError from numpy's fancy indexing for the same inputs)
        samples = np.asarray(samples)
        T = int(self.array.shape[0])
        if n:
            last = max([iv * (its - 1)] + [t_off + iv * (steps - 1) for t_off, steps in self._windows()])
            if self._sequence is not None and self._add_insolation and self._sequence > 1:
                last = max(last, iv * (its * (self._sequence - 1) + its - 1))
            lo, hi = int(samples.min()), int(samples.max())
            if lo < 0 or hi + last >= T:
                raise IndexError('index %d is out of bounds for axis 0 with size %d' % (lo if lo < 0 else hi + last, T))
        # (pinned + non_blocking: a pageable upload is a SYNCHRONOUS copy on the current stream, i.e. the host waited here for the
        # training step of the previous batch to finish before it could even enqueue this batch's gathers: generator-fed training
        # ran at 1.25 ms per step where the step itself takes 0.65 -- round 6)
        if smp_dev is not None:
            smp = smp_dev                       # (a slice of the epoch's order, already on the device: __getitem__)
        else:
            smp = torch.from_numpy(samples.astype(np.int32))
            smp = smp.pin_memory().to(self.device, non_blocking=True) if self.device.type == 'cuda' else smp.to(self.device)
        cl = self.channels_last
        vin_n, add = self._input_size, self._add_insolation
        cin = its * (vin_n + add)

        def empty(c, dt, lead=()):
            return torch.empty(((n,) + lead + space + (c,)) if cl else ((n,) + lead + (c,) + space), dtype=dt,
                               device=self.device)
        p = empty(cin, d['pdtype'])
        ops.batch_gather(d['array'], smp, d['vin'], p, its, 0, iv, 0, vin_n + add, cl)
        if add:
            ops.batch_gather(d['sol'], smp, d['zero'], p, its, 0, iv, vin_n, vin_n + add, cl)
        targets = []
        for t_off, steps in self._windows():
            t = empty(steps * self._output_size, torch.float32)
            ops.batch_gather(d['array'], smp, d['vout'], t, steps, t_off, iv, 0, self._output_size, cl)
            targets.append(t)
        plist = [p]
        if self._sequence is not None and add:
            for s in range(1, self._sequence):
                # solar inputs keep their time axis: (n, its, *space, 1) / (n, its, 1, *space) == its one-channel gathers
                sol = empty(1, d['pdtype'], lead=(its,))
                flat = sol.view((n * its,) + tuple(sol.shape[2:]))
                idx = (smp.view(-1, 1) + iv * (its * s + torch.arange(its, device=self.device, dtype=torch.int32))).view(-1)
                ops.batch_gather(d['sol'], idx.contiguous(), d['zero'], flat, 1, 0, 1, 0, 1, cl)
                plist.append(sol)
        if 'const' in d:
            plist.append(d['const'].unsqueeze(0).expand((n,) + tuple(d['const'].shape)))
        p_out = plist if len(plist) > 1 else p
        return p_out, (targets if self._sequence is not None else targets[0])


def tf_data_generator(generator, batch_size=None, input_names=None, output_names=None):
    """
    Counterpart of the reference's tf.data wrapper (DLWP/model/generators.py:1014-1074): an iterable over the generator's
    batches whose list-valued inputs / outputs become dicts keyed by the model's Input / output layer names, which
    `DLWPFunctional.fit_generator` consumes directly.  Same argument checks and default names as the reference.
    """
    p, t = generator.generate([0])
    p_is_list, t_is_list = isinstance(p, list), isinstance(t, list)
    if p_is_list:
        if input_names is None:
            input_names = ['input_%d' % (i + 1) for i in range(len(p))]
        if len(input_names) != len(p):
            raise ValueError("mismatched length of input names relative to generated data; got %d but expected %d" %
                             (len(input_names), len(p)))
    if t_is_list:
        if output_names is None:
            output_names = ['output'] + ['output_%d' % i for i in range(1, len(t))]
        if len(output_names) != len(t):
            raise ValueError("mismatched length of input names relative to generated data; got %d but expected %d" %
                             (len(output_names), len(t)))
    del p, t

    class _Dataset(object):
        def __len__(self):
            return len(generator)

        def __iter__(self):
            for x, y in generator:
                if p_is_list:
                    x = {input_names[i]: a for i, a in enumerate(x)}
                if t_is_list:
                    y = {output_names[i]: a for i, a in enumerate(y)}
                yield x, y

        def on_epoch_end(self):
            generator.on_epoch_end()
    return _Dataset()
