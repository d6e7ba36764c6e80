"""
`TimeSeriesEstimator` of the MI355X engine (reference DLWP/model/extensions.py:23-481): roll a trained cubed-sphere model
forward in time, feeding its predictions back in as inputs and re-injecting the KNOWN inputs (top-of-atmosphere insolation,
constants) at every step.  This is the production inference path of the reference (Tutorial 4).

What is built: the configuration the DLWP-CS scripts use (Azure/train_cs.py, Tutorial 3/4) --
  * `DLWPFunctional` model (one or several integration steps, `_n_steps` outputs), `channels_last` generator without a time
    axis, input time steps == output time steps, every input variable predicted by the model;
  * inputs `[main_input, solar_1.., constants]` in the generator's order.
On the device the whole rollout runs through `DLWP.keras.Model.rollout_with_forcing`: the state never leaves HBM between
steps (the reference pays a numpy round trip, three concatenates and two transposes per step: extensions.py:273-306), the
series is downloaded once.  Coordinates are plain numpy arrays (xarray is not part of this stack): the result is a small
`Forecast` record with `.values`, `.dims` and `.coords` laid out exactly like the reference's DataArray.
Not built (raises NotImplementedError): imputation, generators whose inputs are not all predicted, rank-2 lat/lon data,
`keep_time_axis` generators, output time steps != input time steps.  `interval` != 1 is served as the reference serves it
(see `predict`).
"""
import numpy as np

from .models import DLWPFunctional


class Forecast(object):
    """values + named dims + coordinate arrays (the subset of xarray.DataArray the verification code reads)."""

    def __init__(self, values, dims, coords, name='forecast'):
        self.values = values
        self.dims = tuple(dims)
        self.coords = dict(coords)
        self.name = name

    @property
    def shape(self):
        return self.values.shape

    def __array__(self, dtype=None):
        return np.asarray(self.values, dtype=dtype)

    def isel(self, **indexers):
        idx = [slice(None)] * self.values.ndim
        coords = dict(self.coords)
        for k, v in indexers.items():
            ax = self.dims.index(k)
            idx[ax] = v
            coords[k] = np.asarray(self.coords[k])[v]
        vals = self.values[tuple(idx)]
        dims = tuple(d for d, i in zip(self.dims, idx) if not isinstance(i, (int, np.integer)))
        return Forecast(vals, dims, {d: coords[d] for d in dims if d in coords}, self.name)


def _values(x):
    return np.asarray(getattr(x, 'values', x))


class TimeSeriesEstimator(object):
    """
    Sophisticated wrapper class for producing time series forecasts from a DLWP model, using a Generator with metadata
    (reference DLWP/model/extensions.py:23-30).
    """

    def __init__(self, model, generator, sample_times=None, dt=None, lat=None, lon=None, varlev=None):
        """
        :param model: DLWPFunctional instance
        :param generator: ArrayDataGenerator (or any object with its attributes: `generate`, `_input_time_steps`,
            `_output_time_steps`, `_interval`, `_add_insolation`, `insolation_array`, `constants`, `channels_last`,
            `convolution_shape`, `output_convolution_shape`, `_n_sample`); a `.ds` with `.sample / .lat / .lon` is used for
            the coordinates when present
        :param sample_times: optional datetime64 array, time of every row of the generator's array (coordinate `time`;
            also lets the insolation be computed beyond the end of the data from `lat` / `lon`)
        :param dt: optional timedelta64 between rows (default: from sample_times)
        :param lat, lon: optional (6, N, N) coordinates for DLWP.util.insolation
        :param varlev: optional names of the output channels per time step
        """
        if not isinstance(model, DLWPFunctional):
            raise NotImplementedError('TimeSeriesEstimator: the engine serves DLWPFunctional models')
        self.model = model
        self.generator = generator
        g = generator
        self.rank = len(g.convolution_shape) - 1
        self._add_insolation = bool(g._add_insolation)
        self._input_time_steps = int(g._input_time_steps)
        self._output_time_steps = int(g._output_time_steps)
        self._interval = int(getattr(g, '_interval', 1))
        self.channels_last = bool(getattr(g, 'channels_last', False))
        if not self.channels_last or getattr(g, '_keep_time_axis', False):
            raise NotImplementedError('TimeSeriesEstimator: channels_last generators without a time axis (the cubed-sphere '
                                      'configuration, Azure/train_cs.py:150-160)')
        if self._output_time_steps != self._input_time_steps:
            raise NotImplementedError('TimeSeriesEstimator: output_time_steps must equal input_time_steps')
        ds = getattr(g, 'ds', None)
        if sample_times is None and ds is not None and hasattr(ds, 'sample'):
            sample_times = _values(ds.sample)
        self._sample_times = None if sample_times is None else np.asarray(sample_times)
        if dt is None and self._sample_times is not None and len(self._sample_times) > 1:
            dt = self._sample_times[1] - self._sample_times[0]
        self._dt = dt
        self._lat = _values(lat) if lat is not None else (_values(ds.lat) if ds is not None and hasattr(ds, 'lat') else None)
        self._lon = _values(lon) if lon is not None else (_values(ds.lon) if ds is not None and hasattr(ds, 'lon') else None)
        n_var_out = g.output_convolution_shape[-1] // self._output_time_steps
        n_var_in = g.convolution_shape[-1] // self._input_time_steps - self._add_insolation
        if n_var_in != n_var_out:
            raise NotImplementedError('TimeSeriesEstimator: every input variable must be predicted by the model '
                                      '(%d inputs, %d outputs per time step)' % (n_var_in, n_var_out))
        self._output_sel = {'varlev': np.arange(n_var_out) if varlev is None else np.asarray(varlev)}
        if hasattr(g, 'constants') and g.constants is not None:
            self.constants = np.asarray(g.constants).transpose(tuple(range(1, 1 + self.rank)) + (0,))
        else:
            self.constants = None

    @property
    def shape(self):
        return (self.generator._n_sample,) + tuple(self.generator.shape)

    @property
    def convolution_shape(self):
        return (self.generator._n_sample,) + tuple(self.generator.convolution_shape)

    # ------------------------------------------------------------------------------------------------------------- #
    def _insolation_rows(self, n_rows):
        """insolation on the data's time grid for rows [0, n_rows): the generator's array, extended past its end with
        DLWP.util.insolation when times and coordinates are known (the reference always recomputes, extensions.py:279-287)."""
        have = np.asarray(self.generator.insolation_array, dtype=np.float32)
        if n_rows <= have.shape[0]:
            return have
        if self._sample_times is None or self._dt is None or self._lat is None or self._lon is None:
            raise IndexError('TimeSeriesEstimator: the forecast needs insolation for %d time rows but the generator holds %d; '
                             'pass sample_times / lat / lon so that it can be computed' % (n_rows, have.shape[0]))
        from ..util import insolation
        t0 = self._sample_times[0]
        extra_t = [t0 + k * self._dt for k in range(have.shape[0], n_rows)]
        return np.concatenate([have, insolation(extra_t, self._lat, self._lon)], axis=0)

    def predict(self, steps, samples=(), impute=False, keep_time_dim=False, prefer_first_times=True,
                f_hour_timedelta_type=False, **kwargs):
        """
        Step forward the time series prediction from the model 'steps' times, feeding predictions back in as inputs
        (reference extensions.py:162-190).

        :param steps: int: number of times to step forward
        :param samples: list of int: which samples in the generator to predict for; () = all
        :param keep_time_dim: bool: keep the time_step dimension instead of integrating it with f_hour
        :param f_hour_timedelta_type: bool: f_hour as timedelta instead of float hours
        :return: Forecast with dims ('f_hour', 'time', ['time_step',] 'x0', 'x1', 'x2', 'varlev')
        """
        if int(steps) < 1:
            raise ValueError('must use positive integer for steps')
        if impute:
            raise NotImplementedError('TimeSeriesEstimator: impute')
        steps = int(steps)
        g = self.generator
        its, ots, iv = self._input_time_steps, self._output_time_steps, self._interval
        # interval != 1: the reference's feedback loop never reads it (the insolation times of extensions.py:279-287 and the
        # re-assembly :289-306 advance by dt, not interval * dt) -- it reaches the initial batches (the generator's) and the
        # f_hour coordinates only.  Same here, pinned by g10_estimator_interval2.npz.
        es = ots                                                   # keep_inputs branch (ots <= its), extensions.py:196-199
        effective_steps = int(np.ceil(steps / es))
        samples = np.arange(g._n_sample, dtype=np.int64) if len(samples) == 0 else np.asarray(samples, dtype=np.int64)
        predictors, t = g.generate(samples)
        t0 = t[0] if isinstance(t, (list, tuple)) else t
        t_shape = tuple(t0.shape)
        n_steps, time_dim = self.model._n_steps, self.model.time_dim
        if iv != 1 and n_steps == 1:
            # extensions.py:333,362: the single-step branch advances its sample coordinate by (es + interval - 1) * dt
            raise NotImplementedError('TimeSeriesEstimator: interval != 1 with a single-step model')
        verbose = kwargs.get('verbose', 0)
        sol = None
        if n_steps > 1 and not self._add_insolation:
            # the reference falls back to predict_timeseries here (extensions.py:255-262)
            result = self.model.predict_timeseries(predictors, steps, keep_time_dim=True, **kwargs)
            result = np.asarray(result).reshape((-1,) + t_shape)[:effective_steps]
            result = np.moveaxis(result, 0, 1)
        else:
            sequence_steps = int(np.ceil(steps / n_steps / time_dim)) if n_steps > 1 else effective_steps
            if self._add_insolation:
                need = int(samples.max()) + sequence_steps * its * n_steps + n_steps * its
                sol = self._insolation_rows(need)
            plist = list(predictors) if isinstance(predictors, (list, tuple)) else [predictors]
            rollout = getattr(self.model.model, 'rollout_with_forcing', None)
            if rollout is not None:
                series = rollout(plist if len(plist) > 1 else plist[0], sequence_steps, insolation=sol,
                                 start_index=samples if sol is not None else None, io_time_steps=its, verbose=verbose)
                series = series.cpu().numpy()                      # ONE download: (sequence_steps, n_steps, B, ...)
                result = np.moveaxis(series, 2, 0)
            else:
                result = self._host_loop(plist, sequence_steps, sol, samples, **kwargs)
            result = result.reshape((t_shape[0], -1) + t_shape[1:])[:, :effective_steps]
        B = result.shape[0]
        space = tuple(g.output_convolution_shape[-self.rank - 1:-1])
        rv = result.reshape((B, effective_steps) + space + (ots, -1))
        if self._dt is None:
            dt_h = 1.0
        elif f_hour_timedelta_type:
            dt_h = self._dt
        else:
            dt_h = float(np.timedelta64(self._dt) / np.timedelta64(1, 'h'))
        if self._sample_times is not None:
            time_coord = self._sample_times[samples] + (its - 1) * self._dt
        else:
            time_coord = samples + (its - 1)
        grid = [np.arange(d) for d in space]
        xdims = ['x%d' % d for d in range(self.rank)]
        if keep_time_dim:
            vals = rv.transpose((1, 0, -2) + tuple(range(2, 2 + self.rank)) + (-1,))
            f_hour = np.arange(1, effective_steps * (es + iv - 1) + 1, es + iv - 1) * dt_h
            return Forecast(vals, ['f_hour', 'time', 'time_step'] + xdims + ['varlev'],
                            dict(zip(['f_hour', 'time', 'time_step'] + xdims + ['varlev'],
                                     [f_hour, time_coord, np.arange(ots)] + grid + [self._output_sel['varlev']])))
        vals = rv.transpose((1, -2, 0) + tuple(range(2, 2 + self.rank)) + (-1,)).reshape(
            (rv.shape[1] * rv.shape[-2], rv.shape[0]) + space + (-1,))
        f_hour = np.array([(np.arange(0, es) + iv + e * (es - 1 + iv)) for e in range(effective_steps)]).flatten() * dt_h
        fc = Forecast(vals, ['f_hour', 'time'] + xdims + ['varlev'],
                      dict(zip(['f_hour', 'time'] + xdims + ['varlev'],
                               [f_hour, time_coord] + grid + [self._output_sel['varlev']])))
        return fc.isel(f_hour=slice(0, steps))

    def _host_loop(self, plist, sequence_steps, sol, samples, **kwargs):
        """The reference's own loop (extensions.py:268-306) around `model.predict`, for model objects that do not expose
        the device-resident rollout (e.g. a wrapped third-party predictor): same bookkeeping, one host round trip per step."""
        its, ots = self._input_time_steps, self._output_time_steps
        n_steps = self.model._n_steps
        rank = self.rank
        fwd_tr = (0, rank + 1) + tuple(range(1, 1 + rank)) + (-1,)
        bwd_tr = (0,) + tuple(range(2, 2 + rank)) + (1, -1)
        p = [np.asarray(a) for a in plist]
        B = p[0].shape[0]
        space = p[0].shape[1:-1]
        result = None
        new_t = np.asarray(samples, dtype=np.int64).copy()
        for s in range(sequence_steps):
            if kwargs.get('verbose', 0) > 0:
                print('Time step %d/%d' % (s + 1, sequence_steps))
            outs = self.model.predict(p if len(p) > 1 else p[0], **kwargs)
            outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
            if result is None:
                result = np.full((B, sequence_steps, n_steps) + tuple(outs[0].shape[1:]), np.nan, dtype=np.float32)
            result[:, s] = np.stack(outs, axis=1)
            if s + 1 == sequence_steps:
                break
            new_t = new_t + ots * n_steps
            last = result[:, s, -1]
            if sol is None:
                p = [last] + p[1:]
                continue
            new_ins = [np.concatenate([np.expand_dims(sol[new_t + n + m * its][:, None], axis=-1) for n in range(its)], axis=1)
                       for m in range(n_steps)]
            r = last.reshape(tuple(last.shape[:-1]) + (ots, -1)).transpose(fwd_tr)
            p = [np.concatenate([r, new_ins[0]], axis=-1).transpose(bwd_tr).reshape((B,) + tuple(space) + (-1,))] + new_ins[1:]
            if self.constants is not None:
                p.append(np.repeat(np.expand_dims(self.constants, axis=0), B, axis=0))
        return result
