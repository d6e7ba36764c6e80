#
# MI355X-native host API of the DLWP-CS functional model wrapper.
#

"""
`DLWPFunctional`: the host-side wrapper the reference scripts drive (reference DLWP/model/models.py:320-472), re-hosted
on `DLWP.keras.Model` (HIP kernels underneath).  Method names, argument meaning, attributes and error behaviour are those
of the reference so that training / inference scripts are drop-in.
"""

import numpy as np


class DLWPFunctional(object):
    """
    DLWP model class wrapping a functional model built with the `DLWP.keras` shim.  Like the reference class it does
    NOT scale or impute data; that is the caller's business.
    """

    def __init__(self, is_convolutional=True, is_recurrent=False, time_dim=1):
        """
        :param is_convolutional: bool: the model consumes / produces spatial shapes
        :param is_recurrent: bool: the model has a recurrent time axis
        :param time_dim: int >= 1: number of time steps in the model's input and output
        """
        self.is_convolutional = is_convolutional
        self.is_recurrent = is_recurrent
        if int(time_dim) < 1:
            raise ValueError("'time_dim' must be >= 1")
        self.time_dim = time_dim

        # attributes other DLWP components read (generators: is_convolutional / is_recurrent / impute)
        self.scaler = None
        self.scaler_y = None
        self.impute = False
        self.imputer = None
        self.imputer_y = None
        self._n_steps = 1

        self.base_model = None
        self.model = None
        self.gpus = 1

        # DLWP >= 0.9.0 compatibility flag
        self.FHW_DIMS = True

    def build_model(self, model, gpus=1, **compile_kwargs):
        """
        Compile a functional model.

        :param model: DLWP.keras.Model
        :param gpus: int: number of GPUs the model is trained on.  The MI355X engine is one-process-per-GPU: with
            gpus > 1 this process must be one rank of a `torch.distributed` (RCCL) job of that world size (launch with
            `python -m torch.distributed.run --nproc-per-node <gpus> ...`); gradients are all-reduced over xGMI.
            (The reference cloned the model onto the CPU and wrapped it in `multi_gpu_model`, models.py:369-374.)
        :param compile_kwargs: passed to the model's `compile`
        """
        if type(gpus) is not int:
            raise TypeError("'gpus' argument must be an int")
        self.base_model = model
        self._n_steps = len(model.outputs)
        if gpus > 1:
            import torch.distributed as dist
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            if world != gpus:
                raise RuntimeError("gpus=%d requested but this process is part of a torch.distributed world of size %d; "
                                   "launch one process per GPU (python -m torch.distributed.run --nproc-per-node %d ...)"
                                   % (gpus, world, gpus))
            self.gpus = gpus
        self.model = self.base_model
        self.model.compile(**compile_kwargs)

    def scaler_transform(self, X, y=None):
        """Identity, for API compatibility with the scaling wrappers."""
        if y is not None:
            return X, y
        else:
            return X

    def fit(self, predictors, targets, **kwargs):
        """
        :param predictors: ndarray (or list of): predictor data
        :param targets: ndarray (or list of): target data
        :param kwargs: passed to the model's `fit`
        """
        self.model.fit(predictors, targets, **kwargs)

    def fit_generator(self, generator, **kwargs):
        """
        :param generator: batch producer: a Sequence-like object (`__len__`/`__getitem__` -> (inputs, targets)) such as
            DLWP's ArrayDataGenerator, or any iterable of (inputs, targets)
        :param kwargs: passed to the model's `fit`
        """
        self.model.fit(generator, **kwargs)

    def predict(self, predictors, **kwargs):
        """
        :param predictors: ndarray (or list of): predictor data
        :param kwargs: passed to the model's `predict`
        :return: ndarray (list of ndarray for a multi-output model)
        """
        return self.model.predict(predictors, **kwargs)

    def predict_timeseries(self, predictors, time_steps, keep_time_dim=False, **kwargs):
        """
        Iterate the model on its own output to produce a forecast of `time_steps` steps: the model is run
        ceil(time_steps / n_outputs / time_dim) times and the outputs are stacked along a leading forecast axis.
        Same bookkeeping as the reference (models.py:418-460): requires output shape == input shape; the first feature
        axis is split by `time_dim` (meaningful for channels_first inputs).

        On the device the state stays resident in HBM between steps (one upload, one download per step for the
        returned series) instead of a numpy round trip per step.

        :param predictors: ndarray: predictor data
        :param time_steps: int: number of time steps to predict forward
        :param keep_time_dim: if True, keep the time_step dimension in the output, otherwise integrates it into the
            forecast_hour (first) dimension
        :param kwargs: passed to `predict` (`verbose` > 0 prints progress)
        :return: ndarray: model prediction; first dim is time
        """
        if isinstance(predictors, (list, tuple)):
            raise NotImplementedError('DLWPFunctional.predict_timeseries cannot use extra inputs at the moment. '
                                      'Use TimeSeriesEstimator instead.')
        time_steps = int(time_steps)
        if time_steps < 1:
            raise ValueError("time_steps must be an int > 0")
        steps = int(np.ceil(time_steps / self._n_steps / self.time_dim))
        out_steps = steps * self._n_steps
        sample_dim = predictors.shape[0]
        feature_shape = predictors.shape[2:] if self.is_recurrent else predictors.shape[1:]
        time_series = np.full((out_steps,) + predictors.shape, np.nan, dtype=np.float32)
        verbose = kwargs.get('verbose', 0)
        rollout = getattr(self.model, 'rollout_on_device', None)
        if rollout is not None:
            rollout(predictors, steps, self._n_steps, time_series, verbose=verbose,
                    batch_size=kwargs.get('batch_size'))
        else:
            state = predictors.copy()
            for t in range(steps):
                if verbose > 0:
                    print('Prediction step %d/%d' % (t + 1, steps))
                result = self.predict(state, **kwargs)
                if self._n_steps == 1:
                    state[:] = result[:]
                    time_series[t] = result
                else:
                    state[:] = result[-1]
                    for s in range(self._n_steps):
                        time_series[t * self._n_steps + s] = result[s]
        time_series = time_series.reshape((out_steps, sample_dim, self.time_dim, -1) + feature_shape[1:])
        if not keep_time_dim:
            time_series = time_series.transpose((0, 2, 1) + tuple(range(3, 3 + len(feature_shape))))
            time_series = time_series.reshape((out_steps * self.time_dim, sample_dim, -1) + feature_shape[1:])
        return time_series

    def evaluate(self, predictors, targets, **kwargs):
        """
        :param predictors: ndarray: predictor data
        :param targets: ndarray: target data
        :param kwargs: passed to the model's `evaluate`
        :return: loss (and metrics)
        """
        score = self.model.evaluate(predictors, targets, **kwargs)
        return score
