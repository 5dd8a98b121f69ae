"""
DLWPNeuralNet / DLWPFunctional: the reference's model wrappers (DLWP/model/models.py:21-316, 319-464) over the HIP back
end.  Same constructor flags, methods, attributes and exception types; the differences are underneath:

  * build_model resolves the (name, args, kwargs) triples in this package's registries (dlwp_amd.layers, then
    dlwp_amd.custom) instead of keras.layers / DLWP.custom, and lowers the stack to fused gfx950 kernels;
  * predict_timeseries keeps the forecast state in HBM for the whole rollout and replays ONE captured hipGraph
    (the reference crosses the host<->device boundary twice per step per 32-sample chunk, models.py:277-293);
  * gpus > 1 means data parallelism with one process per GPU under torch.distributed (RCCL), not
    keras.utils.multi_gpu_model's CPU-hosted replication (models.py:104-109).
"""
import math

import numpy as np

from .. import util


class _Wrapper(object):
    """State and helpers shared by both wrappers."""

    def __init__(self, is_convolutional, is_recurrent, time_dim):
        if int(time_dim) < 1:
            raise ValueError("'time_dim' must be >= 1")
        self.is_convolutional = is_convolutional
        self.is_recurrent = is_recurrent
        self.time_dim = time_dim
        self.scaler = None
        self.scaler_y = None
        self.impute = False
        self.imputer = None
        self.imputer_y = None
        self.base_model = None
        self.model = None
        self.gpus = 1

    # -- rollout plumbing -------------------------------------------------------------------------------------------- #
    def _feature_shape(self, predictors):
        return predictors.shape[2:] if self.is_recurrent else predictors.shape[1:]

    def _finish_series(self, series, n_slots, n_sample, feature_shape, keep_time_dim):
        """(slots, N, <state>) -> (slots, N, time_dim, V, ...) or (slots*time_dim, N, V, ...); never truncated to the
        requested number of steps (reference models.py:294-300, 448-451)."""
        series = series.reshape((n_slots, n_sample, self.time_dim, -1) + tuple(feature_shape[1:]))
        if keep_time_dim:
            return series
        order = (0, 2, 1) + tuple(range(3, series.ndim))
        return series.transpose(order).reshape((n_slots * self.time_dim, n_sample, -1) + tuple(feature_shape[1:]))

    def _device_rollout_ok(self, predictors):
        net = self.model
        return (hasattr(net, 'rollout_on_device') and tuple(predictors.shape[1:]) == tuple(net.inputs[0].shape)
                and all(tuple(o.shape) == tuple(net.inputs[0].shape) for o in net.outputs))

    def _rollout_chunk(self, x, calls, keep_time_dim, fresh=False):
        """All `calls` model applications of the members in device tensor x as one hipGraph; merge of the time axis on the
        device too.  fresh: the result must not alias the graph's cached series buffer."""
        from .. import ops
        series = self.model.rollout_on_device(x, calls)          # (calls*n_out, N) + state shape
        n_slots, n_sample = series.shape[0], series.shape[1]
        fs = tuple(series.shape[3:]) if self.is_recurrent else tuple(series.shape[2:])
        if keep_time_dim:
            out = series.reshape((n_slots, n_sample, self.time_dim, -1) + fs[1:])
            return out.clone() if fresh else out
        flat = series.reshape((n_slots, n_sample, -1) + tuple(series.shape[-2:]))
        out = ops.series_merge_time(flat.contiguous(), self.time_dim)
        return out.reshape((n_slots * self.time_dim, n_sample, -1) + fs[1:])

    #: members per hipGraph launch when the series goes back to the host: the device-to-host copy of one chunk runs
    #: on its own stream under the rollout of the next (results do not depend on the chunking: tests/test_gpu_model.py)
    host_chunk_members = 64

    def _rollout_device(self, predictors, calls, keep_time_dim, return_device=False):
        import torch
        net = self.model
        on_host = not isinstance(predictors, torch.Tensor)
        n = int(predictors.shape[0])
        chunk = int(self.host_chunk_members)
        if return_device or n < 2 * chunk:
            x = predictors if not on_host else \
                torch.from_numpy(np.ascontiguousarray(predictors, dtype=np.float32)).to(net.device)
            out = self._rollout_chunk(x, calls, keep_time_dim)
            return out if return_device else out.cpu().numpy()
        # large ensembles: pipelined over member chunks into ONE pinned host array (time first, as the reference returns
        # it); slot t of a chunk is a contiguous block of it, so every copy is a plain asynchronous DMA
        parts = -(-n // chunk)
        chunk = -(-n // parts)                       # even chunks: one graph shape (plus at most one remainder shape)
        xh = np.ascontiguousarray(predictors, dtype=np.float32) if on_host else None
        copy_stream, up_probed = util.io_streams(net.device)      # (hardware queues of their own: util.distinct_streams)
        copy_stream.wait_stream(torch.cuda.current_stream(net.device))
        host = None
        # host inputs: chunk k + 1 is staged into page-locked memory and uploaded on its own stream while chunk k rolls out (a
        # synchronous upload from pageable memory in front of every chunk left the GPU idle for ~1.5 ms each)
        up_stream = up_probed if on_host else None
        stage = [util.pinned_results.take((chunk,) + tuple(predictors.shape[1:])) for _ in range(2)] if on_host else None
        stage_ev = [None, None]
        bounds = [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]

        def upload(k):
            lo, hi = bounds[k]
            buf = stage[k % 2]
            if stage_ev[k % 2] is not None:
                stage_ev[k % 2].synchronize()             # the upload that last read this staging buffer has drained
            buf[:hi - lo].numpy()[...] = xh[lo:hi]
            with torch.cuda.stream(up_stream):
                t = buf[:hi - lo].to(net.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(up_stream)
            stage_ev[k % 2] = ev
            return t, ev
        try:
          nxt = upload(0) if on_host else None
          for k, (lo, hi) in enumerate(bounds):
              if on_host:
                  xc, ev = nxt
                  torch.cuda.current_stream(net.device).wait_event(ev)
                  xc.record_stream(torch.cuda.current_stream(net.device))
              else:
                  xc = predictors[lo:hi]
              out = self._rollout_chunk(xc, calls, keep_time_dim, fresh=True)
              if on_host and k + 1 < len(bounds):
                  nxt = upload(k + 1)                         # host copy + DMA under this chunk's kernels
              if host is None:         # page-locked, recycled once the caller lets the previous result go (util._PinnedPool)
                  host = util.pinned_results.take((out.shape[0], n) + tuple(out.shape[2:]))
              done = torch.cuda.Event()
              done.record()
              copy_stream.wait_event(done)
              with torch.cuda.stream(copy_stream):
                  # the chunk's (T, members, ...) block -> rows lo:hi of every time slot: one strided DMA (row-by-row otherwise)
                  if not (host.is_pinned() and util.copy2d_d2h_async(host[:, lo:hi], out, copy_stream)):
                      for t in range(out.shape[0]):
                          host[t, lo:hi].copy_(out[t], non_blocking=True)
              out.record_stream(copy_stream)
          copy_stream.synchronize()
        finally:
            if stage is not None:          # (also when a chunk fails: the staging buffers go back to the pool)
                for buf in stage:
                    util.pinned_results._give_back(buf.view(-1))
        return util.pinned_results.lend(host)


class DLWPNeuralNet(_Wrapper):
    """DLWP model class around a Sequential network built from (layer_name, args, kwargs) triples."""

    def __init__(self, is_convolutional=True, is_recurrent=False, time_dim=1, scaler_type='StandardScaler',
                 scale_targets=True, apply_same_y_scaling=True, impute_missing=False):
        super(DLWPNeuralNet, self).__init__(is_convolutional, is_recurrent, time_dim)
        self.scaler_type = scaler_type
        self.scale_targets = scale_targets
        self.apply_same_y_scaling = apply_same_y_scaling
        self.impute = impute_missing
        self._is_init_fit = scaler_type is None

    # -- model construction ------------------------------------------------------------------------------------------ #
    @staticmethod
    def _check_layers(layers):
        if type(layers) not in [list, tuple]:
            raise TypeError("'layers' argument must be a tuple")
        checked = []
        for i, spec in enumerate(layers):
            if type(spec) not in [list, tuple]:
                raise TypeError("each element of 'layers' must be a tuple")
            if len(spec) != 3:
                raise ValueError("each layer must be specified by three elements (name, args, kwargs)")
            name, args, kwargs = spec
            args = () if args is None else args
            kwargs = {} if kwargs is None else kwargs
            if type(args) is not tuple:
                raise TypeError("the 'args' element of layer %d must be a tuple" % i)
            if type(kwargs) is not dict:
                raise TypeError("the 'kwargs' element of layer %d must be a dict" % i)
            checked.append((name, args, kwargs))
        return checked

    def build_model(self, layers=(), gpus=1, **compile_kwargs):
        """Build and compile the network.  Each element of `layers` is (layer_name, layer_args, layer_kwargs) with the
        Keras layer names the reference uses, e.g. ('Conv2D', (32, 3), {'dilation_rate': 2, ...})."""
        if type(gpus) is not int:
            raise TypeError("'gpus' argument must be an int")
        specs = self._check_layers(layers)
        from ..engine import Sequential
        net = Sequential()
        self.base_model = net            # kept current so a failed build can be inspected (examples/train.py:241-244)
        for name, args, kwargs in specs:
            try:
                cls = util.get_from_class('dlwp_amd.layers', name)
            except (ImportError, AttributeError):
                cls = util.get_from_class('dlwp_amd.custom', name)
            net.add(cls(*args, **kwargs))
        self.model = net
        self.gpus = gpus
        if gpus > 1:
            from .. import parallel
            parallel.attach(net, gpus)
        net.compile(**compile_kwargs)

    # -- scaling / imputing (host side, scikit-learn; identity for every convolutional example: scaler_type=None) ----- #
    @staticmethod
    def _flat(a, with_shape=False):
        shape = a.shape
        a = a.reshape((shape[0], -1))
        return (a, shape) if with_shape else a

    def scaler_fit(self, X, y, **kwargs):
        if self.scaler_type is None:
            return
        cls = util.get_from_class('sklearn.preprocessing', self.scaler_type)
        self.scaler, self.scaler_y = cls(**kwargs), cls(**kwargs)
        self.scaler.fit(self._flat(X))
        if self.scale_targets:
            if self.apply_same_y_scaling:
                self.scaler_y = self.scaler
            else:
                self.scaler_y.fit(self._flat(y))

    def scaler_transform(self, X, y=None):
        if self.scaler_type is None:
            return X if y is None else (X, y)
        Xf, xs = self._flat(X, True)
        Xt = self.scaler.transform(Xf).reshape(xs)
        if y is None:
            return Xt
        if not self.scale_targets:
            return Xt, y
        yf, ys = self._flat(y, True)
        return Xt, self.scaler_y.transform(yf).reshape(ys)

    def imputer_fit(self, X, y):
        try:
            cls = util.get_from_class('sklearn.impute', 'SimpleImputer')     # sklearn.preprocessing.Imputer is gone
            make = lambda: cls(missing_values=np.nan, strategy='mean', copy=False)  # noqa: E731
        except (ImportError, AttributeError):
            cls = util.get_from_class('sklearn.preprocessing', 'Imputer')
            make = lambda: cls(missing_values=np.nan, strategy='mean', axis=0, copy=False)  # noqa: E731
        self.imputer, self.imputer_y = make(), make()
        self.imputer.fit(self._flat(X))
        if self.apply_same_y_scaling:
            self.imputer_y = self.imputer
        else:
            self.imputer_y.fit(self._flat(y))

    def imputer_transform(self, X, y=None):
        Xf, xs = self._flat(X, True)
        Xt = self.imputer.transform(Xf).reshape(xs)
        if y is None:
            return Xt
        yf, ys = self._flat(y, True)
        return Xt, self.imputer_y.transform(yf).reshape(ys)

    def init_fit(self, predictors, targets, scaler_kwargs=None):
        """Fit the imputer and scaler without training (for later fit(..., initialize=False) / fit_generator)."""
        if self.impute:
            self.imputer_fit(predictors, targets)
            predictors, targets = self.imputer_transform(predictors, y=targets)
        self.scaler_fit(predictors, targets, **(scaler_kwargs or {}))
        self._is_init_fit = True

    # -- fit / predict / evaluate ------------------------------------------------------------------------------------- #
    def _prepare(self, predictors, targets=None):
        if self.impute:
            if targets is None:
                predictors = self.imputer_transform(predictors)
            else:
                predictors, targets = self.imputer_transform(predictors, targets)
        if targets is None:
            return self.scaler_transform(predictors)
        return self.scaler_transform(predictors, targets)

    def fit(self, predictors, targets, initialize=True, **kwargs):
        if initialize:
            self.init_fit(predictors, targets)
        elif not self._is_init_fit:
            raise AttributeError('DLWPNeuralNet has not been initialized for fitting with init_fit()')
        X, y = self._prepare(predictors, targets)
        val = kwargs.get('validation_data')
        if val is not None:
            kwargs['validation_data'] = self._prepare(*val)
        return self.model.fit(X, y, **kwargs)

    def fit_generator(self, generator, **kwargs):
        from .generators import DataGenerator
        if isinstance(generator, DataGenerator) and not self._is_init_fit:
            raise AttributeError('DLWPNeuralNet has not been initialized for fitting with init_fit()')
        return self.model.fit_generator(generator, **kwargs)

    def predict(self, predictors, **kwargs):
        predicted = self.model.predict(self._prepare(predictors), **kwargs)
        if self.scale_targets and self.scaler_type is not None:
            return self.scaler_y.inverse_transform(predicted)
        return predicted

    def evaluate(self, predictors, targets, **kwargs):
        X, y = self._prepare(predictors, targets)
        return self.model.evaluate(X, y, **kwargs)

    def predict_timeseries(self, predictors, time_steps, step_sequence=False, keep_time_dim=False, **kwargs):
        """Autoregressive forecast of `time_steps` steps.  Returns float32 with time first:
        (ceil(time_steps/time_dim)*time_dim, N, V, ...) -- or (.., N, time_dim, V, ...) with keep_time_dim -- exactly
        the reference's layout (models.py:247-301).  With step_sequence only the first predicted step of each call is
        kept and fed back."""
        time_steps = int(time_steps)
        if time_steps < 1:
            raise ValueError("time_steps must be an int > 0")
        return_device = bool(kwargs.pop('return_device', False))
        n_calls = time_steps if step_sequence else int(math.ceil(1. * time_steps / self.time_dim))
        identity_io = (not self.impute) and (self.scaler_type is None)
        if identity_io and not step_sequence and self._device_rollout_ok(predictors):
            return self._rollout_device(predictors, n_calls, keep_time_dim, return_device)
        # generic host loop (foreign model objects, scalers, step_sequence)
        verbose = kwargs.get('verbose', 0)
        n_sample = predictors.shape[0]
        feature_shape = self._feature_shape(predictors)
        series = np.full((n_calls,) + predictors.shape, np.nan, dtype=np.float32)
        state = np.array(predictors, copy=True)
        for t in range(n_calls):
            if verbose > 0:
                print('Time step %d/%d' % (t + 1, n_calls))
            out = self.predict(state, **kwargs)
            series[t] = out
            if not step_sequence:
                state = np.array(out, copy=True)
            elif self.is_recurrent:
                state = np.concatenate([state[:, 1:], out[:, :1]], axis=1)
            else:
                split = (n_sample, self.time_dim, -1) + tuple(feature_shape[1:])
                state = np.concatenate([state.reshape(split)[:, 1:], out.reshape(split)[:, :1]],
                                       axis=1).reshape(predictors.shape)
        merged = self._finish_series(series, n_calls, n_sample, feature_shape, keep_time_dim or step_sequence)
        if step_sequence and not keep_time_dim:
            merged = merged[:, :, 0]
        return merged


class DLWPFunctional(_Wrapper):
    """DLWP model class around a functional dlwp_amd.engine.Model, possibly with several chained outputs
    (examples/train_functional.py:281-285).  No scaling / imputing, as in the reference."""

    def __init__(self, is_convolutional=True, is_recurrent=False, time_dim=1):
        super(DLWPFunctional, self).__init__(is_convolutional, is_recurrent, time_dim)
        self._n_steps = 1

    def build_model(self, model, gpus=1, **compile_kwargs):
        if type(gpus) is not int:
            raise TypeError("'gpus' argument must be an int")
        self.base_model = model
        self.model = model
        self._n_steps = len(model.outputs)
        self.gpus = gpus
        if gpus > 1:
            from .. import parallel
            parallel.attach(model, gpus)
        model.compile(**compile_kwargs)

    def scaler_transform(self, X, y=None):
        return X if y is None else (X, y)

    def fit(self, predictors, targets, **kwargs):
        return self.model.fit(predictors, targets, **kwargs)

    def fit_generator(self, generator, **kwargs):
        return self.model.fit_generator(generator, **kwargs)

    def predict(self, predictors, **kwargs):
        return self.model.predict(predictors, **kwargs)

    def evaluate(self, predictors, targets, **kwargs):
        return self.model.evaluate(predictors, targets, **kwargs)

    def predict_timeseries(self, predictors, time_steps, keep_time_dim=False, **kwargs):
        """Forecast with a model that emits `_n_steps` consecutive states per call; the last one seeds the next call
        (reference models.py:414-452).  Output layout as DLWPNeuralNet.predict_timeseries."""
        time_steps = int(time_steps)
        if time_steps < 1:
            raise ValueError("time_steps must be an int > 0")
        return_device = bool(kwargs.pop('return_device', False))
        n_calls = int(math.ceil(time_steps / self._n_steps / self.time_dim))
        if self._device_rollout_ok(predictors):
            return self._rollout_device(predictors, n_calls, keep_time_dim, return_device)
        verbose = kwargs.get('verbose', 0)
        n_slots = n_calls * self._n_steps
        n_sample = predictors.shape[0]
        series = np.full((n_slots,) + predictors.shape, np.nan, dtype=np.float32)
        state = np.array(predictors, copy=True)
        for t in range(n_calls):
            if verbose > 0:
                print('Prediction step %d/%d' % (t + 1, n_calls))
            result = self.predict(state, **kwargs)
            if self._n_steps == 1:
                series[t] = result
                state = np.array(result, copy=True)
            else:
                series[t * self._n_steps:(t + 1) * self._n_steps] = np.stack(result, axis=0)
                state = np.array(result[-1], copy=True)
        return self._finish_series(series, n_slots, n_sample, self._feature_shape(predictors), keep_time_dim)
