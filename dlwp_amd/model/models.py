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

    #: a series of at least this many bytes goes back to the host SLOT BY SLOT while the rollout runs (_rollout_streamed); smaller
    #: ones after it, in one copy
    host_stream_bytes = 64 << 20
    #: member chunks the first model call of a streamed rollout is cut into when the predictors come from the host: the call
    #: starts on the first chunk while the others are still being uploaded
    host_head_chunks = 4

    def _rollout_device(self, predictors, calls, keep_time_dim, return_device=False):
        import torch
        net = self.model
        on_host = not isinstance(predictors, torch.Tensor)
        n = int(predictors.shape[0])
        member = int(np.prod(predictors.shape[1:]))
        n_out = len(net.outputs)
        run = member // int(self.time_dim)
        streamed = (not return_device and calls >= 2 and 4 * n * member * calls * n_out >= int(self.host_stream_bytes)
                    and net.device.type == 'cuda' and (keep_time_dim or run % 4 == 0) and member % 4 == 0)
        if not streamed:
            x = predictors if not on_host else \
                torch.from_numpy(np.ascontiguousarray(predictors, dtype=np.float32)).to(net.device)
            out = self._rollout_chunk(x, calls, keep_time_dim)
            return out if return_device else out.cpu().numpy()
        return self._rollout_streamed(predictors, calls, keep_time_dim)

    def _rollout_streamed(self, predictors, calls, keep_time_dim):
        """The rollout with its series going home WHILE it runs (the metric's own API -- numpy in, numpy out,
        DLWP/model/models.py:265-301, call site examples/plot_forecasts.py:234-239 -- at the device-resident speed).
        The rollout is one hipGraph PER MODEL CALL (engine.StreamedRollout), all members in each, launched back to back on the
        main stream; behind call j an event, and on a copy stream (two, alternating; hardware queues of their own) the slots call j
        wrote leave for ONE page-locked result array laid out as the reference returns it (time first) under call j + 1:
        keep_time_dim: one contiguous asynchronous copy (hipMemcpyAsync) straight from the series slot; otherwise the sample <->
        time transposition of the slot into a staging buffer (dlwp_series_merge_time, ~25 us) and ONE contiguous copy of both
        time steps.  What the runtime does with such a copy on this ROCm (rocprofv3 --memory-copy-trace,
        profiles/r5_host_visible_copies.txt): NOT a copy-engine (SDMA) transfer -- uploads are, downloads into page-locked memory
        run as the runtime's blit kernel, `__amd_rocclr_copyBuffer`, 1.0 ms per 65 MB slot = 56 GB/s, and no switch of the
        runtime moves them (GPU_FORCE_BLIT_COPY_SIZE=0, DEBUG_CLR_LIMIT_BLIT_WG: same trace, same rate).  It shares the chip
        gracefully: the 34.6 ms rollout stays 34.6 ms beside 1 GB of such copies, where r5's own store kernel (a few workgroups
        writing the mapped result array, transposition folded in; 55-58 GB/s alone) took it to 50.9-53.4 ms at 8, 16 or 32
        workgroups (profiles/r5_d2h_forms.json) -- removed.  The rollout is now LINK-bound: 1.82 GB leave in ~34.5 ms = 52.7 GB/s
        of the 56.7 GB/s a lone copy reaches on this box.
        The pipeline's fill is one model call and its drain one slot's transfer (r4's member chunks: a quarter of the rollout
        each).  Host predictors: call 0 is cut into member chunks, chunk c + 1 is gathered into page-locked staging by the
        library's host threads and uploaded while chunk c computes.  Same kernels on the same data as the one-graph rollout:
        bit-identical (tests/test_gpu_model.py)."""
        import ctypes
        import os
        import torch
        from .. import _lib
        net = self.model
        dev = net.device
        on_host = not isinstance(predictors, torch.Tensor)
        n = int(predictors.shape[0])
        td = int(self.time_dim)
        sr = net.streamed_rollout(n, calls, int(self.host_head_chunks) if on_host else 1)
        n_out = sr.n_out
        member = int(sr.s0[0].numel())
        run = member // td
        feat = tuple(predictors.shape[1:])
        fs = feat[1:] if self.is_recurrent else feat
        slots = calls * n_out
        if keep_time_dim:
            out_shape = (slots, n, td, member // td // int(np.prod(fs[1:]))) + tuple(fs[1:])
        else:
            out_shape = (slots * td, n, run // int(np.prod(fs[1:]))) + tuple(fs[1:])
        host = util.pinned_results.take(out_shape)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        h = _lib.handle(idx)
        main = torch.cuda.current_stream(dev)
        down, up = util.d2h_streams(dev), util.io_streams(dev)[1]
        for s_ in down:
            s_.wait_stream(main)
        stage = None
        if not keep_time_dim:          # one transposed call per copy stream in flight
            stage = [sr.__dict__.setdefault('_stage%d' % k, torch.empty(n_out * td * n * run, dtype=torch.float32, device=dev))
                     for k in range(len(down))]
        series_ptr = sr.series.data_ptr()
        slot_bytes = 4 * n * member

        def send(call, k):
            """(copy stream k) the slots of `call` -> their place in the result array"""
            st = down[k]
            sp = ctypes.c_void_p(st.cuda_stream)
            src = series_ptr + call * n_out * slot_bytes
            with torch.cuda.stream(st):
                if keep_time_dim:
                    host.view(-1)[call * n_out * n * member:(call + 1) * n_out * n * member].copy_(
                        sr.series.view(-1)[call * n_out * n * member:(call + 1) * n_out * n * member], non_blocking=True)
                else:
                    _lib.check(_lib.lib.dlwp_series_merge_time(h, ctypes.c_void_p(src), ctypes.c_void_p(stage[k].data_ptr()), n_out, n,
                                                               td, 1, run, _lib.F32, sp))
                    host.view(-1)[call * n_out * n * member:(call + 1) * n_out * n * member].copy_(stage[k], non_blocking=True)

        pins = []
        try:
            # ---- call 0: in member chunks, each behind its own upload
            if on_host:
                xh = np.ascontiguousarray(predictors, dtype=np.float32).reshape(n, member)
                bounds = sr.chunk_bounds()
                pins = [util.pinned_results.take((hi - lo, member)) for lo, hi in bounds[:2]]
                pin_ev = [None, None]
                up.wait_stream(main)
                threads = max(1, min(8, (os.cpu_count() or 2) // 2))
                for c, (lo, hi) in enumerate(bounds):
                    buf = pins[c % 2]
                    if pin_ev[c % 2] is not None:
                        pin_ev[c % 2].synchronize()           # the upload that last read this staging buffer has drained
                    rows = np.arange(lo, hi, dtype=np.int64)
                    if buf.is_pinned():
                        _lib.check(_lib.lib.dlwp_host_gather_rows(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(xh.ctypes.data),
                                                                  rows.ctypes.data_as(ctypes.c_void_p), hi - lo, 4 * member, n, threads))
                    else:
                        buf.numpy()[...] = xh[lo:hi]
                    with torch.cuda.stream(up):
                        sr.s0.view(n, member)[lo:hi].copy_(buf, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(up)
                    pin_ev[c % 2] = ev
                    main.wait_event(ev)
                    sr.head[c].launch()
            else:
                sr.s0.copy_(predictors.reshape(sr.s0.shape))
                for g in sr.head:
                    g.launch()
            ev = torch.cuda.Event()
            ev.record(main)
            down[0].wait_event(ev)
            send(0, 0)
            # ---- calls 1 ...: all members, slot j leaves under call j + 1
            for j, g in enumerate(sr.tail, start=1):
                g.launch()
                ev = torch.cuda.Event()
                ev.record(main)
                k = j % len(down)
                down[k].wait_event(ev)
                send(j, k)
            for s_ in down:
                s_.synchronize()
            # (the cached series / staging buffers are rewritten by the next call on the main stream: order it behind the copies)
            for s_ in down:
                main.wait_stream(s_)
        except BaseException:
            #  a failed call: copies on the download streams may still be writing into the result array -- let them
            # drain, then the array goes back to the pool instead of leaking its page-locked memory
            for s_ in down:
                try:
                    s_.synchronize()
                except Exception:  # noqa: BLE001
                    pass
            util.pinned_results._give_back(host.view(-1))
            raise
        finally:
            if pins:                       # (also when a call fails: no upload may still be reading a staging buffer that goes
                up.synchronize()           #  back to the pool)
            for buf in pins:
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
            # (from a plain single process the other ranks are started here and build the same network from this specification:
            #  the reference's keras.utils.multi_gpu_model is a single-process call too, models.py:104-109)
            wargs = dict(is_convolutional=self.is_convolutional, is_recurrent=self.is_recurrent, time_dim=self.time_dim,
                         scaler_type=self.scaler_type, scale_targets=self.scale_targets,
                         apply_same_y_scaling=self.apply_same_y_scaling, impute_missing=self.impute)
            parallel.attach(net, gpus, spec=('sequential', wargs, tuple(specs), dict(compile_kwargs)), wrapper=self)
            with parallel.unmirrored(net):
                net.compile(**compile_kwargs)
            return
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
        drv = getattr(self.model, '__dict__', {}).get('_driver')
        if drv is not None and not drv.in_call and not drv.closed and not return_device and drv.world > 1:
            return drv.sharded_rollout(self, predictors, time_steps, dict(kwargs, step_sequence=step_sequence,
                                                                           keep_time_dim=keep_time_dim))
        n_calls = time_steps if step_sequence else int(math.ceil(1. * time_steps / self.time_dim))
        identity_io = (not self.impute) and (self.scaler_type is None)
        if identity_io and not step_sequence and self._device_rollout_ok(predictors):
            return self._rollout_device(predictors, n_calls, keep_time_dim, return_device)
        if identity_io and step_sequence and self._device_rollout_ok(predictors) and self.model.device.type == 'cuda' and \
                hasattr(self.model, 'fed_rollout_on_device'):
            # one predicted step per call: the next input is the last time_dim - 1 input steps + the FIRST predicted step
            # (reference models.py:280-290: two host reshapes + a concatenate around a predict round trip per step) -- a channel
            # shift of the state and a window of the output, done by the feedback launch between the calls of ONE hipGraph
            td = int(self.time_dim)
            state_c = int(np.prod(predictors.shape[1:-2]))
            v = state_c // td
            if state_c <= 128 and v * td == state_c:
                src = [c + v for c in range(state_c - v)] + [-1 - j for j in range(v)]
                if kwargs.get('verbose', 0) > 0:
                    for t in range(n_calls):
                        print('Time step %d/%d' % (t + 1, n_calls))
                series = self.model.fed_rollout_on_device(predictors, n_calls, src)
                merged = series.reshape((n_calls, int(predictors.shape[0]), td, -1) + tuple(predictors.shape[-2:]))
                if not keep_time_dim:
                    merged = merged[:, :, 0]
                return merged.contiguous() if return_device else merged.cpu().numpy()
        # generic host loop (foreign model objects, scalers)
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
            import os
            from .. import parallel
            spec = None
            if parallel.needs_spawn():          # a plain single process: the workers load the graph from a file
                path = os.path.join(parallel._shm_dir(), 'dlwp_graph_%d_%d.npz' % (os.getpid(), id(model) & 0xffff))
                model.save(path)
                wargs = dict(is_convolutional=self.is_convolutional, is_recurrent=self.is_recurrent, time_dim=self.time_dim)
                spec = ('functional', wargs, path, dict(compile_kwargs))
            try:
                parallel.attach(model, gpus, spec=spec, wrapper=self)
                with parallel.unmirrored(model):
                    model.compile(**compile_kwargs)       # (its parameter broadcast ends only when the workers have built theirs)
            finally:
                if spec is not None:
                    try:
                        os.unlink(spec[2])
                    except OSError:
                        pass
            return
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
        drv = getattr(self.model, '__dict__', {}).get('_driver')
        if drv is not None and not drv.in_call and not drv.closed and not return_device and drv.world > 1:
            return drv.sharded_rollout(self, predictors, time_steps, dict(kwargs, keep_time_dim=keep_time_dim))
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
