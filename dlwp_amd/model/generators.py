"""
Batch feed for fit_generator.

  DataGenerator   the reference's keras.utils.Sequence (DLWP/model/generators.py:19-159): same constructor, shape
                  properties, shuffle order, NaN-sample removal and (X, y) batches of shape (n,)+convolution_shape.
  ArrayDataset    an in-memory stand-in for the xarray Dataset the reference reads (dims, predictors, targets, isel):
                  xarray / netCDF4 are absent from this image and stay out of scope.
  DeviceLoader    what replaces Keras' worker *processes* + per-batch feed_dict copy: a background thread gathers batch
                  i+1 into a pinned host buffer and a copy stream moves it to HBM while batch i trains
                  (pinned-host -> HBM double buffering).
"""
import threading

import numpy as np

from ..util import delete_nan_samples


class _Var(object):
    def __init__(self, values):
        self.values = values

    @property
    def shape(self):
        return self.values.shape


class ArrayDataset(object):
    """Duck-typed predictor file: arrays laid out (sample, [time_step,] varlev..., lat, lon), float32."""

    def __init__(self, predictors, targets, dims=None):
        predictors = np.asarray(predictors)
        targets = np.asarray(targets)
        if dims is None:
            dims = ('sample', 'time_step', 'varlev', 'lat', 'lon') if predictors.ndim == 5 else \
                ('sample', 'varlev', 'lat', 'lon')
        if len(dims) != predictors.ndim:
            raise ValueError('dims %r do not match predictors of rank %d' % (dims, predictors.ndim))
        self._dims = tuple(dims)
        self.predictors = _Var(predictors)
        self.targets = _Var(targets)
        self.dims = dict(zip(dims, predictors.shape))

    def isel(self, sample=slice(None)):
        return ArrayDataset(self.predictors.values[sample], self.targets.values[sample], self._dims)

    def close(self):
        pass


class DataGenerator(object):
    """Generates (predictors, targets) batches on the fly from a dataset with `predictors` and `targets` variables."""

    def __init__(self, model, ds, batch_size=32, shuffle=False, remove_nan=True):
        if not hasattr(ds, 'predictors') or not hasattr(ds, 'targets'):
            raise ValueError("dataset must have 'predictors' and 'targets' variables")
        self.model = model
        self.ds = ds
        self._batch_size = batch_size
        self._shuffle = shuffle
        self._remove_nan = remove_nan
        self._is_convolutional = model.is_convolutional
        self._keep_time_axis = model.is_recurrent
        self._impute_missing = model.impute
        self._n_sample = ds.dims['sample']
        self._has_time_step = 'time_step' in ds.dims
        self._indices = []
        self.on_epoch_end()

    # -- shapes ------------------------------------------------------------------------------------------------------ #
    @property
    def batch_size(self):
        return self._batch_size

    @property
    def shape(self):
        """(time_step, [variable, level,] lat, lon); a singleton time_step is added when the file has none."""
        s = tuple(self.ds.predictors.shape[1:])
        return s if self._has_time_step else (1,) + s

    @property
    def n_features(self):
        return int(np.prod(self.shape))

    @property
    def dense_shape(self):
        if self._keep_time_axis:
            return (self.shape[0], self.n_features // self.shape[0])
        return (self.n_features,)

    def _conv_shape(self, keep_time_axis):
        s = self.shape
        if keep_time_axis:
            return (s[0], int(np.prod(s[1:-2]))) + tuple(s[-2:])
        return (int(np.prod(s[:-2])),) + tuple(self.ds.predictors.shape[-2:])

    @property
    def convolution_shape(self):
        """(channels, y, x), or (time_step, channels, y, x) for a recurrent model; channels = time_step-major."""
        return self._conv_shape(self._keep_time_axis)

    @property
    def shape_2d(self):
        return self._conv_shape(False)

    # -- batches ----------------------------------------------------------------------------------------------------- #
    def on_epoch_end(self):
        self._indices = np.arange(self._n_sample)
        if self._shuffle:
            np.random.shuffle(self._indices)     # legacy global RandomState, as the reference (generators.py:103-106)

    def generate(self, samples, scale_and_impute=True):
        """Batch for the given sample indices; an empty list means every sample."""
        ds = self.ds.isel(sample=samples if len(samples) > 0 else slice(None))
        p = ds.predictors.values
        t = ds.targets.values
        ds.close()
        p = p.reshape((p.shape[0], -1))
        t = t.reshape((t.shape[0], -1))
        if self._remove_nan:
            p, t = delete_nan_samples(p, t)
        if scale_and_impute:
            if self._impute_missing:
                p, t = self.model.imputer_transform(p, t)
            p, t = self.model.scaler_transform(p, t)
        # the reference reshapes with the PRE-deletion sample count (generators.py:113,129) and therefore raises as soon
        # as a NaN sample was actually dropped; use the surviving count
        n = p.shape[0]
        if self._is_convolutional:
            p = p.reshape((n,) + self.convolution_shape)
            t = t.reshape((n,) + self.convolution_shape)
        elif self._keep_time_axis:
            p = p.reshape((n,) + self.dense_shape)
            t = t.reshape((n,) + self.dense_shape)
        return p, t

    def __len__(self):
        return int(np.ceil(self._n_sample / self._batch_size))

    def __getitem__(self, index):
        if int(index) < 0:
            index = len(self) + index
        # the reference tests `index > len(self)`, so index == len(self) silently returns the WHOLE dataset (empty index
        # list -> "all samples"); raise instead
        if index >= len(self) or index < 0:
            raise IndexError('batch index out of range')
        return self.generate(self._indices[index * self._batch_size:(index + 1) * self._batch_size])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class DeviceLoader(object):
    """Pinned-host -> HBM double-buffered feed over any Sequence-like generator (`__len__`, `__getitem__`).

        for X, y in DeviceLoader(gen, device):   # X, y are device tensors, valid until the next iteration
            ...

    A worker thread runs gen[i+1] (numpy gather on the host) and stages it into one of two pinned buffers while the
    consumer works on batch i; the H2D copies are issued on a private copy stream and the consumer's stream waits on
    their event, so compute and transfer overlap.  `order` optionally restricts / permutes the batch indices (rank
    sharding for data-parallel training)."""

    def __init__(self, generator, device, order=None, depth=2):
        import torch
        self.gen, self.device, self.depth = generator, device, max(2, int(depth))
        self.order = list(range(len(generator))) if order is None else list(order)
        self._torch = torch
        self._copy_stream = torch.cuda.Stream(device=device) if device.type == 'cuda' else None
        self._slots = [None] * self.depth

    def __len__(self):
        return len(self.order)

    def _stage(self, slot, X, y):
        torch = self._torch
        bufs = self._slots[slot]
        need = (tuple(X.shape), tuple(y.shape))
        if bufs is None or bufs['cap'][0] < X.size or bufs['cap'][1] < y.size:
            pin = self.device.type == 'cuda'
            bufs = {'cap': (X.size, y.size),
                    'hx': torch.empty(X.size, dtype=torch.float32, pin_memory=pin),
                    'hy': torch.empty(y.size, dtype=torch.float32, pin_memory=pin),
                    'dx': torch.empty(X.size, dtype=torch.float32, device=self.device),
                    'dy': torch.empty(y.size, dtype=torch.float32, device=self.device),
                    'ev': torch.cuda.Event() if pin else None, 'free': None}
            self._slots[slot] = bufs
        if bufs['ev'] is not None and bufs.get('recorded'):
            bufs['ev'].synchronize()        # the previous H2D copy out of this pinned slot must have drained
        hx = bufs['hx'][:X.size].view(need[0])
        hy = bufs['hy'][:y.size].view(need[1])
        hx.numpy()[...] = X
        hy.numpy()[...] = y
        dx = bufs['dx'][:X.size].view(need[0])
        dy = bufs['dy'][:y.size].view(need[1])
        if self._copy_stream is not None:
            if bufs['free'] is not None:
                self._copy_stream.wait_event(bufs['free'])      # the consumer is done with this slot's device buffers
            with torch.cuda.stream(self._copy_stream):
                dx.copy_(hx, non_blocking=True)
                dy.copy_(hy, non_blocking=True)
                bufs['ev'].record(self._copy_stream)
                bufs['recorded'] = True
        else:
            dx.copy_(hx)
            dy.copy_(hy)
        return dx, dy, bufs

    def __iter__(self):
        torch = self._torch
        results = {}
        lock = threading.Condition()
        n = len(self.order)

        def worker():
            try:
                for k, idx in enumerate(self.order):
                    with lock:
                        while k - state['consumed'] >= self.depth:
                            lock.wait()
                    X, y = self.gen[idx]
                    X = np.ascontiguousarray(X, dtype=np.float32)
                    y = np.ascontiguousarray(y, dtype=np.float32)
                    item = self._stage(k % self.depth, X, y)
                    with lock:
                        results[k] = item
                        lock.notify_all()
            except BaseException as e:  # noqa: BLE001
                with lock:
                    results['error'] = e
                    lock.notify_all()

        state = {'consumed': 0}
        th = threading.Thread(target=worker, daemon=True)
        th.start()
        for k in range(n):
            with lock:
                while k not in results and 'error' not in results:
                    lock.wait()
                if 'error' in results:
                    raise results['error']
                dx, dy, bufs = results.pop(k)
            if bufs['ev'] is not None:
                torch.cuda.current_stream(self.device).wait_event(bufs['ev'])
            yield dx, dy
            if bufs['ev'] is not None:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                bufs['free'] = done
            with lock:
                state['consumed'] = k + 1
                lock.notify_all()
        th.join()
