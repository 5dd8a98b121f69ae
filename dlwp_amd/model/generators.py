"""
Batch feed for fit_generator.

  DataGenerator   the reference's keras.utils.Sequence (DLWP/model/generators.py:19-159): same constructor, shape
                  properties, shuffle order, NaN-sample removal and (X, y) batches of shape (n,)+convolution_shape.
  ArrayDataset    an in-memory stand-in for the xarray Dataset the reference reads (dims, predictors, targets, isel):
                  xarray / netCDF4 are absent from this image and stay out of scope.
  SeriesDataGenerator  the generator examples/train.py:120 and validate.py:191 construct (DLWP/model/generators.py:
                  323-629): one continuous time series, variable / level selection, input / output time windows,
                  `interval`, multi-target `sequence`, optional insolation channel.
  SeriesDataset / LabeledArray   in-memory stand-ins for the xarray objects it reads (label selection, coordinates).
  DeviceLoader    what replaces Keras' worker *processes* + per-batch feed_dict copy: a background thread gathers batch
                  i+1 into a pinned host buffer and a copy stream moves it to HBM while batch i trains
                  (pinned-host -> HBM double buffering).
"""
import ctypes
import os
import threading

import numpy as np

from ..util import delete_nan_samples, insolation


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


def _has_nan(a, chunk=1 << 24):
    """np.isnan(a).any() without a full-size boolean temporary: 16 M elements at a time, stops at the first hit"""
    flat = a.reshape(-1)
    for lo in range(0, flat.size, chunk):
        if np.isnan(flat[lo:lo + chunk]).any():
            return True
    return False


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

    def batch_sources(self):
        """DeviceLoader's fast path: [(array, per-sample shape)] for predictors and targets when generate(samples) is nothing but
        a row gather -- float32 C-contiguous arrays without NaN samples to drop, no imputer or scaler in the model -- so that the
        rows can be copied straight into pinned memory by the library's host threads (dlwp_host_gather_rows); None otherwise.
        Same values as generate(): every step of it is then a copy or a reshape.  The decision is cached against what it was
        made FROM -- the two arrays' identity and address, the model's scaler / imputer switches -- and re-taken when any of
        them changes; a subclass that overrides generate() or __getitem__ (augmentation) never takes the fast path.  The NaN
        scan happens once per array; writing NaNs into a set afterwards is not seen (as little as by a cached dataset)."""
        if type(self).generate is not DataGenerator.generate or type(self).__getitem__ is not DataGenerator.__getitem__:
            return None
        p, t = getattr(self.ds.predictors, 'values', None), getattr(self.ds.targets, 'values', None)
        if not (isinstance(p, np.ndarray) and isinstance(t, np.ndarray)):
            return None
        key = (p.ctypes.data, p.shape, t.ctypes.data, t.shape, bool(self._impute_missing),
               getattr(self.model, 'scaler_type', None), bool(self._remove_nan))
        cached = self.__dict__.get('_fast_sources')
        if cached is not None and (cached[0] == key or cached[0] == 'off'):
            if cached[0] == key:
                self._fast_misses = 0          #  a key that repeats: only CONSECUTIVE changes count against the fast path
            return cached[1]
        # a dataset whose `.values` hands out a fresh array every time (a lazy file-backed variable) changes the key on every call:
        # the decision -- a full NaN scan -- must not be re-taken per batch.  Three different keys: generate() it is.
        misses = self.__dict__.get('_fast_misses', 0) + (1 if cached is not None else 0)
        self._fast_misses = misses
        if misses >= 3:
            self._fast_sources = ('off', None)
            return None
        fast = None
        ok = (p.dtype == np.float32 and t.dtype == np.float32 and
              p.flags['C_CONTIGUOUS'] and t.flags['C_CONTIGUOUS'] and p.shape[0] == t.shape[0] and p.shape[0] > 0 and
              not self._impute_missing and getattr(self.model, 'scaler_type', None) is None)
        if ok and self._remove_nan:           # (one pass over the arrays, once: a set without NaNs drops nothing)
            ok = not (_has_nan(p) or _has_nan(t))
        if ok:
            if self._is_convolutional:
                shp = tuple(self.convolution_shape)
            elif self._keep_time_axis:
                shp = tuple(self.dense_shape)
            else:
                shp = (self.n_features,)
            if int(np.prod(shp)) == p[0].size == t[0].size:
                fast = [(p.reshape(p.shape[0], -1), shp), (t.reshape(t.shape[0], -1), shp)]
        self._fast_sources = (key, fast)
        return fast

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


class LabeledArray(object):
    """The part of xarray.DataArray the series generator uses: `values`, `shape`, named dimensions with coordinate
    labels, `.sel(dim=[labels])`, `.isel(dim=index)`, `.load()`, and coordinates as attributes (`.sample.values` ...)."""

    class _Coord(object):
        def __init__(self, values):
            self.values = values

    def __init__(self, values, coords, dims):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        if len(self.dims) != self.values.ndim:
            raise ValueError('dims %r do not match an array of rank %d' % (self.dims, self.values.ndim))
        self.coords = {k: np.asarray(v) for k, v in (coords or {}).items()}
        for d, n in zip(self.dims, self.values.shape):
            if d in self.coords and len(self.coords[d]) != n:
                raise ValueError('coordinate %r has %d labels for an axis of length %d' % (d, len(self.coords[d]), n))

    @property
    def shape(self):
        return self.values.shape

    def __getattr__(self, name):
        coords = self.__dict__.get('coords', {})
        if name in coords:
            return LabeledArray._Coord(coords[name])
        raise AttributeError(name)

    def load(self):
        return self

    def sel(self, **selection):
        out = self
        for dim, labels in selection.items():
            if dim not in out.dims:
                raise KeyError('no dimension %r (have %r)' % (dim, out.dims))
            scalar = np.ndim(labels) == 0
            have = out.coords[dim].tolist()
            try:
                idx = [have.index(l) for l in ([labels] if scalar else list(labels))]
            except ValueError:
                raise KeyError('label(s) %r not found on dimension %r' % (labels, dim))
            ax = out.dims.index(dim)
            vals = np.take(out.values, idx[0] if scalar else idx, axis=ax)
            coords = dict(out.coords)
            if scalar:
                coords.pop(dim)
                out = LabeledArray(vals, coords, tuple(d for d in out.dims if d != dim))
            else:
                coords[dim] = out.coords[dim][idx]
                out = LabeledArray(vals, coords, out.dims)
        return out

    def isel(self, **indexers):
        out = self
        for dim, i in indexers.items():
            ax = out.dims.index(dim)
            vals = np.take(out.values, i, axis=ax) if np.ndim(i) == 0 else out.values[(slice(None),) * ax + (i,)]
            coords = dict(out.coords)
            if np.ndim(i) == 0 and not isinstance(i, slice):
                coords.pop(dim, None)
                out = LabeledArray(vals, coords, tuple(d for d in out.dims if d != dim))
            else:
                if dim in coords:
                    coords[dim] = coords[dim][i]
                out = LabeledArray(vals, coords, out.dims)
        return out


class SeriesDataset(object):
    """A predictor file holding ONE continuous time series: `predictors` (sample, [time_step,] selection dims..., lat,
    lon) with coordinates sample (timestamps), lat, lon and the labels of the selection dimensions."""

    def __init__(self, predictors, coords, dims):
        self.predictors = predictors if isinstance(predictors, LabeledArray) else LabeledArray(predictors, coords, dims)
        self.dims = dict(zip(self.predictors.dims, self.predictors.shape))

    def load(self):
        return self

    def close(self):
        pass


class SeriesDataGenerator(object):
    """Batches from a continuous series: inputs = `input_time_steps` consecutive states of the `input_sel` variables
    (+ insolation), targets = `output_time_steps` states of the `output_sel` variables starting `interval` steps after
    the last input step; `sequence` = K gives a list of K consecutive target blocks (multi-output functional models).
    Same constructor, properties and batch layout as the reference (DLWP/model/generators.py:323-629)."""

    #: how much of the dataset is read into memory up front: the whole file / the selected variables / only the series
    LOAD_MODES = ('full', 'required', 'minimal')

    def __init__(self, model, ds, input_sel=None, output_sel=None, input_time_steps=1, output_time_steps=1,
                 sequence=None, interval=1, add_insolation=False, batch_size=32, shuffle=False, remove_nan=True,
                 load='required'):
        if not hasattr(ds, 'predictors'):
            raise ValueError("dataset must have 'predictors' variable")
        for value in (input_time_steps, output_time_steps, batch_size, interval) + (() if sequence is None else (sequence,)):
            assert int(value) > 0
        load = self._load_mode(load)
        self.model, self.ds = model, ds
        self._set_model_flags(model)
        self._batch_size, self._shuffle, self._remove_nan = batch_size, shuffle, remove_nan
        self._set_window(ds.dims['sample'], input_time_steps, output_time_steps, interval, sequence)
        self._select(ds, input_sel, output_sel, load)
        self._indices = []
        self.on_epoch_end()
        self._add_insolation = int(add_insolation)
        if add_insolation:
            self.insolation_da = self._insolation_series(self.da)

    @classmethod
    def _load_mode(cls, load):
        """True / False are accepted for the reference's older boolean argument (both mean 'required')."""
        if not load or load in cls.LOAD_MODES:
            return load
        if isinstance(load, bool):
            return 'required'
        raise ValueError("'load' must be one of 'full', 'required', or 'minimal'")

    def _set_model_flags(self, model):
        self._is_convolutional = model.is_convolutional
        self._keep_time_axis = model.is_recurrent
        self._impute_missing = model.impute

    def _set_window(self, n_series, t_in, t_out, interval, sequence):
        """A sample = t_in input steps, a gap of interval - 1 steps, then t_out target steps, `sequence` times over: the
        number of start positions that fit the series (reference generators.py:389)."""
        self._input_time_steps, self._output_time_steps = t_in, t_out
        self._interval, self._sequence = interval, sequence
        span = t_in + (interval - 1) + t_out * (sequence or 1)
        self._n_sample = n_series - span + 1

    def _select(self, ds, input_sel, output_sel, load):
        """The series itself and its input / output variable selections, loaded as far as `load` asks.  A file written with
        a 'time_step' dimension carries the initialisation time at time_step = -1."""
        if load == 'full':
            ds.load()
        self.da = ds.predictors.isel(time_step=-1) if 'time_step' in ds.dims else ds.predictors
        if load == 'minimal':
            self.da.load()
        self._input_sel, self._output_sel = input_sel or {}, output_sel or {}
        self.input_da, self.output_da = self.da.sel(**self._input_sel), self.da.sel(**self._output_sel)
        if load == 'required':
            self.input_da.load()
            self.output_da.load()

    @staticmethod
    def _insolation_series(da):
        times, lat, lon = da.sample.values, da.lat.values, da.lon.values
        return LabeledArray(insolation(times, lat, lon), {'sample': times, 'lat': lat, 'lon': lon}, ('sample', 'lat', 'lon'))

    # -- shapes ------------------------------------------------------------------------------------------------------ #
    @property
    def batch_size(self):
        return self._batch_size

    @property
    def shape(self):
        """(time_step, [variable, level,] lat, lon) of the inputs; excludes insolation"""
        return (self._input_time_steps,) + tuple(self.input_da.shape[1:])

    @property
    def n_features(self):
        return (int(np.prod(self.shape)) +
                int(np.prod(self.shape[-2:])) * self._input_time_steps * self._add_insolation)

    @property
    def dense_shape(self):
        if self._keep_time_axis:
            return (self.shape[0], self.n_features // self.shape[0])
        return (self.n_features,)

    def _conv_shape(self, keep_time_axis, insolation_channels):
        s = self.shape
        if keep_time_axis:
            return (self._input_time_steps, int(np.prod(s[1:-2])) + insolation_channels) + tuple(s[-2:])
        return (int(np.prod(s[:-2])) + self._input_time_steps * insolation_channels,) + tuple(self.input_da.shape[-2:])

    @property
    def convolution_shape(self):
        """(channels, y, x), or (time_step, channels, y, x) for a recurrent model; includes the insolation channel(s)"""
        return self._conv_shape(self._keep_time_axis, self._add_insolation)

    @property
    def shape_2d(self):
        return self._conv_shape(False, self._add_insolation)

    @property
    def output_shape(self):
        return (self._output_time_steps,) + tuple(self.output_da.shape[1:])

    @property
    def output_n_features(self):
        return int(np.prod(self.output_shape))

    @property
    def output_dense_shape(self):
        if self._keep_time_axis:
            return (self.output_shape[0], self.output_n_features // self.output_shape[0])
        return (self.output_n_features,)

    def _out_conv_shape(self, keep_time_axis):
        s = self.output_shape
        if keep_time_axis:
            return (self._output_time_steps, int(np.prod(s[1:-2]))) + tuple(s[-2:])
        return (int(np.prod(s[:-2])),) + tuple(self.output_da.shape[-2:])

    @property
    def output_convolution_shape(self):
        return self._out_conv_shape(self._keep_time_axis)

    @property
    def output_shape_2d(self):
        return self._out_conv_shape(False)

    # -- batches ----------------------------------------------------------------------------------------------------- #
    def on_epoch_end(self):
        self._indices = np.arange(self._n_sample)
        if self._shuffle:
            np.random.shuffle(self._indices)

    def _window(self, da, samples, first, steps):
        """(n, steps, ...) block: for every sample index the `steps` consecutive series states starting at +first."""
        v = da.values
        return np.stack([v[samples + first + n] for n in range(steps)], axis=1)

    def _finish(self, p, t, scale_and_impute):
        if self._remove_nan:
            p, t = delete_nan_samples(p, t)
        return self._finish_no_nan(p, t, scale_and_impute)

    def _finish_no_nan(self, p, t, scale_and_impute):
        if scale_and_impute:
            if self._impute_missing:
                p, t = self.model.imputer_transform(p, t)
            p, t = self.model.scaler_transform(p, t)
        n = p.shape[0]            # the surviving count (the reference keeps the pre-deletion one and fails to reshape)
        if self._is_convolutional:
            p = p.reshape((n,) + self.convolution_shape)
            t = t.reshape((n,) + self.output_convolution_shape)
        elif self._keep_time_axis:
            p = p.reshape((n,) + self.dense_shape)
            t = t.reshape((n,) + self.output_dense_shape)
        return p, t

    def generate(self, samples, scale_and_impute=True):
        """(predictors, targets) for the given series indices; an empty list means every sample.  With `sequence` the
        targets are a list of arrays (one per forecast block)."""
        samples = np.arange(self._n_sample, dtype=int) if len(samples) == 0 else np.array(samples, dtype=int)
        n_sample = len(samples)
        p = self._window(self.input_da, samples, 0, self._input_time_steps)
        if self._add_insolation:
            # insolation rides as one extra channel per input time step, behind the variables of that step
            s = self._conv_shape(True, 0)
            sol = self._window(self.insolation_da, samples, 0, self._input_time_steps)
            p = np.concatenate([p.reshape((n_sample,) + s), sol[:, :, np.newaxis]], axis=2)
        p = p.reshape((n_sample, -1))
        first = self._input_time_steps + self._interval - 1
        if self._sequence is None:
            t = self._window(self.output_da, samples, first, self._output_time_steps).reshape((n_sample, -1))
            return self._finish(p, t, scale_and_impute)
        blocks = [self._window(self.output_da, samples, first + self._output_time_steps * k,
                               self._output_time_steps).reshape((n_sample, -1)) for k in range(self._sequence)]
        if self._remove_nan:
            # one joint decision per sample (the reference filters inside the loop and loses the row alignment between
            # the predictors and the later target blocks as soon as a sample is actually dropped)
            bad = np.isnan(p).any(axis=1)
            for t in blocks:
                bad |= np.isnan(t).any(axis=1)
            if bad.any():
                keep = np.flatnonzero(~bad)
                p, blocks = p[keep], [t[keep] for t in blocks]
        targets = []
        p_out = None
        for t in blocks:
            pk, t = self._finish_no_nan(p, t, scale_and_impute)
            p_out = pk if p_out is None else p_out
            targets.append(t)
        p = p_out
        return p, targets

    def generate_inputs(self):
        """(predictors of EVERY sample unscaled, shape of the targets) -- what TimeSeriesEstimator.predict takes from
        generate([], scale_and_impute=False) (DLWP/model/extensions.py:171-172) without assembling the targets it never reads.
        Only when no sample can be dropped for NaNs (checked once per dataset array); otherwise None: call generate()."""
        if self._sequence is not None:
            return None
        if self._remove_nan:
            arrays = [self.input_da.values, self.output_da.values] + ([self.insolation_da.values] if self._add_insolation else [])
            key = tuple(id(a) for a in arrays)
            if self.__dict__.get('_nan_free_key') != key:
                self._nan_free, self._nan_free_key = not any(np.isnan(a).any() for a in arrays), key
            if not self._nan_free:
                return None
        n = self._n_sample
        samples = np.arange(n, dtype=int)
        v = self.input_da.values
        t_in = self._input_time_steps
        if self._add_insolation:
            s = self._conv_shape(True, 0)                      # (t_in, channels, y, x) without the insolation
            p = np.empty((n, t_in, s[1] + 1) + tuple(s[2:]), dtype=v.dtype)
            sol = self.insolation_da.values
            for m in range(t_in):
                p[:, m, :s[1]] = v[m:m + n].reshape((n, s[1]) + tuple(s[2:]))
                p[:, m, s[1]] = sol[m:m + n]
        else:
            p = np.stack([v[samples + m] for m in range(t_in)], axis=1)
        p = p.reshape((n,) + self.convolution_shape) if self._is_convolutional else \
            p.reshape((n,) + (self.dense_shape if self._keep_time_axis else (-1,)))
        t_shape = (n,) + (self.output_convolution_shape if self._is_convolutional else
                          (self.output_dense_shape if self._keep_time_axis else (self.output_n_features,)))
        return p, t_shape

    def __len__(self):
        return int(np.ceil(self._n_sample / self._batch_size))

    def __getitem__(self, index):
        if int(index) < 0:
            index = len(self) + index
        if index >= len(self) or index < 0:        # the reference's `index > len(self)` lets index == len through
            raise IndexError('batch index out of range')
        return self.generate(self._indices[index * self._batch_size:(index + 1) * self._batch_size])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class DeviceLoader(object):
    """Pinned-host -> HBM double-buffered feed over any Sequence-like generator (`__len__`, `__getitem__`).

        for X, y in DeviceLoader(gen, device):   # X, y are device tensors, valid until the next iteration
            ...

    Division of labour (r4): a worker thread assembles batch k + 1, k + 2 ... in PINNED host buffers -- host work only, no
    device call ever leaves that thread -- while the consumer thread, the only one that talks to the device, issues the H2D copy
    of batch k + 1 on a private copy stream before it hands out batch k (its own stream waits on the copy's event): transfer,
    host gather and the training step overlap, and nothing races a stream capture or a graph launch in the consumer.  (r3's
    loader copied from its worker thread; the captured training step therefore never ran next to it.)
    Generators with the DataGenerator protocol whose batches are plain row gathers of float32 arrays (`batch_sources`) are
    gathered straight into the pinned buffers by the library's host threads (dlwp_host_gather_rows: the reference's
    use_multiprocessing=True workers, examples/train.py:262-263); everything else goes through `gen[i]` / `generate` and one
    copy.  y may be one array or a list of arrays (SeriesDataGenerator with `sequence`, the multi-output DLWPFunctional
    models): every target gets its own staging buffers and comes out as a list of device tensors.

    `order` optionally restricts / permutes the batch indices.  `shard=(rank, world)` is the data-parallel feed: batch i is
    still the GLOBAL batch i of the generator (same index list on every rank -- the trainer broadcasts rank 0's), but this
    rank gathers and uploads ONLY its contiguous row shard of it (parallel.shard_bounds, the cut
    keras.utils.multi_gpu_model makes inside one process, reference models.py:104-109).  Generators with the
    DataGenerator protocol (`_indices`, `_batch_size`, `generate(samples)`) gather just those samples; any other
    Sequence is asked for the whole batch and sliced on the host before the upload.  iter_batches() yields
    (X, y, n_global) so the training step can weight ragged shards exactly."""

    #: host threads of a native row gather (dlwp_host_gather_rows)
    gather_threads = max(1, min(8, (os.cpu_count() or 2) // 2))

    def __init__(self, generator, device, order=None, depth=3, shard=None, avoid_streams=None):
        import torch
        self.gen, self.device, self.depth = generator, device, max(2, int(depth))
        self.order = list(range(len(generator))) if order is None else list(order)
        self.shard = None if shard is None or int(shard[1]) <= 1 else (int(shard[0]), int(shard[1]))
        self._torch = torch
        # avoid_streams: streams the copy stream must not share a hardware queue with (util.distinct_streams): an upload queued
        # behind a 0.3 ms weight gradient of the step arrives a step late (r4: 2.03 instead of 1.48 ms per step at 64 samples)
        if device.type != 'cuda':
            self._copy_stream = None
        elif avoid_streams:
            from ..util import distinct_streams
            self._copy_stream = distinct_streams(device, 1, list(avoid_streams))[0]
        else:
            self._copy_stream = torch.cuda.Stream(device=device)
        self._slots = [None] * self.depth

    def __len__(self):
        return len(self.order)

    # -- host side: this rank's rows of global batch `idx` -------------------------------------------------------------- #
    def _samples(self, idx):
        """(sample indices of this rank's rows, n_global) for a generator with the DataGenerator protocol, else None"""
        from ..parallel import shard_bounds
        gen = self.gen
        if not all(hasattr(gen, a) for a in ('_indices', '_batch_size', 'generate')):
            return None
        bs = int(gen._batch_size)
        if int(idx) < 0:
            idx = len(gen) + int(idx)
        samples = gen._indices[idx * bs:(idx + 1) * bs]
        n_global = len(samples)
        if self.shard is not None:
            lo, hi = shard_bounds(n_global, *self.shard)
            samples = samples[lo:hi]
        return samples, n_global

    def _fetch(self, idx):
        """(X, [targets], was_list, n_global) as float32 numpy arrays"""
        from ..parallel import shard_bounds
        gen = self.gen
        if self.shard is None:
            X, y = gen[idx]
            n_global = int(np.asarray(X).shape[0])
        elif self._samples(idx) is not None:
            samples, n_global = self._samples(idx)
            if len(samples) > 0:
                X, y = gen.generate(samples)
            else:                               # no rows for this rank: right trailing shape, zero rows
                X, y = gen.generate(gen._indices[:1])      # (generate([]) would mean "every sample")
                X = X[:0]
                y = [t[:0] for t in y] if isinstance(y, (list, tuple)) else y[:0]
        else:
            X, y = gen[idx]
            n_global = int(np.asarray(X).shape[0])
            lo, hi = shard_bounds(n_global, *self.shard)
            X = X[lo:hi]
            y = [t[lo:hi] for t in y] if isinstance(y, (list, tuple)) else y[lo:hi]
        was_list = isinstance(y, (list, tuple))
        ys = [np.ascontiguousarray(t, dtype=np.float32) for t in (y if was_list else [y])]
        return np.ascontiguousarray(X, dtype=np.float32), ys, was_list, n_global

    def _native_plan(self, idx):
        """[(source 2-D float32 array, per-sample shape)] + rows + n_global when batch idx is a plain row gather the library's host
        threads can do (the generator's `batch_sources`), else None"""
        if not hasattr(self.gen, 'batch_sources'):
            return None
        sm = self._samples(idx)
        if sm is None or len(sm[0]) == 0:
            return None
        src = self.gen.batch_sources()
        if src is None:
            return None
        return src, np.ascontiguousarray(sm[0], dtype=np.int64), sm[1]

    # -- staging ---------------------------------------------------------------------------------------------------------- #
    def _alloc_slot(self, sizes):
        """(consumer thread) pinned host + device buffers for arrays of `sizes` elements"""
        torch = self._torch
        pin = self.device.type == 'cuda'
        return {'cap': list(sizes),
                'host': [torch.empty(max(1, s), dtype=torch.float32, pin_memory=pin) for s in sizes],
                'dev': [torch.empty(max(1, s), dtype=torch.float32, device=self.device) for s in sizes],
                'ev': torch.cuda.Event() if pin else None, 'done': None}

    def _fill(self, slot, idx):
        """(worker thread: host work only) batch idx into the pinned buffers of `slot`; returns (shapes, was_list, n_global), or
        the fetched arrays themselves when they do not fit the slot (the consumer re-allocates and copies)"""
        from .. import _lib
        plan = self._native_plan(idx)
        if plan is not None:
            srcs, rows, n_global = plan
            shapes = [(len(rows),) + tuple(shp) for _, shp in srcs]
            sizes = [int(np.prod(sh)) for sh in shapes]
            if slot is not None and len(slot['cap']) == len(sizes) and all(c >= z for c, z in zip(slot['cap'], sizes)):
                for (arr, shp), hbuf in zip(srcs, slot['host']):
                    row_bytes = int(np.prod(shp)) * 4
                    _lib.check(_lib.lib.dlwp_host_gather_rows(ctypes.c_void_p(hbuf.data_ptr()), ctypes.c_void_p(arr.ctypes.data),
                                                              rows.ctypes.data_as(ctypes.c_void_p), len(rows), row_bytes,
                                                              arr.shape[0], int(self.gather_threads)))
                return {'shapes': shapes, 'was_list': False, 'n_global': n_global, 'arrays': None}
        X, ys, was_list, n_global = self._fetch(idx)
        arrays = [X] + ys
        shapes = [a.shape for a in arrays]
        if slot is not None and len(slot['cap']) == len(arrays) and all(c >= a.size for c, a in zip(slot['cap'], arrays)):
            for a, hbuf in zip(arrays, slot['host']):
                hbuf[:a.size].view(a.shape).numpy()[...] = a
            arrays = None
        return {'shapes': shapes, 'was_list': was_list, 'n_global': n_global, 'arrays': arrays}

    def iter_batches(self, order=None):
        """yields (X, y, n_global): device tensors of this rank's rows (y a list when the generator gives a list) and
        the size of the global batch they belong to.  order: this pass's batch indices (default: the loader's); the staging
        buffers stay with the loader, so one loader serves every epoch of a fit_generator call."""
        torch = self._torch
        if order is not None:
            self.order = list(order)
        n = len(self.order)
        lock = threading.Condition()
        state = {'filled': {}, 'free': set(range(self.depth)), 'error': None, 'stop': False}

        def worker():
            try:
                for k, idx in enumerate(self.order):
                    s = k % self.depth
                    with lock:
                        while s not in state['free'] and not state['stop']:
                            lock.wait()
                        if state['stop']:
                            return
                        state['free'].discard(s)
                    res = self._fill(self._slots[s], idx)
                    with lock:
                        state['filled'][k] = res
                        lock.notify_all()
            except BaseException as e:  # noqa: BLE001
                with lock:
                    state['error'] = e
                    lock.notify_all()

        if hasattr(self.gen, 'batch_sources') and self.gen.batch_sources() is not None:
            # plain row gathers: the batch shapes are known without fetching one -- staging buffers for whole batches up front, so
            # that the worker gathers into pinned memory from the first batch on
            bs = int(getattr(self.gen, '_batch_size', 0))
            sizes = [bs * int(np.prod(shp)) for _, shp in self.gen.batch_sources()]
            for i in range(self.depth):
                sl = self._slots[i]
                if bs > 0 and (sl is None or len(sl['cap']) != len(sizes) or
                               any(c < z for c, z in zip(sl['cap'], sizes))):
                    self._slots[i] = self._alloc_slot(sizes)
        th = threading.Thread(target=worker, daemon=True)
        th.start()
        cuda = self._copy_stream is not None

        def issue(k):
            """(consumer thread) wait for the host copy of batch k, start its upload; returns the device views"""
            s = k % self.depth
            with lock:
                while k not in state['filled'] and state['error'] is None:
                    lock.wait()
                if state['error'] is not None:
                    raise state['error']
                res = state['filled'].pop(k)
            slot = self._slots[s]
            if res['arrays'] is not None:            # first use of the slot, or a batch larger than it: (re)allocate, copy here
                sizes = [a.size for a in res['arrays']]
                if slot is None or len(slot['cap']) != len(sizes) or any(c < z for c, z in zip(slot['cap'], sizes)):
                    if slot is not None and slot['ev'] is not None and slot.get('recorded'):
                        slot['ev'].synchronize()
                    slot = self._slots[s] = self._alloc_slot(sizes)
                for a, hbuf in zip(res['arrays'], slot['host']):
                    hbuf[:a.size].view(a.shape).numpy()[...] = a
            hs = [hbuf[:int(np.prod(sh))].view(tuple(sh)) for sh, hbuf in zip(res['shapes'], slot['host'])]
            ds = [dbuf[:int(np.prod(sh))].view(tuple(sh)) for sh, dbuf in zip(res['shapes'], slot['dev'])]
            if cuda:
                if slot['done'] is not None:
                    self._copy_stream.wait_event(slot['done'])      # the consumer is through with this slot's device buffers
                with torch.cuda.stream(self._copy_stream):
                    for d, h in zip(ds, hs):
                        d.copy_(h, non_blocking=True)
                    slot['ev'].record(self._copy_stream)
                    slot['recorded'] = True
            else:
                for d, h in zip(ds, hs):
                    d.copy_(h)
            return ds, slot, res['was_list'], res['n_global']

        def release(k):
            """(consumer thread) batch k's pinned buffers may be refilled once its upload has drained"""
            slot = self._slots[k % self.depth]
            if slot is not None and slot['ev'] is not None and slot.get('recorded'):
                slot['ev'].synchronize()
            with lock:
                state['free'].add(k % self.depth)
                lock.notify_all()

        try:
            nxt = issue(0) if n else None
            for k in range(n):
                cur = nxt
                nxt = issue(k + 1) if k + 1 < n else None        # batch k + 1 uploads under batch k's step
                ds, slot, was_list, n_global = cur
                if slot['ev'] is not None:
                    torch.cuda.current_stream(self.device).wait_event(slot['ev'])
                yield ds[0], (ds[1:] if was_list else ds[1]), n_global
                if slot['ev'] is not None:
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                    slot['done'] = done
                release(k)
        finally:
            with lock:
                state['stop'] = True
                lock.notify_all()
            th.join()
            # a consumer that leaves early (break, exception) has the upload of batch k + 1 in flight: the next pass's worker must
            # not refill that pinned slot under the copy engine
            for sl in self._slots:
                if sl is not None and sl['ev'] is not None and sl.get('recorded'):
                    sl['ev'].synchronize()

    def __iter__(self):
        for X, y, _ in self.iter_batches():
            yield X, y
