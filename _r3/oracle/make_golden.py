#!/usr/bin/env python3
"""
Golden-vector generator.  TEST INFRASTRUCTURE ONLY -- never imported by the product.

Runs ONLY in the build container, where /root/reference exists.  It executes the
reference's own source for the index/bookkeeping half of the hot path
(PeriodicPadding2D/3D.call, FillPadding2D.call, DLWPNeuralNet/DLWPFunctional
.predict_timeseries, DataGenerator, delete_nan_samples) under a numpy-backed stub of
the third-party modules the reference imports but this image lacks (keras, tensorflow,
xarray, netCDF4, dask), and writes the inputs + outputs as small .npz fixtures under
tests/golden/.  Only data travels: no reference source or bytecode is written anywhere
(sys.dont_write_bytecode is set before the import).

The Conv2D / pooling / optimiser arithmetic of the reference lives in unpinned
third-party Keras/TF and cannot be executed here: those parts are "parity unpinned"
 and are NOT covered by these fixtures.

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types
import itertools

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


# --------------------------------------------------------------------------------------------------------------- #
# numpy-backed stub of the reference's missing third-party imports
# --------------------------------------------------------------------------------------------------------------- #

def _normalize_tuple(value, n):
    if isinstance(value, int):
        return (value,) * n
    value = tuple(value)
    assert len(value) == n
    return value


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Var(np.ndarray):
        """ndarray with the .assign() the reference calls on K.zeros(...) results."""
        def assign(self, v):
            self[...] = v
            return self

    def _zeros(shape, *a, **k):
        return np.zeros(shape, dtype=np.float32).view(_Var)

    K = mod('keras.backend',
            backend=lambda: 'numpy',
            floatx=lambda: 'float32',
            concatenate=lambda xs, axis=-1: np.concatenate(xs, axis=axis),
            stack=lambda xs, axis=0: np.stack(xs, axis=axis),
            normalize_data_format=lambda v: 'channels_last' if v is None else v,
            zeros=_zeros,
            ones=lambda shape, *a, **k: np.ones(shape, dtype=np.float32),
            cos=np.cos, sin=np.sin, pow=np.power, sqrt=np.sqrt, square=np.square, abs=np.abs,
            mean=lambda x, axis=None, keepdims=False: np.mean(x, axis=tuple(axis) if isinstance(axis, list) else axis,
                                                             keepdims=keepdims),
            expand_dims=lambda x, axis=-1: np.expand_dims(x, axis),
            repeat_elements=lambda x, rep, axis: np.repeat(x, rep, axis=axis),
            variable=lambda v, name=None: np.asarray(v, dtype=np.float32),
            cast_to_floatx=lambda v: np.asarray(v, dtype=np.float32),
            cast=lambda v, dt: np.asarray(v, dtype=dt),
            eval=lambda v: v,
            int_shape=lambda x: tuple(x.shape))

    class Layer(object):
        def __init__(self, **kwargs):
            self.input_shape_arg = kwargs.get('input_shape')

    class _ZeroPaddingND(Layer):
        """Only what the reference's subclasses rely on: padding-tuple normalisation + data_format."""
        rank = 2

        def __init__(self, padding=1, data_format=None, **kwargs):
            super(_ZeroPaddingND, self).__init__(**kwargs)
            self.data_format = K.normalize_data_format(data_format)
            n = self.rank
            if isinstance(padding, int):
                self.padding = ((padding, padding),) * n
            else:
                if len(padding) != n:
                    raise ValueError('padding should have %d elements' % n)
                self.padding = tuple(_normalize_tuple(p, 2) for p in padding)

    class ZeroPadding2D(_ZeroPaddingND):
        rank = 2

    class ZeroPadding3D(_ZeroPaddingND):
        rank = 3

    class Lambda(Layer):
        def __init__(self, function, **kwargs):
            super(Lambda, self).__init__(**kwargs)
            self.function = function

        def __call__(self, x):
            return self.function(x)

    class Callback(object):
        pass

    class EarlyStopping(Callback):
        def __init__(self, **kwargs):
            self.__dict__.update(kwargs)

    def _conv2d_valid(x, kernel, strides=(1, 1), padding='valid', data_format=None, dilation_rate=(1, 1)):
        """K.conv2d as Keras documents it: cross-correlation, 'valid', kernel (kh, kw, cin, cout).  float64 direct sum
        (our restatement of the third-party op; it only carries the reference's row slicing / kernel indexing)."""
        assert padding == 'valid' and tuple(dilation_rate) == (1, 1)
        x = np.asarray(x, dtype=np.float64)
        if data_format == 'channels_last':
            x = x.transpose(0, 3, 1, 2)
        k = np.asarray(kernel, dtype=np.float64)
        kh, kw = k.shape[:2]
        sr, sc = strides
        ho, wo = (x.shape[2] - kh) // sr + 1, (x.shape[3] - kw) // sc + 1
        y = np.zeros((x.shape[0], k.shape[3], ho, wo))
        for u in range(kh):
            for v in range(kw):
                y += np.einsum('nchw,co->nohw', x[:, :, u:u + (ho - 1) * sr + 1:sr, v:v + (wo - 1) * sc + 1:sc], k[u, v])
        return y if data_format != 'channels_last' else y.transpose(0, 2, 3, 1)

    def _bias_add(x, bias, data_format=None):
        """keras.backend.bias_add (tensorflow backend, Keras 2.2) for a 4-D x: a rank-1 bias broadcasts over the channel
        axis; a rank-3 bias is RESHAPED to (1, b[2], b[0], b[1]) for channels_first, (1,) + shape for channels_last."""
        bs = tuple(np.shape(bias))
        assert x.ndim == 4 and len(bs) in (1, 3)
        if data_format == 'channels_first':
            shp = (1, bs[0], 1, 1) if len(bs) == 1 else (1, bs[2]) + bs[:2]
        else:
            shp = (1, 1, 1, bs[0]) if len(bs) == 1 else (1,) + bs
        return x + np.reshape(bias, shp)

    K.conv2d = _conv2d_valid
    K.bias_add = _bias_add

    class LocallyConnected2D(Layer):
        """keras.layers.local.LocallyConnected2D.__init__ -- argument normalisation only -- and an add_weight that hands
        out the arrays the generator prepared (`weight_source`: name -> callable(shape))."""
        weight_source = None

        def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', data_format=None, activation=None,
                     use_bias=True, kernel_initializer='glorot_uniform', bias_initializer='zeros',
                     kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                     bias_constraint=None, **kwargs):
            super(LocallyConnected2D, self).__init__(**kwargs)
            self.filters = filters
            self.kernel_size = _normalize_tuple(kernel_size, 2)
            self.strides = _normalize_tuple(strides, 2)
            self.padding = padding.lower()
            if self.padding != 'valid':
                raise ValueError('Invalid border mode for LocallyConnected2D (only "valid" is supported): ' + padding)
            self.data_format = K.normalize_data_format(data_format)
            self.activation = {None: (lambda v: v), 'linear': (lambda v: v), 'tanh': np.tanh}[activation]
            self.use_bias = use_bias
            self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
            self.kernel_regularizer = self.bias_regularizer = self.kernel_constraint = self.bias_constraint = None

        def add_weight(self, shape=None, initializer=None, name=None, regularizer=None, constraint=None):
            return type(self).weight_source[name](tuple(shape))

    class Model(object):
        pass

    class InputSpec(object):
        def __init__(self, **kwargs):
            pass

    keras = mod('keras', backend=K)
    keras.callbacks = mod('keras.callbacks', Callback=Callback, EarlyStopping=EarlyStopping)
    keras.layers = mod('keras.layers', Lambda=Lambda, Layer=Layer)
    keras.layers.convolutional = mod('keras.layers.convolutional', ZeroPadding2D=ZeroPadding2D,
                                     ZeroPadding3D=ZeroPadding3D)
    keras.layers.local = mod('keras.layers.local', LocallyConnected2D=LocallyConnected2D)
    keras.losses = mod('keras.losses',
                       mean_absolute_error=lambda t, p: np.mean(np.abs(p - t), axis=-1),
                       mean_squared_error=lambda t, p: np.mean(np.square(p - t), axis=-1))
    conv_utils = types.SimpleNamespace(
        normalize_tuple=lambda v, n, name: _normalize_tuple(v, n),
        conv_output_length=lambda n, k, padding, stride, dilation=1: (n - (k - 1) * dilation - 1 + stride) // stride)
    keras.utils = mod('keras.utils', conv_utils=conv_utils, multi_gpu_model=lambda m, gpus=1: m, Sequence=object)
    keras.engine = mod('keras.engine')
    keras.engine.base_layer = mod('keras.engine.base_layer', InputSpec=InputSpec)
    keras.models = mod('keras.models', Model=Model, Sequential=Model)
    mod('tensorflow', pad=None)
    mod('xarray', DataArray=FakeDataArray)
    mod('dask')
    mod('netCDF4', default_fillvals={'f4': 9.969209968386869e+36})


class _Coord(np.ndarray):
    """A coordinate variable: a 1-d ndarray that also answers `.values`; picking one element gives a _Scalar (the 0-d
    DataArray xarray returns), so `ds['sample'][1] - ds['sample'][0]` has `.values` and scales by integers
    (reference extensions.py:51, 257-262)."""

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    @property
    def values(self):
        return np.asarray(self)

    def __getitem__(self, i):
        r = np.ndarray.__getitem__(self, i)
        return r if isinstance(r, np.ndarray) else _Scalar(r)


class _Scalar(object):
    """0-d coordinate value (datetime64 / timedelta64): `.values`, +, -, * with numbers, arrays and coordinates."""
    __array_ufunc__ = None            # numpy defers to the reflected operators below

    def __init__(self, v):
        self.values = v.values if isinstance(v, _Scalar) else v

    @staticmethod
    def _raw(o):
        return o.values if isinstance(o, (_Scalar, _Coord)) else o

    @staticmethod
    def _wrap(v):
        return _Coord(v) if isinstance(v, np.ndarray) and v.ndim > 0 else _Scalar(v)

    def __add__(self, o):
        return self._wrap(self.values + self._raw(o))

    __radd__ = __add__

    def __sub__(self, o):
        return self._wrap(self.values - self._raw(o))

    def __rsub__(self, o):
        return self._wrap(self._raw(o) - self.values)

    def __mul__(self, o):
        return self._wrap(self.values * self._raw(o))

    __rmul__ = __mul__


def _plain(c):
    """coordinate given as _Coord / FakeDataArray / range / list -> plain ndarray"""
    if isinstance(c, FakeDataArray):
        return np.asarray(c.values)
    if isinstance(c, (_Coord, _Scalar)):
        return np.asarray(c.values)
    return np.asarray(list(c)) if isinstance(c, range) else np.asarray(c)


class FakeDataArray(object):
    """What the reference touches of an xarray DataArray.  SeriesDataGenerator (generators.py:323-629): values / shape,
    label selection on named dimensions (.sel), .isel(time_step=-1), .load(), the coordinate variables .sample / .lat /
    .lon (each with .values), and the constructor xr.DataArray(values, coords=..., dims=...).  TimeSeriesEstimator.predict
    (extensions.py:136-303) on top of that: coords given as a LIST aligned with dims, positional [...] get / set (views
    into the same memory, as numpy basic indexing gives xarray), .loc[{dim: label | labels}] get / set, .reindex(sample=
    new labels) (rows without a match become NaN), .isel(dim=slice), .assign_coords(varlev=MultiIndex) + .unstack('varlev')
    (new coordinates = the MultiIndex LEVELS, i.e. sorted, missing combinations NaN) and .transpose(*dims)."""

    def __init__(self, values, coords=None, dims=None):
        self.values = values if isinstance(values, np.ndarray) else np.asarray(values)
        self.dims = tuple(dims)
        if isinstance(coords, dict):
            self.coords = {k: (v if type(v).__name__ == 'MultiIndex' else _plain(v)) for k, v in coords.items()}
        else:
            self.coords = {d: _plain(c) for d, c in zip(self.dims, coords)}
        for d in self.dims:
            if d in self.coords and type(self.coords[d]).__name__ != 'MultiIndex':
                assert len(self.coords[d]) == self.values.shape[self.dims.index(d)], (d, self.values.shape)

    @property
    def shape(self):
        return self.values.shape

    def __getattr__(self, name):
        coords = self.__dict__.get('coords', {})
        if name in coords:
            return _Coord(coords[name])
        raise AttributeError(name)

    def load(self):
        return self

    def sel(self, **sel):
        out = self
        for dim, labels in sel.items():
            ax = out.dims.index(dim)
            have = list(out.coords[dim])
            idx = [have.index(l) for l in labels]
            coords = dict(out.coords)
            coords[dim] = np.asarray(out.coords[dim])[idx]
            out = FakeDataArray(np.take(out.values, idx, axis=ax), coords, out.dims)
        return out

    def isel(self, **isel):
        out = self
        for dim, i in isel.items():
            ax = out.dims.index(dim)
            if isinstance(i, slice):
                coords = dict(out.coords)
                coords[dim] = coords[dim][i]
                out = FakeDataArray(out.values[(slice(None),) * ax + (i,)], coords, out.dims)
            else:
                coords = {k: v for k, v in out.coords.items() if k != dim}
                out = FakeDataArray(np.take(out.values, i, axis=ax), coords, tuple(d for d in out.dims if d != dim))
        return out

    # -- positional access: views, as xarray over numpy basic indexing ------------------------------------------------- #
    def __getitem__(self, key):
        key = key if isinstance(key, tuple) else (key,)
        key = key + (slice(None),) * (len(self.dims) - len(key))
        coords, dims = {}, []
        for d, k in zip(self.dims, key):
            if isinstance(k, slice):
                dims.append(d)
                if d in self.coords:
                    coords[d] = self.coords[d][k]
            else:
                assert isinstance(k, (int, np.integer)), 'positional lists are not needed by the reference'
        return FakeDataArray(self.values[key], coords, dims)

    def __setitem__(self, key, value):
        self.values[key] = _plain(value)

    # -- label access -------------------------------------------------------------------------------------------------- #
    class _Loc(object):
        def __init__(self, da):
            self.da = da

        def _index(self, sel):
            """per dimension: int (scalar label), list of ints (label list) or slice(None)"""
            da, out = self.da, []
            for d in da.dims:
                if d not in sel:
                    out.append(slice(None))
                    continue
                lab = sel[d]
                have = list(da.coords[d])
                if isinstance(lab, (_Coord, FakeDataArray, np.ndarray, list, tuple)) and np.ndim(_plain(lab)) > 0:
                    out.append([have.index(l) for l in _plain(lab)])
                else:
                    out.append(have.index(_plain(lab)[()] if isinstance(lab, (np.ndarray, _Scalar)) else lab))
            return out

        def __getitem__(self, sel):
            da, idx = self.da, self._index(sel)
            if all(not isinstance(i, list) for i in idx):        # scalars only: a VIEW (so `.loc[..][..] = v` lands)
                coords = {d: da.coords[d] for d, i in zip(da.dims, idx) if isinstance(i, slice) and d in da.coords}
                return FakeDataArray(da.values[tuple(idx)], coords, [d for d, i in zip(da.dims, idx) if isinstance(i, slice)])
            vals, coords, dims = da.values, {}, []
            for ax in range(len(da.dims) - 1, -1, -1):            # right to left so that axes keep their numbers
                i = idx[ax]
                if not isinstance(i, slice):
                    vals = np.take(vals, i, axis=ax)
            for d, i in zip(da.dims, idx):
                if isinstance(i, int):
                    continue
                dims.append(d)
                if d in da.coords:
                    coords[d] = da.coords[d][i] if isinstance(i, list) else da.coords[d]
            return FakeDataArray(vals, coords, dims)

        def __setitem__(self, sel, value):
            da, idx = self.da, self._index(sel)
            mesh = np.ix_(*[np.arange(n) if isinstance(i, slice) else np.atleast_1d(i) for i, n in zip(idx, da.values.shape)])
            block = tuple(n if isinstance(i, slice) else len(np.atleast_1d(i)) for i, n in zip(idx, da.values.shape))
            v = _plain(value)           # positional, as xarray assigns once the coordinates are consistent
            da.values[mesh] = v.reshape(block) if v.size == int(np.prod(block)) else np.broadcast_to(v, block)

    @property
    def loc(self):
        return FakeDataArray._Loc(self)

    def reindex(self, sample=None, method=None):
        assert method is None and self.dims[0] == 'sample'
        new = _plain(sample)
        have = {k: i for i, k in enumerate(self.coords['sample'].tolist())}
        vals = np.full((len(new),) + self.values.shape[1:], np.nan, dtype=self.values.dtype)
        for r, lab in enumerate(new.tolist()):
            if lab in have:
                vals[r] = self.values[have[lab]]
        coords = dict(self.coords)
        coords['sample'] = new
        return FakeDataArray(vals, coords, self.dims)

    def assign_coords(self, **kw):
        coords = dict(self.coords)
        coords.update(kw)
        return FakeDataArray(self.values, coords, self.dims)

    def unstack(self, dim):
        mi = self.coords[dim]
        ax = self.dims.index(dim)
        l0, l1 = [np.asarray(l) for l in mi.levels]
        c0, c1 = [np.asarray(c) for c in mi.codes]
        shape = self.values.shape[:ax] + (len(l0), len(l1)) + self.values.shape[ax + 1:]
        vals = np.full(shape, np.nan, dtype=self.values.dtype)
        for j in range(self.values.shape[ax]):
            vals[(slice(None),) * ax + (c0[j], c1[j])] = np.take(self.values, j, axis=ax)
        coords = {k: v for k, v in self.coords.items() if k != dim}
        coords[mi.names[0]], coords[mi.names[1]] = l0, l1
        return FakeDataArray(vals, coords, self.dims[:ax] + (mi.names[0], mi.names[1]) + self.dims[ax + 1:])

    def transpose(self, *dims):
        perm = [self.dims.index(d) for d in dims]
        return FakeDataArray(self.values.transpose(perm), self.coords, dims)


class FakeSeriesDS(object):
    """Dataset with the single variable 'predictors' (a continuous time series) the SeriesDataGenerator expects; for
    TimeSeriesEstimator also .variables, .coords, ds['sample'] and the coordinate attributes .sample / .lat / .lon."""

    def __init__(self, da):
        self.predictors = da
        self.dims = dict(zip(da.dims, da.shape))
        self.variables = {'predictors': da}
        self.coords = {k: _Coord(v) for k, v in da.coords.items()}

    def __getitem__(self, name):
        return _Coord(self.predictors.coords[name])

    def __getattr__(self, name):
        da = self.__dict__.get('predictors')
        if da is not None and name in da.coords:
            return _Coord(da.coords[name])
        raise AttributeError(name)

    def load(self):
        return self


class FakeDS(object):
    """Duck-typed stand-in for the xarray Dataset the reference DataGenerator consumes."""

    class _Var(object):
        def __init__(self, a):
            self.values = a
            self.shape = a.shape

    def __init__(self, predictors, targets, dims):
        self._p, self._t, self._dimnames = predictors, targets, dims
        self.predictors = FakeDS._Var(predictors)
        self.targets = FakeDS._Var(targets)
        self.dims = dict(zip(dims, predictors.shape))

    def isel(self, sample=slice(None)):
        return FakeDS(self._p[sample], self._t[sample], self._dimnames)

    def close(self):
        pass


def main():
    if not os.path.isdir(REF):
        raise SystemExit('make_golden.py needs %s (build container only)' % REF)
    _install_stubs()
    sys.path.insert(0, REF)
    import DLWP.custom as rc                     # noqa: E402  (reference, under stubs)
    from DLWP.model.models import DLWPNeuralNet, DLWPFunctional
    from DLWP.model.generators import DataGenerator
    from DLWP.util import delete_nan_samples, train_test_split_ind
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20190424)

    # ----- padding -------------------------------------------------------------------------------------------- #
    pad = {}
    x_cf = rng.standard_normal((2, 4, 5, 6)).astype(np.float32)         # channels_first  [N,C,H,W]
    x_cl = np.ascontiguousarray(x_cf.transpose(0, 2, 3, 1))             # channels_last   [N,H,W,C]
    pad['x_cf'], pad['x_cl'] = x_cf, x_cl
    per_cases = [(0, 1), (0, 2), (1, 2), ((1, 2), (3, 1)), 0, 1, (2, 0), ((0, 3), (2, 0)), (5, 6)]
    pad['periodic_n'] = np.int64(len(per_cases))
    for i, p in enumerate(per_cases):
        lay = rc.PeriodicPadding2D(p, data_format='channels_first')
        pad['periodic_%d_padding' % i] = np.asarray(lay.padding, dtype=np.int64)
        pad['periodic_%d_cf' % i] = lay.call(x_cf)
        pad['periodic_%d_cl' % i] = rc.PeriodicPadding2D(p, data_format='channels_last').call(x_cl)
    fill_cases = [(2, 0), (1, 1), ((0, 2), (1, 0)), 0, (0, 3), ((3, 1), (2, 4))]
    pad['fill_n'] = np.int64(len(fill_cases))
    for i, p in enumerate(fill_cases):
        lay = rc.FillPadding2D(p, data_format='channels_first')
        pad['fill_%d_padding' % i] = np.asarray(lay.padding, dtype=np.int64)
        pad['fill_%d_cf' % i] = lay.call(x_cf)
        pad['fill_%d_cl' % i] = rc.FillPadding2D(p, data_format='channels_last').call(x_cl)
    x3 = rng.standard_normal((2, 3, 2, 5, 6)).astype(np.float32)
    pad['x3_cf'] = x3
    p3_cases = [(0, 0, 2), (1, 0, 1), ((0, 1), (2, 0), (1, 3))]
    pad['periodic3_n'] = np.int64(len(p3_cases))
    for i, p in enumerate(p3_cases):
        lay = rc.PeriodicPadding3D(p, data_format='channels_first')
        pad['periodic3_%d_padding' % i] = np.asarray(lay.padding, dtype=np.int64)
        pad['periodic3_%d_cf' % i] = lay.call(x3)
    # the composite every call site uses: Periodic((0,k)) then "ZeroPadding2D((k,0))" (zero rows written with numpy
    # since Keras' own ZeroPadding2D.call is third-party); both orders (train.py:159-163 vs train_functional.py:227)
    for k in (1, 2):
        a = rc.PeriodicPadding2D((0, k), data_format='channels_first').call(x_cf)
        a = np.pad(a, ((0, 0), (0, 0), (k, k), (0, 0)))
        b = np.pad(x_cf, ((0, 0), (0, 0), (k, k), (0, 0)))
        b = rc.PeriodicPadding2D((0, k), data_format='channels_first').call(b)
        pad['composite_pz_%d' % k] = a
        pad['composite_zp_%d' % k] = b
    np.savez_compressed(os.path.join(OUT, 'padding.npz'), **pad)

    # ----- rollout bookkeeping -------------------------------------------------------------------------------- #
    roll = {}

    def step_lin(p, **kw):
        return (0.5 * p + 1.0).astype(np.float32)

    def step_nl(p, **kw):
        return np.tanh(np.roll(p, 1, axis=-1) * 0.75 + 0.1 * p).astype(np.float32)

    n_case = 0
    for time_dim, steps, seq, keep, rec, fn in itertools.product((1, 2, 3), (1, 3, 8), (False, True), (False, True),
                                                                 (False, True), ('lin', 'nl')):
        V = 2
        if rec:
            p0 = rng.standard_normal((3, time_dim, V, 5, 6)).astype(np.float32)
        else:
            p0 = rng.standard_normal((3, time_dim * V, 5, 6)).astype(np.float32)
        m = DLWPNeuralNet(is_convolutional=True, is_recurrent=rec, time_dim=time_dim, scaler_type=None,
                          scale_targets=False)
        m.model = types.SimpleNamespace(predict=step_lin if fn == 'lin' else step_nl)
        out = m.predict_timeseries(p0, steps, step_sequence=seq, keep_time_dim=keep)
        roll['nn_%d_cfg' % n_case] = np.asarray([time_dim, steps, int(seq), int(keep), int(rec), int(fn == 'nl')],
                                                dtype=np.int64)
        roll['nn_%d_in' % n_case] = p0
        roll['nn_%d_out' % n_case] = out
        n_case += 1
    roll['nn_n'] = np.int64(n_case)

    n_case = 0
    for time_dim, steps, n_out, keep, rec in itertools.product((1, 2), (1, 3, 8), (1, 3), (False, True), (False, True)):
        V = 2
        if rec:
            p0 = rng.standard_normal((3, time_dim, V, 5, 6)).astype(np.float32)
        else:
            p0 = rng.standard_normal((3, time_dim * V, 5, 6)).astype(np.float32)
        f = DLWPFunctional(is_convolutional=True, is_recurrent=rec, time_dim=time_dim)
        f._n_steps = n_out

        def predict(p, _n=n_out, **kw):
            outs, q = [], p
            for _ in range(_n):
                q = step_nl(q)
                outs.append(q)
            return outs[0] if _n == 1 else outs
        f.model = types.SimpleNamespace(predict=predict)
        out = f.predict_timeseries(p0, steps, keep_time_dim=keep)
        roll['fn_%d_cfg' % n_case] = np.asarray([time_dim, steps, n_out, int(keep), int(rec)], dtype=np.int64)
        roll['fn_%d_in' % n_case] = p0
        roll['fn_%d_out' % n_case] = out
        n_case += 1
    roll['fn_n'] = np.int64(n_case)
    np.savez_compressed(os.path.join(OUT, 'rollout.npz'), **roll)

    # ----- data generator -------------------------------------------------------------------------------------- #
    gen = {}
    P = rng.standard_normal((10, 2, 2, 6, 8)).astype(np.float32)        # (sample, time_step, varlev, lat, lon)
    T = rng.standard_normal((10, 2, 2, 6, 8)).astype(np.float32)
    gen['P'], gen['T'] = P, T
    dims = ('sample', 'time_step', 'varlev', 'lat', 'lon')
    for rec in (False, True):
        tag = 'rec' if rec else 'conv'
        m = DLWPNeuralNet(is_convolutional=True, is_recurrent=rec, time_dim=2, scaler_type=None, scale_targets=False)
        g = DataGenerator(m, FakeDS(P, T, dims), batch_size=4, shuffle=False)
        gen['%s_len' % tag] = np.int64(len(g))
        gen['%s_shape' % tag] = np.asarray(g.shape, dtype=np.int64)
        gen['%s_n_features' % tag] = np.int64(g.n_features)
        gen['%s_dense_shape' % tag] = np.asarray(g.dense_shape, dtype=np.int64)
        gen['%s_convolution_shape' % tag] = np.asarray(g.convolution_shape, dtype=np.int64)
        gen['%s_shape_2d' % tag] = np.asarray(g.shape_2d, dtype=np.int64)
        for b in range(len(g)):
            X, y = g[b]
            gen['%s_X%d' % (tag, b)], gen['%s_y%d' % (tag, b)] = X, y
        X, y = g[-1]
        gen['%s_Xneg1' % tag] = X
        Xa, ya = g.generate([], scale_and_impute=False)
        gen['%s_Xall' % tag], gen['%s_yall' % tag] = Xa, ya
    # dense (non-convolutional) variants
    m = DLWPNeuralNet(is_convolutional=False, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    g = DataGenerator(m, FakeDS(P, T, dims), batch_size=4)
    gen['dense_X0'] = g[0][0]
    m = DLWPNeuralNet(is_convolutional=False, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    g = DataGenerator(m, FakeDS(P, T, dims), batch_size=4)
    gen['dense_rec_X0'] = g[0][0]
    # shuffle order under the legacy global RandomState (generators.py:103-106)
    for seed in (0, 7):
        np.random.seed(seed)
        m = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
        g = DataGenerator(m, FakeDS(P, T, dims), batch_size=4, shuffle=True)
        gen['shuffle_%d_epoch0' % seed] = np.asarray(g._indices, dtype=np.int64)
        gen['shuffle_%d_X0' % seed] = g[0][0]
        g.on_epoch_end()
        gen['shuffle_%d_epoch1' % seed] = np.asarray(g._indices, dtype=np.int64)
    # no time_step dimension in the dataset
    P4 = rng.standard_normal((7, 3, 6, 8)).astype(np.float32)
    m = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=1, scaler_type=None, scale_targets=False)
    g = DataGenerator(m, FakeDS(P4, P4 * 2, ('sample', 'varlev', 'lat', 'lon')), batch_size=3)
    gen['P4'] = P4
    gen['nots_shape'] = np.asarray(g.shape, dtype=np.int64)
    gen['nots_convolution_shape'] = np.asarray(g.convolution_shape, dtype=np.int64)
    gen['nots_len'] = np.int64(len(g))
    gen['nots_X2'], gen['nots_y2'] = g[2]
    # delete_nan_samples
    Pn, Tn = P.copy(), T.copy()
    Pn[3, 0, 1, 2, 2] = np.nan
    Tn[8, 1, 0, 0, 0] = np.nan
    Tn[3, 1, 1, 5, 7] = np.nan
    pn, tn = delete_nan_samples(Pn.copy(), Tn.copy())
    gen['nan_P'], gen['nan_T'], gen['nan_p_out'], gen['nan_t_out'] = Pn, Tn, pn, tn
    Pl = P.copy()
    Pl[1, 0, 0, 0, 0] = 1.e21
    Pl[5, 1, 1, 1, 1] = -3.e20
    pl, tl = delete_nan_samples(Pl.copy(), T.copy(), large_fill_value=True)
    gen['large_P'], gen['large_p_out'], gen['large_t_out'] = Pl, pl, tl
    Pt = P.copy()
    Pt[2, 0] = np.nan                      # half of sample 2's features
    Pt[6, 0, 0, 0, 0] = np.nan             # a single value
    pt, tt = delete_nan_samples(Pt.copy(), T.copy(), threshold=0.25)
    gen['thr_P'], gen['thr_p_out'], gen['thr_t_out'] = Pt, pt, tt
    # train_test_split_ind deterministic modes
    for method in ('first', 'last'):
        tr, te = train_test_split_ind(10, 3, method=method)
        gen['split_%s_train' % method] = np.asarray(tr, dtype=np.int64)
        gen['split_%s_test' % method] = np.asarray(te, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'generator.npz'), **gen)

    # ----- series generator + insolation (generators.py:323-629, util.py:300-352) --------------------------------- #
    import pandas as pd
    from DLWP.model.generators import SeriesDataGenerator
    from DLWP.util import insolation
    np.int = int                         # generators.py:531 uses the alias numpy >= 1.24 removed
    ser = {}
    n_t = 14
    dates = pd.date_range('2003-02-27 00:00', periods=n_t, freq='6H')       # crosses a month boundary
    lat = np.linspace(87.5, -87.5, 6)
    lon = np.arange(0., 360., 45.)
    S = rng.standard_normal((n_t, 2, 2, 6, 8)).astype(np.float32)            # (sample, variable, level, lat, lon)
    ser['S'], ser['lat'], ser['lon'] = S, lat, lon
    ser['dates'] = dates.values.astype('datetime64[s]').astype(np.int64)       # seconds since the epoch
    ser['insolation'] = insolation(dates.values, lat.copy(), lon.copy())
    ser['insolation_S2'] = insolation(dates.values[:3], lat.copy(), lon.copy(), S=2.)

    def series_ds():
        da = FakeDataArray(S, {'sample': dates.values, 'variable': np.array(['z', 't']), 'level': np.array([500, 850]),
                               'lat': lat.copy(), 'lon': lon.copy()}, ('sample', 'variable', 'level', 'lat', 'lon'))
        return FakeSeriesDS(da)
    cases = {
        'a': dict(rec=False, kw=dict(input_time_steps=2, output_time_steps=2, batch_size=4)),
        'b': dict(rec=True, kw=dict(input_time_steps=2, output_time_steps=2, batch_size=4)),
        'c': dict(rec=False, kw=dict(input_sel={'variable': ['z']}, output_sel={'variable': ['t'], 'level': [850]},
                                     input_time_steps=3, output_time_steps=1, interval=2, batch_size=5)),
        'd': dict(rec=False, kw=dict(input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4)),
        'e': dict(rec=True, kw=dict(input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4)),
        'f': dict(rec=False, kw=dict(input_time_steps=2, output_time_steps=2, sequence=3, batch_size=3)),
        'g': dict(rec=False, kw=dict(input_time_steps=1, output_time_steps=1, batch_size=6, shuffle=True)),
    }
    for tag, case in cases.items():
        m = DLWPNeuralNet(is_convolutional=True, is_recurrent=case['rec'], time_dim=case['kw']['input_time_steps'],
                          scaler_type=None, scale_targets=False)
        np.random.seed(7)
        g = SeriesDataGenerator(m, series_ds(), **case['kw'])
        ser['%s_len' % tag] = np.int64(len(g))
        for prop in ('shape', 'dense_shape', 'convolution_shape', 'shape_2d', 'output_shape', 'output_dense_shape',
                     'output_convolution_shape', 'output_shape_2d'):
            ser['%s_%s' % (tag, prop)] = np.asarray(getattr(g, prop), dtype=np.int64)
        ser['%s_n_features' % tag] = np.int64(g.n_features)
        ser['%s_output_n_features' % tag] = np.int64(g.output_n_features)
        ser['%s_indices' % tag] = np.asarray(g._indices, dtype=np.int64)
        for b in (0, len(g) - 1):
            X, y = g[b]
            ser['%s_X%d' % (tag, b)] = X
            if isinstance(y, list):
                for k, yy in enumerate(y):
                    ser['%s_y%d_%d' % (tag, b, k)] = yy
            else:
                ser['%s_y%d' % (tag, b)] = y
        Xa, ya = g.generate([], scale_and_impute=False)
        ser['%s_Xall' % tag] = Xa
        if not isinstance(ya, list):
            ser['%s_yall' % tag] = ya
    np.savez_compressed(os.path.join(OUT, 'series.npz'), **ser)

    # ----- TimeSeriesEstimator.predict (extensions.py:136-303) under the xarray stub ---------------------------------- #
    from DLWP.model.extensions import TimeSeriesEstimator
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import estimator_cases as EC
    est = {}

    def varlev_ds():
        labels = np.array(['%s/%d' % (v, l) for v in ('z', 't') for l in (500, 850)])
        da = FakeDataArray(S.reshape(n_t, 4, 6, 8).copy(), {'sample': dates.values, 'varlev': labels, 'lat': lat.copy(),
                                                           'lon': lon.copy()}, ('sample', 'varlev', 'lat', 'lon'))
        return FakeSeriesDS(da)

    def run_case(tag, case, make_ds):
        kw = case['gen']
        m = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=kw['input_time_steps'], scaler_type=None,
                          scale_targets=False)
        g = SeriesDataGenerator(m, make_ds(), **kw)
        c_in, c_out = int(g.convolution_shape[0]), int(g.output_convolution_shape[0])
        m.model = types.SimpleNamespace(predict=EC.mixing_model(c_in, c_out))
        with __import__('warnings').catch_warnings():
            __import__('warnings').simplefilter('ignore')
            out = TimeSeriesEstimator(m, g).predict(**case['predict'])
        est['%s_values' % tag] = np.asarray(out.values, dtype=np.float32)
        est['%s_dims' % tag] = np.array(list(out.dims))
        for d in out.dims:
            c = np.asarray(out.coords[d])
            if c.dtype.kind == 'M':
                c = c.astype('datetime64[s]').astype(np.int64)
            elif c.dtype.kind == 'm':
                c = c.astype('timedelta64[s]').astype(np.int64)
            elif c.dtype.kind == 'O':
                c = np.array([str(v) for v in c])
            est['%s_coord_%s' % (tag, d)] = c
        est['%s_channels' % tag] = np.asarray([c_in, c_out], dtype=np.int64)

    for tag, case in EC.CASES.items():
        run_case(tag, case, series_ds)
    for tag, case in EC.VARLEV_CASES.items():
        run_case(tag, case, varlev_ds)
    np.savez_compressed(os.path.join(OUT, 'estimator.npz'), **est)

    # ----- custom losses (numpy-K evaluation of the reference formulas) ----------------------------------------- #
    los = {}
    yt = rng.standard_normal((4, 4, 6, 8)).astype(np.float32)
    yp = (yt + 0.3 * rng.standard_normal((4, 4, 6, 8))).astype(np.float32)
    climo = rng.standard_normal((1, 4, 6, 8)).astype(np.float32) * 0.1
    los['y_true'], los['y_pred'], los['climo'] = yt, yp, climo
    for reg in (None, 'mse', 'mae', 'global'):
        for use_mean in (False, True):
            fn = rc.anomaly_correlation_loss(climo if use_mean else None, regularize_mean=reg, reverse=True)
            v = fn(yt, yp)
            los['acc_%s_%d' % (reg, int(use_mean))] = np.asarray(np.mean(v), dtype=np.float64)
    # mean-ratio regularisers ('global', 'spatial') divide by the mean of the truth: fields with a non-zero mean
    ytp, ypp = (yt + 3.0).astype(np.float32), (yp + 3.1).astype(np.float32)
    los['y_true_pos'], los['y_pred_pos'] = ytp, ypp
    for reg in ('global', 'spatial'):
        for use_mean in (False, True):
            fn = rc.anomaly_correlation_loss(climo + 3.0 if use_mean else None, regularize_mean=reg, reverse=True)
            los['accpos_%s_%d' % (reg, int(use_mean))] = np.asarray(np.mean(fn(ytp, ypp)), dtype=np.float64)
    lats = np.linspace(87.5, -87.5, 6).astype(np.float32)
    los['lats'] = lats
    for weighting in ('cosine', 'midlatitude'):
        fn = rc.latitude_weighted_loss(sys.modules['keras.losses'].mean_squared_error, lats, (4, 6, 8), axis=-2,
                                       weighting=weighting)
        los['latw_%s' % weighting] = np.asarray(np.mean(fn(yt, yp)), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'losses.npz'), **los)

    # ----- forecast error measures (DLWP/model/verify.py:17-102; plain numpy in the reference too) -------------------- #
    from DLWP.model import verify as rv
    ver = {}
    fc = rng.standard_normal((5, 9, 2, 6, 8)).astype(np.float32)          # (forecast step, sample, var, lat, lon)
    va = rng.standard_normal((9, 2, 6, 8)).astype(np.float32)
    va5 = rng.standard_normal((5, 9, 2, 6, 8)).astype(np.float32)
    fc[2, 3, 0, 1, 1] = np.nan                                             # nanmean semantics
    va[7, 1, 2, 2] = np.nan
    ver['forecast'], ver['valid'], ver['valid_steps'] = fc, va, va5
    for method in ('mse', 'mae', 'rmse'):
        ver['fe_series_%s' % method] = rv.forecast_error(fc, va, method=method)
        ver['fe_series_axis_%s' % method] = rv.forecast_error(fc, va, method=method, axis=(0, 2, 3))
        ver['fe_steps_%s' % method] = rv.forecast_error(fc, va5, method=method)
        ver['fe_steps_axis_%s' % method] = rv.forecast_error(fc, va5, method=method, axis=(1, 3, 4))
        ver['pe_%s' % method] = rv.persistence_error(fc[0], va, 4, method=method)
        ver['pe_axis_%s' % method] = rv.persistence_error(fc[0], va, 4, method=method, axis=0)
        ver['ce_%s' % method] = rv.climo_error(va, 3, method=method)
    np.savez_compressed(os.path.join(OUT, 'verify.npz'), **ver)

    # ----- RowConnected2D (DLWP/custom.py:695-896): the reference's build() / call() / row_conv2d run as written; K.conv2d
    #       and K.bias_add underneath are the stub's restatement of Keras (third-party, unpinned) ------------------------ #
    rrng = np.random.default_rng(20190815)
    row = {}
    row_cases = [   # (input (n, c, h, w), filters, kernel_size, strides, activation, use_bias)
        ((2, 3, 9, 10), 4, 5, (1, 1), None, True),          # the call-site form: 5x5, linear (train_functional.py:192)
        ((2, 5, 7, 12), 2, (3, 5), (1, 1), 'tanh', True),
        ((1, 2, 6, 8), 3, 3, (1, 1), 'linear', False),
        ((2, 2, 11, 13), 3, 3, (2, 2), None, True),         # equal strides: one output row per slice
        ((1, 4, 5, 9), 12, (5, 3), (1, 1), None, True),     # a single output row
    ]
    row['n'] = np.int64(len(row_cases))
    for i, (shp, filters, ks, st, act, use_bias) in enumerate(row_cases):
        made = {}

        def src(name):
            def f(shape):
                made[name] = rrng.standard_normal(shape).astype(np.float32) * (0.2 if name == 'kernel' else 1.0)
                return made[name]
            return f
        rc.RowConnected2D.weight_source = {'kernel': src('kernel'), 'bias': src('bias')}
        lay = rc.RowConnected2D(filters, ks, strides=st, padding='valid', activation=act, use_bias=use_bias,
                                data_format='channels_first')
        lay.build((None,) + shp[1:])
        xr = rrng.standard_normal(shp).astype(np.float32)
        yr = lay.call(xr)
        row['%d_x' % i], row['%d_kernel' % i], row['%d_y' % i] = xr, made['kernel'], np.asarray(yr, dtype=np.float64)
        if use_bias:
            row['%d_bias' % i] = made['bias']
        row['%d_kernel_shape' % i] = np.asarray(lay.kernel_shape, dtype=np.int64)
        row['%d_out_rc' % i] = np.asarray([lay.output_row, lay.output_col], dtype=np.int64)
        row['%d_strides' % i] = np.asarray(st, dtype=np.int64)
        row['%d_act' % i] = np.str_(act or 'linear')
        # the same through the functional form with channels_last data (row_conv2d's other branch, :885, :893)
        row['%d_y_cl' % i] = np.asarray(rc.row_conv2d(xr.transpose(0, 2, 3, 1), made['kernel'], lay.kernel_size, lay.strides,
                                                      (lay.output_row, lay.output_col), 'channels_last'), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'row_connected.npz'), **row)

    for f in sorted(os.listdir(OUT)):
        print('%-16s %8d bytes' % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == '__main__':
    main()
