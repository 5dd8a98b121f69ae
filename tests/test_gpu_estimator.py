"""The rollout examples/validate.py:191-205 runs -- TimeSeriesEstimator.predict with inputs != outputs (insolation, variable
selections, interval, fewer / more output steps, impute; DLWP/model/extensions.py:206-240) -- and step_sequence rollouts
(DLWP/model/models.py:280-290) ON THE DEVICE: one hipGraph, a feedback launch between the model calls (csrc/feedback.hip).
Bit-exact on the index work against the oracle's restatement; values against the reference's own predict() (tests/golden/
estimator.npz) within the float32 forward tolerance; the device loop against the reference-form host loop bit for bit."""
import warnings

import numpy as np
import pytest
import torch

from oracle import np_ref
from tests.nets import unet_layers
from tests.test_estimator import _estimator_cases, _golden_dataset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)


@pytest.mark.parametrize('n,t_in,c_in,t_out,c_out,hw,k,es,mode', [
    (9, 2, 3, 2, 2, (6, 8), 2, 2, 'keep'),          # validate.py's form: 2 x (2 variables + insolation) -> 2 x 2 variables
    (9, 2, 3, 1, 2, (6, 8), 1, 1, 'keep'),          # fewer output steps
    (7, 1, 4, 2, 3, (5, 7), 1, 1, 'first'),         # more output steps; a plane of 35 elements: the scalar path
    (7, 1, 4, 2, 3, (5, 7), 2, 2, 'last'),
    (5, 2, 2, 2, 2, (4, 4), 3, 2, 'keep'),          # interval 2: k > es -- rows between stay NaN under impute
    (3, 2, 2, 2, 2, (4, 4), 5, 2, 'keep'),          # k > n: every row runs out
    (1, 2, 3, 2, 2, (6, 8), 2, 2, 'keep'),          # es > n: the tail is the whole state
    (64, 2, 3, 2, 2, (88, 180), 2, 2, 'keep'),      # the headline grid
])
@pytest.mark.parametrize('impute,with_sol', [(False, False), (True, False), (False, True), (True, True)])
def test_state_feedback_is_the_reference_bookkeeping_bit_for_bit(n, t_in, c_in, t_out, c_out, hw, k, es, mode, impute, with_sol):
    from dlwp_amd import ops
    rng = np.random.default_rng(n * 100 + k * 10 + es)
    dev = torch.device('cuda:0')
    p = rng.standard_normal((n, t_in, c_in) + hw).astype(np.float32)
    p[rng.integers(0, n)] = np.nan                                     # a row that already ran out travels as NaN
    r = rng.standard_normal((n, t_out, c_out) + hw).astype(np.float32)
    n_var = c_in - 1 if with_sol else c_in
    shared = min(n_var, c_out)
    idx_in = list(rng.permutation(n_var)[:shared])                     # different orderings on the two sides
    idx_out = list(rng.permutation(c_out)[:shared])
    sol_idx = c_in - 1
    tail = min(es, n)
    sol = rng.standard_normal((tail, t_in) + hw).astype(np.float32) if with_sol else None
    mean = p[np.isfinite(p).all(axis=(1, 2, 3, 4))].mean(axis=0) if impute else None
    keep, first = mode == 'keep', mode == 'first'
    want = np_ref.estimator_next_state(p, r, k, es, idx_in, idx_out, keep_inputs=keep, prefer_first_times=first, mean=mean,
                                       sol=sol, sol_idx=sol_idx)
    src = list(range(t_in * c_in))
    for ji, jo in zip(idx_in, idx_out):
        if keep:                                                       # (keep_inputs: es = t_out <= t_in)
            for m in range(es):
                src[(t_in - es + m) * c_in + ji] = -1 - (m * c_out + jo)
        else:
            f0 = 0 if first else t_out - t_in
            for ts in range(t_in):
                src[ts * c_in + ji] = -1 - ((f0 + ts) * c_out + jo)
    sol_map = None
    if with_sol:
        sol_map = [-1] * (t_in * c_in)
        for ts in range(t_in):
            sol_map[ts * c_in + sol_idx] = ts
    fb = ops.make_feedback(n, t_in * c_in, t_out * c_out, hw[0] * hw[1], src, shift=k, tail=tail, sol=sol_map,
                           sol_planes=t_in if with_sol else 0)
    to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    got = ops.state_feedback(to(p.reshape((n, -1) + hw)), to(r.reshape((n, -1) + hw)), fb, sol=to(sol), mean=to(mean))
    torch.cuda.synchronize()
    assert np.array_equal(_bits(got.cpu().numpy().reshape(p.shape)), _bits(want))       # NaN payloads included


def test_state_feedback_rejects_what_it_cannot_do():
    from dlwp_amd import _lib, ops
    dev = torch.device('cuda:0')
    a = torch.zeros((4, 3, 4, 4), device=dev)
    o = torch.zeros((4, 2, 4, 4), device=dev)
    with pytest.raises(_lib.DlwpError, match='alias'):
        ops.state_feedback(a, o, ops.make_feedback(4, 3, 2, 16, [0, 1, 2]), new_state=a)
    with pytest.raises(_lib.DlwpError, match='takes source'):
        ops.state_feedback(a, o, ops.make_feedback(4, 3, 2, 16, [0, 1, -3]))           # output channel 2 of 2
    with pytest.raises(_lib.DlwpError, match='insolation plane'):
        ops.state_feedback(a, o, ops.make_feedback(4, 3, 2, 16, [0, 1, 2], tail=1, sol=[-1, -1, 1], sol_planes=1))
    with pytest.raises(ValueError, match='at most'):
        ops.make_feedback(4, 200, 2, 16, list(range(200)))


@pytest.mark.parametrize('n,t,c,hw,kept,time_major', [(5, 2, 4, (6, 8), 2, True), (5, 2, 4, (6, 8), 1, True), (3, 3, 2, (5, 7), 3, False),
                                                      (64, 2, 2, (88, 180), 2, True)])
def test_series_arrange_is_the_returned_layout(n, t, c, hw, kept, time_major):
    """DLWP/model/extensions.py:260-263, 298-302: the [:, :, :es] cut, time first, (variable, level) in sorted order -- per call"""
    import ctypes
    from dlwp_amd import _lib
    rng = np.random.default_rng(n + t)
    x = rng.standard_normal((n, t, c) + hw).astype(np.float32)
    perm = [int(v) for v in rng.permutation(c)]
    want = x[:, :kept][:, :, perm]
    if time_major:
        want = want.transpose(1, 0, 2, 3, 4)
    src = torch.from_numpy(x).cuda()
    dst = torch.empty(want.shape, dtype=torch.float32, device='cuda')
    _lib.check(_lib.lib.dlwp_series_arrange(_lib.handle(0), ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), n, t, c,
                                            hw[0] * hw[1], kept, (ctypes.c_int * c)(*perm), int(time_major), _lib.F32,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(_bits(dst.cpu().numpy()), _bits(np.ascontiguousarray(want)))


def _mixing_network(c_in, c_out, hw):
    """oracle.estimator_cases.mixing_model as a REAL network: out[:, j] = tanh(sum_i A[j, i] p[:, i] + b[j]) is a 1 x 1 Conv2D
    with tanh -- the goldens of the reference's predict() then pin the device loop end to end."""
    from dlwp_amd.model import DLWPNeuralNet
    rng = np.random.RandomState(3)                                     # (mixing_model's seed and draw order)
    A = rng.uniform(-0.6, 0.6, size=(c_out, c_in)).astype(np.float32)
    b = rng.uniform(-0.2, 0.2, size=(c_out,)).astype(np.float32)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=1, scaler_type=None, scale_targets=False)
    d.build_model([('Conv2D', (c_out, 1), {'activation': 'tanh', 'data_format': 'channels_first', 'input_shape': (c_in,) + hw})],
                  loss='mse', optimizer='adam')
    d.model.set_weights([A.T.reshape(1, 1, c_in, c_out).copy(), b])
    return d


@pytest.mark.parametrize('tag,case,varlev', _estimator_cases(), ids=[c[0] for c in _estimator_cases()])
def test_device_loop_equals_the_reference_estimator(golden, tag, case, varlev, monkeypatch):
    """Every case of tests/golden/estimator.npz -- what the REFERENCE's TimeSeriesEstimator.predict returns -- through the device
    loop: NaN pattern and coordinates exact, values within the forward tolerance (1 x 1 convolution + tanh on the matrix cores
    against numpy's einsum), and bit-identical to the reference-form host loop around the same device forward."""
    from dlwp_amd.model import SeriesDataGenerator, TimeSeriesEstimator
    g = golden('estimator')
    kw = case['gen']
    c_in, c_out = [int(v) for v in g['%s_channels' % tag]]
    d = _mixing_network(c_in, c_out, (6, 8))
    d.time_dim = kw['input_time_steps']
    gen = SeriesDataGenerator(d, _golden_dataset(golden('series'), varlev), **kw)
    est = TimeSeriesEstimator(d, gen)
    calls = []
    real = d.model._fed_entry
    monkeypatch.setattr(d.model, '_fed_entry', lambda *a, **k: calls.append(1) or real(*a, **k))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out = est.predict(**case['predict'])
    same_io = tag in ('same', 'varlev_same', 'varlev_same_keep')
    assert bool(calls) != same_io                                      # (inputs == outputs is the plain one-graph rollout)
    want = g['%s_values' % tag]
    assert list(out.dims) == [str(v) for v in g['%s_dims' % tag]]
    assert out.values.dtype == np.float32 and out.values.shape == want.shape
    assert np.array_equal(np.isnan(out.values), np.isnan(want))
    assert np.allclose(out.values, want, rtol=0, atol=1e-5, equal_nan=True)
    for dim in out.dims:
        c, ref = np.asarray(out.coords[dim]), g['%s_coord_%s' % (tag, dim)]
        if c.dtype.kind == 'M':
            c = c.astype('datetime64[s]').astype(np.int64)
        elif c.dtype.kind == 'm':
            c = c.astype('timedelta64[s]').astype(np.int64)
        elif c.dtype.kind in 'OU':
            c, ref = np.array([str(v) for v in c]), np.array([str(v) for v in ref])
        assert np.array_equal(c, ref), dim
    monkeypatch.setenv('DLWP_ESTIMATOR_HOST', '1')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        host = est.predict(**case['predict'])
    assert np.array_equal(_bits(out.values), _bits(host.values))


def _series_dataset(rng, n_t, h, w, variables=('z', 't'), levels=(500,)):
    from dlwp_amd.model import SeriesDataset
    dates = (np.datetime64('2010-03-01T00') + np.arange(n_t) * np.timedelta64(6, 'h')).astype('datetime64[s]')
    series = rng.standard_normal((n_t, len(variables), len(levels), h, w)).astype(np.float32)
    return SeriesDataset(series, {'sample': dates, 'variable': np.array(variables), 'level': np.array(levels),
                                  'lat': np.linspace(85., -85., h), 'lon': np.arange(0., 360., 360. / w)},
                         ('sample', 'variable', 'level', 'lat', 'lon'))


@pytest.mark.parametrize('interval,impute', [(1, False), (2, False), (1, True)])
def test_validate_script_rollout_with_insolation_runs_on_the_device(monkeypatch, interval, impute):
    """examples/validate.py:191-205 with a U-Net: SeriesDataGenerator(add_insolation=True) -> TimeSeriesEstimator.predict.  The
    device loop (one hipGraph) == the reference-form host loop (model.predict + numpy re-indexing per step) bit for bit, and
    the first step == the oracle's forward."""
    from dlwp_amd.model import SeriesDataGenerator, TimeSeriesEstimator
    from tests.test_gpu_model import _build, _weights_of
    rng = np.random.default_rng(7)
    h, w = 16, 24
    ds = _series_dataset(rng, 20, h, w)
    d = _build(unet_layers((6, h, w), widths=(8, 16, 16, 16, 8), cout=4), time_dim=2)
    weights = _weights_of(d.model, rng)
    gen = SeriesDataGenerator(d, ds, input_time_steps=2, output_time_steps=2, add_insolation=True, interval=interval, batch_size=4)
    est = TimeSeriesEstimator(d, gen)
    calls = []
    real = d.model._fed_entry
    monkeypatch.setattr(d.model, '_fed_entry', lambda *a, **k: calls.append(1) or real(*a, **k))
    out = est.predict(7, impute=impute)
    assert calls and out.shape[0] == 7
    monkeypatch.setenv('DLWP_ESTIMATOR_HOST', '1')
    host = est.predict(7, impute=impute)
    assert len(calls) == 1
    assert np.array_equal(_bits(out.values), _bits(host.values))
    assert np.array_equal(out.coords['f_hour'], host.coords['f_hour'])
    n = gen._n_sample
    # every variable is predicted and the insolation of the last es rows is refreshed: with k = es nothing ever runs out; with
    # interval 2 (k = es + 1) one row per call keeps a NaN insolation plane
    assert bool(np.isfinite(out.values).all()) == (interval == 1) and np.isfinite(out.values[:2]).all()
    X, _ = gen.generate([], scale_and_impute=False)
    want = np_ref.run_layers(unet_layers((6, h, w), widths=(8, 16, 16, 16, 8), cout=4), X, weights).reshape(n, 2, 2, h, w)
    got0 = out.values[:2, :, ::-1, 0].transpose(1, 0, 2, 3, 4)          # variables come back sorted ('t', 'z')
    assert np.abs(got0 - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # a second call replays the cached graphs
    monkeypatch.delenv('DLWP_ESTIMATOR_HOST')
    again = est.predict(7, impute=impute)
    assert len(calls) == 2
    # the device-resident form (all calls in ONE graph, bench.py's timed region): the same series, un-arranged
    dev = est.predict(7, impute=impute, return_device=True)
    assert isinstance(dev, torch.Tensor) and tuple(dev.shape) == (4, n, 4, h, w)
    ser = dev.cpu().numpy().reshape(4, n, 2, 2, h, w).transpose(0, 2, 1, 3, 4, 5).reshape(8, n, 2, h, w)[:7, :, ::-1]
    assert np.array_equal(_bits(ser), _bits(host.values[:, :, :, 0]))
    assert np.array_equal(_bits(again.values), _bits(host.values))


def test_step_sequence_rollout_runs_on_the_device(monkeypatch):
    """DLWPNeuralNet.predict_timeseries(step_sequence=True), DLWP/model/models.py:280-290: one predicted step per call, the
    other inputs shift by one time slice.  Device loop == the oracle's loop around the same device forward, bit for bit."""
    from tests.test_gpu_model import _build, _weights_of
    rng = np.random.default_rng(9)
    cs = (6, 16, 24)                                                    # 3 time steps x 2 variables
    d = _build(unet_layers(cs, widths=(8, 16, 16, 16, 8)), time_dim=3)
    _weights_of(d.model, rng)
    x = rng.standard_normal((5,) + cs).astype(np.float32)
    calls = []
    real = d.model._fed_entry
    monkeypatch.setattr(d.model, '_fed_entry', lambda *a, **k: calls.append(1) or real(*a, **k))
    for keep in (False, True):
        got = d.predict_timeseries(x, 4, step_sequence=True, keep_time_dim=keep)
        want = np_ref.predict_timeseries_nn(d.predict, x, 4, 3, step_sequence=True, keep_time_dim=keep)
        assert got.shape == want.shape and np.array_equal(_bits(got), _bits(want))
    assert len(calls) == 2
    dev = d.predict_timeseries(x, 4, step_sequence=True, return_device=True)
    assert isinstance(dev, torch.Tensor) and np.array_equal(dev.cpu().numpy(), got[:, :, 0] if got.ndim == 6 else got)


def test_recurrent_model_with_insolation_rolls_out_on_the_device(monkeypatch):
    """The reference's DEFAULT flow (examples/train.py with model_is_recurrent = True, then examples/validate.py): a ConvLSTM2D front
    end on (time, variables + insolation, lat, lon) inputs, outputs (time, variables, lat, lon).  The state's channel axis is the
    flattened (time step, variable) pair either way, so the same feedback launch serves: device loop == reference-form host loop."""
    from dlwp_amd.model import DLWPNeuralNet, SeriesDataGenerator, TimeSeriesEstimator
    rng = np.random.default_rng(17)
    h, w = 16, 24
    ds = _series_dataset(rng, 18, h, w)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    gen = SeriesDataGenerator(d, ds, input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4)
    cs, cso = gen.convolution_shape, gen.output_convolution_shape        # (2, 3, h, w) -> (2, 2, h, w)
    cf = {'data_format': 'channels_first'}

    def block(k, filters, size, dilation, activation):
        return (('PeriodicPadding2D', ((0, k),), dict(cf)), ('ZeroPadding2D', ((k, 0),), dict(cf)),
                ('Conv2D', (filters, size), dict(cf, dilation_rate=dilation, padding='valid', activation=activation)))
    layers = (('PeriodicPadding3D', ((0, 0, 2),), dict(cf, input_shape=cs)), ('ZeroPadding3D', ((0, 2, 0),), dict(cf)),
              ('ConvLSTM2D', (4 * cs[1], 3), dict(cf, dilation_rate=2, padding='valid', activation='tanh', return_sequences=True)),
              ('Reshape', ((4 * cs[0] * cs[1], cs[2], cs[3]),), None)) + block(1, 16, 3, 1, 'tanh') + \
        block(2, cso[0] * cso[1], 5, 1, 'linear') + (('Reshape', (cso,), None),)
    np.random.seed(5)
    d.build_model(layers, loss='mse', optimizer='adam')
    est = TimeSeriesEstimator(d, gen)
    calls = []
    real = d.model._fed_entry
    monkeypatch.setattr(d.model, '_fed_entry', lambda *a, **k: calls.append(1) or real(*a, **k))
    out = est.predict(5)
    assert calls and out.shape == (5, gen._n_sample, 2, 1, h, w) and np.isfinite(out.values).all()
    monkeypatch.setenv('DLWP_ESTIMATOR_HOST', '1')
    host = est.predict(5)
    assert len(calls) == 1 and np.array_equal(_bits(out.values), _bits(host.values))
