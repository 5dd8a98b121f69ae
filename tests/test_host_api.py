"""Host-side logic of the drop-in boundary, on CPU: build_model's contract and error behaviour (reference
DLWP/model/models.py:63-112), the fusion planner, the generic predict_timeseries loop and DataGenerator against the
golden vectors produced from the reference's own source, callbacks, model files."""
import os
import types

import numpy as np
import pytest

from dlwp_amd import custom, layers as L, plan as P, util
from dlwp_amd.engine import Model, Sequential
from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPFunctional, DLWPNeuralNet
from tests.nets import CF, cnn2_layers, unet_layers


def _dlwp(**kw):
    kw.setdefault('scaler_type', None)
    kw.setdefault('scale_targets', False)
    return DLWPNeuralNet(is_convolutional=True, **kw)


# ---- build_model ------------------------------------------------------------------------------------------------- #

def test_build_model_accepts_reference_triples_and_fuses_to_six_launches():
    from dlwp_amd import ops
    ops.set_winograd(False)                                     # direct kernels: every pooling fused into a loader
    try:
        d = _dlwp(time_dim=2)
        d.build_model(unet_layers((4, 88, 180)), loss='mse', optimizer='adam', metrics=['mae'], gpus=1)
    finally:
        ops.set_winograd(True)
    assert d.model is d.base_model and d.gpus == 1
    assert d.model.count_params() == 188996
    assert d.model.output_shape == (None, 4, 88, 180)
    plan = d.model.plan
    # 22 reference layers -> 6 fused convolutions (+ the derived kernels and the interleave of the restated output layer)
    assert [op.kind for op in plan.ops] == ['conv'] * 5 + ['phasew', 'conv', 'd2s']
    assert plan.conv_flops_per_sample() == 1597685760           # 1 597.7 MFLOP
    convs = [op for op in plan.ops if op.kind == 'conv']
    # layer 5 (3x3, dilation 2, on an up-sampled tensor) is its own up-sampling identity: a dilation-1 layer on the
    # 44x90 tensor (halo 1); layer 6 (5x5 on that up-sampled output) runs as 3x3 phase kernels on the same 44x90 tensor
    assert [op.src_mode for op in convs] == [0, 2, 2, 1, 0, 0]
    assert all(op.halo.mode_h == P.PAD_ZERO and op.halo.mode_w == P.PAD_WRAP for op in convs)
    assert [op.halo.left for op in convs] == [2, 1, 1, 1, 1, 1]
    assert convs[4].conv_geometry == (32, (3, 3), (1, 1)) and convs[4].out_shape == (32, 44, 90)
    assert convs[5].conv_geometry == (16, (3, 3), (1, 1)) and convs[5].wparam == 0
    assert plan.ops[-1].dst == P.OUT(0) and plan.ops[0].src == P.STATE_IN
    assert d.model.metrics_names == ['loss', 'mean_absolute_error']
    names = [lay.name for lay in d.base_model.layers]
    assert len(names) == 22 and all(hasattr(lay, 'output_shape') for lay in d.base_model.layers)


def test_default_plan_pools_once_in_front_of_the_winograd_layers():
    """With the Winograd family on (default) the two pooled 3x3 layers (32->64, 64->128) read a pooled tensor written by
    dlwp_maxpool2_fwd: 8 launches, same FLOPs, every halo still fused."""
    d = _dlwp(time_dim=2)
    d.build_model(unet_layers((4, 88, 180)), loss='mse', optimizer='adam', gpus=1)
    plan = d.model.plan
    assert [op.kind for op in plan.ops] == ['conv', 'maxpool', 'conv', 'maxpool', 'conv', 'conv', 'conv', 'phasew', 'conv',
                                            'd2s']
    assert plan.conv_flops_per_sample() == 1597685760
    assert [op.src_mode for op in plan.ops if op.kind == 'conv'] == [0, 0, 0, 1, 0, 0]      # (layers 5, 6 restated, see above)
    assert [op.halo.left for op in plan.ops if op.kind == 'conv'] == [2, 1, 1, 1, 1, 1]


def test_build_model_argument_errors_match_the_reference():
    d = _dlwp()
    with pytest.raises(TypeError, match="'gpus' argument must be an int"):
        d.build_model((), gpus=1.0)
    with pytest.raises(TypeError, match="'layers' argument must be a tuple"):
        d.build_model('Conv2D')
    with pytest.raises(TypeError, match="each element of 'layers' must be a tuple"):
        d.build_model(('Conv2D',))
    with pytest.raises(ValueError, match='three elements'):
        d.build_model((('Conv2D', (4, 3)),))
    with pytest.raises(TypeError, match="'args' element of layer 0 must be a tuple"):
        d.build_model((('Conv2D', [4, 3], {}),))
    with pytest.raises(TypeError, match="'kwargs' element of layer 0 must be a dict"):
        d.build_model((('Conv2D', (4, 3), [1]),))
    with pytest.raises(AttributeError):
        d.build_model((('NoSuchLayer', (), {}),))
    with pytest.raises(ValueError, match="'time_dim' must be >= 1"):
        DLWPNeuralNet(time_dim=0)
    # None for args / kwargs is allowed (examples/train.py:220 passes None)
    d.build_model((('PeriodicPadding2D', None, dict(CF, input_shape=(2, 6, 8))), ('ZeroPadding2D', ((1, 0),), dict(CF)),
                   ('Conv2D', (2, 3), dict(CF))), loss='mse')
    assert d.model.output_shape == (None, 2, 8, 8)     # default PeriodicPadding2D padding is (1, 1)


def test_shape_errors_surface_at_build_time():
    d = _dlwp()
    with pytest.raises(ValueError, match='periodic padding'):
        d.build_model((('PeriodicPadding2D', ((0, 9),), dict(CF, input_shape=(2, 6, 8))),), loss='mse')
    with pytest.raises(ValueError):
        d.build_model((('Conv2D', (2, 7), dict(CF, input_shape=(2, 5, 5))),), loss='mse')
    with pytest.raises(NotImplementedError, match='channels_first'):
        d.build_model((('Conv2D', (2, 3), {'input_shape': (2, 5, 5)}),), loss='mse')


def test_registry_resolves_keras_and_dlwp_custom_names():
    assert util.get_from_class('keras.layers', 'Conv2D') is L.Conv2D
    assert util.get_from_class('DLWP.custom', 'PeriodicPadding2D') is custom.PeriodicPadding2D
    assert util.get_from_class('dlwp_amd.custom', 'FillPadding2D') is custom.FillPadding2D
    with pytest.raises(AttributeError):
        util.get_from_class('keras.layers', 'PeriodicPadding2D')    # custom layers live in the fallback module


def test_padding_argument_forms():
    assert custom.PeriodicPadding2D(2, data_format='channels_first').padding == ((2, 2), (2, 2))
    assert custom.PeriodicPadding2D((0, 2), data_format='channels_first').padding == ((0, 0), (2, 2))
    assert custom.FillPadding2D(((1, 2), (3, 1))).padding == ((1, 2), (3, 1))
    assert custom.FillPadding2D((1, 1)).data_format == 'channels_last'        # Keras default when None
    with pytest.raises(ValueError):
        custom.PeriodicPadding2D((1, 2, 3))
    lay = custom.PeriodicPadding2D(((1, 2), (3, 1)), data_format='channels_first')
    assert lay.compute_output_shape((4, 5, 6)) == (4, 8, 10)
    assert custom.FillPadding2D((2, 1), data_format='channels_last').compute_output_shape((5, 6, 4)) == (9, 8, 4)


# ---- planner ------------------------------------------------------------------------------------------------------- #

def _skip_model(cs=(4, 16, 24)):
    """examples/train_functional.py:248-275 (non-recurrent skip U-Net)."""
    x0 = L.Input(shape=cs)
    pp2, zp2 = custom.PeriodicPadding2D((0, 2), **CF), L.ZeroPadding2D((2, 0), **CF)
    pp1, zp1 = custom.PeriodicPadding2D((0, 1), **CF), L.ZeroPadding2D((1, 0), **CF)
    pool, up = L.MaxPooling2D(2, **CF), L.UpSampling2D(2, **CF)
    c1 = L.Conv2D(32, 3, dilation_rate=2, activation='tanh', **CF)
    c2 = L.Conv2D(64, 3, activation='tanh', **CF)
    c3 = L.Conv2D(128, 3, activation='tanh', **CF)
    c4 = L.Conv2D(32, 3, activation='tanh', **CF)
    c5 = L.Conv2D(16, 3, dilation_rate=2, activation='tanh', **CF)
    c6 = L.Conv2D(cs[0], 5, activation='linear', **CF)
    x = c1(pp2(zp2(x0)))
    x, x1 = custom.slice_layer(0, 16, axis=1)(x), custom.slice_layer(16, 32, axis=1)(x)
    x = c2(pp1(zp1(pool(x))))
    x, x2 = custom.slice_layer(0, 32, axis=1)(x), custom.slice_layer(32, 64, axis=1)(x)
    x = c3(pp1(zp1(pool(x))))
    x = c4(pp1(zp1(up(x))))
    x = L.concatenate([x, x2], axis=1)
    x = c5(pp2(zp2(up(x))))
    x = L.concatenate([x, x1], axis=1)
    x = c6(pp2(zp2(x)))
    return x0, x


def test_planner_skip_unet_slices_are_free_and_concat_is_a_copy():
    from dlwp_amd import ops
    x0, y = _skip_model()
    ops.set_winograd(False)         # the fully fused plan (with Winograd on, pooled slices are materialised first)
    try:
        m = Model(inputs=x0, outputs=y)
    finally:
        ops.set_winograd(True)
    kinds = [op.kind for op in m.plan.ops]
    # (c5 -- dilation 2 on an up-sampled tensor -- is computed at low resolution; its up-sampling lands in the concat buffer)
    assert kinds.count('conv') == 6 and kinds.count('copy') == 4 and set(kinds) == {'conv', 'copy', 'upsample'}
    convs = [op for op in m.plan.ops if op.kind == 'conv']
    assert (convs[1].in_c_off, convs[1].xs[0], convs[1].in_c_total) == (0, 16, 32)     # slice_layer(0,16) read in place
    assert (convs[2].in_c_off, convs[2].xs[0], convs[2].in_c_total) == (0, 32, 64)
    assert convs[4].xs[0] == 64 and convs[5].xs[0] == 32
    assert m.count_params() == sum(int(np.prod(w.shape)) for w in m.weights)


def test_planner_materialises_what_it_cannot_fuse():
    # wrap-of-wrap on the same axis does not compose -> standalone pad, then a fused one
    m = Sequential([custom.PeriodicPadding2D((0, 1), input_shape=(2, 6, 8), **CF), custom.PeriodicPadding2D((0, 1), **CF),
                    L.Conv2D(3, 3, **CF)])
    assert [op.kind for op in m.plan.ops] == ['pad', 'conv']
    assert m.output_shape == (None, 3, 4, 10)
    # zero-of-zero extends
    m = Sequential([L.ZeroPadding2D((1, 0), input_shape=(2, 6, 8), **CF), L.ZeroPadding2D((1, 1), **CF), L.Conv2D(3, 3, **CF)])
    assert [op.kind for op in m.plan.ops] == ['conv'] and tuple(m.plan.ops[0].halo)[:4] == (2, 2, 1, 1)
    # a model that ends in non-conv layers is materialised into the output slot
    m = Sequential([L.Conv2D(4, 3, input_shape=(2, 8, 8), **CF), L.MaxPooling2D(2, **CF), custom.FillPadding2D((1, 0), **CF)])
    assert [op.kind for op in m.plan.ops] == ['conv', 'maxpool', 'pad'] and m.plan.ops[-1].dst == P.OUT(0)
    assert m.output_shape == (None, 4, 5, 3)
    # 'same' padding becomes a zero halo
    m = Sequential([L.Conv2D(4, 3, padding='same', dilation_rate=2, input_shape=(2, 8, 8), **CF)])
    assert tuple(m.plan.ops[0].halo) == (2, 2, 2, 2, 0, 0) and m.output_shape == (None, 4, 8, 8)
    # channels_last padding layers run standalone (rows of W*C floats)
    m = Sequential([custom.PeriodicPadding2D((1, 2), input_shape=(5, 6, 3))])
    assert m.plan.ops[0].kind == 'pad' and m.plan.ops[0].inner == 3 and m.output_shape == (None, 7, 10, 3)


def test_multi_output_functional_model_chains_through_output_slots():
    x0 = L.Input(shape=(2, 8, 12))
    pp, zp = custom.PeriodicPadding2D((0, 1), **CF), L.ZeroPadding2D((1, 0), **CF)
    conv = L.Conv2D(2, 3, activation='tanh', **CF)
    outs = [conv(pp(zp(x0)))]
    for _ in range(2):
        outs.append(conv(pp(zp(outs[-1]))))
    m = Model(inputs=x0, outputs=outs)
    assert [op.dst for op in m.plan.ops] == [P.OUT(0), P.OUT(1), P.OUT(2)]
    assert [op.src for op in m.plan.ops] == [P.STATE_IN, P.OUT(0), P.OUT(1)]
    assert m.count_params() == 3 * 3 * 2 * 2 + 2                  # one shared layer
    f = DLWPFunctional(time_dim=1)
    f.build_model(m, loss='mse', loss_weights=[1. / 3] * 3, optimizer='adam', metrics=['mae'])
    assert f._n_steps == 3
    assert m.metrics_names[0] == 'loss' and len(m.metrics_names) == 1 + 3 + 3


# ---- rollout bookkeeping vs the reference goldens (generic host loop with a foreign model object) ------------------------ #

def _step(kind):
    if kind == 0:
        return lambda p, **kw: (0.5 * p + 1.0).astype(np.float32)
    return lambda p, **kw: np.tanh(np.roll(p, 1, axis=-1) * 0.75 + 0.1 * p).astype(np.float32)


def test_predict_timeseries_matches_reference_goldens(golden):
    g = golden('rollout')
    for i in range(int(g['nn_n'])):
        time_dim, steps, seq, keep, rec, nl = (int(v) for v in g['nn_%d_cfg' % i])
        d = DLWPNeuralNet(is_convolutional=True, is_recurrent=bool(rec), time_dim=time_dim, scaler_type=None,
                          scale_targets=False)
        d.model = types.SimpleNamespace(predict=_step(nl))
        got = d.predict_timeseries(g['nn_%d_in' % i], steps, step_sequence=bool(seq), keep_time_dim=bool(keep))
        want = g['nn_%d_out' % i]
        assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got, want), i


def test_functional_predict_timeseries_matches_reference_goldens(golden):
    g = golden('rollout')
    for i in range(int(g['fn_n'])):
        time_dim, steps, n_out, keep, rec = (int(v) for v in g['fn_%d_cfg' % i])
        f = DLWPFunctional(is_convolutional=True, is_recurrent=bool(rec), time_dim=time_dim)
        f._n_steps = n_out
        step = _step(1)

        def predict(p, _n=n_out, **kw):
            outs, q = [], p
            for _ in range(_n):
                q = step(q)
                outs.append(q)
            return outs[0] if _n == 1 else outs
        f.model = types.SimpleNamespace(predict=predict)
        got = f.predict_timeseries(g['fn_%d_in' % i], steps, keep_time_dim=bool(keep))
        want = g['fn_%d_out' % i]
        assert got.shape == want.shape and np.array_equal(got, want), i


def test_predict_timeseries_error_behaviour():
    d = _dlwp(time_dim=2)
    d.model = types.SimpleNamespace(predict=_step(0))
    with pytest.raises(ValueError, match='time_steps must be an int > 0'):
        d.predict_timeseries(np.zeros((1, 4, 2, 2), np.float32), 0)
    f = DLWPFunctional(time_dim=1)
    f.model = types.SimpleNamespace(predict=_step(0))
    with pytest.raises(ValueError, match='time_steps must be an int > 0'):
        f.predict_timeseries(np.zeros((1, 4, 2, 2), np.float32), -3)
    d2 = DLWPNeuralNet(scaler_type='StandardScaler')
    with pytest.raises(AttributeError, match='init_fit'):
        d2.fit(np.zeros((2, 3)), np.zeros((2, 3)), initialize=False)


# ---- data feed vs the reference goldens ------------------------------------------------------------------------------ #

def test_data_generator_matches_reference_goldens(golden):
    g = golden('generator')
    P_, T_ = g['P'], g['T']
    for tag, rec in (('conv', False), ('rec', True)):
        d = _dlwp(is_recurrent=rec, time_dim=2)
        gen = DataGenerator(d, ArrayDataset(P_, T_), batch_size=4)
        assert len(gen) == int(g['%s_len' % tag])
        assert tuple(gen.shape) == tuple(g['%s_shape' % tag])
        assert gen.n_features == int(g['%s_n_features' % tag])
        assert tuple(gen.dense_shape) == tuple(g['%s_dense_shape' % tag])
        assert tuple(gen.convolution_shape) == tuple(g['%s_convolution_shape' % tag])
        assert tuple(gen.shape_2d) == tuple(g['%s_shape_2d' % tag])
        for b in range(len(gen)):
            X, y = gen[b]
            assert np.array_equal(X, g['%s_X%d' % (tag, b)]) and np.array_equal(y, g['%s_y%d' % (tag, b)])
        assert np.array_equal(gen[-1][0], g['%s_Xneg1' % tag])
        Xa, ya = gen.generate([], scale_and_impute=False)
        assert np.array_equal(Xa, g['%s_Xall' % tag]) and np.array_equal(ya, g['%s_yall' % tag])
    dense = DLWPNeuralNet(is_convolutional=False, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    assert np.array_equal(DataGenerator(dense, ArrayDataset(P_, T_), batch_size=4)[0][0], g['dense_X0'])
    for seed in (0, 7):
        np.random.seed(seed)
        gen = DataGenerator(_dlwp(time_dim=2), ArrayDataset(P_, T_), batch_size=4, shuffle=True)
        assert np.array_equal(gen._indices, g['shuffle_%d_epoch0' % seed])
        assert np.array_equal(gen[0][0], g['shuffle_%d_X0' % seed])
        gen.on_epoch_end()
        assert np.array_equal(gen._indices, g['shuffle_%d_epoch1' % seed])
    P4 = g['P4']
    gen = DataGenerator(_dlwp(time_dim=1), ArrayDataset(P4, P4 * 2), batch_size=3)
    assert tuple(gen.shape) == tuple(g['nots_shape']) and len(gen) == int(g['nots_len'])
    assert tuple(gen.convolution_shape) == tuple(g['nots_convolution_shape'])
    X, y = gen[2]
    assert np.array_equal(X, g['nots_X2']) and np.array_equal(y, g['nots_y2'])


def test_data_generator_edge_cases_and_documented_deviations(golden):
    g = golden('generator')
    d = _dlwp(time_dim=2)
    with pytest.raises(ValueError, match="'predictors' and 'targets'"):
        DataGenerator(d, object())
    gen = DataGenerator(d, ArrayDataset(g['P'], g['T']), batch_size=4)
    with pytest.raises(IndexError):
        gen[len(gen)]           # the reference returns the WHOLE dataset here (off-by-one)
    with pytest.raises(IndexError):
        gen[-len(gen) - 1]
    # a batch that really contains NaN samples: the reference raises a reshape error; here the samples are dropped
    gen = DataGenerator(d, ArrayDataset(g['nan_P'], g['nan_T']), batch_size=10)
    X, y = gen[0]
    assert X.shape == (8, 4, 6, 8) and not np.isnan(X).any() and not np.isnan(y).any()
    assert np.array_equal(X.reshape(8, -1), g['nan_p_out'].reshape(8, -1))


def test_delete_nan_samples_and_split_match_reference_goldens(golden):
    g = golden('generator')
    p, t = util.delete_nan_samples(g['nan_P'].copy(), g['nan_T'].copy())
    assert np.array_equal(p, g['nan_p_out']) and np.array_equal(t, g['nan_t_out'])
    p, t = util.delete_nan_samples(g['large_P'].copy(), g['T'].copy(), large_fill_value=True)
    assert np.array_equal(p, g['large_p_out']) and np.array_equal(t, g['large_t_out'])
    p, t = util.delete_nan_samples(g['thr_P'].copy(), g['T'].copy(), threshold=0.25)
    assert np.array_equal(p, g['thr_p_out'], equal_nan=True) and np.array_equal(t, g['thr_t_out'])
    with pytest.raises(ValueError):
        util.delete_nan_samples(g['P'], g['T'], threshold=2)
    for method in ('first', 'last'):
        tr, te = util.train_test_split_ind(10, 3, method=method)
        assert np.array_equal(tr, g['split_%s_train' % method]) and np.array_equal(te, g['split_%s_test' % method])
    tr, te = util.train_test_split_ind(10, 3, method='random')
    assert sorted(tr + te) == list(range(10)) and len(te) == 3
    with pytest.raises(ValueError):
        util.train_test_split_ind(10, 3, method='middle')


# ---- callbacks / persistence ------------------------------------------------------------------------------------------ #

def test_early_stopping_min_semantics():
    class M(object):
        stop_training = False
        w = [np.zeros(2)]

        def get_weights(self):
            return [a.copy() for a in self.w]

        def set_weights(self, ws):
            self.w = ws
    m = M()
    cb = custom.EarlyStoppingMin(min_epochs=3, monitor='loss', min_delta=0., patience=2, restore_best_weights=True)
    cb.set_model(m)
    cb.on_train_begin()
    losses = [5., 6., 7., 4., 4.5, 4.6, 1.0]
    stopped_at = None
    for ep, lv in enumerate(losses):
        m.w = [np.full(2, float(ep))]
        cb.on_epoch_end(ep, {'loss': lv})
        if m.stop_training:
            stopped_at = ep
            break
    assert stopped_at == 5 and cb.stopped_epoch == 5            # epochs 0-2 ignored; best at 3; patience 2
    assert np.array_equal(m.w[0], np.full(2, 3.0))              # best weights restored
    with pytest.raises(ValueError):
        custom.EarlyStoppingMin(min_epochs=-1)
    h = custom.History()
    h.on_train_begin()
    h.on_epoch_end(0, {'loss': 1.0})
    h.on_epoch_end(1, {'loss': 0.5})
    assert h.history == {'loss': [1.0, 0.5]} and h.epoch == [0, 1]


def test_save_and_load_model_round_trip(tmp_path):
    d = _dlwp(time_dim=2)
    d.build_model(cnn2_layers((2, 9, 12), hidden=8), loss='mse', optimizer='adam', metrics=['mae'])
    w0 = d.model.get_weights()
    assert [w.shape for w in w0] == [(5, 5, 2, 8), (8,), (5, 5, 8, 2), (2,)]          # Keras HWIO layout
    hist = custom.History()
    hist.on_train_begin()
    hist.on_epoch_end(0, {'loss': 2.0})
    base = os.path.join(str(tmp_path), 'model')
    util.save_model(d, base, history=hist)
    assert all(os.path.exists(base + ext) for ext in ('.keras', '.pkl', '.history'))
    d2, h2 = util.load_model(base, history=True)
    assert isinstance(d2, DLWPNeuralNet) and d2.time_dim == 2 and h2 == {'loss': [2.0]}
    assert all(np.array_equal(a, b) for a, b in zip(w0, d2.model.get_weights()))
    assert [op.kind for op in d2.model.plan.ops] == ['conv', 'conv']
    assert d2.model.optimizer.lr == pytest.approx(1e-3) and d2.model.metrics_names == ['loss', 'mean_absolute_error']
    # a custom loss survives the round trip (examples/train.py saves models compiled with the ACC loss)
    climo = np.random.default_rng(0).standard_normal((1, 2, 9, 12)).astype(np.float32)
    lats = np.linspace(80, -80, 9)
    d3 = _dlwp(time_dim=2)
    d3.build_model(cnn2_layers((2, 9, 12), hidden=8), optimizer='adam', metrics=['mae'],
                   loss=custom.latitude_weighted_loss(custom.anomaly_correlation_loss(climo, regularize_mean='mse'), lats,
                                                      (2, 9, 12), axis=-2, weighting='midlatitude'))
    util.save_model(d3, base + '_acc')
    d4 = util.load_model(base + '_acc')
    sp = d4.model.loss
    assert isinstance(sp, custom.LossSpec) and (sp.kind, sp.regularize, sp.scale) == (1, 1, 1.0)
    assert np.array_equal(sp.mean, climo[0]) and np.allclose(sp.row_weights, custom.latitude_weights(lats, 'midlatitude'))
    assert custom.anomaly_correlation_loss(None, regularize_mean='global').regularize == 3
    assert custom.anomaly_correlation_loss(None, regularize_mean='spatial').regularize == 4
    with pytest.raises(AssertionError):
        custom.anomaly_correlation_loss(None, regularize_mean='median')
    # functional graph with shared layers and skips
    x0, y = _skip_model((4, 8, 12))
    m = Model(inputs=x0, outputs=y)
    path = os.path.join(str(tmp_path), 'skip.keras')
    m.save(path)
    from dlwp_amd.serialization import load_model_file
    m2 = load_model_file(path)
    assert [op.kind for op in m2.plan.ops] == [op.kind for op in m.plan.ops]
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), m2.get_weights()))


def test_compute_paths_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    d = _dlwp(time_dim=1)
    d.build_model(cnn2_layers((2, 9, 12), hidden=8), loss='mse')
    with pytest.raises(RuntimeError):
        d.predict(np.zeros((1, 2, 9, 12), np.float32))


def test_compat_shim_registers_reference_module_names():
    import importlib
    import dlwp_amd.compat  # noqa: F401
    from DLWP.model import DLWPNeuralNet as A, DataGenerator as G
    from DLWP.custom import PeriodicPadding2D as PP, EarlyStoppingMin, slice_layer  # noqa: F401
    from DLWP.util import save_model, load_model, get_from_class  # noqa: F401
    from keras.layers import Input, ZeroPadding2D, Conv2D, MaxPooling2D, UpSampling2D, concatenate  # noqa: F401
    from keras.models import Model as KM
    from keras.callbacks import History  # noqa: F401
    from keras.losses import mean_squared_error
    assert A is DLWPNeuralNet and G is DataGenerator and PP is custom.PeriodicPadding2D and KM is Model
    d = A(is_convolutional=True, time_dim=1, scaler_type=None, scale_targets=False)
    d.build_model(cnn2_layers((2, 9, 12), hidden=8), loss=mean_squared_error, optimizer='adam', metrics=['mae'])
    assert d.model.metrics_names == ['loss', 'mean_absolute_error']
    assert importlib.import_module('DLWP.model.models').DLWPFunctional is DLWPFunctional


def test_recurrent_front_end_lowers_to_convs_and_gate_updates():
    """examples/train.py:142-157: PeriodicPadding3D + ZeroPadding3D + ConvLSTM2D + Reshape.  Both 3-D pads become the
    fused halo of the input convolution; each time step is input conv (+ recurrent 'same' conv) + one gate kernel."""
    from tests.nets import lstm_unet_layers
    cs = (2, 3, 16, 24)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(lstm_unet_layers(cs, widths=(8, 16, 32, 16, 8)), loss='mse', optimizer='adam')
    plan = d.model.plan
    ops5 = plan.ops[:5]
    assert [op.kind for op in ops5] == ['conv', 'lstm', 'conv', 'conv', 'lstm']
    xc0, g0, xc1, hc1, g1 = ops5
    assert xc0.src == P.STATE_IN and (xc0.in_c_off, xc0.xs[0], xc0.in_c_total) == (0, 3, 6) and xc1.in_c_off == 3
    assert tuple(xc0.halo) == (2, 2, 2, 2, P.PAD_ZERO, P.PAD_WRAP) and xc0.layer.dilation_rate == (2, 2)
    assert tuple(hc1.halo) == (1, 1, 1, 1, P.PAD_ZERO, P.PAD_ZERO) and hc1.layer.dilation_rate == (1, 1)
    assert hc1.src == g0.dst and (hc1.in_c_off, hc1.xs[0], hc1.in_c_total) == (0, 12, 24)
    assert g0.aux[0] is None and g0.aux[1] is None and g1.aux[1] == g0.aux[2] and g1.out_c_off == 12
    lstm = d.model.layers[2]
    assert [tuple(w.shape) for w in lstm.weights] == [(3, 3, 3, 48), (3, 3, 12, 48), (48,)]
    b = lstm.get_weights()[2]
    assert np.all(b[12:24] == 1) and b.sum() == 12                      # unit_forget_bias
    r = lstm.get_weights()[1].reshape(-1, 48)
    assert np.allclose(r.T @ r, np.eye(48), atol=1e-5)                  # orthogonal recurrent initialiser
    assert d.model.output_shape == (None,) + cs
    with pytest.raises(NotImplementedError, match='first'):
        d2 = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
        d2.build_model((('PeriodicPadding3D', ((1, 0, 2),), dict(CF, input_shape=cs)),
                        ('ConvLSTM2D', (4, 3), dict(CF, return_sequences=True))), loss='mse')


def test_recurrent_model_file_round_trip(tmp_path):
    """util.save_model / load_model (reference DLWP/util.py:126-174) keep the ConvLSTM2D front end: layer configs, l2
    regulariser, Keras-ordered weights (kernel, recurrent_kernel, bias)."""
    from dlwp_amd.regularizers import l2
    from tests.nets import lstm_unet_layers
    cs = (2, 3, 16, 24)
    layers = list(lstm_unet_layers(cs, widths=(8, 16, 32, 16, 8)))
    layers[2] = (layers[2][0], layers[2][1], dict(layers[2][2], kernel_regularizer=l2(1e-4)))
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(tuple(layers), loss='mse', optimizer='adam')
    util.save_model(d, str(tmp_path / 'lstm'))
    d2 = util.load_model(str(tmp_path / 'lstm'))
    assert d2.is_recurrent and d2.model.output_shape == (None,) + cs
    assert [op.kind for op in d2.model.plan.ops] == [op.kind for op in d.model.plan.ops]
    lstm = [lay for lay in d2.model.layers if isinstance(lay, L.ConvLSTM2D)][0]
    assert lstm.kernel_regularizer.l2 == 1e-4 and lstm.return_sequences and lstm.dilation_rate == (2, 2)
    assert all(np.array_equal(a, b) for a, b in zip(d.model.get_weights(), d2.model.get_weights()))


def test_inference_plan_moves_the_pooling_into_the_producers():
    """predict / predict_timeseries run `model.infer_plan`: a MaxPooling2D(2) that is the only consumer of a convolution is
    applied in that convolution's epilogue, so the reference's 22 layers are 6 launches again and the pre-pooling tensors
    are never written; the training plan keeps them (the backward pass needs them)."""
    d = _dlwp(time_dim=2)
    d.build_model(unet_layers((4, 88, 180)), loss='mse', optimizer='adam', gpus=1)
    ip = d.model.infer_plan
    # 6 convolutions; the 5x5 output layer reads an up-sampled tensor and is restated on its low-resolution source:
    # derived kernels ('phasew', once per graph launch) + a 3x3 convolution with 4 x 4 phase channels whose epilogue stores
    # them interleaved (dlwp_conv2d.out_d2s: no depth-to-space pass)
    assert [op.kind for op in ip.ops] == ['conv'] * 5 + ['phasew', 'conv']
    convs = [op for op in ip.ops if op.kind == 'conv']
    assert [op.out_pool for op in convs] == [True, True, False, False, False, False]
    assert ip.buffers[0] == (32, 44, 90) and ip.buffers[1] == (64, 22, 45)
    # layer 5 (3x3, dilation 2 on the up-sampled 44x90 tensor) == UpSampling2D(3x3, dilation 1, on the 44x90 tensor)
    assert convs[4].conv_geometry == (32, (3, 3), (1, 1)) and convs[4].xs == (64, 44, 90) and convs[4].src_mode == 0
    assert tuple(convs[4].halo)[:4] == (1, 1, 1, 1) and convs[4].out_shape == (32, 44, 90)
    # layer 6 (5x5 on the up-sampled output of layer 5): 3x3 phase kernels, 16 = 4 phases x 4 channels
    assert convs[5].conv_geometry == (16, (3, 3), (1, 1)) and convs[5].xs == (32, 44, 90) and convs[5].wparam == 0
    assert tuple(convs[5].halo)[:4] == (1, 1, 1, 1) and ip.ops[-1].out_shape == (4, 88, 180) and convs[5].out_d2s
    assert convs[5].conv_out_shape == (16, 44, 90) and [op.kind for op in d.model.plan.ops][-2:] == ['conv', 'd2s']
    # the ALGORITHMIC count is the reference graph's, whatever is executed
    assert ip.conv_flops_per_sample() == d.model.plan.conv_flops_per_sample() == 1597685760
    assert not any(op.out_pool for op in d.model.plan.ops) and len(d.model.plan.ops) == 10
    assert d.model.executor.plan is ip and d.model.train_executor.plan is d.model.plan


def test_remaining_reference_callbacks():
    """RunHistory (custom.py:71-91), Adam / SGD learning-rate trackers (custom.py:32-51)."""
    logged = []
    run = types.SimpleNamespace(log=lambda k, v: logged.append((k, v)))
    h = custom.RunHistory(run)
    h.on_train_begin()
    h.on_epoch_end(0, {'loss': 2.0, 'val_loss': 3.0})
    h.on_epoch_end(1, {'loss': 1.0})
    assert h.history == {'loss': [2.0, 1.0], 'val_loss': [3.0]} and h.epoch == [0, 1]
    assert logged == [('loss', 2.0), ('val_loss', 3.0), ('loss', 1.0)]
    opt = types.SimpleNamespace(lr=1e-3, decay=0.01, iterations=100)
    t = custom.AdamLearningRateTracker()
    t.model = types.SimpleNamespace(optimizer=opt)
    t.on_epoch_end(0)
    want = (1e-3 / (1 + 0.01 * 100)) * np.sqrt(1 - 0.999 ** 101) / (1 - 0.9 ** 101)
    assert t.last_lr == pytest.approx(want, rel=1e-12)
    s = custom.SGDLearningRateTracker()
    s.model = t.model
    s.on_epoch_end(0)
    assert s.last_lr == pytest.approx(5e-4, rel=1e-12)
    assert util.get_from_class('DLWP.custom', 'RunHistory') is custom.RunHistory


def test_tf_padding3d_lowers_to_a_mirror_halo_of_the_recurrent_front_end():
    """TFPadding3D (reference custom.py:602-672) in front of ConvLSTM2D: a REFLECT halo of the (T*C, H, W) store, fused into
    the input convolution's loader like PeriodicPadding3D."""
    x0 = L.Input(shape=(2, 3, 8, 12))
    y = L.ConvLSTM2D(4, 3, padding='valid', return_sequences=True, **CF)(
        custom.TFPadding3D((0, 1, 1), mode='REFLECT', **CF)(x0))
    m = Model(inputs=x0, outputs=y)
    convs = [op for op in m.plan.ops if op.kind == 'conv']
    assert tuple(convs[0].halo) == (1, 1, 1, 1, 3, 3) and m.output_shape == (None, 2, 4, 8, 12)
    assert not any(op.kind == 'pad' for op in m.plan.ops)
    with pytest.raises(ValueError):
        custom.TFPadding3D((0, 1, 1), mode='WRAP', **CF)
    with pytest.raises(NotImplementedError):
        Model(inputs=x0, outputs=custom.TFPadding3D((1, 0, 0), **CF)(x0))       # padding of the first (channel) axis


def test_output_written_in_place_behind_a_reshape_is_what_a_later_branch_reads():
    """a convolution behind a Reshape writes the model's output slot itself; a second branch lowered AFTER that
    redirect must read the slot, not the abandoned scratch buffer."""
    x0 = L.Input(shape=(2, 8, 12))
    a = L.Conv2D(3, 3, padding='same', activation='tanh', **CF)(x0)
    out0 = L.Reshape((1, 3, 8, 12))(a)
    out1 = L.Conv2D(2, 3, padding='same', **CF)(a)
    m = Model(inputs=x0, outputs=[out0, out1])
    for plan in (m.plan, m.infer_plan):
        convs = [op for op in plan.ops if op.kind == 'conv']
        assert [op.kind for op in plan.ops] == ['conv', 'conv']
        assert convs[0].dst == P.OUT(0) and convs[1].src == P.OUT(0) and convs[1].dst == P.OUT(1)


def test_member_chain_rule_and_its_overrides(monkeypatch):
    """Executor.member_groups: one chain below 0.5 M grid points per launch, two from there on, always a divisor of the member
    count; DLWP_ROLLOUT_GROUPS=<g> asks for g chains, 'split' (two graphs on two probed streams, Executor.make_rollout) leaves the
    rule alone.  The range in which make_rollout measures instead lies below the rule's threshold."""
    from dlwp_amd.engine import Executor
    monkeypatch.delenv('DLWP_ROLLOUT_GROUPS', raising=False)
    assert Executor.member_groups(8, 88 * 180) == 1 and Executor.member_groups(32, 88 * 180) == 2
    assert Executor.member_groups(4, 180 * 360) == 1 and Executor.member_groups(8, 180 * 360) == 2
    assert Executor.member_groups(33, 88 * 180) == 1            # odd: no two equal chains
    monkeypatch.setenv('DLWP_ROLLOUT_GROUPS', '4')
    assert Executor.member_groups(8, 88 * 180) == 4 and Executor.member_groups(6, 88 * 180) == 3
    monkeypatch.setenv('DLWP_ROLLOUT_GROUPS', 'split')
    assert Executor.member_groups(8, 88 * 180) == 1 and Executor.member_groups(32, 88 * 180) == 2
    lo, hi = Executor.tune_groups_between
    assert 0 < lo < hi <= 500000
