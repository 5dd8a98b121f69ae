"""RowConnected2D / row_conv2d (reference DLWP/custom.py:695-896) through the C ABI (dlwp_rowconv2d_*): forward, data and
weight gradient against the float64 oracle (oracle/np_ref.py: row_connected2d*, pinned against the reference's own build() /
call() / row_conv2d by tests/golden/row_connected.npz), the golden vectors themselves, and the matrix-core kernels against
the library's one-thread-per-output kernels at the model's full size.  fp32: within 1e-5 of the output scale."""
import numpy as np
import pytest
import torch

from oracle import np_ref

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from dlwp_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def close(got, ref, what):
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(np.asarray(got, dtype=np.float64) - ref).max() / scale
    assert err <= RTOL, '%s: max error %.3g of the output scale' % (what, err)


# (n, cin, h, w, cout, (kh, kw), (top, bottom, left, right), (mode_h, mode_w), activation, bias)
CASES = [
    (3, 32, 16, 36, 4, (5, 5), (2, 2, 2, 2), (0, 1), 'linear', True),    # the call-site form: zero rows, periodic columns
    (2, 32, 12, 40, 12, (5, 5), (2, 2, 2, 2), (0, 1), 'linear', True),   # 12 fields (config 4 / 5): unpacked columns
    (5, 8, 9, 21, 2, (3, 3), (1, 1, 1, 1), (2, 1), 'tanh', True),        # 8 pixels per instruction row, edge rows
    (2, 5, 10, 17, 1, (3, 5), (0, 0, 0, 0), (0, 0), 'linear', False),    # one field, no halo, cin not a multiple of 4
    (1, 16, 8, 150, 20, (3, 3), (1, 1, 1, 1), (3, 4), 'relu', True),     # two cout groups, mirror halos, ragged column blocks
    (4, 12, 7, 33, 7, (5, 3), (2, 2, 1, 1), (1, 1), 'tanh', True),       # cout 7 -> 8 columns x 2 pixels
    (2, 40, 11, 70, 3, (5, 5), (3, 1, 0, 4), (0, 1), 'linear', True),    # asymmetric halo
    (1, 4, 5, 9, 12, (5, 3), (0, 0, 0, 0), (0, 0), 'linear', True),      # a single output row
]


def _setup(ops, case, seed):
    n, cin, h, w, cout, (kh, kw), pads, modes, act, use_bias = case
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    t, b, l, r = pads
    ho, wo = h + t + b - kh + 1, w + l + r - kw + 1
    k, bias = np_ref.init_row_connected_weights(ho, (kh, kw), cin, cout, rng, bias_scale=0.5)
    k = (k * 4).astype(np.float32)
    cd = ops.make_conv(cout, kh, kw, 1, ops.make_pad(t, b, l, r, modes[0], modes[1]),
                       {'linear': ops.ACT_LINEAR, 'tanh': ops.ACT_TANH, 'relu': ops.ACT_RELU}[act])
    xp = np_ref.pad2d_modes(x.astype(np.float64), pads, modes[0], modes[1])
    return x, xp, k, (bias if use_bias else None), cd, (ho, wo)


@pytest.mark.parametrize('ci', range(len(CASES)))
def test_forward_matches_the_oracle(ops, ci):
    case = CASES[ci]
    x, xp, k, bias, cd, _ = _setup(ops, case, 100 + ci)
    ref = np_ref.row_connected2d(xp, k, bias, case[8])
    for direct in (False, True):
        y = ops.rowconv2d(dev(x), dev(k), dev(bias) if bias is not None else None, cd, direct=direct)
        close(host(y), ref, 'case %d direct=%s' % (ci, direct))


def test_forward_golden_vectors_of_the_reference(ops, golden):
    """RowConnected2D.build / call / row_conv2d of the reference itself (oracle/make_golden.py) at stride 1."""
    g = golden('row_connected')
    seen = 0
    for i in range(int(g['n'])):
        if tuple(g['%d_strides' % i]) != (1, 1):
            continue
        x, k = g['%d_x' % i], g['%d_kernel' % i]
        bias = g['%d_bias' % i] if '%d_bias' % i in g.files else None
        rows, kh, kw, cin, cout = k.shape
        act = {'linear': ops.ACT_LINEAR, 'tanh': ops.ACT_TANH}[str(g['%d_act' % i])]
        cd = ops.make_conv(cout, kh, kw, 1, None, act)
        for direct in (False, True):
            y = ops.rowconv2d(dev(x), dev(k), dev(bias) if bias is not None else None, cd, direct=direct)
            assert tuple(y.shape[2:]) == tuple(g['%d_out_rc' % i])
            close(host(y), g['%d_y' % i], 'golden %d direct=%s' % (i, direct))
        seen += 1
    assert seen >= 4


def test_upsampled_source_in_the_loaders(ops):
    """keras UpSampling2D(2) in front (the decoder's output layer reads an up-sampled tensor): forward and weight gradient read
    the stored low-resolution tensor; same numbers as on the materialised up-sampled one."""
    from dlwp_amd._lib import Shape4
    rng = np.random.default_rng(11)
    for (n, cin, h, w, cout, k) in ((3, 32, 11, 18, 4, 5), (2, 8, 6, 45, 12, 3)):
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        pad = k // 2
        ho, wo = 2 * h, 2 * w
        kern, b = np_ref.init_row_connected_weights(ho, (k, k), cin, cout, rng, bias_scale=0.3)
        cd = ops.make_conv(cout, k, k, 1, ops.make_pad(pad, pad, pad, pad, 0, 1), ops.ACT_LINEAR, src_mode=ops.SRC_UPSAMPLE2)
        xup = np_ref.upsample2(x.astype(np.float64))
        ref = np_ref.row_connected2d(np_ref.pad2d_modes(xup, (pad,) * 4, 0, 1), kern, b)
        for direct in (False, True):
            close(host(ops.rowconv2d(dev(x), dev(kern), dev(b), cd, direct=direct)), ref, 'upsampled forward direct=%s' % direct)
        dz = rng.standard_normal((n, cout, ho, wo)).astype(np.float32)
        _, dk, db = np_ref.row_connected2d_grads(np_ref.pad2d_modes(xup, (pad,) * 4, 0, 1), kern, dz)
        dw, dbt = torch.empty(kern.shape, device='cuda'), torch.empty((ho, 1, cout), device='cuda')
        ops.rowconv2d_bwd_weight(dev(x), dev(dz), dw, dbt, cd, Shape4(n, cin, h, w))
        close(host(dw), dk, 'upsampled dw')
        close(host(dbt), db, 'upsampled db')


def test_channel_windows(ops):
    """slice_layer in front (input channel window) and concatenate behind (output channel window), as for Conv2D."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 24, 10, 20)).astype(np.float32)
    k, b = np_ref.init_row_connected_weights(10, (3, 3), 16, 4, rng, bias_scale=0.3)
    cd = ops.make_conv(4, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR, in_c_off=5, in_c_total=24, out_c_off=2,
                       out_c_total=9)
    out = torch.full((2, 9, 10, 20), 7.0, device='cuda')
    ops.rowconv2d(dev(x), dev(k), dev(b), cd, out=out, x_channels=16)
    ref = np_ref.row_connected2d(np_ref.pad2d_modes(x[:, 5:21].astype(np.float64), (1, 1, 1, 1), 0, 1), k, b)
    got = host(out)
    close(got[:, 2:6], ref, 'window')
    assert (got[:, :2] == 7.0).all() and (got[:, 6:] == 7.0).all()


@pytest.mark.parametrize('ci', range(len(CASES)))
def test_gradients_match_the_oracle(ops, ci):
    from dlwp_amd._lib import Shape4
    case = CASES[ci]
    n, cin, h, w, cout, (kh, kw), pads, modes, act, use_bias = case
    x, xp, k, bias, cd, (ho, wo) = _setup(ops, case, 200 + ci)
    rng = np.random.default_rng(300 + ci)
    dz = rng.standard_normal((n, cout, ho, wo)).astype(np.float32)
    dxp, dk, db = np_ref.row_connected2d_grads(xp, k, dz)
    dx_ref = np_ref.pad2d_modes_grad(dxp, x.shape, pads, modes[0], modes[1])
    xs = Shape4(n, cin, h, w)
    dx = torch.empty((n, cin, h, w), device='cuda')
    ops.rowconv2d_bwd_data(dev(dz), dev(k), cd, xs, dx)
    close(host(dx), dx_ref, 'dx case %d' % ci)
    dw = torch.empty(k.shape, device='cuda')
    dbt = torch.empty((ho, 1, cout), device='cuda')
    ops.rowconv2d_bwd_weight(dev(x), dev(dz), dw, dbt, cd, xs)
    close(host(dw), dk, 'dw case %d' % ci)
    close(host(dbt), db, 'db case %d' % ci)
    # accumulate adds; a second run is bit-identical (fixed summation order)
    dw2, db2 = dw.clone(), dbt.clone()
    ops.rowconv2d_bwd_weight(dev(x), dev(dz), dw2, db2, cd, xs, accumulate=True)
    close(host(dw2), 2 * dk, 'dw accumulate case %d' % ci)
    close(host(db2), 2 * db, 'db accumulate case %d' % ci)
    dw3 = torch.empty_like(dw)
    ops.rowconv2d_bwd_weight(dev(x), dev(dz), dw3, None, cd, xs)
    assert torch.equal(dw3, dw)


def test_full_size_layer_runs_on_the_matrix_cores_and_agrees_with_the_vector_kernels(ops):
    """The call-site layer at the model's own size (32 -> 4 fields, 5x5, 88x180 behind the periodic / zero halo; 32 -> 12 at
    180x360 for the 1-degree configurations): all three passes take the MFMA route, and agree with the library's
    one-thread-per-output forward / with linearity-derived checks of the gradients."""
    from dlwp_amd._lib import Shape4
    for (n, h, w, cout) in ((8, 88, 180, 4), (2, 180, 360, 12)):
        rng = np.random.default_rng(h)
        cin = 32
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        k, b = np_ref.init_row_connected_weights(h, (5, 5), cin, cout, rng, bias_scale=0.1)
        cd = ops.make_conv(cout, 5, 5, 1, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_LINEAR)
        for which in (0, 1, 2):
            assert ops.rowconv2d_uses_matrix_cores((n, cin, h, w), cd, which), which
        xd, kd, bd = dev(x), dev(k), dev(b)
        y = ops.rowconv2d(xd, kd, bd, cd)
        yd = ops.rowconv2d(xd, kd, bd, cd, direct=True)
        close(host(y), host(yd), 'forward %dx%d' % (h, w))
        # one sample against the float64 oracle
        ref = np_ref.row_connected2d(np_ref.pad2d_modes(x[:1].astype(np.float64), (2, 2, 2, 2), 0, 1), k, b)
        close(host(y)[:1], ref, 'forward vs oracle %dx%d' % (h, w))
        # adjoint identities: <dz, J dx> = <J^T dz, dx> for the data path, <dz, y(k') - bias> = <dW, k'> for the weights
        dz = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        xs = Shape4(n, cin, h, w)
        dzd = dev(dz)
        dx = torch.empty_like(xd)
        ops.rowconv2d_bwd_data(dzd, kd, cd, xs, dx)
        x2 = dev(rng.standard_normal(x.shape))
        y2 = ops.rowconv2d(x2, kd, None, cd)
        lhs = float((y2.double() * dzd.double()).sum())
        rhs = float((dx.double() * x2.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (lhs, rhs)
        dw = torch.empty_like(kd)
        db = torch.empty_like(bd)
        ops.rowconv2d_bwd_weight(xd, dzd, dw, db, cd, xs)
        k2 = dev(rng.standard_normal(k.shape) * 0.05)
        y3 = ops.rowconv2d(xd, k2, None, cd)
        lhs = float((y3.double() * dzd.double()).sum())
        rhs = float((dw.double() * k2.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (lhs, rhs)
        close(host(db).reshape(cout, h), dz.astype(np.float64).sum(axis=(0, 3)), 'db %dx%d' % (h, w))


def test_errors(ops):
    from dlwp_amd._lib import DlwpError
    x = torch.zeros((1, 4, 8, 8), device='cuda')
    k = torch.zeros((6, 3, 3, 4, 2), device='cuda')
    with pytest.raises(DlwpError):      # dilation 2: not a row-connected layer
        ops.rowconv2d(x, torch.zeros((4, 3, 3, 4, 2), device='cuda'), None, ops.make_conv(2, 3, 3, 2))
    with pytest.raises(ValueError):     # kernel rows != output rows
        ops.rowconv2d(x, torch.zeros((5, 3, 3, 4, 2), device='cuda'), None, ops.make_conv(2, 3, 3, 1))
    with pytest.raises(ValueError):     # bfloat16 storage is not offered for this layer
        ops.rowconv2d(x.to(torch.bfloat16), k, None, ops.make_conv(2, 3, 3, 1))
    y = ops.rowconv2d(torch.zeros((0, 4, 8, 8), device='cuda'), k, None, ops.make_conv(2, 3, 3, 1))    # empty batch
    assert tuple(y.shape) == (0, 2, 6, 6)


# ----------------------------------------------------------------------------------------------------------------- #
# the layer inside a model: DLWPNeuralNet.build_model by name (DLWP.custom registry), predict, hipGraph rollout, fit
# ----------------------------------------------------------------------------------------------------------------- #
CF = {'data_format': 'channels_first'}


def _row_net(cs, hidden=16):
    """A small stack in the reference's layer-triple form whose last layer is RowConnected2D behind the periodic / zero halo
    (examples/train_functional.py:191-196, 275)."""
    return (
        ('PeriodicPadding2D', ((0, 1),), dict(CF, input_shape=cs)),
        ('ZeroPadding2D', ((1, 0),), CF),
        ('Conv2D', (hidden, 3), dict(CF, activation='tanh')),
        ('MaxPooling2D', (2,), CF),
        ('PeriodicPadding2D', ((0, 1),), CF),
        ('ZeroPadding2D', ((1, 0),), CF),
        ('Conv2D', (hidden, 3), dict(CF, activation='tanh')),
        ('UpSampling2D', (2,), CF),
        ('PeriodicPadding2D', ((0, 2),), CF),
        ('ZeroPadding2D', ((2, 0),), CF),
        ('RowConnected2D', (cs[0], 5), dict(CF, padding='valid', activation='linear')),
    )


def _build_row_model(cs, seed=0):
    from dlwp_amd.model import DLWPNeuralNet
    np.random.seed(seed)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(_row_net(cs), loss='mse', optimizer='adam', metrics=['mae'])
    rng = np.random.default_rng(seed + 1)
    ws = d.model.get_weights()
    ws = [w if i % 2 == 0 else (0.2 * rng.standard_normal(w.shape)).astype(np.float32)
          for i, w in enumerate(ws)]          # (kernel, bias) pairs: biases randomised so that their layout is tested
    d.model.set_weights(ws)
    pairs = [(ws[i], ws[i + 1]) for i in range(0, len(ws), 2)]
    return d, pairs


def test_model_with_a_row_connected_output_layer_matches_the_oracle():
    cs = (4, 12, 20)
    d, pairs = _build_row_model(cs)
    assert [op.kind for op in d.model.infer_plan.ops][-1] == 'rowconv'
    assert pairs[-1][0].shape == (12, 5, 5, 16, 4) and pairs[-1][1].shape == (12, 1, 4)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((5,) + cs).astype(np.float32)
    ref = np_ref.run_layers(_row_net(cs), x, pairs)
    got = d.predict(x)
    assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)
    # the hipGraph rollout replays the same launches: bit-identical to the step-by-step host loop
    series = d.predict_timeseries(x, 6)
    p = x
    for t in range(3):
        p = d.predict(p)
        assert np.array_equal(series[2 * t:2 * t + 2].transpose(1, 0, 2, 3, 4).reshape(p.shape), p), t


def test_training_through_the_row_connected_layer_matches_autograd():
    from oracle import torch_ref
    cs = (4, 12, 20)
    d, pairs = _build_row_model(cs, seed=3)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((6,) + cs).astype(np.float32)
    y = rng.standard_normal((6,) + cs).astype(np.float32)
    tw = torch_ref.to_torch_weights(pairs, dtype=torch.float64, requires_grad=True)
    out = torch_ref.run_layers(_row_net(cs), torch.tensor(x, dtype=torch.float64), tw)
    yt = torch.tensor(y, dtype=torch.float64)
    loss = ((out - yt) ** 2).mean()
    loss.backward()
    grads_ref = []
    for (w, b) in tw:
        gw = w.grad.numpy()
        grads_ref += [gw if gw.ndim == 5 else gw.transpose(2, 3, 1, 0), b.grad.numpy()]
    vals = d.model.train_on_batch(x, y)
    assert vals[0] == pytest.approx(float(loss.detach()), rel=2e-5)
    torch.cuda.synchronize()
    tr, off = d.model._trainer, 0
    for g_ref in grads_ref:
        g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
        off += g_ref.size
        assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape
    # and the loss goes down over a few steps
    l0 = d.model.test_on_batch(x, y)[0]
    for _ in range(10):
        d.model.train_on_batch(x, y)
    assert d.model.test_on_batch(x, y)[0] < l0


def test_row_conv2d_function_on_device_tensors(golden):
    """DLWP.custom.row_conv2d's functional form, both data formats, against the reference's golden outputs."""
    from dlwp_amd import custom
    g = golden('row_connected')
    i = 0
    x, k = g['%d_x' % i], g['%d_kernel' % i]
    out_rc = tuple(int(v) for v in g['%d_out_rc' % i])
    y = custom.row_conv2d(dev(x.transpose(0, 2, 3, 1)), dev(k), (5, 5), (1, 1), out_rc, 'channels_last')
    close(host(y), g['%d_y_cl' % i], 'channels_last')
    with pytest.raises(NotImplementedError):
        custom.row_conv2d(dev(x), dev(k), (5, 5), (2, 2), out_rc, 'channels_first')


def test_bfloat16_activation_mode_keeps_the_row_connected_layer_in_float32():
    """Model.set_activation_dtype('bfloat16') (BASELINE config 4): the buffers the row kernels touch stay float32; the rest of
    the stack stores bfloat16 -- the forecast moves by bf16 rounding only, and the rollout graph still equals the host loop."""
    cs = (4, 12, 20)
    d, pairs = _build_row_model(cs, seed=5)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((4,) + cs).astype(np.float32)
    y32 = d.predict(x)
    d.model.set_activation_dtype('bfloat16')
    plan = d.model.infer_plan
    rc = [op for op in plan.ops if op.kind == 'rowconv'][0]
    assert rc.src not in d.model.executor._bf16 and len(d.model.executor._bf16) >= 1
    y16 = d.predict(x)
    assert np.isfinite(y16).all()
    assert np.abs(y16 - y32).max() <= 3e-2 * max(np.abs(y32).max(), 1.0)
    assert np.abs(y16 - y32).max() > 0          # (it really ran in the other storage mode)
    series = d.predict_timeseries(x, 4)
    p = d.predict(x)
    assert np.array_equal(series[0:2].transpose(1, 0, 2, 3, 4).reshape(p.shape), p)
    d.model.set_activation_dtype('float32')
    assert np.array_equal(d.predict(x), y32)


def test_functional_skip_unet_with_the_latitude_dependent_output_layer():
    """examples/train_functional.py with latitude_dependent = True (:53, 191-196) and skip_connections = True (:248-275): the
    RowConnected2D output layer reads the concatenation of the decoder and the first skip; DLWPFunctional with two chained
    outputs (integration_steps = 2: the SAME layer objects applied twice), predict against the oracle, the rollout against the
    host loop, and training through both applications of the shared row-connected kernel."""
    from dlwp_amd import custom, layers as L
    from dlwp_amd.engine import Model
    from dlwp_amd.model import DLWPFunctional
    rng = np.random.default_rng(15)
    cs = (4, 16, 24)
    x0 = L.Input(shape=cs)
    pp2, zp2 = custom.PeriodicPadding2D((0, 2), **CF), L.ZeroPadding2D((2, 0), **CF)
    pp1, zp1 = custom.PeriodicPadding2D((0, 1), **CF), L.ZeroPadding2D((1, 0), **CF)
    pool, up = L.MaxPooling2D(2, **CF), L.UpSampling2D(2, **CF)
    c1 = L.Conv2D(32, 3, dilation_rate=2, activation='tanh', **CF)
    c2 = L.Conv2D(32, 3, activation='tanh', **CF)
    c5 = L.Conv2D(16, 3, dilation_rate=2, activation='tanh', **CF)
    row = custom.RowConnected2D(cs[0], 5, padding='valid', activation='linear', **CF)
    s11, s12 = custom.slice_layer(0, 16, axis=1), custom.slice_layer(16, 32, axis=1)

    def net(x):
        x = c1(pp2(zp2(x)))
        x, x1 = s11(x), s12(x)
        x = c2(pp1(zp1(pool(x))))
        x = c5(pp2(zp2(up(x))))
        x = L.concatenate([x, x1], axis=1)
        return row(pp2(zp2(x)))
    outs = [net(x0)]
    outs.append(net(outs[0]))
    np.random.seed(15)
    m = Model(inputs=x0, outputs=outs)
    f = DLWPFunctional(is_convolutional=True, time_dim=2)
    f.build_model(m, loss='mse', loss_weights=[0.5, 0.5], optimizer='adam', metrics=['mae'])
    ws = m.get_weights()
    assert [w.shape for w in ws][-2:] == [(16, 5, 5, 32, 4), (16, 1, 4)]
    ws = [w if i % 2 == 0 else (0.1 * rng.standard_normal(w.shape)).astype(np.float32) for i, w in enumerate(ws)]
    m.set_weights(ws)
    p1, p2, p5, pr = [(ws[i], ws[i + 1]) for i in range(0, 8, 2)]

    def ref(x):
        def halo(t, k):
            return np_ref.zero_padding2d(np_ref.periodic_padding2d(t, (0, k)), (k, 0))
        a = np_ref.conv2d(halo(x, 2), *p1, 2, 'tanh')
        a, a1 = a[:, :16], a[:, 16:]
        b = np_ref.conv2d(halo(np_ref.maxpool2(a), 1), *p2, 1, 'tanh')
        g = np_ref.conv2d(halo(np_ref.upsample2(b), 2), *p5, 2, 'tanh')
        return np_ref.row_connected2d(halo(np.concatenate([g, a1], axis=1), 2), *pr)
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    y1, y2 = f.predict(x)
    r1 = ref(x.astype(np.float64))
    assert np.abs(y1 - r1).max() <= 2e-5 * max(np.abs(r1).max(), 1.0)
    assert np.abs(y2 - ref(r1)).max() <= 8e-5 * max(np.abs(r1).max(), 1.0)
    ts = f.predict_timeseries(x, 5)
    p, slots = x, []
    for _ in range(2):
        o1, o2 = f.predict(p)
        slots += [o1, o2]
        p = o2
    assert np.array_equal(ts, np_ref._merge_time(np.stack(slots), 4, 3, 2, cs, False))
    # training: the shared row-connected kernel collects the gradients of both applications
    yt = [rng.standard_normal((3,) + cs).astype(np.float32) for _ in range(2)]
    l0 = m.test_on_batch(x, yt)[0]
    for _ in range(12):
        m.train_on_batch(x, yt)
    assert m.test_on_batch(x, yt)[0] < l0


def test_random_geometries_against_the_oracle(ops):
    """Seeded sweep over the planners' corner cases: kernel sizes 1 ... 5 x 1 ... 7, 1 ... 40 output fields (every packing
    factor, several 16-channel groups), channel counts that are not multiples of 4, widths from one fragment to several column
    blocks, batches that do not fill the sample groups, every halo mode, asymmetric halos -- forward and all three gradients."""
    from dlwp_amd._lib import Shape4
    rng = np.random.default_rng(2024)
    n_mfma = 0
    for trial in range(30):
        kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 8))
        if trial >= 28:                 # taller than the matrix-core kernels unroll: the vector-ALU kernels of all three passes
            kh = 6 + (trial - 28)
        cout = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 33, 40]))
        cin = int(rng.choice([1, 3, 4, 6, 8, 13, 24]))
        n = int(rng.integers(1, 6))
        h = int(rng.integers(kh, kh + 9))
        w = int(rng.choice([kw, kw + 3, 17, 33, 70, 131]))
        mh, mw = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        lim_h = h - 1 if mh == 3 else h
        lim_w = w - 1 if mw == 3 else w
        t, b = (int(rng.integers(0, min(3, lim_h) + 1)) for _ in range(2))
        l, r = (int(rng.integers(0, min(4, lim_w) + 1)) for _ in range(2))
        ho, wo = h + t + b - kh + 1, w + l + r - kw + 1
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        k, bias = np_ref.init_row_connected_weights(ho, (kh, kw), cin, cout, rng, bias_scale=0.5)
        k = (k * 3).astype(np.float32)
        cd = ops.make_conv(cout, kh, kw, 1, ops.make_pad(t, b, l, r, mh, mw), ops.ACT_TANH)
        on_mfma = [ops.rowconv2d_uses_matrix_cores((n, cin, h, w), cd, which) for which in (0, 1, 2)]
        assert kh <= 5 or not (on_mfma[0] or on_mfma[1]), (kh, kw)
        n_mfma += sum(on_mfma)
        what = 'trial %d: k%dx%d cin %d cout %d n %d %dx%d halo %r modes %r' % (trial, kh, kw, cin, cout, n, h, w, (t, b, l, r), (mh, mw))
        xp = np_ref.pad2d_modes(x.astype(np.float64), (t, b, l, r), mh, mw)
        close(host(ops.rowconv2d(dev(x), dev(k), dev(bias), cd)), np_ref.row_connected2d(xp, k, bias, 'tanh'), what)
        dz = rng.standard_normal((n, cout, ho, wo)).astype(np.float32)
        dxp, dk, db = np_ref.row_connected2d_grads(xp, k, dz)
        xs = Shape4(n, cin, h, w)
        dx = torch.empty((n, cin, h, w), device='cuda')
        ops.rowconv2d_bwd_data(dev(dz), dev(k), cd, xs, dx)
        close(host(dx), np_ref.pad2d_modes_grad(dxp, x.shape, (t, b, l, r), mh, mw), what + ' dx')
        dw, dbt = torch.empty(k.shape, device='cuda'), torch.empty((ho, 1, cout), device='cuda')
        ops.rowconv2d_bwd_weight(dev(x), dev(dz), dw, dbt, cd, xs)
        close(host(dw), dk, what + ' dw')
        close(host(dbt), db, what + ' db')
    assert n_mfma >= 70          # (of 90 passes: the rest are the LDS-footprint and kernel-height fall-backs)
