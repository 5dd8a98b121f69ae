"""GPU parity at the model level, through the reference's own API (DLWPNeuralNet / DLWPFunctional build_model, predict,
predict_timeseries, fit, evaluate) with everything below it running in libdlwp_hip.so, against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref
from tests.nets import CF, cnn2_layers, unet_layers

pytestmark = pytest.mark.gpu

#: whole forward vs the float64 oracle, relative to the output scale: BASELINE.md's bar (1e-5).  Measured on the MI355X
#: (gpurun_out/forward_errors.json, written by every GPU test session; r4): 0.7e-6 - 1.4e-6 on the full-size configurations
#: (73x144 two-layer CNN 0.7-0.9e-6, 180x360 recurrent stack 0.8-1.2e-6, 180x360x12 U-Net 0.7-1.4e-6); r3 ran with 2e-5
FWD_TOL = 1e-5


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _build(layers, time_dim=2, seed=0, **compile_kw):
    from dlwp_amd.model import DLWPNeuralNet
    np.random.seed(seed)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=time_dim, scaler_type=None, scale_targets=False)
    compile_kw.setdefault('loss', 'mse')
    compile_kw.setdefault('optimizer', 'adam')
    compile_kw.setdefault('metrics', ['mae'])
    d.build_model(layers, **compile_kw)
    return d


def _weights_of(model, rng=None, bias_scale=0.1):
    """[(w_hwio, b)] of the Conv2D layers in order; biases randomised so they are actually tested."""
    ws = model.get_weights()
    pairs = [(ws[i], ws[i + 1]) for i in range(0, len(ws), 2)]
    if rng is not None:
        pairs = [(w, (bias_scale * rng.standard_normal(b.shape)).astype(np.float32)) for w, b in pairs]
        model.set_weights([a for p in pairs for a in p])
    return pairs


def host_t(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


#: every relative error the forward / rollout parity tests measured in this session: {test id: [values]} -- written to
#: gpurun_out/forward_errors.json when the session ends (tests/conftest.py), so that the tolerance can be read against what the
#: kernels actually deliver
MEASURED = {}


def _rel(a, b):
    v = float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    import os
    MEASURED.setdefault(os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0], []).append(v)
    return v


def _bf16_weight_indices(model, n):
    """Indices (among the weighted layers, the oracle's numbering) of the layers the product multiplies with bf16-rounded
    kernels: bf16-stored input on the bf16 matrix cores."""
    weighted = [lay for lay in model.layers if len(lay._weights)]
    on16 = model.executor.bf16_weight_layers(n)
    return set(i for i, lay in enumerate(weighted) if any(lay is o for o in on16))


def _bf16_lstm_parts(model, n):
    """Which convolutions of the ConvLSTM2D front end run on the bf16 matrix cores: subset of ('kernel', 'recurrent')."""
    names = {'kernel': 'kernel', 'recurrent_kernel': 'recurrent'}
    return tuple(names[o.which] for o in model.executor.bf16_weight_layers(n) if hasattr(o, 'which'))


def test_unet_forward_matches_float64_oracle():
    rng = np.random.default_rng(0)
    cs = (4, 16, 24)
    layers = unet_layers(cs)
    d = _build(layers)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    got = d.predict(x)
    want = np_ref.run_layers(layers, x, weights)
    assert got.shape == want.shape == (3,) + cs and got.dtype == np.float32
    assert _rel(got, want) < FWD_TOL
    # second opinion: the unfused torch-CPU float32 graph agrees to the same level
    want32 = torch_ref.run_layers(layers, torch.from_numpy(x), torch_ref.to_torch_weights(weights)).numpy()
    assert _rel(got, want32) < FWD_TOL


def test_config1_cnn_on_odd_grid_matches_oracle():
    """Config 1 shape class: 73x144-like odd height, 2 channels, two 5x5 convs (examples/train.py plumbing path)."""
    rng = np.random.default_rng(1)
    cs = (2, 19, 36)
    layers = cnn2_layers(cs, hidden=32)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((2,) + cs).astype(np.float32)
    assert _rel(d.predict(x), np_ref.run_layers(layers, x, weights)) < FWD_TOL


def test_predict_does_not_depend_on_batch_chunking_or_batch_mates():
    rng = np.random.default_rng(2)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs))
    _weights_of(d.model, rng)
    x = rng.standard_normal((300,) + cs).astype(np.float32)
    from dlwp_amd import ops
    full = d.predict(x)
    assert np.array_equal(full, d.predict(x, batch_size=256))      # chunked (256 + 44) == one pass, bit for bit
    # a member alone == the member inside a batch, BIT FOR BIT, at every batch size -- the default since r5 (split-K,
    # which divides a long input-channel sum over several workgroups on launches of a handful of samples, is opt-in again:
    # VERDICT r4 weak c)
    assert ops.set_splitk(0) == 0                                    # (the default IS off)
    for lo, hi in ((7, 8), (7, 9), (5, 9), (0, 16)):
        assert np.array_equal(d.predict(x[lo:hi]), full[lo:hi]), (lo, hi)
    # opted in (DLWP_OPT_SPLITK = 1): within one split regime a member's bits do not depend on its batch mates; across regimes
    # they agree to float32 round-off
    prev = ops.set_splitk(1)
    try:
        alone, pair = d.predict(x[7:8]), d.predict(x[7:9])
        assert np.array_equal(alone, pair[:1])                       # 1 and 2 members: the same regime
        assert _rel(alone, full[7:8]) < 2e-6
    finally:
        ops.set_splitk(prev)


def test_predict_timeseries_graph_equals_host_loop_and_tracks_the_oracle():
    rng = np.random.default_rng(3)
    cs = (4, 16, 24)                                                  # time_dim 2 x 2 variables
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((5,) + cs).astype(np.float32)
    steps = 7                                                         # -> ceil(7/2) = 4 forwards, 8 output steps
    got = d.predict_timeseries(x, steps)
    assert got.shape == (8, 5, 2, 16, 24) and got.dtype == np.float32
    kept = d.predict_timeseries(x, steps, keep_time_dim=True)
    assert kept.shape == (4, 5, 2, 2, 16, 24)
    # (1) the captured hipGraph rollout == the reference-style host loop over our own predict(), bit for bit
    def host_loop(p):
        ser = []
        for _ in range(4):
            p = d.predict(p)
            ser.append(p)
        return np.stack(ser)
    ser = host_loop(x)
    assert np.array_equal(kept, ser.reshape(4, 5, 2, 2, 16, 24))
    assert np.array_equal(got, np_ref._merge_time(ser, 4, 5, 2, cs, False))
    # (2) against the float64 oracle rollout: tight on the first forward, bounded growth afterwards
    want = np_ref.predict_timeseries_nn(lambda p: np_ref.run_layers(layers, p, weights), x.astype(np.float64), steps, 2)
    assert _rel(got[:2], want[:2]) < FWD_TOL
    assert _rel(got, want) < FWD_TOL
    # replay: same graph, new initial state
    x2 = rng.standard_normal((5,) + cs).astype(np.float32)
    got2 = d.predict_timeseries(x2, steps)
    assert np.array_equal(got2[:2], np_ref._merge_time(d.predict(x2)[None], 1, 5, 2, cs, False))
    # step_sequence goes through the generic loop and keeps only the first predicted step
    seq = d.predict_timeseries(x, 3, step_sequence=True)
    assert seq.shape == (3, 5, 2, 16, 24)
    assert np.array_equal(seq[0], d.predict(x).reshape(5, 2, 2, 16, 24)[:, 0])


def test_predict_timeseries_pipelined_host_copy_is_bit_identical():
    """A large series goes back to the host slot by slot WHILE the rollout runs (DLWPNeuralNet._rollout_streamed: one graph per
    model call, copy streams, one pinned result array; r5): same bits as the one-graph rollout with a single copy behind it, for
    both output layouts, member chunks in the first call (host predictors) or not (7 members: no even cut; device predictors)."""
    import torch
    rng = np.random.default_rng(5)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs), time_dim=2)
    _weights_of(d.model, rng)
    for n in (8, 7):
        x = rng.standard_normal((n,) + cs).astype(np.float32)
        d.host_stream_bytes = 1 << 60               # never streamed: one graph, one copy
        whole = d.predict_timeseries(x, 6)
        whole_kept = d.predict_timeseries(x, 6, keep_time_dim=True)
        d.host_stream_bytes = 0                     # always streamed
        try:
            for _ in range(2):                  # (the second call reuses the cached graphs, staging buffers and streams)
                assert np.array_equal(d.predict_timeseries(x, 6), whole), n
                assert np.array_equal(d.predict_timeseries(x, 6, keep_time_dim=True), whole_kept), n
            assert np.array_equal(d.predict_timeseries(torch.from_numpy(x).cuda(), 6), whole), n
            sr = d.model.streamed_rollout(n, 3, d.host_head_chunks)
            assert len(sr.head) == (4 if n == 8 else 1) and len(sr.tail) == 2
            dev = d.predict_timeseries(x, 6, return_device=True)
            assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), whole)
        finally:
            d.host_stream_bytes = 64 << 20


@pytest.mark.parametrize('n', [1031, 2101])
def test_full_size_ensemble_beyond_a_thousand_members(n):
    """1031 / 2101 members (odd: one unpaired member after the sample pairs, a ragged last host chunk) of the full 88 x 180
    grid in one rollout: activations of 2.1 / 4.3 GB per layer (past 32-bit byte offsets), per-sample buffer descriptors
    on 64-bit sample bases.  The first, a middle
    and the last members equal the same members run as a small ensemble (other tile configurations are chosen there: equal
    to rounding, not to the bit), device-resident and through the chunked host return."""
    import torch
    rng = np.random.default_rng(77)
    cs = (4, 88, 180)
    d = _build(unet_layers(cs), time_dim=2)
    _weights_of(d.model, rng)
    x = torch.from_numpy(rng.standard_normal((8,) + cs).astype(np.float32)).cuda()
    x = (x.repeat((n + 7) // 8, 1, 1, 1)[:n] * torch.linspace(0.5, 1.5, n, device='cuda')[:, None, None, None]).contiguous()
    dev = d.predict_timeseries(x, 4, return_device=True)
    assert tuple(dev.shape) == (4, n) + (2,) + cs[1:] and bool(torch.isfinite(dev).all())
    for lo, hi in ((0, 5), (511, 518), (n - 7, n)):
        small = d.predict_timeseries(x[lo:hi].contiguous(), 4, return_device=True)
        assert _rel(dev[:, lo:hi].cpu().numpy(), small.cpu().numpy()) < 2e-5, (lo, hi)
    host = d.predict_timeseries(x, 4)                                 # streamed: slot by slot while the rollout runs
    assert isinstance(host, np.ndarray) and np.array_equal(host, dev.cpu().numpy())


def test_headline_ensemble_of_256_members_against_the_oracle_directly():
    """the bench's operating point -- 256 members of the full 88 x 180 grid in one rollout graph, i.e. the
    streaming layer-1 kernel, sample pairs, the 256-member tile choices, the streaming output layer -- checked against the
    float64 oracle DIRECTLY (not through a small-ensemble run): members 0, 131 and 255 after the first forward to the parity
    bar, member 0 after the second forward (its own output fed back) to twice that."""
    import torch
    rng = np.random.default_rng(256)
    cs = (4, 88, 180)
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((256,) + cs).astype(np.float32)
    dev = d.predict_timeseries(torch.from_numpy(x).cuda(), 4, keep_time_dim=True, return_device=True)   # 2 forwards
    got = dev.cpu().numpy().reshape((2, 256) + cs)
    worst = 0.0
    for m in (0, 131, 255):
        want = np_ref.run_layers(layers, x[m:m + 1].astype(np.float64), weights)
        err = _rel(got[0, m:m + 1], want)
        worst = max(worst, err)
        assert err < FWD_TOL, (m, err)
        if m == 0:
            want2 = np_ref.run_layers(layers, want, weights)
            err2 = _rel(got[1, :1], want2)
            assert err2 < FWD_TOL, err2
            MEASURED['headline_256_members_second_forward_member_0'] = [err2]
    MEASURED['headline_256_members_first_forward_members_0_131_255'] = [worst]


def test_member_sharding_is_bit_identical():
    """Ensemble members are independent: a rollout of a shard equals the same rows of the full rollout (what lets the
    8-GPU run be compared member by member with the 1-GPU run)."""
    from dlwp_amd.parallel import shard_bounds
    rng = np.random.default_rng(4)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs), time_dim=2)
    _weights_of(d.model, rng)
    base = rng.standard_normal((1,) + cs).astype(np.float32)
    members = base + 0.01 * rng.standard_normal((8,) + cs).astype(np.float32)
    full = d.predict_timeseries(members, 6)
    for rank in range(4):
        lo, hi = shard_bounds(8, rank, 4)
        assert np.array_equal(d.predict_timeseries(members[lo:hi], 6), full[:, lo:hi])


def test_rollout_captured_as_parallel_member_chains_is_bit_identical():
    """dlwp_rollout_create_grouped: the members split into equal groups captured as parallel graph branches (each chain runs
    the plan on its own members of every buffer) -- same forecasts, bit for bit, as one chain."""
    rng = np.random.default_rng(6)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs), time_dim=2)
    _weights_of(d.model, rng)
    net = d.model
    x = torch.from_numpy(rng.standard_normal((6,) + cs).astype(np.float32)).to(net.device)
    from dlwp_amd import ops
    # (r4) forked graphs never hold split-K launches (csrc/rollout.hip), a single chain of this few members does: the chains equal
    # the single chain of the same -- unsplit -- regime bit for bit, and the split one to float32 round-off
    prev = ops.set_splitk(1)                 # (opt-in since r5)
    try:
        want_split = net.rollout_on_device(x, 5, graph_cache=False).clone()
        ops.set_splitk(0)
        want = net.rollout_on_device(x, 5, graph_cache=False).clone()
    finally:
        ops.set_splitk(prev)
    assert _rel(host_t(want_split[0]), host_t(want[0])) < 2e-6
    for groups in (2, 3, 6):
        s0 = torch.empty_like(x)
        ser = torch.empty_like(want)
        g = net.executor.make_rollout(s0, ser, 5, groups=groups)
        s0.copy_(x)
        g.launch()
        torch.cuda.synchronize()
        assert torch.equal(ser, want), groups
        g.close()
    # r5: a forked graph goes through a stream of the rollout's own only when the caller sits on the null stream (the runtime fault of
    # r4); on a caller's REAL stream it is launched directly (csrc/rollout.hip) -- the same bits either way
    s0 = torch.empty_like(x)
    ser = torch.empty_like(want)
    g = net.executor.make_rollout(s0, ser, 5, groups=2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s0.copy_(x)
        g.launch()
    side.synchronize()
    assert torch.equal(ser, want)
    ser.zero_()
    s0.copy_(x)
    g.launch()                          # ... and from the null stream
    torch.cuda.synchronize()
    assert torch.equal(ser, want)
    g.close()
    from dlwp_amd._lib import DlwpError
    with pytest.raises(DlwpError, match='equal groups'):
        net.executor.make_rollout(torch.empty_like(x), torch.empty_like(want), 5, groups=4)


def test_member_chains_chosen_by_measurement_never_change_a_bit():
    """Executor.make_rollout(groups=None) in the range where one chain is MEASURED against two (members x grid points in
    Executor.tune_groups_between: 16 members of the 88 x 180 grid): whichever it keeps gives the bits of the single chain; outside the
    range, with an odd member count, or with DLWP_ROLLOUT_GROUPS set nothing is measured (groups as member_groups' rule says)."""
    import torch
    from dlwp_amd.engine import Executor
    rng = np.random.default_rng(11)
    cs = (4, 88, 180)
    d = _build(unet_layers(cs), time_dim=2)
    _weights_of(d.model, rng)
    net = d.model
    lo, hi = Executor.tune_groups_between
    n = 16
    assert lo <= n * 88 * 180 < hi
    x = torch.from_numpy(rng.standard_normal((n,) + cs).astype(np.float32)).to(net.device)
    outs = {}
    for groups in (1, None, 'split'):         # 'split': two single-chain graphs of 8 members on two probed streams
        s0 = torch.empty_like(x)
        ser = torch.empty((3, n) + cs, device=net.device)
        g = net.executor.make_rollout(s0, ser, 3, groups=groups)
        assert g.groups in (1, 2, 'split') if groups is None else g.groups == groups
        s0.copy_(x)
        g.launch()
        torch.cuda.synchronize()
        outs[groups] = ser.clone()
        g.close()
    assert torch.equal(outs[1], outs[None]) and torch.equal(outs[1], outs['split'])
    for m in (4, 15):                      # below the range / odd: the rule, no measurement
        s0 = torch.empty((m,) + cs, device=net.device)
        g = net.executor.make_rollout(s0, torch.empty((3, m) + cs, device=net.device), 3)
        assert g.groups == Executor.member_groups(m, 88 * 180)
        g.close()


def test_functional_skip_unet_and_chained_outputs():
    from dlwp_amd import custom, layers as L
    from dlwp_amd.engine import Model
    from dlwp_amd.model import DLWPFunctional
    rng = np.random.default_rng(5)
    cs = (4, 16, 24)
    x0 = L.Input(shape=cs)
    pp2, zp2 = custom.PeriodicPadding2D((0, 2), **CF), L.ZeroPadding2D((2, 0), **CF)
    pp1, zp1 = custom.PeriodicPadding2D((0, 1), **CF), L.ZeroPadding2D((1, 0), **CF)
    pool, up = L.MaxPooling2D(2, **CF), L.UpSampling2D(2, **CF)
    convs = [L.Conv2D(32, 3, dilation_rate=2, activation='tanh', **CF), L.Conv2D(64, 3, activation='tanh', **CF),
             L.Conv2D(128, 3, activation='tanh', **CF), L.Conv2D(32, 3, activation='tanh', **CF),
             L.Conv2D(16, 3, dilation_rate=2, activation='tanh', **CF), L.Conv2D(4, 5, activation='linear', **CF)]
    s11, s12 = custom.slice_layer(0, 16, axis=1), custom.slice_layer(16, 32, axis=1)
    s21, s22 = custom.slice_layer(0, 32, axis=1), custom.slice_layer(32, 64, axis=1)

    def skip_model(x):                                               # examples/train_functional.py:248-275
        x = convs[0](pp2(zp2(x)))
        x, x1 = s11(x), s12(x)
        x = convs[1](pp1(zp1(pool(x))))
        x, x2 = s21(x), s22(x)
        x = convs[2](pp1(zp1(pool(x))))
        x = convs[3](pp1(zp1(up(x))))
        x = L.concatenate([x, x2], axis=1)
        x = convs[4](pp2(zp2(up(x))))
        x = L.concatenate([x, x1], axis=1)
        return convs[5](pp2(zp2(x)))
    outs = [skip_model(x0)]
    outs.append(skip_model(outs[0]))                                 # integration_steps = 2
    np.random.seed(5)
    m = Model(inputs=x0, outputs=outs)
    f = DLWPFunctional(is_convolutional=True, time_dim=2)
    f.build_model(m, loss='mse', loss_weights=[0.5, 0.5], optimizer='adam', metrics=['mae'])
    ws = m.get_weights()
    pairs = [(ws[i], (0.1 * rng.standard_normal(ws[i + 1].shape)).astype(np.float32)) for i in range(0, 12, 2)]
    m.set_weights([a for p in pairs for a in p])

    def ref(x):
        def halo(t, k):
            return np_ref.zero_padding2d(np_ref.periodic_padding2d(t, (0, k)), (k, 0))
        a = np_ref.conv2d(halo(x, 2), *pairs[0], 2, 'tanh')
        a, a1 = a[:, :16], a[:, 16:]
        b = np_ref.conv2d(halo(np_ref.maxpool2(a), 1), *pairs[1], 1, 'tanh')
        b, b2 = b[:, :32], b[:, 32:]
        c = np_ref.conv2d(halo(np_ref.maxpool2(b), 1), *pairs[2], 1, 'tanh')
        e = np_ref.conv2d(halo(np_ref.upsample2(c), 1), *pairs[3], 1, 'tanh')
        e = np.concatenate([e, b2], axis=1)
        g = np_ref.conv2d(halo(np_ref.upsample2(e), 2), *pairs[4], 2, 'tanh')
        g = np.concatenate([g, a1], axis=1)
        return np_ref.conv2d(halo(g, 2), *pairs[5], 1, 'linear')
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    y1, y2 = f.predict(x)
    r1 = ref(x.astype(np.float64))
    r2 = ref(r1)
    assert _rel(y1, r1) < FWD_TOL and _rel(y2, r2) < FWD_TOL
    # rollout: 2 outputs per call, time_dim 2 -> 5 steps need ceil(5/2/2) = 2 calls = 4 slots = 8 steps
    ts = f.predict_timeseries(x, 5)
    assert ts.shape == (8, 3, 2, 16, 24)
    want = np_ref.predict_timeseries_functional(lambda p: [ref(p), ref(ref(p))], x.astype(np.float64), 5, 2, n_outputs=2)
    assert _rel(ts[:4], want[:4]) < FWD_TOL
    # and it is exactly the host loop over predict()
    p, slots = x, []
    for _ in range(2):
        o1, o2 = f.predict(p)
        slots += [o1, o2]
        p = o2
    assert np.array_equal(ts, np_ref._merge_time(np.stack(slots), 4, 3, 2, cs, False))
    # r5: the streamed return (one graph per model call, BOTH output slots of a call leaving under the next call) gives the same array
    kept = f.predict_timeseries(x, 5, keep_time_dim=True)
    f.host_stream_bytes = 0
    try:
        assert np.array_equal(f.predict_timeseries(x, 5), ts)
        assert np.array_equal(f.predict_timeseries(x, 5, keep_time_dim=True), kept)
        assert m.streamed_rollout(3, 2, 1).n_out == 2
    finally:
        f.host_stream_bytes = 64 << 20


def test_fit_generator_with_sequence_targets_of_a_multi_output_model():
    """examples/train_functional.py flow: SeriesDataGenerator(sequence=2) hands a LIST of target arrays per batch and a
    DLWPFunctional model with two chained outputs trains on them through fit_generator -- i.e. through the DeviceLoader,
    which stages every target on its own.  The device feed equals feeding the generator's batches by hand."""
    from dlwp_amd import layers as L
    from dlwp_amd.engine import Model
    from dlwp_amd.model import DLWPFunctional, SeriesDataGenerator, SeriesDataset
    from dlwp_amd.training import Adam
    rng = np.random.default_rng(15)
    n_t, h, w = 30, 12, 16
    dates = (np.datetime64('2012-01-01T00') + np.arange(n_t) * np.timedelta64(6, 'h')).astype('datetime64[s]')
    series = np.cumsum(0.3 * rng.standard_normal((n_t, 2, 1, h, w)), axis=0).astype(np.float32)
    ds = SeriesDataset(series, {'sample': dates, 'variable': np.array(['z', 't']), 'level': np.array([500]),
                                'lat': np.linspace(60., -60., h), 'lon': np.arange(0., 360., 22.5)},
                       ('sample', 'variable', 'level', 'lat', 'lon'))

    def build():
        np.random.seed(8)
        x0 = L.Input(shape=(4, h, w))
        c1 = L.Conv2D(8, 3, padding='same', activation='tanh', **CF)
        c2 = L.Conv2D(4, 3, padding='same', activation='linear', **CF)
        o1 = c2(c1(x0))
        o2 = c2(c1(o1))
        f = DLWPFunctional(is_convolutional=True, time_dim=2)
        f.build_model(Model(inputs=x0, outputs=[o1, o2]), loss='mse', loss_weights=[0.5, 0.5], optimizer=Adam(lr=2e-3),
                      metrics=['mae'])
        return f
    f = build()
    gen = SeriesDataGenerator(f, ds, input_time_steps=2, output_time_steps=2, sequence=2, batch_size=6)
    X, y = gen[0]
    assert isinstance(y, list) and len(y) == 2 and y[0].shape == X.shape == (6, 4, h, w)
    hist = f.fit_generator(gen, epochs=3, verbose=0)
    losses = f.model.history.history['loss']
    assert len(losses) == 3 and np.isfinite(losses).all() and losses[-1] < losses[0]
    # the same three epochs, batches fed by hand
    g = build()
    for _ in range(3):
        for i in range(len(gen)):
            Xi, yi = gen[i]
            g.model.train_on_batch(Xi, yi)
    assert g.model.optimizer.iterations == f.model.optimizer.iterations == 3 * len(gen)
    for a, b in zip(f.model.get_weights(), g.model.get_weights()):
        assert np.array_equal(a, b)


def test_tf_padding2d_modes_forward_and_training_match_the_oracle():
    """TFPadding2D (reference DLWP/custom.py:527-600: tf.pad CONSTANT / REFLECT / SYMMETRIC) as the halo of a small
    encoder-decoder: fused into the convolution loaders, forward against the float64 oracle (numpy's pad modes of the same
    names), one train step's loss and gradients against torch autograd.  The REFLECT layer behind the UpSampling2D is not
    restated on its low-resolution source (that identity does not hold for a mirror without the border element)."""
    rng = np.random.default_rng(23)
    cs = (3, 12, 20)

    def blk(f, k, mode, act):
        return [('TFPadding2D', (k // 2,), dict(CF, mode=mode)),
                ('Conv2D', (f, k), dict(CF, padding='valid', activation=act))]
    layers = blk(16, 3, 'REFLECT', 'tanh')
    layers[0][2]['input_shape'] = cs
    layers += [('MaxPooling2D', (2,), dict(CF))] + blk(32, 3, 'SYMMETRIC', 'tanh') + [('UpSampling2D', (2,), dict(CF))]
    layers += blk(16, 3, 'REFLECT', 'tanh') + blk(3, 5, 'SYMMETRIC', 'linear') + blk(3, 3, 'CONSTANT', 'linear')
    layers = tuple(layers)
    d = _build(layers, time_dim=1)
    assert all(op.kind in ('conv', 'maxpool') for op in d.model.infer_plan.ops if op.kind not in ('phasew', 'd2s'))
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    got = d.predict(x)
    want = np_ref.run_layers(layers, x, weights)
    assert got.shape == want.shape and _rel(got, want) < FWD_TOL
    y = rng.standard_normal(got.shape).astype(np.float32)
    tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
    out = torch_ref.run_layers(layers, torch.tensor(x, dtype=torch.float64), tw)
    assert _rel(out.detach().numpy(), want) < 1e-12                      # the two oracles agree on the mirror modes
    loss = ((out - torch.tensor(y, dtype=torch.float64)) ** 2).mean()
    loss.backward()
    vals = d.model.train_on_batch(x, y)
    assert vals[0] == pytest.approx(float(loss.detach()), rel=2e-5)
    tr = d.model._trainer
    off = 0
    for w, b in tw:
        for g_ref in (w.grad.numpy().transpose(2, 3, 1, 0), b.grad.numpy()):
            g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
            off += g_ref.size
            assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape
    with pytest.raises(NotImplementedError, match='constant_values'):
        _build((('TFPadding2D', (1,), dict(CF, mode='CONSTANT', constant_values=1.0, input_shape=cs)),))


def test_standalone_layers_run_when_nothing_fuses():
    """Padding-only / pooling-tail models exercise the standalone kernels through build_model (both data formats)."""
    rng = np.random.default_rng(6)
    from dlwp_amd.model import DLWPNeuralNet
    d = DLWPNeuralNet(scaler_type=None, scale_targets=False)
    d.build_model((('PeriodicPadding2D', ((1, 2),), {'input_shape': (5, 6, 3)}),), loss='mse')     # channels_last default
    x = rng.standard_normal((4, 5, 6, 3)).astype(np.float32)
    assert np.array_equal(d.predict(x), np_ref.periodic_padding2d(x, (1, 2), 'channels_last'))
    d.build_model((('Conv2D', (4, 3), dict(CF, input_shape=(2, 8, 8))), ('MaxPooling2D', (2,), dict(CF)),
                   ('FillPadding2D', ((1, 0),), dict(CF))), loss='mse')
    w, b = d.model.get_weights()
    x = rng.standard_normal((2, 2, 8, 8)).astype(np.float32)
    want = np_ref.fill_padding2d(np_ref.maxpool2(np_ref.conv2d(x, w, b)), (1, 0))
    assert _rel(d.predict(x), want) < FWD_TOL


def test_full_size_unet_longitude_shift_equivariance_and_spot_parity():
    """BASELINE.json config-2 size (4 x 88 x 180): size-independent property + one sample against the oracle."""
    rng = np.random.default_rng(7)
    cs = (4, 88, 180)
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((4,) + cs).astype(np.float32)
    y = d.predict(x)
    ys = d.predict(np.roll(x, 8, axis=-1))                          # shift by a multiple of the pooling factor (4)
    # The direct kernels are equivariant bit for bit.  The Winograd kernels are, too, wherever the 2x2 tile lattice is
    # itself periodic (even width); the quarter-resolution layer is 45 columns wide, so its tiles pair columns up
    # differently after the wrap and the two rollouts differ by fp32 round-off only.
    assert _rel(ys, np.roll(y, 8, axis=-1)) < 1e-5
    from dlwp_amd import ops
    ops.set_winograd(False)
    try:
        dd = _build(layers, time_dim=2)
        dd.model.set_weights(d.model.get_weights())
        yd = dd.predict(x)
        assert np.array_equal(dd.predict(np.roll(x, 8, axis=-1)), np.roll(yd, 8, axis=-1))
        assert _rel(yd, y) < 1e-5                                   # direct vs Winograd family: round-off only
    finally:
        ops.set_winograd(True)
    want = torch_ref.run_layers(layers, torch.from_numpy(x[:1]), torch_ref.to_torch_weights(weights)).numpy()
    assert _rel(y[:1], want) < FWD_TOL


# ----------------------------------------------------------------------------------------------------------------- #
# training
# ----------------------------------------------------------------------------------------------------------------- #

def _torch_step(layers, weights, x, y, lr=1e-3):
    """One oracle train step: torch autograd on the unfused float64 graph + Keras-form Adam.  Returns loss, mae, grads,
    new weights (all numpy, HWIO)."""
    tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
    out = torch_ref.run_layers(layers, torch.tensor(x, dtype=torch.float64), tw)
    yt = torch.tensor(y, dtype=torch.float64)
    loss = ((out - yt) ** 2).mean()
    mae = (out - yt).abs().mean()
    loss.backward()
    grads, new = [], []
    for (w, b), (w0, b0) in zip(tw, weights):
        gw = w.grad.numpy().transpose(2, 3, 1, 0)
        gb = b.grad.numpy()
        grads += [gw, gb]
        for p0, g in ((w0, gw), (b0, gb)):
            p1, _, _ = np_ref.adam_keras_step(p0.astype(np.float64), np.zeros_like(g), np.zeros_like(g), g, 0, lr=lr)
            new.append(p1)
    return float(loss.detach()), float(mae.detach()), grads, new


def test_train_step_gradients_loss_and_adam_match_autograd_oracle():
    rng = np.random.default_rng(8)
    cs = (4, 16, 24)
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((6,) + cs).astype(np.float32)
    y = rng.standard_normal((6,) + cs).astype(np.float32)
    loss_ref, mae_ref, grads_ref, new_ref = _torch_step(layers, weights, x, y)
    tr = d.model._trainer
    vals = d.model.train_on_batch(x, y)
    assert vals[0] == pytest.approx(loss_ref, rel=2e-5) and vals[1] == pytest.approx(mae_ref, rel=2e-5)
    torch.cuda.synchronize()
    off = 0
    for g_ref in grads_ref:
        g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
        off += g_ref.size
        assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape
    for w_new, w_ref in zip(d.model.get_weights(), new_ref):
        assert np.abs(w_new - w_ref).max() < 5e-6                    # first Adam step moves every weight by ~lr = 1e-3
    assert d.model.optimizer.iterations == 1
    # evaluate() at the updated weights == the oracle's loss at ITS updated weights (the first Adam step at lr=1e-3 moves
    # all 189k weights coherently and overshoots on a 6-sample batch; the oracle shows the same jump)
    ev = d.evaluate(x, y, verbose=0)
    new_pairs = [(new_ref[i], new_ref[i + 1]) for i in range(0, len(new_ref), 2)]
    out1 = np_ref.run_layers(layers, x, new_pairs)
    assert len(ev) == 2
    assert ev[0] == pytest.approx(np_ref.mse(y, out1), rel=1e-3) and ev[1] == pytest.approx(np_ref.mae(y, out1), rel=1e-3)


def test_full_size_training_batch_of_520_equals_its_eight_distinct_samples():
    """A batch of 520 at 88 x 180 (65 copies of 8 samples; 1 GB per activation, 4 160 weight-gradient tiles per layer-2
    channel block, the partial-sum workspaces at their large-batch sizes) has the loss and the mean gradient of the 8
    samples: a size-independent check of the whole training step far above the batches the benchmarks use."""
    rng = np.random.default_rng(21)
    cs = (4, 88, 180)
    x8 = rng.standard_normal((8,) + cs).astype(np.float32)
    y8 = rng.standard_normal((8,) + cs).astype(np.float32)
    got = []
    for reps in (1, 65):
        d = _build(unet_layers(cs), time_dim=2, seed=5)
        vals = d.model.train_on_batch(np.tile(x8, (reps, 1, 1, 1)), np.tile(y8, (reps, 1, 1, 1)))
        torch.cuda.synchronize()
        got.append((vals, d.model._trainer.flat_grads.cpu().numpy().copy()))
    (v8, g8), (v520, g520) = got
    assert v520[0] == pytest.approx(v8[0], rel=2e-5) and v520[1] == pytest.approx(v8[1], rel=2e-5)
    assert np.isfinite(g520).all() and np.abs(g8).max() > 0
    assert np.abs(g520 - g8).max() <= 2e-4 * np.abs(g8).max()


def test_captured_training_step_equals_the_eager_step(monkeypatch):
    """A batch shape seen more than twice runs as one captured hipGraph (Trainer._graph_step: forward, loss, backward and the
    Adam update with the step number in device memory).  Same kernels, same order: weights, reported loss and the step
    counter follow the eager run (the only arithmetic difference is lr_t, evaluated in double on the device instead of on
    the host); a second batch shape gets its own graph; set_weights / evaluate in between do not disturb it."""
    rng = np.random.default_rng(12)
    cs = (4, 16, 24)
    layers = unet_layers(cs, widths=(8, 16, 16, 16, 8))
    xs = [rng.standard_normal((n,) + cs).astype(np.float32) for n in (6, 6, 6, 4, 6, 4, 4, 6, 4, 6)]
    ys = [rng.standard_normal(x.shape).astype(np.float32) for x in xs]

    def run(graph):
        monkeypatch.setenv('DLWP_TRAIN_GRAPH', '1' if graph else '0')
        d = _build(layers, time_dim=2, seed=3)
        tr = d.model._trainer
        logs = [d.model.train_on_batch(x, y) for x, y in zip(xs, ys)]
        ev = d.evaluate(xs[0], ys[0], verbose=0)
        logs.append(d.model.train_on_batch(xs[0], ys[0]))
        return d, tr, logs, ev
    de, tre, logs_e, ev_e = run(False)
    dg, trg, logs_g, ev_g = run(True)
    assert not tre._graphs and sorted(k[0] for k in trg._graphs) == [4, 6]
    assert dg.model.optimizer.iterations == de.model.optimizer.iterations == len(xs) + 1
    for a, b in zip(logs_g, logs_e):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7), (a, b)
    assert np.allclose(ev_g, ev_e, rtol=1e-5)
    for a, b in zip(dg.model.get_weights(), de.model.get_weights()):
        assert np.abs(a - b).max() <= 2e-7 * (len(xs) + 1)
    # a restored step counter is picked up by the captured step
    dg.model.optimizer.iterations = 1000
    de.model.optimizer.iterations = 1000
    monkeypatch.setenv('DLWP_TRAIN_GRAPH', '1')
    lg = dg.model.train_on_batch(xs[0], ys[0])
    monkeypatch.setenv('DLWP_TRAIN_GRAPH', '0')
    le = de.model.train_on_batch(xs[0], ys[0])
    assert np.allclose(lg, le, rtol=1e-5) and dg.model.optimizer.iterations == 1001
    for a, b in zip(dg.model.get_weights(), de.model.get_weights()):
        assert np.abs(a - b).max() <= 2e-7 * (len(xs) + 2)


def test_fit_and_fit_generator_reduce_the_loss():
    from dlwp_amd import custom
    from dlwp_amd.model import ArrayDataset, DataGenerator
    rng = np.random.default_rng(9)
    cs = (2, 12, 16)
    d = _build(cnn2_layers(cs, hidden=16), time_dim=2)
    # a learnable target: a fixed smoothing of the input
    P = rng.standard_normal((48, 2, 1, 12, 16)).astype(np.float32)
    T = (0.5 * P + 0.25 * np.roll(P, 1, axis=-1) + 0.25 * np.roll(P, -1, axis=-1)).astype(np.float32)
    gen = DataGenerator(d, ArrayDataset(P, T), batch_size=16, shuffle=True)
    assert gen.convolution_shape == cs
    hist = custom.History()
    early = custom.EarlyStoppingMin(min_epochs=2, monitor='val_loss', min_delta=0., patience=50, restore_best_weights=True)
    X, y = gen.generate([], scale_and_impute=False)
    d.fit_generator(gen, epochs=12, verbose=0, validation_data=gen, use_multiprocessing=True,
                    callbacks=[hist, custom.RNNResetStates(), early])
    assert len(hist.history['loss']) == 12 and 'val_loss' in hist.history and 'mean_absolute_error' in hist.history
    assert hist.history['loss'][-1] < 0.5 * hist.history['loss'][0]
    before = d.evaluate(X, y, verbose=0)[0]
    d.fit(X, y, batch_size=16, epochs=6, verbose=0, validation_data=(X, y), shuffle=True, callbacks=[hist])
    after = d.evaluate(X, y, verbose=0)[0]
    assert after < before
    assert d.model.optimizer.iterations == 12 * 3 + 6 * 3


def test_multi_output_training_accumulates_shared_layer_gradients():
    from dlwp_amd import custom, layers as L
    from dlwp_amd.engine import Model
    rng = np.random.default_rng(10)
    cs = (2, 8, 12)
    x0 = L.Input(shape=cs)
    pp, zp = custom.PeriodicPadding2D((0, 1), **CF), L.ZeroPadding2D((1, 0), **CF)
    c1 = L.Conv2D(8, 3, activation='tanh', **CF)
    c2 = L.Conv2D(2, 3, activation='linear', **CF)

    def f(t):
        return c2(pp(zp(c1(pp(zp(t))))))
    o1 = f(x0)
    o2 = f(o1)
    np.random.seed(10)
    m = Model(inputs=x0, outputs=[o1, o2])
    m.compile(optimizer='adam', loss='mse', loss_weights=[0.5, 0.5], metrics=['mae'])
    ws = m.get_weights()
    pairs = [(ws[0], (0.1 * rng.standard_normal(8)).astype(np.float32)), (ws[2], (0.1 * rng.standard_normal(2)).astype(np.float32))]
    m.set_weights([a for p in pairs for a in p])
    x = rng.standard_normal((4,) + cs).astype(np.float32)
    y1 = rng.standard_normal((4,) + cs).astype(np.float32)
    y2 = rng.standard_normal((4,) + cs).astype(np.float32)
    tw = torch_ref.to_torch_weights(pairs, dtype=torch.float64, requires_grad=True)
    blk = (('PeriodicPadding2D', ((0, 1),), CF), ('ZeroPadding2D', ((1, 0),), CF),
           ('Conv2D', (8, 3), dict(CF, activation='tanh')), ('PeriodicPadding2D', ((0, 1),), CF),
           ('ZeroPadding2D', ((1, 0),), CF), ('Conv2D', (2, 3), dict(CF, activation='linear')))
    t1 = torch_ref.run_layers(blk, torch.tensor(x, dtype=torch.float64), tw)
    t2 = torch_ref.run_layers(blk, t1, tw)
    l1 = ((t1 - torch.tensor(y1, dtype=torch.float64)) ** 2).mean()
    l2 = ((t2 - torch.tensor(y2, dtype=torch.float64)) ** 2).mean()
    (0.5 * l1 + 0.5 * l2).backward()
    vals = m.train_on_batch(x, [y1, y2])
    assert vals[0] == pytest.approx(float(0.5 * l1 + 0.5 * l2), rel=2e-5)
    assert vals[1] == pytest.approx(float(l1), rel=2e-5) and vals[2] == pytest.approx(float(l2), rel=2e-5)
    torch.cuda.synchronize()
    tr = m._trainer
    refs = [tw[0][0].grad.numpy().transpose(2, 3, 1, 0), tw[0][1].grad.numpy(),
            tw[1][0].grad.numpy().transpose(2, 3, 1, 0), tw[1][1].grad.numpy()]
    off = 0
    for g_ref in refs:
        g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
        off += g_ref.size
        assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6)


def test_conv_backward_kernels_against_oracle_all_halo_modes():
    """dlwp_conv2d_bwd_data / _bwd_weight directly through the C ABI: symmetric fast path and the general
    (asymmetric / edge halo) path, with pooled and up-sampled loaders."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(11)
    cases = [  # cin, cout, k, dil, pads, mh, mw, src
        (8, 24, 3, 1, (1, 1, 1, 1), 0, 1, 0), (20, 36, 3, 2, (2, 2, 2, 2), 0, 1, 0), (6, 4, 5, 1, (2, 2, 2, 2), 0, 1, 0),
        (8, 16, 3, 1, (1, 1, 1, 1), 0, 1, 1), (8, 16, 3, 1, (1, 1, 1, 1), 0, 1, 2), (5, 7, 3, 2, (2, 1, 3, 2), 2, 1, 0),
        (4, 8, 3, 1, (0, 0, 0, 0), 0, 0, 0), (6, 8, 3, 1, (1, 1, 1, 1), 2, 2, 0), (33, 40, 3, 1, (1, 1, 1, 1), 1, 1, 0),
        (8, 32, 3, 1, (1, 1, 1, 1), 3, 3, 0), (6, 8, 5, 1, (2, 2, 2, 2), 4, 3, 0), (8, 16, 3, 1, (1, 1, 1, 1), 4, 1, 1)]
    for cin, cout, k, dil, pads, mh, mw, src in cases:
        n, h, w = 3, 12, 20
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
        xs64 = np.asarray(x, np.float64)
        xt = {0: xs64, 1: np_ref.upsample2(xs64), 2: np_ref.maxpool2(xs64)}[src]
        xp = np_ref.pad2d_modes(xt, pads, mh, mw)
        ho, wo = xp.shape[2] - dil * (k - 1), xp.shape[3] - dil * (k - 1)
        dz = rng.standard_normal((n, cout, ho, wo)).astype(np.float32)
        dxp, dw_ref, _ = np_ref.conv2d_grads(xp, wt, dz, dil)
        dx_ref = np_ref.pad2d_modes_grad(dxp, xt.shape, pads, mh, mw)
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mh, mw), ops.ACT_LINEAR, src_mode=src)
        xs = _lib.Shape4(n, cin, h, w)
        dxd = torch.empty(xt.shape, dtype=torch.float32, device='cuda')
        ops.conv2d_bwd_data(torch.from_numpy(dz).cuda(), torch.from_numpy(wt).cuda(), cd, xs, dxd)
        dwd = torch.empty(wt.shape, dtype=torch.float32, device='cuda')
        ops.conv2d_bwd_weight(torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda(), dwd, cd, xs)
        torch.cuda.synchronize()
        case = (cin, cout, k, dil, pads, mh, mw, src)
        assert np.abs(dxd.cpu().numpy() - dx_ref).max() <= 2e-5 * max(1., np.abs(dx_ref).max()), case
        assert np.abs(dwd.cpu().numpy() - dw_ref).max() <= 2e-5 * max(1., np.abs(dw_ref).max()), case


def test_data_gradient_of_an_upsampled_source_with_the_fused_sum_epilogue():
    """dlwp_conv2d_bwd_data_stored == dlwp_conv2d_bwd_data followed by dlwp_upsample2_bwd (float64 oracle); layers without
    a summing kernel report 'unsupported' and leave the two-call route to the caller."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(12)
    for cin, cout, h, w, mh, mw, expect in [(32, 16, 6, 10, 0, 1, True), (64, 24, 11, 23, 0, 1, True),
                                            (96, 8, 5, 45, 0, 0, True), (8, 16, 6, 10, 0, 1, False),
                                            (128, 64, 22, 45, 0, 1, True)]:
        n = 2
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
        xt = np_ref.upsample2(np.asarray(x, np.float64))
        xp = np_ref.pad2d_modes(xt, (1, 1, 1, 1), mh, mw)
        dz = rng.standard_normal((n, cout, 2 * h, 2 * w)).astype(np.float32)
        dxp, _, _ = np_ref.conv2d_grads(xp, wt, dz, 1)
        dx_ref = np_ref.pad2d_modes_grad(dxp, xt.shape, (1, 1, 1, 1), mh, mw)
        dx_ref = dx_ref.reshape(n, cin, h, 2, w, 2).sum(axis=(3, 5))
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, mh, mw), ops.ACT_LINEAR, src_mode=ops.SRC_UPSAMPLE2)
        xs = _lib.Shape4(n, cin, h, w)
        dxd = torch.full((n, cin, h, w), float('nan'), dtype=torch.float32, device='cuda')
        dzd, wd = torch.from_numpy(dz).cuda(), torch.from_numpy(wt).cuda()
        ok = ops.conv2d_bwd_data_stored(dzd, wd, cd, xs, dxd)
        assert ok == expect, (cin, cout)
        if not ok:
            assert torch.isnan(dxd).all()
            continue
        two = ops.upsample2_bwd(ops.conv2d_bwd_data(dzd, wd, cd, xs, torch.empty((n, cin, 2 * h, 2 * w), device='cuda')))
        torch.cuda.synchronize()
        scale = max(1., np.abs(dx_ref).max())
        assert np.abs(dxd.cpu().numpy() - dx_ref).max() <= 4e-5 * scale, (cin, cout, h, w)
        assert (dxd - two).abs().max().item() <= 4e-5 * scale


def test_conv_weight_gradient_random_shapes_against_oracle():
    """40 seeded random layer geometries through the heuristic's choice of weight-gradient kernel (direct, packed-N,
    Winograd): odd sizes, ragged channels, both dilations, all halo modes, all source modes, asymmetric halos."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(20240611)
    kinds = set()
    cfgs = ops.wgrad_configs()
    for case in range(40):
        k = int(rng.choice([3, 3, 3, 5]))
        dil = int(rng.choice([1, 2])) if k == 3 else 1
        cin = int(rng.choice([1, 3, 4, 8, 16, 20, 32, 40]))
        cout = int(rng.choice([2, 4, 12, 32, 36, 64]))
        src = int(rng.choice([0, 0, 1, 2]))
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(6, 26)), int(rng.integers(8, 44))
        if src == 2:
            h, w = h + 6, w + 8
        mode_h, mode_w = int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2]))
        p = dil * (k - 1) // 2
        pads = (p, p, p, p) if rng.integers(0, 3) else (p, p + 1, p + 1, p)       # sometimes asymmetric
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        xs64 = np.asarray(x, np.float64)
        xt = {0: xs64, 1: np_ref.upsample2(xs64), 2: np_ref.maxpool2(xs64)}[src]
        if (mode_h == 1 and max(pads[:2]) > xt.shape[2]) or (mode_w == 1 and max(pads[2:]) > xt.shape[3]):
            continue
        xp = np_ref.pad2d_modes(xt, pads, mode_h, mode_w)
        ho, wo = xp.shape[2] - dil * (k - 1), xp.shape[3] - dil * (k - 1)
        dz = rng.standard_normal((n, cout, ho, wo)).astype(np.float32)
        _, dw_ref, _ = np_ref.conv2d_grads(xp, np.zeros((k, k, cin, cout)), dz, dil)
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mode_h, mode_w), ops.ACT_LINEAR, src_mode=src)
        xs = _lib.Shape4(n, cin, h, w)
        pick = _lib.lib.dlwp_conv2d_wgrad_pick_config(_lib.handle(0), xs, ctypes.byref(cd))
        assert pick >= 0
        kinds.add('packed' if cfgs[pick][4] < 0 else 'other')
        dwd = torch.empty((k, k, cin, cout), dtype=torch.float32, device='cuda')
        ops.conv2d_bwd_weight(torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda(), dwd, cd, xs)
        err = np.abs(dwd.cpu().numpy() - dw_ref).max()
        assert err <= 3e-5 * max(1., np.abs(dw_ref).max()), (case, n, cin, cout, k, dil, src, h, w, pads, mode_h, mode_w, pick, err)
    assert 'packed' in kinds and 'other' in kinds


def test_conv_weight_gradient_full_size_properties():
    """Size-independent properties of dlwp_conv2d_bwd_weight at BASELINE.json's full 88x180 grid (config 2, layer 5
    shape; the Winograd weight-gradient kernel): linearity in dz, additivity over the batch, and the adjoint identity
    <conv(x; w), dz> == <w, dW(x, dz)> against the forward kernel."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(23)
    n, cin, h, w, cout = 4, 64, 88, 180, 32
    x = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32)).cuda()
    dz1 = torch.from_numpy(rng.standard_normal((n, cout, h, w)).astype(np.float32)).cuda()
    dz2 = torch.from_numpy(rng.standard_normal((n, cout, h, w)).astype(np.float32)).cuda()
    cd = ops.make_conv(cout, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_LINEAR)
    xs = _lib.Shape4(n, cin, h, w)

    def dw_of(xx, dz, shape=xs):
        out = torch.empty((3, 3, cin, cout), dtype=torch.float32, device='cuda')
        ops.conv2d_bwd_weight(xx, dz, out, cd, shape)
        return out
    a, b = dw_of(x, dz1), dw_of(x, dz2)
    scale = float(a.abs().max())
    assert float((dw_of(x, dz1 + dz2) - (a + b)).abs().max()) <= 2e-5 * scale * 2           # linear in dz
    assert float((dw_of(x, 2 * dz1) - 2 * a).abs().max()) == 0.0                              # exactly, for a power of two
    halves = dw_of(x[:2].contiguous(), dz1[:2].contiguous(), _lib.Shape4(2, cin, h, w)) + \
        dw_of(x[2:].contiguous(), dz1[2:].contiguous(), _lib.Shape4(2, cin, h, w))
    assert float((halves - a).abs().max()) <= 2e-5 * scale                                     # additive over samples
    wt = torch.from_numpy(np_ref.glorot_uniform((3, 3, cin, cout), rng)).cuda()
    y = ops.conv2d(x, wt, None, cd)
    lhs = float((y.double() * dz1.double()).sum())
    rhs = float((wt.double() * a.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), float(y.double().abs().sum()) * 1e-3)       # adjoint of the forward conv


@pytest.mark.parametrize('cin,cout,h,w,src,n', [(64, 128, 22, 45, 0, 8), (128, 64, 22, 45, 1, 4), (32, 64, 44, 90, 0, 4),
                                                (64, 32, 44, 90, 0, 3)])
def test_channel_block_weight_gradient_full_size_properties(cin, cout, h, w, src, n):
    """The config-3 U-Net's Winograd weight gradients at their full grids (csrc/conv_wgrad_cb_kernel.h: odd widths -> element
    loads, 90-wide rows -> ragged last pixel quad, the up-sampled source's 9-position form): additivity over the batch (other
    splits, other tile walks), linearity in dz, the adjoint identity against the forward kernel, and a float64 oracle on a
    corner of the kernel tensor (two input x three output channels: cheap at this size)."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(cin + cout + src)
    x = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32)).cuda()
    ho, wo = (2 * h, 2 * w) if src == 1 else (h, w)
    dz1 = torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)).cuda()
    dz2 = torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)).cuda()
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR, src_mode=src)

    def dw_of(xx, dz):
        out = torch.empty((3, 3, cin, cout), dtype=torch.float32, device='cuda')
        ops.conv2d_bwd_weight(xx, dz, out, cd, _lib.Shape4(xx.shape[0], cin, h, w))
        return out
    a, b = dw_of(x, dz1), dw_of(x, dz2)
    scale = float(a.abs().max())
    assert float((dw_of(x, dz1 + dz2) - (a + b)).abs().max()) <= 2e-5 * scale * 2
    assert float((dw_of(x, 2 * dz1) - 2 * a).abs().max()) == 0.0
    k = n // 2
    halves = dw_of(x[:k].contiguous(), dz1[:k].contiguous()) + dw_of(x[k:].contiguous(), dz1[k:].contiguous())
    assert float((halves - a).abs().max()) <= 2e-5 * scale
    wt = torch.from_numpy(np_ref.glorot_uniform((3, 3, cin, cout), rng)).cuda()
    y = ops.conv2d(x, wt, None, cd)
    lhs = float((y.double() * dz1.double()).sum())
    rhs = float((wt.double() * a.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), float(y.double().abs().sum()) * 1e-3)
    # float64 oracle on channels (0, cin - 1) x (0, 17, cout - 1)
    ci_s, co_s = [0, cin - 1], [0, 17, cout - 1]
    xs64 = x[:, ci_s].double().cpu().numpy()
    xt = np_ref.upsample2(xs64) if src == 1 else xs64
    xp = np_ref.pad2d_modes(xt, (1, 1, 1, 1), 0, 1)
    _, dw_ref, _ = np_ref.conv2d_grads(xp, np.zeros((3, 3, 2, 3)), dz1[:, co_s].double().cpu().numpy(), 1)
    got = a.cpu().numpy()[:, :, ci_s][:, :, :, co_s]
    assert np.abs(got - dw_ref).max() <= 3e-5 * max(1., np.abs(dw_ref).max())


def test_fused_activation_backward_and_bias_gradient_equal_the_two_separate_kernels():
    """dlwp_act_bwd_bias_grad == dlwp_act_bwd followed by dlwp_bias_grad: dz bit for bit, db to float32 rounding (fixed
    but differently ordered partial sums), on a channel window, with vector (hw % 4 == 0) and scalar planes, in place."""
    from dlwp_amd import ops
    rng = np.random.default_rng(3)
    for (n, c_total, h, w, c, c_off) in [(5, 12, 6, 10, 12, 0), (3, 9, 7, 9, 4, 3), (64, 32, 11, 20, 32, 0)]:
        for act in (ops.ACT_TANH, ops.ACT_RELU):
            y = torch.from_numpy(np.tanh(rng.standard_normal((n, c_total, h, w))).astype(np.float32)).cuda()
            dy = torch.from_numpy(rng.standard_normal((n, c_total, h, w)).astype(np.float32)).cuda()
            dz_ref = ops.act_bwd(y, dy, act)
            db_ref = torch.empty(c, device='cuda')
            ops.bias_grad(dz_ref, db_ref, c, c_off)
            g = dy.clone()
            db = torch.empty(c, device='cuda')
            ops.act_bwd_bias_grad(y, g, act, db, c, c_off, out=g)
            assert torch.equal(g[:, c_off:c_off + c], dz_ref[:, c_off:c_off + c])
            assert torch.equal(g[:, :c_off], dy[:, :c_off]) and torch.equal(g[:, c_off + c:], dy[:, c_off + c:])
            assert torch.allclose(db, db_ref, rtol=1e-5, atol=1e-4)
            want = dz_ref[:, c_off:c_off + c].double().sum(dim=(0, 2, 3))
            assert torch.allclose(db.double(), want, rtol=1e-5, atol=1e-4)


def test_fused_pooling_activation_backward_equals_the_three_separate_kernels():
    """dlwp_pool_act_bwd_bias_grad == dlwp_maxpool2_bwd, dlwp_act_bwd, dlwp_bias_grad in turn: dz bit for bit (ties
    included: the first maximum of a window takes the gradient), db to float32 rounding; even and odd planes."""
    from dlwp_amd import ops
    rng = np.random.default_rng(4)
    for (n, c, h, w) in [(3, 5, 6, 10), (2, 4, 7, 9), (4, 3, 8, 11), (64, 32, 22, 44)]:
        for act in (ops.ACT_TANH, ops.ACT_RELU, ops.ACT_LINEAR):
            yv = np.tanh(rng.standard_normal((n, c, h, w))).astype(np.float32)
            yv[rng.random(yv.shape) < 0.3] = 0.25          # plenty of ties inside windows
            y = torch.from_numpy(yv).cuda()
            dp = torch.from_numpy(rng.standard_normal((n, c, h // 2, w // 2)).astype(np.float32)).cuda()
            dz_ref = ops.act_bwd(y, ops.maxpool2_bwd(y, dp), act) if act != ops.ACT_LINEAR else ops.maxpool2_bwd(y, dp)
            db_ref = torch.empty(c, device='cuda')
            ops.bias_grad(dz_ref, db_ref, c)
            db = torch.empty(c, device='cuda')
            dz = ops.pool_act_bwd_bias_grad(y, dp, act, db)
            assert torch.equal(dz, dz_ref), (n, c, h, w, act)
            assert torch.allclose(db, db_ref, rtol=1e-5, atol=1e-4)
            assert torch.equal(ops.pool_act_bwd_bias_grad(y, dp, act), dz_ref)       # without a bias


def test_conv_weight_gradient_every_compiled_tile_configuration():
    """Force each weight-gradient tile configuration in turn: ragged tiles, ragged channel groups, odd AND even widths
    (column-pair loads vs their element-wise form), output widths that are / are not multiples of 4 (pixel-quad loads),
    periodic + zero halo, direct / pooled / up-sampled source."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(17)
    cfgs = ops.wgrad_configs()
    forms = ops.wgrad_config_forms()
    geoms = [(3, 19, 50, 0), (2, 11, 21, 0), (2, 18, 44, 2), (2, 7, 13, 1)]      # (n, h, w stored, src_mode)
    cache = {}
    try:
        for i, (ks, dil, th, tw, nt, waves, lds) in enumerate(cfgs):
            c4 = forms[i][1] == 4                                   # the streaming form: at most 4 input channels
            for gi, (n, h, w, src) in enumerate(geoms):
                key = (ks, dil, gi, nt < 0, c4)
                if key not in cache:
                    cin, cout = (20, 36) if nt > 0 else (20, 3)     # packed-N instances: at most 4 output channels
                    if c4:
                        cin, cout = 3 + gi % 2, 36                  # 3 or 4 channels, a ragged second cout tile
                    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
                    xs64 = np.asarray(x, np.float64)
                    xt = {0: xs64, 1: np_ref.upsample2(xs64), 2: np_ref.maxpool2(xs64)}[src]
                    p = dil * (ks - 1) // 2
                    pads = (p, p, p, p)
                    xp = np_ref.pad2d_modes(xt, pads, 0, 1)
                    dz = rng.standard_normal((n, cout, xt.shape[2], xt.shape[3])).astype(np.float32)
                    _, dw_ref, _ = np_ref.conv2d_grads(xp, np.zeros((ks, ks, cin, cout)), dz, dil)
                    cache[key] = (torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda(), dw_ref, pads, cin, cout)
                xd, dzd, dw_ref, pads, cin, cout = cache[key]
                cd = ops.make_conv(cout, ks, ks, dil, ops.make_pad(*pads, 0, 1), ops.ACT_LINEAR, src_mode=src)
                dwd = torch.empty((ks, ks, cin, cout), dtype=torch.float32, device='cuda')
                ops.force_wgrad_config(i)
                ops.conv2d_bwd_weight(xd, dzd, dwd, cd, _lib.Shape4(n, cin, h, w))
                err = np.abs(dwd.cpu().numpy() - dw_ref).max()
                assert err <= 2e-5 * max(1., np.abs(dw_ref).max()), (i, cfgs[i], geoms[gi], err)
    finally:
        ops.force_wgrad_config(-1)


@pytest.mark.parametrize('mode_h,mode_w', [(0, 1), (2, 1), (0, 0), (1, 1)])
def test_weight_gradient_on_an_up_sampled_source_every_channel_block_instance_long_tile_walks(mode_h, mode_w):
    """The 9-position body for an up-sampled source (r6: conv_wgrad_cbu_kernel.h -- source-resolution LDS window, two buffers, one
    barrier per tile) against the float64 oracle on every channel-block instance: zero / periodic / edge (pole rows) halos at source
    resolution, quads that straddle the periodic seam, a ragged last dz quad (output width 4 k + 2), ragged channel groups -- and with
    DLWP_OPT_WGRAD_FILL = 1, so that a workgroup walks MANY tiles and both buffers change roles several times (at the default fill
    these small launches give every workgroup one or two tiles)."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(23 + 5 * mode_h + mode_w)
    cfgs, forms = ops.wgrad_configs(), ops.wgrad_config_forms()
    n, cin, cout, hs, ws = 5, 40, 72, 11, 23                  # up-sampled: 22 x 46 (46 = 4 k + 2); 40 = 32 + 8, 72 = 64 + 8 channels
    x = rng.standard_normal((n, cin, hs, ws)).astype(np.float32)
    xu = np_ref.upsample2(np.asarray(x, np.float64))
    xp = np_ref.pad2d_modes(xu, (1, 1, 1, 1), mode_h, mode_w)
    dz = rng.standard_normal((n, cout, 2 * hs, 2 * ws)).astype(np.float32)
    _, dw_ref, _ = np_ref.conv2d_grads(xp, np.zeros((3, 3, cin, cout)), dz, 1)
    xd, dzd = torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda()
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, mode_h, mode_w), ops.ACT_LINEAR, src_mode=1)
    seen = 0
    try:
        for fill in (1, 4):
            _lib.set_option(_lib.OPT_WGRAD_FILL, fill)
            for i, (ks, dil, th, tw, nt, waves, lds) in enumerate(cfgs):
                if forms[i][1] != 3 or (ks, dil) != (3, 1):
                    continue
                ops.force_wgrad_config(i)
                dwd = torch.full((3, 3, cin, cout), float('nan'), dtype=torch.float32, device='cuda')
                ops.conv2d_bwd_weight(xd, dzd, dwd, cd, _lib.Shape4(n, cin, hs, ws))
                err = np.abs(dwd.cpu().numpy() - dw_ref).max()
                assert err <= 2e-5 * max(1., np.abs(dw_ref).max()), (fill, i, cfgs[i], err)
                seen += 1
    finally:
        ops.force_wgrad_config(-1)
        _lib.set_option(_lib.OPT_WGRAD_FILL, 4)
    assert seen >= 2 * 9                                       # the nine channel-block instances, at both fills


def test_reference_style_example_script_runs_end_to_end(tmp_path):
    """examples/train_and_forecast.py is written with the reference's imports (DLWP.*, keras.*) through the compat shim:
    data generator -> build_model -> fit_generator with callbacks -> save / load -> predict_timeseries."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'train_and_forecast.py')
    spec = importlib.util.spec_from_file_location('train_and_forecast', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ['x', '--grid', '16x24', '--samples', '48', '--epochs', '2', '--batch-size', '16', '--model-file',
                os.path.join(str(tmp_path), 'm')]
    try:
        score, series = mod.main()
    finally:
        sys.argv = argv
    assert len(score) == 2 and np.isfinite(score[0]) and series.shape == (8, 12, 2, 16, 24)
    # ... and with the reference's `latitude_dependent` switch: RowConnected2D as the output layer, looked up by name in
    # DLWP.custom, trained, saved, reloaded and rolled out
    sys.argv = ['x', '--grid', '16x24', '--samples', '48', '--epochs', '2', '--batch-size', '16', '--model-file',
                os.path.join(str(tmp_path), 'mrow'), '--latitude-dependent']
    try:
        score, series = mod.main()
    finally:
        sys.argv = argv
    assert len(score) == 2 and np.isfinite(score[0]) and series.shape == (8, 12, 2, 16, 24)


def test_custom_losses_match_reference_goldens_and_autograd(golden):
    """anomaly_correlation_loss / latitude_weighted_loss on the device: VALUES pinned by tests/golden/losses.npz (produced by
    running the reference's own loss functions under the numpy-K shim), gradients against torch autograd of the oracle
    formula."""
    from dlwp_amd import custom, ops
    g = golden('losses')
    yt, yp, climo, lats = g['y_true'], g['y_pred'], g['climo'], g['lats']
    ypd, ytd = torch.from_numpy(yp).cuda(), torch.from_numpy(yt).cuda()
    stats = torch.zeros(7, device='cuda')

    def run(spec):
        dy = torch.empty_like(ypd)
        mean = None if spec.mean is None else torch.from_numpy(spec.mean).cuda()
        roww = None if spec.row_weights is None else torch.from_numpy(spec.row_weights).cuda()
        ops.loss_custom(ypd, ytd, stats, dy, spec.scale, mean, roww, spec.kind, spec.regularize)
        torch.cuda.synchronize()
        return spec.scale * float(stats[0]), dy.cpu().numpy(), stats.cpu().numpy()

    def torch_loss(spec, p):
        t = torch.tensor(yt, dtype=torch.float64)
        w = torch.ones(1, dtype=torch.float64) if spec.row_weights is None else \
            torch.tensor(spec.row_weights, dtype=torch.float64)[None, None, :, None]
        m = torch.zeros(1, dtype=torch.float64) if spec.mean is None else torch.tensor(spec.mean, dtype=torch.float64)[None]
        pw, tw = p * w, t * w
        if spec.kind == 0:
            return ((pw - tw) ** 2).mean()
        P, T = pw - m, tw - m
        a = (P * T).mean() / torch.sqrt((P * P).mean() * (T * T).mean())
        reg = {0: 0.0, 1: ((pw - tw) ** 2).mean(), 2: (pw - tw).abs().mean()}[spec.regularize]
        return spec.scale * (reg - a)

    for reg in (None, 'mse', 'mae'):
        for use_mean in (False, True):
            spec = custom.anomaly_correlation_loss(climo if use_mean else None, regularize_mean=reg, reverse=True)
            val, dy, st = run(spec)
            want = float(g['acc_%s_%d' % (reg, int(use_mean))])
            assert val == pytest.approx(want, rel=2e-5, abs=2e-6), (reg, use_mean)
            assert st[1] == pytest.approx(np_ref.mse(yt, yp), rel=1e-5) and st[2] == pytest.approx(np_ref.mae(yt, yp), rel=1e-5)
            p = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
            torch_loss(spec, p).backward()
            assert np.abs(dy - p.grad.numpy()).max() <= 2e-5 * np.abs(p.grad.numpy()).max(), (reg, use_mean)
    # 'global' / 'spatial' mean-ratio regularisers (custom.py:1070-1074) on fields with a non-zero mean, also under latitude
    # weights
    ytp, ypp = g['y_true_pos'], g['y_pred_pos']
    yp_keep, yt_keep = (ypd, ytd), (yp, yt)
    ypd, ytd = torch.from_numpy(ypp).cuda(), torch.from_numpy(ytp).cuda()
    yp, yt = ypp, ytp

    def torch_loss_ratio(spec, p):
        t = torch.tensor(ytp, dtype=torch.float64)
        w = torch.ones(1, dtype=torch.float64) if spec.row_weights is None else \
            torch.tensor(spec.row_weights, dtype=torch.float64)[None, None, :, None]
        m = torch.zeros(1, dtype=torch.float64) if spec.mean is None else torch.tensor(spec.mean, dtype=torch.float64)[None]
        pw, tw = p * w, t * w
        P, T = pw - m, tw - m
        a = (P * T).mean() / torch.sqrt((P * P).mean() * (T * T).mean())
        if spec.regularize == 3:
            reg = ((tw.mean() - pw.mean()) / tw.mean()).abs()
        else:
            mt, mp = tw.mean(dim=(-2, -1)), pw.mean(dim=(-2, -1))
            reg = ((mt - mp) / mt).abs().mean()
        return spec.scale * (reg - a)
    for reg in ('global', 'spatial'):
        for use_mean in (False, True):
            spec = custom.anomaly_correlation_loss((climo + 3.0) if use_mean else None, regularize_mean=reg)
            val, dy, st = run(spec)
            assert val == pytest.approx(float(g['accpos_%s_%d' % (reg, int(use_mean))]), rel=2e-5, abs=2e-6), (reg, use_mean)
            p = torch.tensor(ypp, dtype=torch.float64, requires_grad=True)
            torch_loss_ratio(spec, p).backward()
            assert np.abs(dy - p.grad.numpy()).max() <= 2e-5 * np.abs(p.grad.numpy()).max(), (reg, use_mean)
        spec = custom.latitude_weighted_loss(custom.anomaly_correlation_loss(climo + 3.0, regularize_mean=reg), lats,
                                             (4, 6, 8), axis=-2, weighting='cosine')
        val, dy, _ = run(spec)
        p = torch.tensor(ypp, dtype=torch.float64, requires_grad=True)
        ref = torch_loss_ratio(spec, p)
        ref.backward()
        assert val == pytest.approx(float(ref.detach()), rel=2e-5)
        assert np.abs(dy - p.grad.numpy()).max() <= 2e-5 * np.abs(p.grad.numpy()).max()
    (ypd, ytd), (yp, yt) = yp_keep, yt_keep
    for weighting in ('cosine', 'midlatitude'):
        spec = custom.latitude_weighted_loss(None, lats, (4, 6, 8), axis=-2, weighting=weighting)
        val, dy, _ = run(spec)
        assert val == pytest.approx(float(g['latw_%s' % weighting]), rel=2e-5)
        p = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
        torch_loss(spec, p).backward()
        assert np.abs(dy - p.grad.numpy()).max() <= 2e-5 * np.abs(p.grad.numpy()).max()
    # nested, as examples/train.py:224-234 builds it, and trained through build_model
    spec = custom.latitude_weighted_loss(custom.anomaly_correlation_loss(climo, regularize_mean='mse'), lats, (4, 6, 8),
                                         axis=-2, weighting='midlatitude')
    val, dy, _ = run(spec)
    p = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
    ref = torch_loss(spec, p)
    ref.backward()
    assert val == pytest.approx(float(ref.detach()), rel=2e-5)
    assert np.abs(dy - p.grad.numpy()).max() <= 2e-5 * np.abs(p.grad.numpy()).max()
    rng = np.random.default_rng(12)
    cs = (4, 6, 8)
    d = _build(cnn2_layers(cs, hidden=8), time_dim=2, loss=spec)
    x = rng.standard_normal((8,) + cs).astype(np.float32)
    l0 = d.model.train_on_batch(x, yt[:1].repeat(8, axis=0) * 0 + x)
    for _ in range(30):
        l1 = d.model.train_on_batch(x, x)
    assert d.model.metrics_names == ['loss', 'mean_absolute_error'] and l1[0] < l0[0] and -1.1 < l1[0]


# ----------------------------------------------------------------------------------------------------------------- #
# recurrent front end: PeriodicPadding3D + ZeroPadding3D + ConvLSTM2D  (examples/train.py:142-157)
# ----------------------------------------------------------------------------------------------------------------- #

def _lstm_weights(model, rng):
    """Weights of a ConvLSTM2D + Conv2D stack in the oracle's form [(k, r, b) | (w, b)], biases randomised."""
    ws = model.get_weights()
    out, i = [], 0
    for lay in model.layers:
        n = len(lay._weights)
        if n == 0:
            continue
        arrs = [a.copy() for a in ws[i:i + n]]
        arrs[-1] = (arrs[-1] + 0.1 * rng.standard_normal(arrs[-1].shape)).astype(np.float32)
        out.append(tuple(arrs))
        i += n
    model.set_weights([a for item in out for a in item])
    return out


def test_convlstm_unet_forward_and_rollout_match_oracle():
    from dlwp_amd.model import DLWPNeuralNet
    from tests.nets import lstm_unet_layers
    rng = np.random.default_rng(21)
    cs = (2, 2, 16, 24)                                               # (time_dim, variables, lat, lon)
    layers = lstm_unet_layers(cs, widths=(16, 32, 64, 32, 16))
    np.random.seed(3)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(layers, loss='mse', optimizer='adam')
    assert d.model.output_shape == (None,) + cs
    kinds = [op.kind for op in d.model.plan.ops]
    assert kinds[:5] == ['conv', 'lstm', 'conv', 'conv', 'lstm']      # x-conv, gates | x-conv, h-conv, gates
    weights = _lstm_weights(d.model, rng)
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    got = d.predict(x)
    want = np_ref.run_layers(layers, x, weights)
    assert got.shape == want.shape == (3,) + cs
    assert _rel(got, want) < FWD_TOL
    want32 = torch_ref.run_layers(layers, torch.from_numpy(x), torch_ref.to_torch_weights(weights)).numpy()
    assert _rel(got, want32) < FWD_TOL
    # rollout: the hipGraph replay equals the host loop over predict(), and follows the reference's bookkeeping
    series = d.predict_timeseries(x, 5)                               # ceil(5/2) = 3 forwards -> 6 steps
    assert series.shape == (6, 3, 2, 16, 24)
    p, ser = x, []
    for _ in range(3):
        p = d.predict(p)
        ser.append(p)
    ser = np.stack(ser)
    assert np.array_equal(series, np_ref._merge_time(ser, 3, 3, 2, cs[1:], False).reshape(series.shape))
    # r5: the streamed return on a RECURRENT state (time_step, variable, lat, lon): the same arrays, both layouts
    kept = d.predict_timeseries(x, 5, keep_time_dim=True)
    d.host_stream_bytes = 0
    try:
        assert np.array_equal(d.predict_timeseries(x, 5), series)
        got_kept = d.predict_timeseries(x, 5, keep_time_dim=True)
        assert got_kept.shape == kept.shape and np.array_equal(got_kept, kept)
    finally:
        d.host_stream_bytes = 64 << 20


def test_convlstm_train_step_matches_autograd_oracle_with_l2():
    """Back-propagation through time on the recurrent front end (gate backward kernel + the two convolutions per step,
    weight gradients accumulated over the steps), l2 kernel regulariser on the ConvLSTM2D kernel as in
    examples/train.py:154, against torch autograd on the float64 restatement."""
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.regularizers import l2
    from tests.nets import lstm_unet_layers
    rng = np.random.default_rng(31)
    cs = (3, 2, 8, 12)                                                # three time steps: h_0, h_1 feed recurrent convs
    lam = 1e-3
    layers = list(lstm_unet_layers(cs, widths=(8, 16, 16, 16, 8)))
    layers[2] = (layers[2][0], layers[2][1], dict(layers[2][2], kernel_regularizer=l2(lam)))
    layers = tuple(layers)
    np.random.seed(4)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=3, scaler_type=None, scale_targets=False)
    d.build_model(layers, loss='mse', optimizer='adam', metrics=['mae'])
    weights = _lstm_weights(d.model, rng)
    x = rng.standard_normal((4,) + cs).astype(np.float32)
    y = rng.standard_normal((4,) + cs).astype(np.float32)
    # oracle: float64 autograd
    tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
    out = torch_ref.run_layers(layers, torch.tensor(x, dtype=torch.float64), tw)
    yt = torch.tensor(y, dtype=torch.float64)
    mse = ((out - yt) ** 2).mean()
    reg = lam * (tw[0][0] ** 2).sum()
    (mse + reg).backward()
    grads_ref = []
    for item in tw:
        for k, a in enumerate(item):
            g = a.grad.numpy()
            grads_ref.append(g.transpose(2, 3, 1, 0) if g.ndim == 4 else g)
    vals = d.model.train_on_batch(x, y)
    assert vals[0] == pytest.approx(float(mse + reg), rel=2e-5)
    assert vals[1] == pytest.approx(float((out - yt).abs().mean()), rel=2e-5)
    torch.cuda.synchronize()
    tr = d.model._trainer
    off = 0
    for g_ref in grads_ref:
        g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
        off += g_ref.size
        assert np.abs(g - g_ref).max() <= 3e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape
    assert off == tr.flat_grads.numel()
    # a few more steps reduce the loss
    first = vals[0]
    for _ in range(20):
        vals = d.model.train_on_batch(x, y)
    assert vals[0] < first


def test_time_series_estimator_runs_the_device_rollout_for_matching_io():
    """examples/validate.py:191-205: SeriesDataGenerator + TimeSeriesEstimator.  With identical inputs and outputs the
    estimator is the plain rollout (device hipGraph) plus coordinates; with an insolation input it steps on the host,
    one fused device forward per step."""
    from dlwp_amd.model import SeriesDataGenerator, SeriesDataset, TimeSeriesEstimator
    rng = np.random.default_rng(41)
    n_t, h, w = 14, 16, 24
    dates = (np.datetime64('2010-01-01T00') + np.arange(n_t) * np.timedelta64(6, 'h')).astype('datetime64[s]')
    series = rng.standard_normal((n_t, 2, 1, h, w)).astype(np.float32)
    ds = SeriesDataset(series, {'sample': dates, 'variable': np.array(['z', 't']), 'level': np.array([500]),
                                'lat': np.linspace(80., -80., h), 'lon': np.arange(0., 360., 15.)},
                       ('sample', 'variable', 'level', 'lat', 'lon'))
    d = _build(unet_layers((4, h, w), widths=(8, 16, 16, 16, 8)), time_dim=2)
    _weights_of(d.model, rng)
    g = SeriesDataGenerator(d, ds, input_time_steps=2, output_time_steps=2, batch_size=4)
    est = TimeSeriesEstimator(d, g)
    out = est.predict(6)
    n = g._n_sample
    assert out.shape == (6, n, 2, 1, h, w)
    X, _ = g.generate([], scale_and_impute=False)
    ser = d.predict_timeseries(X, 6)                                  # (6, n, 2, h, w)
    # every row at every lead (the forecast overwrites all inputs: tests/golden/estimator.npz 'same'); the variables come
    # back in sorted label order ('t', 'z'), as the reference's unstack gives them
    assert list(out.coords['variable']) == ['t', 'z']
    assert np.array_equal(out.values[:, :, ::-1, 0], ser) and np.isfinite(out.values).all()
    # insolation as an extra input channel per time step: in != out -> host stepping around the device forward
    d2 = _build(unet_layers((6, h, w), widths=(8, 16, 16, 16, 8), cout=4), time_dim=2)
    _weights_of(d2.model, rng)
    g2 = SeriesDataGenerator(d2, ds, input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4)
    out2 = TimeSeriesEstimator(d2, g2).predict(4)
    X2, _ = g2.generate([], scale_and_impute=False)
    first = d2.predict(X2).reshape(n, 2, 2, h, w)
    assert np.array_equal(out2.values[0, :, ::-1, 0], first[:, 0]) and np.array_equal(out2.values[1, :, ::-1, 0], first[:, 1])
    assert np.isfinite(out2.values[2:, :n - 2]).all()


def test_bfloat16_activation_storage_matches_the_rounding_oracle():
    """BASELINE.json config 4: bf16 activations between the layers, fp32 at the model boundary.  Oracle = the float64
    graph with every intermediate Conv2D output rounded to bf16 (ties to even)."""
    rng = np.random.default_rng(51)
    cs = (4, 16, 24)
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((5,) + cs).astype(np.float32)
    y32 = d.predict(x)
    d.model.set_activation_dtype('bfloat16')
    # every scratch buffer is bf16 except the phase-major output of the restated 5x5 layer (read by depth-to-space)
    d2s_src = set(op.src for op in d.model.infer_plan.ops if op.kind == 'd2s')
    assert [b.dtype for b in d.model.executor.scratch(5)] == \
        [torch.float32 if i in d2s_src else torch.bfloat16 for i in range(len(d.model.infer_plan.buffers))]
    y16 = d.predict(x)
    on16 = _bf16_weight_indices(d.model, 5)
    assert 0 in on16 and len(on16) >= 3        # the first layer too: its float32 input is rounded by the loader
    want = np_ref.run_layers(layers, x, weights, bf16_activations=True, bf16_weights=on16)
    assert y16.dtype == np.float32 and y16.shape == y32.shape
    # different summation order -> a few intermediate values round to the neighbouring bf16; the effect on the output is
    # far below the bf16-vs-fp32 difference itself
    # (the restated 5x5 output layer multiplies with bf16-rounded SUMS of taps where the oracle sums bf16-rounded taps:
    # another 2^-9-relative effect of the same kind)
    assert _rel(y16, want) < 6e-3
    assert 1e-4 < _rel(y16, y32) < 3e-2
    # rollout: captured graph == host loop over predict(), bit for bit, in bf16 mode too
    series = d.predict_timeseries(x, 4)
    p, ser = x, []
    for _ in range(2):
        p = d.predict(p)
        ser.append(p)
    assert np.array_equal(series, np_ref._merge_time(np.stack(ser), 2, 5, 2, cs, False))
    # training is unaffected (float32 activations), and switching back restores the float32 results exactly
    d.model.train_on_batch(x, x)
    d.model.set_activation_dtype('float32')
    assert d.predict(x).dtype == np.float32


def test_bfloat16_mode_is_batch_invariant_and_deterministic():
    """The kernel family AND the tile instance of the bf16 matrix-core path follow from the layer alone (never from the
    batch size): a member's forecast does not depend on its batch mates, and repeated runs are bit-identical."""
    rng = np.random.default_rng(61)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs), time_dim=2)
    _weights_of(d.model, rng)
    d.model.set_activation_dtype('bfloat16')
    x = rng.standard_normal((7,) + cs).astype(np.float32)
    y = d.predict(x)
    assert np.array_equal(y, d.predict(x))
    for lo, hi in ((0, 1), (2, 5), (6, 7)):
        assert np.array_equal(d.predict(x[lo:hi]), y[lo:hi])
    d.model.set_activation_dtype('float32')


def test_bfloat16_storage_with_the_recurrent_front_end():
    from dlwp_amd.model import DLWPNeuralNet
    from tests.nets import lstm_unet_layers
    rng = np.random.default_rng(52)
    cs = (2, 4, 16, 24)                 # F = 16 recurrent channels: enough for a bf16 matrix-core K slice
    layers = lstm_unet_layers(cs, widths=(16, 32, 64, 32, 16))
    np.random.seed(5)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(layers, loss='mse', optimizer='adam')
    weights = _lstm_weights(d.model, rng)
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    d.model.set_activation_dtype('bfloat16')
    kinds = {i: b.dtype for i, b in enumerate(d.model.executor.scratch(3))}
    # cell state float32; the h sequence, the stored gate pre-activations and the convolution stack bfloat16.  The cell
    # update rides in the epilogue of the convolution that completes a step's pre-activations (dlwp_convlstm_conv_fwd):
    # input convolution on the first step, recurrent convolution on the second -- no gate kernel, one stored z tensor
    plan = d.model.infer_plan
    fused = [op for op in plan.ops if op.kind == 'conv' and op.lstm_f]
    assert len(fused) == 2 and fused[0].aux[:2] == (None, None)
    assert not any(op.kind == 'lstm' for op in plan.ops)
    # r3: with the h sequence in octets the SECOND step is one launch (dlwp_convlstm_step_fwd: recurrent + input convolution +
    # cell update) and no pre-activation tensor is stored at all; otherwise the input convolution's z is (bfloat16)
    whole = fused[1].src2 is not None
    assert whole == bool(d.model.executor._oct)
    assert [op.kind for op in plan.ops][:2 if whole else 3] == ['conv'] * (2 if whole else 3)
    h_buf, c_bufs = fused[0].dst, [op.aux[2] for op in fused]
    z_bufs = [] if whole else [fused[1].aux[0]]
    assert fused[1].dst == h_buf and fused[1].src == h_buf and fused[1].aux[1] == c_bufs[0]
    assert kinds[h_buf] == torch.bfloat16 and all(kinds[b] == torch.float32 for b in c_bufs)
    assert all(kinds[b] == torch.bfloat16 for b in z_bufs)
    on16 = _bf16_weight_indices(d.model, 3)
    parts = _bf16_lstm_parts(d.model, 3)
    assert 1 in on16 and set(parts) == {'kernel', 'recurrent'}   # both ConvLSTM convolutions and the first Conv2D
    got = d.predict(x)
    want = np_ref.run_layers(layers, x, weights, bf16_activations=True, bf16_weights=on16, bf16_lstm=parts,
                             lstm_fused='step' if whole else True)
    assert _rel(got, want) < 4e-3
    if whole:      # the two-launch steps (DLWP_LSTM_STEP=0) agree with ITS oracle, and with the one-launch step to the rounding of z
        import os as _os
        _os.environ['DLWP_LSTM_STEP'] = '0'
        try:
            d.model.set_activation_dtype('float32').set_activation_dtype('bfloat16')
            assert not any(op.src2 is not None for op in d.model.infer_plan.ops if op.kind == 'conv')
            two = d.predict(x)
            assert _rel(two, np_ref.run_layers(layers, x, weights, bf16_activations=True, bf16_weights=on16, bf16_lstm=parts,
                                               lstm_fused=True)) < 4e-3
            assert _rel(two, got) < 8e-3
        finally:
            del _os.environ['DLWP_LSTM_STEP']
            d.model.set_activation_dtype('float32').set_activation_dtype('bfloat16')
    # ... and the separate gate kernel (DLWP_LSTM_FUSE=0: the plan of the float32 mode, stored and rounded z_x, z_h) agrees
    # with it to the rounding of those tensors
    import os
    os.environ['DLWP_LSTM_FUSE'] = '0'
    try:
        d.model.set_activation_dtype('float32').set_activation_dtype('bfloat16')
        assert any(op.kind == 'lstm' for op in d.model.infer_plan.ops)
        unfused = d.predict(x)
        assert _rel(unfused, np_ref.run_layers(layers, x, weights, bf16_activations=True, bf16_weights=on16, bf16_lstm=parts)) < 4e-3
        assert _rel(unfused, got) < 8e-3
    finally:
        del os.environ['DLWP_LSTM_FUSE']
        d.model.set_activation_dtype('float32').set_activation_dtype('bfloat16')
    # rollout graph == eager forward, bit for bit, with the bf16 h sequence too
    series = d.predict_timeseries(x, 2, keep_time_dim=True)
    assert np.array_equal(np.asarray(series)[0].reshape(got.shape), got)


def test_recurrent_reference_style_example_runs_end_to_end():
    """examples/train_recurrent_and_validate.py: the reference's default flow (ConvLSTM2D front end + l2, SeriesDataGenerator
    with insolation, latitude-weighted anomaly-correlation loss, fit_generator, TimeSeriesEstimator) through the compat shim."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples',
                        'train_recurrent_and_validate.py')
    spec = importlib.util.spec_from_file_location('train_recurrent_and_validate', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ['x', '--grid', '16x24', '--times', '64', '--epochs', '2', '--batch-size', '8', '--forecast-steps', '4']
    try:
        hist, series = mod.main()
    finally:
        sys.argv = argv
    assert len(hist['loss']) == 2 and np.isfinite(hist['loss']).all() and 'val_loss' in hist
    assert series.dims == ('f_hour', 'time', 'variable', 'level', 'lat', 'lon')
    assert series.shape == (4, 13, 2, 1, 16, 24)
    assert np.isfinite(series.values[:2]).all()                        # later steps run out of data for the last samples


def test_imported_keras_hdf5_checkpoint_forecasts_like_the_oracle():
    """A Keras HDF5 checkpoint (reference DLWP/util.py:141-144 writes them; tests/golden/keras_sequential.h5 comes out of a real
    libhdf5) imported through dlwp_amd.hdf5_lite: the rebuilt model's forecast equals the oracle run on the file's weights."""
    import os
    from dlwp_amd import serialization
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    m = serialization.load_model_file(os.path.join(golden, 'keras_sequential.h5'))
    e = np.load(os.path.join(golden, 'keras_h5_expected.npz'))
    pairs = [(e['seq|conv2d_1/kernel'], e['seq|conv2d_1/bias']), (e['seq|conv2d_2/kernel'], e['seq|conv2d_2/bias'])]
    layers = (('PeriodicPadding2D', ((0, 2),), CF), ('ZeroPadding2D', ((2, 0),), CF),
              ('Conv2D', (8, 5), dict(CF, activation='tanh')), ('MaxPooling2D', (2,), CF), ('UpSampling2D', (2,), CF),
              ('PeriodicPadding2D', ((0, 1),), CF), ('ZeroPadding2D', ((1, 0),), CF),
              ('Conv2D', (2, 3), dict(CF, activation='linear')))
    x = np.random.default_rng(3).standard_normal((5, 2, 10, 12)).astype(np.float32)
    ref = np_ref.run_layers(layers, x, pairs)
    got = m.predict(x)
    assert _rel(got, ref) <= FWD_TOL
    mf = serialization.import_keras_hdf5(os.path.join(golden, 'keras_functional.h5'))
    xf = np.random.default_rng(4).standard_normal((3, 3, 8, 12)).astype(np.float32)
    k, b = e['fun|shared/kernel'], e['fun|shared/bias']
    blk = (('PeriodicPadding2D', ((0, 1),), CF), ('ZeroPadding2D', ((1, 0),), CF), ('Conv2D', (3, 3), dict(CF, activation='tanh')))
    h1 = np_ref.run_layers(blk, xf, [(k, b)])
    h2 = np_ref.run_layers(blk, h1, [(k, b)])
    cat = np.concatenate([h1, h2], axis=1)
    out = np_ref.run_layers((('PeriodicPadding2D', ((0, 2),), CF), ('ZeroPadding2D', ((2, 0),), CF),
                             ('RowConnected2D', (3, 5), dict(CF, activation='linear'))), cat,
                            [(e['fun|row/kernel'], e['fun|row/bias'])])
    assert _rel(mf.predict(xf), out) <= FWD_TOL


def test_keras_hdf5_checkpoint_written_here_resumes_training_where_it_stopped(tmp_path):
    """The reference's save_model / load_model pair (DLWP/util.py:126-192) through the Keras HDF5 files THIS package writes (r6):
    architecture, weights, training_config, iteration count and Adam moments (group optimizer_weights, Keras' order) -- a run that
    is saved after 3 steps, loaded and continued equals the uninterrupted run; the custom loss with its arrays survives too."""
    from dlwp_amd import custom, hdf5_lite, util
    from dlwp_amd.training import Adam
    rng = np.random.default_rng(8)
    cs = (4, 16, 24)
    x = [rng.standard_normal((6,) + cs).astype(np.float32) for _ in range(6)]
    climo = rng.standard_normal((1,) + cs).astype(np.float32)
    loss = custom.anomaly_correlation_loss(climo, regularize_mean='mse')
    d = _build(unet_layers(cs, widths=(8, 16, 16, 16, 8)), time_dim=2, seed=4, loss=loss, optimizer=Adam(lr=2e-3))
    _weights_of(d.model, rng)
    w_start = d.model.get_weights()
    for k in range(3):
        d.model.train_on_batch(x[k], x[k])
    base = str(tmp_path / 'ckpt')
    util.save_model(d, base)
    assert hdf5_lite.is_hdf5(base + '.keras')
    f = hdf5_lite.File(base + '.keras')
    names = [n.decode() for n in np.asarray(f['optimizer_weights'].attrs['weight_names']).reshape(-1)]
    n_par = len(d.model._trainer.entries)
    assert names[0] == 'Adam/iterations:0' and len(names) == 1 + 3 * n_par
    assert int(np.asarray(f['optimizer_weights'][names[0]][...]).reshape(-1)[0]) == 3
    d2 = util.load_model(base)
    assert isinstance(d2.model.loss, custom.LossSpec) and np.array_equal(d2.model.loss.mean, climo[0])
    assert d2.model.optimizer.iterations == 3 and d2.model.optimizer.lr == pytest.approx(2e-3)
    assert all(np.array_equal(a, b) for a, b in zip(d.model.get_weights(), d2.model.get_weights()))
    for a, b in zip(d.model._trainer.opt_state, d2.model._trainer.opt_state):
        assert torch.equal(a, b)
    for k in range(3, 6):
        la, lb = d.model.train_on_batch(x[k], x[k]), d2.model.train_on_batch(x[k], x[k])
        assert np.allclose(la, lb, rtol=1e-6, atol=1e-7)
    for a, b in zip(d.model.get_weights(), d2.model.get_weights()):
        assert np.abs(a - b).max() <= 1e-7
    assert not all(np.array_equal(a, b) for a, b in zip(w_start, d2.model.get_weights()))


def test_host_series_come_from_recycled_pinned_buffers_without_aliasing_a_live_result():
    """predict_timeseries (streamed return) gives a numpy array on page-locked memory that goes back to a pool when the caller
    lets it go (util._PinnedPool): a result that is still referenced -- even through a view -- is never overwritten by a later
    call, and a released one is reused (same address) instead of page-locking 1.8 GB again."""
    import gc
    from dlwp_amd import util
    rng = np.random.default_rng(3)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs, widths=(8, 16, 16, 16, 8)), time_dim=2)
    d.host_stream_bytes = 0                      # the streamed return (its result array comes from the pool)
    x1 = rng.standard_normal((16,) + cs).astype(np.float32)
    x2 = rng.standard_normal((16,) + cs).astype(np.float32)
    a = d.predict_timeseries(x1, 4)
    keep = a[1].copy()
    view = a[1]                                   # a view keeps the whole buffer on loan
    addr = a.ctypes.data
    del a
    gc.collect()
    b = d.predict_timeseries(x2, 4)              # must NOT land in the buffer `view` still looks at
    assert b.ctypes.data != addr and np.array_equal(view, keep)
    del view
    gc.collect()
    c = d.predict_timeseries(x1, 4)              # the first buffer is free again: reused
    assert c.ctypes.data == addr and np.array_equal(c[1], keep)
    assert util.pinned_results.per_size >= 2
