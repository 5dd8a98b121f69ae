"""Split-K launches of the Winograd kernel on small grids (csrc/conv_fwd_k3d1s.hip, DLWP_OPT_SPLITK): the input
channels of an output tile are divided over S workgroups, the last arrival sums the S partial tiles in index order, adds the bias,
activates, pools and stores.  Parity: every split count within the float32 tolerance of the float64 oracle AND of the unsplit
launch; the same split count twice -> the same bits (the order of the sum never depends on the arrival order); the counters are
back at zero after every launch (back-to-back launches).  Reference semantics: Keras Conv2D as assembled in examples/train.py:164-219."""
import numpy as np
import pytest
import torch

from oracle import np_ref

pytestmark = pytest.mark.gpu

CONV_RTOL = 1e-5


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from dlwp_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def _ref(x, w, b, pads, mh, mw, act, src):
    xs = np.asarray(x, np.float64)
    if src == 1:
        xs = np_ref.upsample2(xs)
    return np_ref.conv2d(np_ref.pad2d_modes(xs, pads, mh, mw), w, b, 1, act)


def _close(got, want, what):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape, (got.shape, want.shape, what)
    assert err <= CONV_RTOL * scale, (what, err, scale)


# n, cin, h, w, cout, mode_h, mode_w, act, src_mode, out_pool
CASES = [
    (2, 64, 22, 45, 128, 0, 1, 'tanh', 0, False),      # U-Net layer 3: an even batch -> sample pairs side by side
    (1, 64, 22, 45, 128, 0, 1, 'tanh', 0, False),      # ... one member: no pairs, ragged last column tile
    (3, 32, 44, 90, 64, 0, 1, 'tanh', 0, True),        # layer 2 with MaxPooling2D(2) in the epilogue (the finishing block pools)
    (2, 128, 22, 45, 64, 0, 1, 'relu', 1, False),      # layer 4: up-sampled source, 9 live Winograd positions
    (2, 64, 44, 90, 32, 0, 1, 'tanh', 0, False),       # layer 5 restated on the 44 x 90 tensor
    (2, 22, 19, 50, 64, 2, 1, 'linear', 0, False),     # ragged input channels (3 chunks, the last one with 6 of 8), edge rows
    (2, 40, 17, 33, 32, 1, 1, 'tanh', 0, True),        # odd map under a pooling epilogue, periodic in both axes
]


def _split_instances(ops):
    """registry indices of the Winograd instances with a compiled split-K variant (dlwp_conv2d_config_flags bit 5)"""
    return [i for i, c in enumerate(ops.conv_configs()) if c[5] == 0 and c[10] & 32]


@pytest.mark.parametrize('case', CASES)
def test_split_counts_match_the_oracle_and_the_unsplit_launch(ops, case):
    """every instance with a split variant that covers the layer, forced, at split counts 2 ... 16"""
    from dlwp_amd._lib import DlwpError
    n, cin, h, w, cout, mh, mw, act, src, pool = case
    rng = np.random.default_rng(7000 + CASES.index(case))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    want = _ref(x, wt, b, (1, 1, 1, 1), mh, mw, act, src)
    if pool:
        want = np_ref.maxpool2(want)
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, mh, mw), ops.ACTIVATIONS[act], src_mode=src, out_pool=pool)
    xd, wd, bd = dev(x), dev(wt), dev(b)
    chunks = -(-cin // 8)
    prev = ops.set_splitk(0)
    tried = 0
    try:
        for cfg in _split_instances(ops):
            ops.force_conv_config(cfg)
            ops.set_splitk(0)
            try:
                base = ops.conv2d(xd, wd, bd, cd).clone()
            except DlwpError:
                continue                       # this instance does not cover the layer (64-channel blocks: 9-position layers only)
            assert ops.conv_split_count(x.shape, cd) == 1
            _close(host(base), want, 'config %d unsplit' % cfg)
            tried += 1
            for s in (2, 3, 4, 8, 16):
                ops.set_splitk(s)
                eff = ops.conv_split_count(x.shape, cd)
                if eff == 1 and ops.conv_configs()[cfg][6] == 4:
                    break                      # 64-channel blocks split only in their 9-position form (as they are registered)
                assert 2 <= eff <= min(s, chunks), (cfg, s, eff, chunks)
                outs = [ops.conv2d(xd, wd, bd, cd).clone() for _ in range(3)]      # back to back: the counters reset themselves
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ('not reproducible', cfg, s)
                _close(host(outs[0]), want, 'config %d split %d' % (cfg, s))
                _close(host(outs[0]), host(base), 'config %d split %d vs unsplit' % (cfg, s))
    finally:
        ops.force_conv_config(-1)
        ops.set_splitk(prev)
    assert tried >= 2


def test_split_launch_into_channel_windows(ops):
    """reads 24 of 40 stored channels from channel 8, writes its 32 outputs into channels 16.. of a 64-channel tensor"""
    rng = np.random.default_rng(7100)
    n, h, w = 2, 20, 40
    x = rng.standard_normal((n, 40, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, 24, 32), rng)
    b = (0.1 * rng.standard_normal(32)).astype(np.float32)
    cd = ops.make_conv(32, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, in_c_off=8, in_c_total=40, out_c_off=16,
                       out_c_total=64)
    want = _ref(x[:, 8:32], wt, b, (1, 1, 1, 1), 0, 1, 'tanh', 0)
    prev = ops.set_splitk(3)
    ops.force_conv_config(_split_instances(ops)[0])
    try:
        assert ops.conv_split_count((n, 24, h, w), cd) == 3
        y = torch.full((n, 64, h, w), 7.0, device='cuda')
        ops.conv2d(dev(x), dev(wt), dev(b), cd, out=y, x_channels=24)
    finally:
        ops.force_conv_config(-1)
        ops.set_splitk(prev)
    _close(host(y[:, 16:48]), want, 'channel windows')
    assert float(y[:, :16].min()) == 7.0 and float(y[:, 48:].max()) == 7.0


def test_split_data_gradients(ops):
    """the data gradients run the forward kernels (flipped kernels; the 2x2-sum epilogue for an up-sampled source): split launches
    of both against the unsplit ones"""
    from dlwp_amd import _lib
    rng = np.random.default_rng(7200)
    for src, (n, cin, h, w, cout) in ((0, (2, 64, 22, 46, 128)), (1, (2, 128, 22, 45, 64))):
        ho, wo = (2 * h, 2 * w) if src else (h, w)
        wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
        dz = rng.standard_normal((n, cout, ho, wo)).astype(np.float32)
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR, src_mode=src)
        xs = _lib.Shape4(n, cin, h, w)
        res = []
        prev = ops.set_splitk(0)
        ops.force_conv_config(_split_instances(ops)[0])
        try:
            for s in (0, 4):
                ops.set_splitk(s)
                dx = torch.empty((n, cin, h, w), device='cuda')
                if src:
                    assert ops.conv2d_bwd_data_stored(dev(dz), dev(wt), cd, xs, dx)
                else:
                    ops.conv2d_bwd_data(dev(dz), dev(wt), cd, xs, dx)
                res.append(host(dx))
        finally:
            ops.force_conv_config(-1)
            ops.set_splitk(prev)
        assert np.isfinite(res[1]).all()
        _close(res[1], res[0], 'data gradient, src_mode %d' % src)


def test_rule_splits_long_chains_on_tiny_grids_only(ops):
    """DLWP_OPT_SPLITK = 1: the exchange between the workgroups of a tile costs ~5 us, a chunk of 8 input channels ~1.2 us of a
    workgroup's life -- the rule splits (S = 3) launches of at least 12 chunks on at most a third of a workgroup per CU, nothing
    else: from 4 members of the 88 x 180 grid on, and for every layer of at most 88 input channels, a sample's bits do not depend on
    its batch size at all"""
    cd4 = ops.make_conv(64, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, src_mode=1)       # layer 4: 128 -> 64, up-sampled
    assert ops.conv_split_count((1, 128, 22, 45), cd4) == 1          # r5: OFF unless asked for (batch-invariant bits by default)
    prev = ops.set_splitk(1)
    assert prev == 0
    try:
        assert ops.conv_split_count((1, 128, 22, 45), cd4) == 3
        assert ops.conv_split_count((2, 128, 22, 45), cd4) == 3
        for n in (4, 8, 64, 256, 1024):
            assert ops.conv_split_count((n, 128, 22, 45), cd4) == 1
        cd3 = ops.make_conv(128, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)                    # layer 3: 8 chunks
        for n in (1, 2, 8, 256):
            assert ops.conv_split_count((n, 64, 22, 45), cd3) == 1
        cd1 = ops.make_conv(32, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH, out_pool=True)     # 4 input channels: one chunk
        assert ops.conv_split_count((1, 4, 88, 180), cd1) == 1
    finally:
        ops.set_splitk(prev)


def test_rollout_graph_with_split_launches_equals_the_eager_loop_and_the_unsplit_rollout(ops):
    """2 members of the config-2 U-Net (its Winograd layers split): the captured rollout (split-K regions of its own in the caller's
    workspace) equals the eager forwards bit for bit, and the first forward of the unsplit rollout to float32 round-off; against the
    float64 oracle both keep the forward tolerance"""
    from dlwp_amd.model import DLWPNeuralNet
    from tests.nets import unet_layers
    cs = (4, 88, 180)
    layers = unet_layers(cs)
    np.random.seed(3)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(layers, loss='mse', optimizer='adam', metrics=['mae'])
    m = d.model
    rng = np.random.default_rng(11)
    ws = m.get_weights()
    pairs = [(ws[i], (0.1 * rng.standard_normal(ws[i + 1].shape)).astype(np.float32)) for i in range(0, len(ws), 2)]
    m.set_weights([a for p in pairs for a in p])
    x = rng.standard_normal((2,) + cs).astype(np.float32)
    x0 = dev(x)
    calls = 3
    outs = {}
    prev = ops.set_splitk(1)
    try:
        for mode in (1, 0):
            ops.set_splitk(mode)
            series = m.rollout_on_device(x0, calls, graph_cache=False).clone()
            outs[mode] = series
            eager = x0
            for t in range(calls):
                eager = m.predict_on_device(eager)
                assert torch.equal(series[t].reshape(eager.shape), eager), 'graph != eager at forward %d (split mode %d)' % (t, mode)
    finally:
        ops.set_splitk(prev)
    a, b = host(outs[1]), host(outs[0])
    want = np_ref.run_layers(layers, x, pairs)
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(a[0] - want).max()) <= 2e-5 * scale and float(np.abs(b[0] - want).max()) <= 2e-5 * scale
    assert float(np.abs(a[0] - b[0]).max()) <= 2e-5 * scale
