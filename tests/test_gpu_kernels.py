"""GPU parity tests proper: every libdlwp_hip.so kernel, called through the C ABI, against the oracle on the same
seeded inputs and against the committed golden fixtures.  Bit-exact for padding / pooling / copies; fp32 convolution
within 1e-5 of the float64 direct sum (relative to the output scale; the tolerance BASELINE.md states)."""
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


# ----------------------------------------------------------------------------------------------------------------- #
# padding
# ----------------------------------------------------------------------------------------------------------------- #

def _pad_struct(ops, padding, mode):
    (t, b), (l, r) = padding
    return ops.make_pad(t, b, l, r, mode, mode)


def test_pad2d_golden_periodic_and_fill(ops, golden):
    g = golden('padding')
    x_cf, x_cl = g['x_cf'], g['x_cl']
    for kind, mode in (('periodic', ops.PAD_WRAP), ('fill', ops.PAD_EDGE)):
        for i in range(int(g['%s_n' % kind])):
            padding = tuple(map(tuple, g['%s_%d_padding' % (kind, i)]))
            (t, b), (l, r) = padding
            if kind == 'periodic' and (max(t, b) > x_cf.shape[2] or max(l, r) > x_cf.shape[3]):
                continue
            p = _pad_struct(ops, padding, mode)
            assert np.array_equal(host(ops.pad2d(dev(x_cf), p)), g['%s_%d_cf' % (kind, i)]), (kind, padding)
            assert np.array_equal(host(ops.pad2d(dev(x_cl), p, channels_last=True)), g['%s_%d_cl' % (kind, i)])


def test_pad2d_golden_composite_halo(ops, golden):
    g = golden('padding')
    for k in (1, 2):
        p = ops.make_pad(k, k, k, k, ops.PAD_ZERO, ops.PAD_WRAP)
        assert np.array_equal(host(ops.pad2d(dev(g['x_cf']), p)), g['composite_pz_%d' % k])


def test_pad2d_rejects_periodic_pad_larger_than_axis(ops):
    from dlwp_amd._lib import DlwpError
    x = torch.zeros((1, 1, 3, 4), device='cuda')
    with pytest.raises(DlwpError, match='periodic column padding'):
        ops.pad2d(x, ops.make_pad(0, 0, 5, 5, ops.PAD_ZERO, ops.PAD_WRAP))


@pytest.mark.parametrize('shape', [(3, 5, 7, 9), (2, 4, 88, 180), (1, 3, 73, 144), (2, 2, 16, 20),
                                   # r5: rows that are not whole 16-byte units -- the flat-vector instances of pad2d_fwd / pad2d_bwd
                                   # (aligned windows in, groups of 4 rows out; csrc/halo.hip) -- on the U-Net's inner maps
                                   (2, 3, 22, 45), (2, 2, 44, 90), (5, 1, 3, 5), (1, 1, 1, 7)])
def test_pad2d_all_modes_fwd_bwd(ops, shape):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    for mh in (0, 1, 2, 3, 4):           # zero, periodic, edge, tf.pad REFLECT, tf.pad SYMMETRIC
        for mw in (0, 1, 2, 3, 4):
            pads = (2, 1, 3, 2) if shape[-1] % 4 else (2, 2, 2, 2)
            if shape[-2] < 3:
                pads = (1, 0, 3, 2) if mh in (1, 2, 4) else (0, 0, 3, 2)      # (a 1-row axis: what each mode admits)
            p = ops.make_pad(*pads, mh, mw)
            want = np_ref.pad2d_modes(x, pads, mh, mw)
            got = host(ops.pad2d(dev(x), p))
            assert np.array_equal(got, want), (mh, mw)
            dy = rng.standard_normal(want.shape).astype(np.float32)
            dx = host(ops.pad2d_bwd(dev(dy), shape, p))
            dx_ref = np_ref.pad2d_modes_grad(dy, shape, pads, mh, mw)
            assert np.abs(dx - dx_ref).max() <= 1e-5 * max(1., np.abs(dx_ref).max())


@pytest.mark.parametrize('c,h,w,k', [(32, 44, 90, 1), (64, 22, 45, 1), (128, 44, 90, 1)])
def test_pad2d_composite_halos_of_the_unets_inner_maps(ops, c, h, w, k):
    """Periodic(0, k) + Zero(k, 0) on the 44 x 90 and 22 x 45 maps (92- / 47-float rows: the flat-vector instances), 7 samples so
    that the last group of four rows is partial: bit-exact forward, adjoint backward."""
    rng = np.random.default_rng(c + w)
    x = rng.standard_normal((7, c, h, w)).astype(np.float32)
    p = ops.make_pad(k, k, k, k, ops.PAD_ZERO, ops.PAD_WRAP)
    want = np_ref.pad2d_modes(x, (k, k, k, k), 0, 1)
    assert np.array_equal(host(ops.pad2d(dev(x), p)), want)
    dy = rng.standard_normal(want.shape).astype(np.float32)
    dx = host(ops.pad2d_bwd(dev(dy), x.shape, p))
    dx_ref = np_ref.pad2d_modes_grad(dy, x.shape, (k, k, k, k), 0, 1)
    assert np.abs(dx - dx_ref).max() <= 1e-5 * max(1., np.abs(dx_ref).max())


def test_pad2d_empty_and_single(ops):
    p = ops.make_pad(1, 1, 1, 1, ops.PAD_ZERO, ops.PAD_WRAP)
    y = ops.pad2d(torch.zeros((0, 3, 4, 5), device='cuda'), p)
    assert tuple(y.shape) == (0, 3, 6, 7)
    x = np.arange(1, dtype=np.float32).reshape(1, 1, 1, 1) + 5
    assert np.array_equal(host(ops.pad2d(dev(x), p)), np_ref.pad2d_modes(x, (1, 1, 1, 1), 0, 1))


# ----------------------------------------------------------------------------------------------------------------- #
# pooling / up-sampling / copies
# ----------------------------------------------------------------------------------------------------------------- #

@pytest.mark.parametrize('shape', [(2, 3, 8, 10), (1, 2, 7, 9), (2, 32, 88, 180), (1, 5, 45, 22)])
def test_maxpool2_and_upsample2(ops, shape):
    rng = np.random.default_rng(shape[-1])
    x = rng.standard_normal(shape).astype(np.float32)
    y = host(ops.maxpool2(dev(x)))
    assert np.array_equal(y, np_ref.maxpool2(x))
    dy = rng.standard_normal(y.shape).astype(np.float32)
    assert np.array_equal(host(ops.maxpool2_bwd(dev(x), dev(dy))), np_ref.maxpool2_grad(x, dy).astype(np.float32))
    u = host(ops.upsample2(dev(x)))
    assert np.array_equal(u, np_ref.upsample2(x))
    du = rng.standard_normal(u.shape).astype(np.float32)
    got = host(ops.upsample2_bwd(dev(du)))
    assert np.abs(got - np_ref.upsample2_grad(du)).max() < 1e-5


def test_maxpool2_bwd_ties_route_to_first_maximum(ops):
    x = np.zeros((1, 1, 4, 4), np.float32)
    dy = np.arange(1, 5, dtype=np.float32).reshape(1, 1, 2, 2)
    got = host(ops.maxpool2_bwd(dev(x), dev(dy)))
    assert np.array_equal(got, np_ref.maxpool2_grad(x, dy).astype(np.float32))
    assert got[0, 0, 0, 0] == 1 and got[0, 0, 0, 1] == 0


def test_copy_channels_slice_and_concat(ops):
    rng = np.random.default_rng(11)
    a = rng.standard_normal((3, 32, 6, 10)).astype(np.float32)
    b = rng.standard_normal((3, 16, 6, 10)).astype(np.float32)
    # slice_layer(16, 32) of a, concatenated behind b  (train_functional.py:203-206, 259)
    out = torch.full((3, 32, 6, 10), float('nan'), device='cuda')
    ops.copy_channels(dev(b), out, 16, 0, 0)
    ops.copy_channels(dev(a), out, 16, 16, 16)
    assert np.array_equal(host(out), np.concatenate([b, a[:, 16:32]], axis=1))
    # odd sizes take the scalar path
    a2 = rng.standard_normal((2, 5, 3, 3)).astype(np.float32)
    out2 = torch.zeros((2, 7, 3, 3), device='cuda')
    ops.copy_channels(dev(a2), out2, 3, 1, 4)
    want = np.zeros((2, 7, 3, 3), np.float32)
    want[:, 4:7] = a2[:, 1:4]
    assert np.array_equal(host(out2), want)


@pytest.mark.parametrize('t,n,td,v,h,w', [(3, 4, 2, 2, 5, 6), (2, 3, 3, 1, 4, 4), (4, 1, 1, 4, 8, 12), (5, 2, 2, 3, 3, 5)])
def test_series_merge_time_matches_reference_reshape(ops, t, n, td, v, h, w):
    rng = np.random.default_rng(t * 100 + n)
    s = rng.standard_normal((t, n, td * v, h, w)).astype(np.float32)
    want = np_ref._merge_time(s, t, n, td, (td * v, h, w), False)
    got = host(ops.series_merge_time(dev(s), td))
    assert got.shape == want.shape and np.array_equal(got, want)


# ----------------------------------------------------------------------------------------------------------------- #
# convolution
# ----------------------------------------------------------------------------------------------------------------- #

def _conv_ref(x, w, b, dil, pads, mh, mw, act, src_mode):
    xs = np.asarray(x, np.float64)
    if src_mode == 1:
        xs = np_ref.upsample2(xs)
    elif src_mode == 2:
        xs = np_ref.maxpool2(xs)
    xp = np_ref.pad2d_modes(xs, pads, mh, mw)
    return np_ref.conv2d(xp, w, b, dil, act)


def _check_conv(ops, got, want, what=''):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape, (got.shape, want.shape, what)
    assert err <= CONV_RTOL * scale, (what, err, scale)


CASES = [
    # (n, cin, h, w, cout, k, dil, pads(t,b,l,r), mode_h, mode_w, act, src_mode)
    (2, 4, 16, 36, 32, 3, 2, (2, 2, 2, 2), 0, 1, 'tanh', 0),          # U-Net L1
    (2, 8, 16, 36, 24, 3, 1, (1, 1, 1, 1), 0, 1, 'tanh', 2),          # pooled input
    (2, 16, 6, 10, 40, 3, 1, (1, 1, 1, 1), 0, 1, 'tanh', 1),          # up-sampled input
    (1, 32, 16, 36, 4, 5, 1, (2, 2, 2, 2), 0, 1, 'linear', 0),        # output layer, cout=4
    (1, 2, 13, 17, 32, 5, 1, (2, 2, 2, 2), 0, 1, 'tanh', 0),          # config-1 first layer, odd sizes, cin=2
    (2, 5, 9, 11, 7, 3, 2, (2, 1, 3, 2), 2, 1, 'relu', 0),            # asymmetric, edge rows, odd channels
    (1, 6, 12, 12, 16, 3, 1, (1, 1, 1, 1), 1, 1, 'linear', 0),        # periodic in both axes
    (1, 4, 10, 14, 8, 3, 1, (0, 0, 0, 0), 0, 0, 'linear', 0),         # plain valid conv, output smaller than input
    (1, 12, 10, 20, 32, 3, 2, (2, 2, 2, 2), 2, 2, 'tanh', 0),         # fill both axes, cin=12
    (3, 4, 8, 36, 32, 3, 2, (2, 2, 2, 2), 0, 1, 'tanh', 0),
    (2, 8, 12, 20, 32, 3, 1, (1, 1, 1, 1), 3, 3, 'tanh', 0),          # TFPadding2D REFLECT halo, Winograd family
    (2, 6, 9, 14, 20, 5, 1, (2, 2, 2, 2), 4, 3, 'linear', 0),          # SYMMETRIC rows, REFLECT columns, direct family
    (2, 8, 6, 10, 32, 3, 1, (1, 1, 1, 1), 4, 4, 'tanh', 1),            # SYMMETRIC halo on an up-sampled source
    (1, 8, 12, 20, 16, 3, 2, (2, 2, 2, 2), 3, 1, 'relu', 2),           # REFLECT rows on a pooled source, dilation 2
]


@pytest.mark.parametrize('case', CASES)
def test_conv2d_fused_vs_float64_oracle(ops, case):
    n, cin, h, w, cout, k, dil, pads, mh, mw, act, src = case
    rng = np.random.default_rng(1000 + CASES.index(case))      # fixed per case: a failure reproduces with the same inputs
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    want = _conv_ref(x, wt, b, dil, pads, mh, mw, act, src)
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mh, mw), ops.ACTIVATIONS[act], src_mode=src)
    got = host(ops.conv2d(dev(x), dev(wt), dev(b), cd))
    _check_conv(ops, got, want, 'mfma')
    got_d = host(ops.conv2d(dev(x), dev(wt), dev(b), cd, direct=True))
    _check_conv(ops, got_d, want, 'direct')


def test_conv2d_every_compiled_tile_configuration(ops):
    """Force each MFMA tile configuration in turn on a shape with ragged tile edges and ragged channel counts."""
    rng = np.random.default_rng(99)
    cfgs = ops.conv_configs()
    problems = {}
    try:
        for i, (ks, dil, th, tw, waves, fa, bnf, ck, pool, lds, flags) in enumerate(cfgs):
            if pool >= 2:
                continue                                # bf16 matrix-core instances: test_conv2d_bf16_mfma_* below
            cmax = 16 // (-bnf) if bnf < 0 else 0       # packed-N instances cover cout <= 16/S
            wino = fa == 0                              # Winograd instances: whole chunks of 8 in / 32 out channels
            plain_split = (flags & 5) == 1              # position-split Winograd, own arithmetic: layers WITHOUT whole 32-channel tiles
            key = (ks, dil, pool, cmax, wino, plain_split)
            src = 2 if pool else 0                      # pooled-loader instances only run the fused max-pool source
            if key not in problems:
                n, cin, h, w, cout = 2, 20, (39 if pool else 19), (101 if pool else 50), (36 if not cmax else max(1, cmax - 1))
                if wino:
                    cin, cout = 24, (48 if plain_split else 64)
                x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
                wt = np_ref.glorot_uniform((ks, ks, cin, cout), rng)
                b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
                p = dil * (ks - 1) // 2
                pads = (p, p, p, p)
                want = _conv_ref(x, wt, b, dil, pads, 0, 1, 'tanh', src)
                problems[key] = (dev(x), dev(wt), dev(b), pads, want, cout)
            xd, wd, bd, pads, want, cout = problems[key]
            ops.force_conv_config(i)
            cd = ops.make_conv(cout, ks, ks, dil, ops.make_pad(*pads, 0, 1), ops.ACT_TANH, src_mode=src)
            got = host(ops.conv2d(xd, wd, bd, cd))
            _check_conv(ops, got, want, 'config %d %r' % (i, cfgs[i]))
    finally:
        ops.force_conv_config(-1)


@pytest.mark.parametrize('dil', [1, 2])
def test_few_channel_streaming_kernel_equals_the_general_instance(ops, dil):
    """csrc/conv_fwd_few.hip (pooled 3x3 layers of at most four input channels; weights in registers, a workgroup walks over the
    samples of one tile position): the bits of the direct family's instance it stands in for, and the float64 oracle -- ragged
    tiles in both directions, 1 / 3 / 4 input channels, 16 / 32 / 40 output channels, every activation, wrap / zero / edge halos,
    channel windows on both sides, a batch that leaves workgroups with one item and with several (position changes inside a
    workgroup's share)."""
    rng = np.random.default_rng(400 + dil)
    p = dil
    cases = [  # n, cin, h, w, cout, mode_h, mode_w, act
        (5, 4, 20, 52, 32, 0, 1, 'tanh'),
        (3, 3, 18, 76, 40, 2, 1, 'relu'),
        (70, 1, 10, 36, 16, 1, 0, 'linear'),
        (300, 4, 12, 68, 32, 0, 1, 'tanh'),
        (4, 2, 16, 44, 32, 0, 2, 'tanh'),       # edge columns: the dword loader (no aligned 16-byte quads across the halo)
        (3, 4, 24, 72, 32, 0, 1, 'relu'),       # pooled width 36: 16-byte stores
        # r6: 5-8 input channels = two groups of four (the 6-channel first layer of examples/validate.py's network)
        (5, 6, 20, 52, 32, 0, 1, 'tanh'),
        (300, 6, 12, 68, 32, 0, 1, 'tanh'),
        (70, 5, 18, 76, 64, 2, 1, 'relu'),
        (4, 8, 16, 44, 32, 0, 2, 'linear'),
        (40, 7, 24, 72, 32, 1, 0, 'tanh'),
    ]
    for n, cin, h, w, cout, mh, mw, act in cases:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
        b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(p, p, p, p, mh, mw), ops.ACTIVATIONS[act], out_pool=True)
        xd, wd, bd = dev(x), dev(wt), dev(b)
        prev = ops.set_few_stream(0)
        try:
            base = ops.conv2d(xd, wd, bd, cd)
            assert ops.conv_launch_info(x.shape, cd)[0][0] >= 0
            ops.set_few_stream(2)
            info = ops.conv_launch_info(x.shape, cd)
            assert info[0][0] == -2 and info[0][2] == 256, info
            got = ops.conv2d(xd, wd, bd, cd)
        finally:
            ops.set_few_stream(prev)
        assert torch.equal(got, base), (n, cin, h, w, cout, float((got - base).abs().max()))
        if n <= 5:
            want = np_ref.maxpool2(_conv_ref(x, wt, b, dil, (p, p, p, p), mh, mw, act, 0))
            _check_conv(ops, host(got), want, 'few-channel stream %r' % ((n, cin, h, w, cout),))
    # channel windows: 3 of 6 stored input channels from channel 2, the 32 outputs into channels 8.. of a 48-channel tensor
    n, h, w = 9, 16, 40
    x6 = rng.standard_normal((n, 6, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, 3, 32), rng)
    b = (0.1 * rng.standard_normal(32)).astype(np.float32)
    cd = ops.make_conv(32, 3, 3, dil, ops.make_pad(p, p, p, p, 0, 1), ops.ACT_TANH, in_c_off=2, in_c_total=6, out_c_off=8,
                       out_c_total=48, out_pool=True)
    outs = []
    prev = ops.set_few_stream(0)
    try:
        for mode in (0, 2):
            ops.set_few_stream(mode)
            y = torch.full((n, 48, h // 2, w // 2), 7.0, device='cuda')
            ops.conv2d(dev(x6), dev(wt), dev(b), cd, out=y, x_channels=3)
            outs.append(y)
    finally:
        ops.set_few_stream(prev)
    assert torch.equal(outs[0], outs[1])
    want = np_ref.maxpool2(_conv_ref(x6[:, 2:5], wt, b, dil, (p, p, p, p), 0, 1, 'tanh', 0))
    _check_conv(ops, host(outs[1][:, 8:40]), want, 'few-channel stream, channel windows')
    assert float(outs[1][:, :8].min()) == 7.0 and float(outs[1][:, 40:].max()) == 7.0


@pytest.mark.parametrize('dil', [1, 2])
def test_few_channel_streaming_kernel_unpooled_and_both_outputs(ops, dil):
    """r4: the streaming kernel also stores the layer's own output -- alone (a first layer without pooling: the
    91 x 180 sub-record) or beside its MaxPooling2D(2) image (the training forward, dlwp_conv2d_fwd_pool2).  The bits of the
    general instance in both modes; odd heights (91 rows), ragged channel tiles, every activation, widths that cut the last tile."""
    rng = np.random.default_rng(500 + dil)
    p = dil
    cases = [  # n, cin, h, w, cout, mode_h, mode_w, act
        (40, 4, 91, 180, 32, 0, 1, 'tanh'),
        (70, 3, 18, 76, 40, 2, 1, 'relu'),
        (300, 1, 10, 36, 16, 1, 0, 'linear'),
        (90, 4, 24, 72, 32, 0, 1, 'tanh'),
        (40, 6, 91, 180, 32, 0, 1, 'tanh'),      # r6: two channel groups
        (90, 8, 24, 72, 32, 0, 1, 'relu'),
    ]
    for n, cin, h, w, cout, mh, mw, act in cases:
        x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
        b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
        cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(p, p, p, p, mh, mw), ops.ACTIVATIONS[act])
        res = {}
        prev = ops.set_few_stream(0)
        try:
            for mode in (0, 2):
                ops.set_few_stream(mode)
                info = ops.conv_launch_info(tuple(x.shape), cd)
                assert (info[0][0] == -2) == (mode == 2), (mode, info)
                y = ops.conv2d(x, wt, b, cd).clone()
                y2 = torch.full((n, cout, h, w), 7.0, device='cuda')
                pl = torch.full((n, cout, h // 2, w // 2), 7.0, device='cuda')
                both = ops.conv2d(x, wt, b, cd, out=y2, out_pool2=pl)
                res[mode] = (y, y2, pl, both is not None)
        finally:
            ops.set_few_stream(prev)
        assert torch.equal(res[0][0], res[2][0]), ('unpooled', n, cin, h, w, cout)
        assert res[0][3] and res[2][3]
        assert torch.equal(res[0][1], res[2][1]) and torch.equal(res[0][2], res[2][2]), ('both outputs', n, cin, h, w, cout)
        assert torch.equal(res[2][1], res[2][0]) and torch.equal(res[2][2], ops.maxpool2(res[2][0]))
        if n <= 40:
            want = _conv_ref(host(x)[:2], host(wt), host(b), dil, (p, p, p, p), mh, mw, act, 0)
            _check_conv(ops, host(res[2][0])[:2], want, 'streaming kernel, unpooled')


def test_few_channel_streaming_kernel_is_chosen_by_batch_size(ops):
    """DLWP_OPT_FEW_STREAM = 1 (default): layer 1 of the 88 x 180 U-Net goes to the streaming kernel from 2.5 tiles per resident
    workgroup on (3 per CU), the general instance below -- with the pooling epilogue; the unpooled output (r4: a width that is a
    multiple of 4) only with DLWP_OPT_FEW_STREAM = 2; never for more than eight input channels (r6: 5-8 run as two groups of four,
    and such a layer belongs to the direct family at EVERY batch size -- a member's bits do not depend on its batch), 5x5, or a
    width that cuts a quad."""
    cd = ops.make_conv(32, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH, out_pool=True)
    assert ops.conv_launch_info((256, 4, 88, 180), cd)[0][0] == -2
    g = ops.conv_launch_info((256, 4, 88, 180), cd)[0]
    assert g[2] == 256 and g[1] % 3 == 0 and g[3] == 2.0 * 256 * 32 * 36 * 11 * 6 * 256      # 3 workgroups per CU; 72 MFMAs per wave and tile
    assert ops.conv_launch_info((64, 4, 88, 180), cd)[0][0] == -2
    assert ops.conv_launch_info((16, 4, 88, 180), cd)[0][0] >= 0
    assert ops.conv_launch_info((1, 4, 88, 180), cd)[0][0] >= 0
    g6 = ops.conv_launch_info((256, 6, 88, 180), cd)[0]
    assert g6[0] == -2 and g6[3] == 2 * g[3]                                   # 144 MFMAs per wave and tile
    cfgs = ops.conv_configs()
    small = ops.conv_launch_info((8, 6, 88, 180), cd)[0]
    assert small[0] >= 0 and cfgs[small[0]][5] > 0                             # the general DIRECT instance (not Winograd: fa > 0)
    assert ops.conv_launch_info((256, 9, 88, 180), cd)[0][0] >= 0
    plain = ops.make_conv(32, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH)
    assert ops.conv_launch_info((256, 4, 88, 180), plain)[0][0] >= 0          # (measured no faster unpooled: only when asked for)
    prev = ops.set_few_stream(2)
    try:
        assert ops.conv_launch_info((256, 4, 88, 180), plain)[0][0] == -2
        assert ops.conv_launch_info((256, 4, 91, 180), plain)[0][0] == -2
        assert ops.conv_launch_info((256, 4, 88, 178), plain)[0][0] >= 0      # 178 columns: the last quad is cut
    finally:
        ops.set_few_stream(prev)
    five = ops.make_conv(32, 5, 5, 1, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH, out_pool=True)
    assert ops.conv_launch_info((256, 4, 88, 180), five)[0][0] != -2
    prev = ops.set_few_stream(0)
    try:
        assert ops.conv_launch_info((256, 4, 88, 180), cd)[0][0] >= 0
    finally:
        ops.set_few_stream(prev)


def test_winograd_nine_position_variants_of_every_instance(ops):
    """The WinoCfg::UPS variants (up-sampled source with an odd halo; 2x2 summing epilogue) of every dilation-1 Winograd
    instance, forced in turn, against the float64 oracle -- and bit-identical across instances."""
    rng = np.random.default_rng(98)
    cfgs = ops.conv_configs()
    n, cin, h, w, cout = 2, 24, 9, 26, 64
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    want_up = _conv_ref(x, wt, b, 1, (1, 1, 1, 1), 0, 1, 'tanh', 1)
    xh = rng.standard_normal((n, cin, 2 * h, 2 * w)).astype(np.float32)
    want_sum = _conv_ref(xh, wt, None, 1, (1, 1, 1, 1), 0, 1, 'linear', 0).reshape(n, cout, h, 2, w, 2).sum(axis=(3, 5))
    seen_up, seen_sum, tried = None, None, 0
    try:
        for i, c in enumerate(cfgs):
            ks, dil, fa, pool = c[0], c[1], c[5], c[8]
            if not (ks == 3 and dil == 1 and fa == 0 and pool < 2) or c[6] == 1:   # (16-channel blocks: no such variant)
                continue
            tried += 1
            ops.force_conv_config(i)
            cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, src_mode=1)
            got = ops.conv2d(dev(x), dev(wt), dev(b), cd)
            _check_conv(ops, host(got), want_up, 'up-sampled source, config %d %r' % (i, c))
            assert seen_up is None or torch.equal(got, seen_up), 'config %d differs from the other Winograd instances' % i
            seen_up = got
            cs = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR)
            cs.out_pool = 2
            got = ops.conv2d(dev(xh), dev(wt), None, cs)
            assert tuple(got.shape) == (n, cout, h, w)
            scale = max(1.0, float(np.abs(want_sum).max()))
            assert np.abs(host(got) - want_sum).max() <= 4e-5 * scale, 'summing epilogue, config %d %r' % (i, c)
            assert seen_sum is None or torch.equal(got, seen_sum)
            seen_sum = got
    finally:
        ops.force_conv_config(-1)
    assert tried >= 4


@pytest.mark.parametrize('pool,in16', [(False, False), (True, False), (False, True)])
def test_winograd_position_split_instances_for_16_output_channels(ops, pool, in16):
    """conv_fwd_wino2_kernel.h (layers with 16 output channels per block, e.g. the restated output layer 32 -> 4 x 4): two
    waves per tile fragment, 8 of the 16 transformed positions each, partial output transforms combined through LDS.
    Every tile shape gives the same bits (the heuristic picks by batch size, so a member's forecast must not depend on
    it) and equals the float64 oracle to fp32 round-off; plain and pooled epilogues, float32 and bfloat16 input; 16 and 48
    output channels (one and three channel tiles)."""
    rng = np.random.default_rng(96)
    cfgs = ops.conv_configs()
    for cout in (16, 48):
        n, cin, h, w = 3, 24, 20, 70
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        if in16:
            x = np_ref.round_bf16(x).astype(np.float32)
        wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
        b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        want = _conv_ref(x, wt, b, 1, (1, 1, 1, 1), 0, 1, 'tanh', 0)
        if pool:
            want = np_ref.maxpool2(want)
        xd = dev(x).to(torch.bfloat16) if in16 else dev(x)
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, out_pool=pool)
        seen, tried = None, 0
        try:
            for i, c in enumerate(cfgs):
                if not (c[0] == 3 and c[1] == 1 and c[5] == 0 and c[6] == 1 and c[10] & 1) or c[10] & 4:
                    continue              # (bit 2: the variants for 32-channel layers, not offered here)
                tried += 1
                ops.force_conv_config(i)
                got = ops.conv2d(xd, dev(wt), dev(b), cd, out=torch.empty(want.shape, dtype=torch.float32, device='cuda'))
                _check_conv(ops, host(got), want, 'config %d %r' % (i, c))
                assert seen is None or torch.equal(got, seen), 'split instance %d %r differs from the others' % (i, c)
                seen = got
        finally:
            ops.force_conv_config(-1)
        assert tried >= 2
        if not in16:      # the heuristic's choice is one of them (a bfloat16-stored input goes to the bf16 matrix-core family)
            import ctypes
            from dlwp_amd import _lib
            pick = _lib.lib.dlwp_conv2d_pick_config(_lib.handle(0), ops.Shape4(n, cin, h, w), ctypes.byref(cd))
            assert pick >= 0 and cfgs[pick][10] & 1, 'expected a position-split Winograd instance, got %r' % (cfgs[pick],)
            assert torch.equal(ops.conv2d(xd, dev(wt), dev(b), cd, out=torch.empty_like(seen)), seen)


@pytest.mark.parametrize('pool', [False, True])
def test_winograd_compat_split_instances_give_the_bits_of_the_32_channel_kernel(ops, pool):
    """A layer with whole 32-channel tiles runs on conv_fwd_wino_kernel.h, or -- while its grid is small -- on the COMPAT
    position-split instances of conv_fwd_wino2_kernel.h, which evaluate both transforms in that kernel's order of operations:
    every such instance gives the SAME BITS (plain and pooled epilogue, tanh, ragged map, two channel tiles), so a member's
    forecast does not depend on the batch it is in; the plain split instances (their own arithmetic) are refused for such a
    layer; a 2-member grid really takes a COMPAT instance, a 256-member one the 32-channel kernel."""
    import ctypes
    from dlwp_amd import _lib
    rng = np.random.default_rng(99)
    cfgs = ops.conv_configs()
    n, cin, h, w, cout = 2, 40, 22, 46, 64
    x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
    b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, out_pool=pool)
    want = _conv_ref(host(x), host(wt), host(b), 1, (1, 1, 1, 1), 0, 1, 'tanh', 0)
    if pool:
        want = np_ref.maxpool2(want)
    seen, kinds, refused = None, set(), 0
    try:
        for i, c in enumerate(cfgs):
            if not (c[0] == 3 and c[1] == 1 and c[5] == 0 and c[8] < 2 and c[6] in (1, 2)):
                continue
            ops.force_conv_config(i)
            try:
                got = ops.conv2d(x, wt, b, cd)
            except _lib.DlwpError:
                refused += (c[10] & 5) == 1   # (a plain split instance, or an instance without this epilogue)
                continue
            _check_conv(ops, host(got), want, 'config %d %r' % (i, c))
            assert seen is None or torch.equal(got, seen), 'Winograd instance %d %r differs from the others' % (i, c)
            seen = got
            assert not (c[10] & 1) or c[10] & 4
            kinds.add(c[10] & 1)
    finally:
        ops.force_conv_config(-1)
    assert kinds == {0, 1} and refused >= 3, (kinds, refused)
    pick = _lib.lib.dlwp_conv2d_pick_config(_lib.handle(0), ops.Shape4(n, cin, h, w), ctypes.byref(cd))
    assert cfgs[pick][10] & 1, 'a 2-member grid should run on a COMPAT split instance, got %r' % (cfgs[pick],)
    big = _lib.lib.dlwp_conv2d_pick_config(_lib.handle(0), ops.Shape4(256, cin, h, w), ctypes.byref(cd))
    assert not cfgs[big][10] & 1 and cfgs[big][6] == 2
    assert torch.equal(ops.conv2d(x, wt, b, cd), seen)


def test_position_split_streaming_kernel_equals_the_general_instance(ops):
    """csrc/conv_fwd_wino2s.hip (16-output-channel Winograd blocks with the transformed filters resident in LDS, a workgroup walks
    over the samples of one tile position -- the restated output layer at large batches): the bits of the position-split instance
    it stands in for, and the float64 oracle.  The depth-to-space store and the plain one, 2 and 4 channel chunks with ragged
    channel counts, ragged tiles in both directions, 16 / 48 output channels (a workgroup's share crosses filter blocks), every
    activation, channel windows on both sides, batches that leave a workgroup one item and several (position changes)."""
    rng = np.random.default_rng(4242)
    cases = [  # n, cin, h, w, cout, mode_h, mode_w, act, d2s
        (24, 32, 44, 90, 16, 0, 1, 'linear', True),      # the U-Net's restated output layer
        (5, 16, 19, 44, 48, 0, 1, 'tanh', False),
        (7, 30, 20, 70, 16, 2, 1, 'relu', True),
        (130, 14, 10, 36, 16, 1, 0, 'tanh', False),
        (3, 32, 16, 64, 48, 0, 2, 'linear', False),
        (1, 29, 9, 34, 16, 0, 1, 'tanh', True),
    ]
    for n, cin, h, w, cout, mh, mw, act, d2s in cases:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
        b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, mh, mw), ops.ACTIVATIONS[act], out_d2s=d2s)
        xd, wd, bd = dev(x), dev(wt), dev(b)
        prev = ops.set_few_stream(0)
        try:
            base = ops.conv2d(xd, wd, bd, cd)
            assert ops.conv_launch_info(x.shape, cd)[0][0] >= 0
            ops.set_few_stream(2)
            info = ops.conv_launch_info(x.shape, cd)
            assert info[0][0] == -3 and info[0][2] == 512, info
            got = ops.conv2d(xd, wd, bd, cd)
        finally:
            ops.set_few_stream(prev)
        assert torch.equal(got, base), (n, cin, h, w, cout, float((got - base).abs().max()))
        if n <= 7:
            want = _conv_ref(x, wt, b, 1, (1, 1, 1, 1), mh, mw, act, 0)
            if d2s:
                want = np_ref.depth_to_space2(want, cout // 4)
            _check_conv(ops, host(got), want, 'position-split stream %r' % ((n, cin, h, w, cout),))
    # channel windows: 24 of 40 stored input channels from channel 8; 4 fields into channels 2.. of a 7-field tensor
    n, h, w = 9, 20, 70
    x40 = rng.standard_normal((n, 40, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, 24, 16), rng)
    b = (0.1 * rng.standard_normal(16)).astype(np.float32)
    cd = ops.make_conv(16, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, in_c_off=8, in_c_total=40, out_c_off=2,
                       out_c_total=7, out_d2s=True)
    outs = []
    prev = ops.set_few_stream(0)
    try:
        for mode in (0, 2):
            ops.set_few_stream(mode)
            y = torch.full((n, 7, 2 * h, 2 * w), 7.0, device='cuda')
            ops.conv2d(dev(x40), dev(wt), dev(b), cd, out=y, x_channels=24)
            outs.append(y)
    finally:
        ops.set_few_stream(prev)
    assert torch.equal(outs[0], outs[1])
    want = np_ref.depth_to_space2(_conv_ref(x40[:, 8:32], wt, b, 1, (1, 1, 1, 1), 0, 1, 'tanh', 0), 4)
    _check_conv(ops, host(outs[1][:, 2:6]), want, 'position-split stream, channel windows')
    assert float(outs[1][:, :2].min()) == 7.0 and float(outs[1][:, 6:].max()) == 7.0


def test_position_split_streaming_kernel_is_chosen_by_batch_size(ops):
    """DLWP_OPT_FEW_STREAM = 1 (default): the restated output layer of the 88 x 180 U-Net (32 -> 16 phase channels at 44 x 90,
    stored depth-to-space) goes to the streaming kernel from 2.5 tiles per resident workgroup on (2 per CU), the general instance
    below; never with more than 32 input channels, a pooling epilogue, 32-channel blocks, or bfloat16 storage."""
    cd = ops.make_conv(16, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR, out_d2s=True)
    g = ops.conv_launch_info((256, 32, 44, 90), cd)[0]
    assert g[0] == -3 and g[2] == 512 and g[1] % 2 == 0, g
    assert g[3] == 2.0 * 1024 * 16 * 32 * 6 * 3 * 256       # 64 tiles x 16 positions x 16 channels x 32 inputs per item
    assert ops.conv_launch_info((128, 32, 44, 90), cd)[0][0] == -3
    assert ops.conv_launch_info((32, 32, 44, 90), cd)[0][0] >= 0
    assert ops.conv_launch_info((1, 32, 44, 90), cd)[0][0] >= 0
    assert ops.conv_launch_info((256, 40, 44, 90), cd)[0][0] >= 0
    assert ops.conv_launch_info((256, 32, 44, 90), ops.make_conv(32, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH))[0][0] >= 0
    assert ops.conv_launch_info((256, 32, 44, 90), ops.make_conv(16, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH,
                                                                   out_pool=True))[0][0] >= 0
    prev = ops.set_few_stream(0)
    try:
        assert ops.conv_launch_info((256, 32, 44, 90), cd)[0][0] >= 0
    finally:
        ops.set_few_stream(prev)


@pytest.mark.parametrize('fields,hw', [(4, (20, 70)), (12, (19, 45)), (4, (44, 90))])
def test_winograd_phase_channels_stored_interleaved(ops, fields, hw):
    """dlwp_conv2d.out_d2s: the 4 F output channels are the 2x2 phases of F fields (phase-major) and the 16-channel Winograd
    instances store them interleaved into (n, c_total, 2 ho, 2 wo) -- the same bits as the convolution followed by
    dlwp_depth_to_space2, ragged edges and a channel window of a wider output included; every tile shape agrees."""
    rng = np.random.default_rng(98)
    cfgs = ops.conv_configs()
    n, cin = 3, 24
    h, w = hw
    cout = 4 * fields
    x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
    b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
    plain = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)
    assert ops.supports_out_d2s((cin, h, w), plain)
    y4 = ops.conv2d(x, wt, b, plain)
    want = torch.full((n, fields + 3, 2 * h, 2 * w), 7.0, device='cuda')
    ops.depth_to_space2(y4, fields, out=want, c_off=2)
    ref = np_ref.depth_to_space2(_conv_ref(host(x), host(wt), host(b), 1, (1, 1, 1, 1), 0, 1, 'tanh', 0), fields)
    _check_conv(ops, host(want[:, 2:2 + fields]), ref, 'unfused reference path')
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, out_c_off=2, out_c_total=fields + 3,
                       out_d2s=True)
    tried = 0
    try:
        for i, c in enumerate([None] + list(cfgs)):
            if c is not None and (not (c[0] == 3 and c[1] == 1 and c[5] == 0 and c[6] == 1 and c[10] & 1) or c[10] & 4):
                continue
            ops.force_conv_config(i - 1)
            tried += 1
            got = ops.conv2d(x, wt, b, cd, out=torch.full_like(want, 7.0))
            assert torch.equal(got, want), 'config %r: interleaved stores differ from conv + depth_to_space2' % (c,)
    finally:
        ops.force_conv_config(-1)
    assert tried >= 3
    # layers the 16-channel instances do not take keep the separate pass
    assert not ops.supports_out_d2s((cin, h, w), ops.make_conv(32, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH))
    assert not ops.supports_out_d2s((cin, h, w), ops.make_conv(16, 5, 5, 1, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH))


@pytest.mark.parametrize('f,hw,first', [(24, (20, 72), True), (24, (20, 72), False), (16, (16, 24), False), (40, (9, 44), False)])
def test_convlstm_cell_update_in_the_convolution_epilogue(ops, f, hw, first):
    """dlwp_convlstm_conv_fwd (bfloat16 inference): the convolution that completes a step's gate pre-activations applies the
    cell update in its epilogue.  Against the float64 oracle with the product's roundings (bf16 input / kernel / stored z_add,
    float32 cell state) and against the unfused sequence conv -> bf16 z -> dlwp_convlstm_gates (which rounds z once more)."""
    rng = np.random.default_rng(99)
    n = 3
    h, w = hw
    if first:       # input convolution: float32 model input rounded by the loader, dilation 2, periodic + zero halo
        cin, dil, halo = 6, 2, (2, 2, 2, 2, 0, 1)
        x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        xq = np_ref.round_bf16(host(x))
    else:           # recurrent convolution: bf16 h, 'same' zero halo
        cin, dil, halo = f, 1, (1, 1, 1, 1, 0, 0)
        xq = np_ref.round_bf16(rng.standard_normal((n, cin, h, w)))
        x = dev(xq.astype(np.float32)).to(torch.bfloat16)
    wt = np_ref.glorot_uniform((3, 3, cin, 4 * f), rng)
    b = (0.1 * rng.standard_normal(4 * f)).astype(np.float32)
    b[f:2 * f] += 1.0
    zadd = None if first else np_ref.round_bf16(0.5 * rng.standard_normal((n, 4 * f, h, w)))
    cprev = None if first else rng.standard_normal((n, f, h, w)).astype(np.float32)
    cd = ops.make_conv(4 * f, 3, 3, dil, ops.make_pad(*halo), ops.ACT_TANH, out_c_off=f, out_c_total=3 * f, lstm_f=f)
    assert ops.convlstm_conv_supported((cin, h, w), cd, in_bf16=not first, compute_bf16=first)
    h_out = torch.full((n, 3 * f, h, w), 7.0, device='cuda').to(torch.bfloat16)
    c_out = torch.empty((n, f, h, w), device='cuda')
    ops.convlstm_conv(x, dev(wt), dev(b), cd, h_out, c_out, z_add=dev(zadd.astype(np.float32)).to(torch.bfloat16) if zadd is not None else None,
                      c_prev=dev(cprev) if cprev is not None else None, compute_bf16=first)
    # oracle: exact products of bf16 values, float accumulation
    z = _conv_ref(xq, np_ref.round_bf16(wt), b, dil, halo[:4], halo[4], halo[5], 'linear', 0)
    if zadd is not None:
        z = z + zadd
    zi, zf, zc, zo = z[:, :f], z[:, f:2 * f], z[:, 2 * f:3 * f], z[:, 3 * f:]
    c_want = np_ref.hard_sigmoid(zi) * np.tanh(zc) + (np_ref.hard_sigmoid(zf) * cprev if cprev is not None else 0.0)
    h_want = np_ref.hard_sigmoid(zo) * np.tanh(c_want)
    assert np.abs(host(c_out) - c_want).max() < 2e-5 * max(1.0, np.abs(c_want).max())
    got_h = host(h_out.float())
    assert np.abs(got_h[:, f:2 * f] - h_want).max() < 4.1e-3          # one bf16 ulp of values below 1
    assert np.all(got_h[:, :f] == 7.0) and np.all(got_h[:, 2 * f:] == 7.0)   # the window only
    # unfused: conv -> bf16 z -> gate kernel
    plain = ops.make_conv(4 * f, 3, 3, dil, ops.make_pad(*halo), ops.ACT_LINEAR)
    zx = ops.conv2d(x, dev(wt), dev(b), plain, out=torch.empty((n, 4 * f, h, w), device='cuda', dtype=torch.bfloat16), compute_bf16=first)
    h2 = torch.zeros((n, 3 * f, h, w), device='cuda', dtype=torch.bfloat16)
    c2 = torch.empty_like(c_out)
    ops.convlstm_gates(zx, dev(zadd.astype(np.float32)).to(torch.bfloat16) if zadd is not None else None,
                       dev(cprev) if cprev is not None else None, c2, h2, f, h_c_off=f, act=ops.ACT_TANH, rec_act=0)
    assert np.abs(host(c2) - host(c_out)).max() < 2e-2 and np.abs(host(h2.float())[:, f:2 * f] - got_h[:, f:2 * f]).max() < 2e-2
    # every compiled cell-update instance of this geometry (tile shape, channel chunk): same result up to summation order
    tried = 0
    try:
        for i, c in enumerate(ops.conv_configs()):
            if not (c[10] & 2 and c[1] == dil and (c[8] == 3) == first) or c[10] & 24:     # (& 24: octet-layout instances)
                continue
            ops.force_conv_config(i)
            h3, c3 = torch.zeros_like(h_out), torch.empty_like(c_out)
            ops.convlstm_conv(x, dev(wt), dev(b), cd, h3, c3, z_add=dev(zadd.astype(np.float32)).to(torch.bfloat16) if zadd is not None else None,
                              c_prev=dev(cprev) if cprev is not None else None, compute_bf16=first)
            assert np.abs(host(c3) - c_want).max() < 2e-5 * max(1.0, np.abs(c_want).max()), 'config %d %r' % (i, c)
            assert np.abs(host(h3.float())[:, f:2 * f] - h_want).max() < 4.1e-3, 'config %d %r' % (i, c)
            tried += 1
    finally:
        ops.force_conv_config(-1)
    assert tried >= 2
    # widths that are not a multiple of 4 keep the separate gate kernel
    assert not ops.convlstm_conv_supported((cin, h, w + 2), cd, in_bf16=not first, compute_bf16=first)


def test_winograd_wide_plus_narrow_launch_is_bit_identical(ops):
    """A 22x45 map at a batch that fills the chip: the 32 whole columns go to the 8x32 instance, the last 13 to the 16-wide
    one in a second launch.  Same bits as the single forced instance; plain, pooled-epilogue and up-sampled launches."""
    rng = np.random.default_rng(97)
    cfgs = ops.conv_configs()
    wide = [i for i, c in enumerate(cfgs) if c[0] == 3 and c[1] == 1 and c[5] == 0 and c[2:5] == (8, 32, 4) and c[6] == 2]
    assert wide
    for src, hw, pool in ((0, (22, 45), False), (0, (22, 90), True), (1, (11, 45), False), (0, (22, 77), False)):
        n, cin, cout = 96, 16, 128
        x = dev(rng.standard_normal((n, cin) + hw).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
        b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, src_mode=src, out_pool=pool)
        got = ops.conv2d(x, wt, b, cd)
        ops.force_conv_config(wide[0])
        try:
            want = ops.conv2d(x, wt, b, cd)
        finally:
            ops.force_conv_config(-1)
        assert torch.equal(got, want), (src, hw, pool)
        ref = _conv_ref(host(x[:2]), host(wt), host(b), 1, (1, 1, 1, 1), 0, 1, 'tanh', src)
        if pool:
            ref = np_ref.maxpool2(ref)
        _check_conv(ops, host(got[:2]), ref)
    # float32, plain source, even batch on a 22x45 map: SAMPLE PAIRS side by side (a virtual row of 2 x 48 = 3 x 32 columns,
    # the gap holds the halos) on the wide instance -- the case above; here with channel windows on both sides, a zero
    # column halo, and an odd batch (which keeps the wide + narrow launches): always the bits of the single instance
    for n, mode_w in ((96, 1), (96, 0), (95, 1)):
        cin, cout, hw = 16, 64, (22, 45)
        xw = dev(rng.standard_normal((n, cin + 5) + hw).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
        b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, mode_w), ops.ACT_TANH, in_c_off=3, in_c_total=cin + 5,
                           out_c_off=2, out_c_total=cout + 7)
        got = ops.conv2d(xw, wt, b, cd, out=torch.full((n, cout + 7) + hw, 5.0, device='cuda'), x_channels=cin)
        ops.force_conv_config(wide[0])
        try:
            want = ops.conv2d(xw, wt, b, cd, out=torch.full((n, cout + 7) + hw, 5.0, device='cuda'), x_channels=cin)
        finally:
            ops.force_conv_config(-1)
        assert torch.equal(got, want), (n, mode_w)
        ref = _conv_ref(host(xw[-3:, 3:3 + cin]), host(wt), host(b), 1, (1, 1, 1, 1), 0, mode_w, 'tanh', 0)
        _check_conv(ops, host(got[-3:, 2:2 + cout]), ref)
        assert bool((got[:, :2] == 5.0).all()) and bool((got[:, 2 + cout:] == 5.0).all())


@pytest.mark.parametrize('cin', [5, 6, 7, 13, 22, 30])
@pytest.mark.parametrize('dil,src', [(1, 0), (2, 0), (1, 1)])
def test_winograd_with_ragged_input_channels(ops, cin, dil, src):
    """Input-channel counts that are not multiples of 8 run zero-padded to whole chunks (the ConvLSTM2D input convolution of
    config 4 has 6): out-of-range planes and filter rows read as 0.  9 or 12 input channels stay on the direct family,
    whose chunks of 4 waste less; r6: so do 5-8 channels of a plain source (the first layer of a network with an insolation input)."""
    import ctypes
    from dlwp_amd import _lib
    rng = np.random.default_rng(100 + cin)
    n, h, w, cout = 2, 13, 38, 64
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    pads = (dil, dil, dil, dil)
    cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(*pads, 0, 1), ops.ACT_TANH, src_mode=src)
    pick = _lib.lib.dlwp_conv2d_pick_config(_lib.handle(0), ops.Shape4(n, cin, h, w), ctypes.byref(cd))
    if cin <= 8 and src == 0:
        # r6: a plain-source layer of 5-8 input channels belongs to the DIRECT family (the streaming first-layer kernel stands in
        # for its instances at large batches with the same bits: csrc/conv_fwd.hip few_family)
        assert pick >= 0 and ops.conv_configs()[pick][5] != 0, 'expected a direct-family instance'
    else:
        assert pick >= 0 and ops.conv_configs()[pick][5] == 0, 'expected a Winograd instance'
    # the planes behind the window must not leak in: poison what follows the last channel
    xd = torch.full((n, cin + 3, h, w), 1e6, dtype=torch.float32, device='cuda')
    xd[:, :cin] = dev(x)
    got = ops.conv2d(xd, dev(wt), dev(b), cd, x_channels=cin)
    _check_conv(ops, host(got), _conv_ref(x, wt, b, dil, pads, 0, 1, 'tanh', src), 'cin %d' % cin)
    for c_direct in (9, 12):
        c9 = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)
        p9 = _lib.lib.dlwp_conv2d_pick_config(_lib.handle(0), ops.Shape4(n, c_direct, h, w), ctypes.byref(c9))
        assert p9 < 0 or ops.conv_configs()[p9][5] != 0


def test_conv2d_channel_windows_slice_and_concat(ops):
    """slice_layer on the input side and concatenate on the output side without copies (custom.py:675-692)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 32, 8, 12)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, 16, 24), rng)
    b = (0.1 * rng.standard_normal(24)).astype(np.float32)
    want = _conv_ref(x[:, 16:32], wt, b, 1, (1, 1, 1, 1), 0, 1, 'tanh', 0)
    cd = ops.make_conv(24, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH, in_c_off=16, in_c_total=32,
                       out_c_off=8, out_c_total=40)
    out = torch.full((2, 40, 8, 12), 7.0, device='cuda')
    ops.conv2d(dev(x), dev(wt), dev(b), cd, out=out, x_channels=16)
    got = host(out)
    _check_conv(ops, got[:, 8:32], want)
    assert np.all(got[:, :8] == 7.0) and np.all(got[:, 32:] == 7.0)


def test_conv2d_linearity_and_longitude_shift_equivariance_full_size(ops):
    """Size-independent properties at BASELINE.json's full 88x180 grid (config 2, layer 5 shape)."""
    rng = np.random.default_rng(2)
    n, cin, h, w, cout = 2, 64, 88, 180, 32
    x1 = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    x2 = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
    cd = ops.make_conv(cout, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_LINEAR)
    y1 = host(ops.conv2d(dev(x1), wt, None, cd))
    y2 = host(ops.conv2d(dev(x2), wt, None, cd))
    y12 = host(ops.conv2d(dev(x1 + 2 * x2), wt, None, cd))
    assert np.abs(y12 - (y1 + 2 * y2)).max() < 2e-4
    # periodic halo: exact shift equivariance, bit for bit, for shifts by the Winograd tile period (2 * dilation = 4)
    ys = host(ops.conv2d(dev(np.roll(x1, 8, axis=-1)), wt, None, cd))
    assert np.array_equal(ys, np.roll(y1, 8, axis=-1))
    ops.set_winograd(False)                                 # the direct kernel is equivariant under ANY shift
    try:
        yd = host(ops.conv2d(dev(x1), wt, None, cd))
        assert np.array_equal(host(ops.conv2d(dev(np.roll(x1, 7, axis=-1)), wt, None, cd)), np.roll(yd, 7, axis=-1))
        assert np.abs(yd - y1).max() < 1e-5 * max(1.0, np.abs(yd).max())     # Winograd vs direct: fp32 round-off only
    finally:
        ops.set_winograd(True)
    # spot-check against the float64 oracle on one sample / a few channels (full tensor would take minutes on CPU)
    want = _conv_ref(x1[:1], host(wt)[..., :4], None, 2, (2, 2, 2, 2), 0, 1, 'linear', 0)
    _check_conv(ops, y1[:1, :4], want)


def test_conv2d_batch_invariance(ops):
    """A sample's result must not depend on what else is in the batch (member sharding across GPUs relies on it):
    direct family (48 output channels) and Winograd family (64)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, 16, 22, 45)).astype(np.float32)
    for cout in (48, 64):
        wt = dev(np_ref.glorot_uniform((3, 3, 16, cout), rng))
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)
        full = host(ops.conv2d(dev(x), wt, None, cd))
        for i in (0, 3):
            one = host(ops.conv2d(dev(x[i:i + 1]), wt, None, cd))
            assert np.array_equal(one[0], full[i])


def test_conv2d_with_prepared_weights_is_bit_identical(ops):
    """dlwp_conv2d_prepare + dlwp_conv2d_fwd_prepared == dlwp_conv2d_fwd: Winograd (plain and up-sampled source), packed-N
    (5x5, 4 output channels), bf16 arrangement, a direct layer that needs no preparation (prepared is None: 40 output channels
    are not whole Winograd channel tiles) and the 16-channel position-split Winograd instance (48 = 3 tiles)."""
    rng = np.random.default_rng(31)
    cases = [(16, 64, 3, 1, 0, False, True), (64, 64, 3, 1, 1, False, True), (32, 4, 5, 1, 0, False, None),
             (16, 40, 3, 1, 0, False, False), (16, 48, 3, 1, 0, False, True), (32, 32, 3, 1, 0, True, True)]
    for cin, cout, k, dil, src, cbf16, expect_prep in cases:
        x = dev(rng.standard_normal((3, cin, 16, 36)).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((k, k, cin, cout), rng))
        b = dev(rng.standard_normal(cout).astype(np.float32))
        p = dil * (k - 1) // 2
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, 0, 1), ops.ACT_TANH, src_mode=src)
        prep = ops.conv2d_prepare(x, wt, cd, compute_bf16=cbf16)
        assert expect_prep is None or (prep is not None) == expect_prep, (cin, cout, k)
        want = ops.conv2d(x, wt, b, cd, compute_bf16=cbf16)
        got = ops.conv2d(x, wt, b, cd, compute_bf16=cbf16, prepared=prep)
        assert torch.equal(got, want), (cin, cout, k)


# ----------------------------------------------------------------------------------------------------------------- #
# ConvLSTM2D cell update
# ----------------------------------------------------------------------------------------------------------------- #

@pytest.mark.parametrize('n,f,h,w,first,rec', [(2, 3, 5, 8, True, 'hard_sigmoid'), (3, 4, 6, 7, False, 'hard_sigmoid'),
                                                (1, 8, 16, 24, False, 'sigmoid'), (0, 2, 4, 4, False, 'hard_sigmoid')])
def test_convlstm_gates_match_oracle(ops, n, f, h, w, first, rec):
    """c = rec(z_f) c' + rec(z_i) tanh(z_c), h = rec(z_o) tanh(c) (keras ConvLSTM2DCell.call); h lands in a channel
    window of the return_sequences buffer.  Vector (hw % 4 == 0) and scalar paths, first step (no zh / c')."""
    rng = np.random.default_rng(n * 100 + f)
    zx = (2 * rng.standard_normal((n, 4 * f, h, w))).astype(np.float32)
    zh = None if first else (2 * rng.standard_normal((n, 4 * f, h, w))).astype(np.float32)
    cp = None if first else rng.standard_normal((n, f, h, w)).astype(np.float32)
    z = zx.astype(np.float64) + (0 if zh is None else zh)
    r = np_ref.hard_sigmoid if rec == 'hard_sigmoid' else (lambda v: 1 / (1 + np.exp(-v)))
    c_want = r(z[:, :f]) * np.tanh(z[:, 2 * f:3 * f]) + (0 if cp is None else r(z[:, f:2 * f]) * cp)
    h_want = r(z[:, 3 * f:]) * np.tanh(c_want)
    c_out = torch.empty((n, f, h, w), device='cuda')
    h_out = torch.full((n, 3 * f, h, w), 7.0, device='cuda')
    ops.convlstm_gates(dev(zx), None if zh is None else dev(zh), None if cp is None else dev(cp), c_out, h_out, f,
                       h_c_off=f, act=ops.ACT_TANH, rec_act=ops.REC_HARD_SIGMOID if rec == 'hard_sigmoid' else ops.REC_SIGMOID)
    if n == 0:
        return
    assert np.abs(host(c_out) - c_want).max() < 2e-6
    got = host(h_out)
    assert np.abs(got[:, f:2 * f] - h_want).max() < 2e-6
    assert np.all(got[:, :f] == 7.0) and np.all(got[:, 2 * f:] == 7.0)
    # h stored as bfloat16 (config 4): the same values rounded once; c is float32 and unchanged
    c16 = torch.empty((n, f, h, w), device='cuda')
    h16 = torch.full((n, 3 * f, h, w), 7.0, device='cuda', dtype=torch.bfloat16)
    ops.convlstm_gates(dev(zx), None if zh is None else dev(zh), None if cp is None else dev(cp), c16, h16, f,
                       h_c_off=f, act=ops.ACT_TANH, rec_act=ops.REC_HARD_SIGMOID if rec == 'hard_sigmoid' else ops.REC_SIGMOID)
    assert torch.equal(c16, c_out)
    assert torch.equal(h16, h_out.to(torch.bfloat16))
    # gate pre-activations stored as bfloat16 too: the float32 arithmetic on the rounded values
    zx16 = dev(zx).to(torch.bfloat16)
    zh16 = None if zh is None else dev(zh).to(torch.bfloat16)
    cz = torch.empty((n, f, h, w), device='cuda')
    hz = torch.full((n, 3 * f, h, w), 7.0, device='cuda', dtype=torch.bfloat16)
    ops.convlstm_gates(zx16, zh16, None if cp is None else dev(cp), cz, hz, f, h_c_off=f, act=ops.ACT_TANH,
                       rec_act=ops.REC_HARD_SIGMOID if rec == 'hard_sigmoid' else ops.REC_SIGMOID)
    z = np_ref.round_bf16(zx) + (0 if zh is None else np_ref.round_bf16(zh))
    c_want = r(z[:, :f]) * np.tanh(z[:, 2 * f:3 * f]) + (0 if cp is None else r(z[:, f:2 * f]) * cp)
    h_want = r(z[:, 3 * f:]) * np.tanh(c_want)
    assert np.abs(host(cz) - c_want).max() < 2e-6
    hg = hz.to(torch.float32).cpu().numpy()[:, f:2 * f]
    assert np.all(np.abs(hg - h_want) <= 2.0 ** -8 * np.abs(h_want) + 2e-6)


# ----------------------------------------------------------------------------------------------------------------- #
# bfloat16 storage of the activations (BASELINE.json config 4); arithmetic stays fp32
# ----------------------------------------------------------------------------------------------------------------- #

BF16_CASES = [
    # (n, cin, h, w, cout, k, dil, src_mode)         kernel family the shape selects
    (2, 4, 12, 20, 32, 3, 2, 0),                     # direct MFMA (cin = 4)
    (2, 24, 12, 20, 64, 3, 1, 0),                    # Winograd
    (2, 32, 12, 20, 32, 3, 2, 1),                    # Winograd, dilation 2, fused up-sampling
    (2, 20, 12, 20, 36, 3, 1, 2),                    # direct MFMA with the pooled loader (ragged channels)
    (2, 32, 12, 20, 4, 5, 1, 0),                     # packed-N 5x5 output layer
    (1, 3, 9, 11, 5, 7, 1, 0),                       # no MFMA instance: the one-thread-per-output kernel
]


def _weights_as_multiplied(ops, wt, x_shape, cd, in16, out16, compute_bf16=False):
    """The kernel the oracle has to use: rounded to bfloat16 when the layer runs on the bf16 matrix cores."""
    from dlwp_amd import _lib
    dt = _lib.dtype_io(_lib.BF16 if in16 else _lib.F32, _lib.BF16 if out16 else _lib.F32, compute_bf16)
    if ops.uses_bf16_weights(x_shape, cd, dt):
        return np_ref.round_bf16(wt), True
    return wt, False


@pytest.fixture(params=[True, False], ids=['bf16-mfma', 'fp32-families'])
def bf16_mfma(request, ops):
    prev = ops.set_bf16_mfma(request.param)
    yield request.param
    ops.set_bf16_mfma(prev)


@pytest.mark.parametrize('case', BF16_CASES)
@pytest.mark.parametrize('io', [('bf16', 'bf16'), ('f32', 'bf16'), ('bf16', 'f32')])
def test_conv2d_bfloat16_storage(ops, case, io, bf16_mfma):
    n, cin, h, w, cout, k, dil, src = case
    rng = np.random.default_rng(sum(case))
    x = np_ref.round_bf16(rng.standard_normal((n, cin, h, w))).astype(np.float32)     # exactly representable inputs
    wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    p = dil * (k - 1) // 2
    pads = (p, p, p, p)
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, 0, 1), ops.ACT_TANH, src_mode=src)
    w_ref, on16 = _weights_as_multiplied(ops, wt, x.shape, cd, io[0] == 'bf16', io[1] == 'bf16')
    assert on16 == (bf16_mfma and io[0] == 'bf16' and cin >= 12 and src != 2 and k in (3, 5))
    want = _conv_ref(x, w_ref, b, dil, pads, 0, 1, 'tanh', src)
    xd = dev(x).to(torch.bfloat16) if io[0] == 'bf16' else dev(x)
    out = torch.empty(want.shape, dtype=torch.bfloat16 if io[1] == 'bf16' else torch.float32, device='cuda')
    ops.conv2d(xd, dev(wt), dev(b), cd, out=out)
    got = out.to(torch.float32).cpu().numpy()
    if io[1] == 'f32':
        _check_conv(ops, got, want, 'bf16 in')
    else:
        # the stored value is the fp32 result rounded to bf16: at most one bf16 ulp (2^-8 relative) from the oracle, and
        # almost everywhere exactly the oracle's own rounding
        assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 1e-6)
        assert np.mean(got == np_ref.round_bf16(want)) > 0.99


BF16_MFMA_CASES = [
    # (n, cin, c_off, c_total, h, w, cout, k, dil, pads(t,b,l,r), mode_h, mode_w, act, src_mode, out16)
    (2, 32, 0, 32, 19, 50, 36, 3, 1, (1, 1, 1, 1), 0, 1, 'tanh', 0, True),      # odd left halo, ragged cout and tiles
    (2, 48, 0, 48, 20, 72, 32, 3, 1, (1, 1, 1, 1), 0, 1, 'tanh', 0, True),      # config-4 conv2d_1 channels (32 + 16)
    (1, 24, 8, 40, 11, 34, 96, 3, 1, (1, 1, 1, 1), 0, 0, 'linear', 0, False),   # ConvLSTM recurrent conv: zero 'same', window
    (2, 20, 0, 20, 9, 12, 7, 3, 2, (2, 2, 2, 2), 2, 1, 'relu', 0, True),        # dilation 2, edge rows, ragged channels
    (2, 16, 0, 16, 7, 9, 40, 3, 1, (1, 1, 1, 1), 0, 1, 'tanh', 1, True),        # fused up-sampling from an ODD width
    (1, 64, 0, 64, 12, 20, 32, 3, 2, (2, 2, 2, 2), 0, 1, 'tanh', 1, False),     # dilated + up-sampled, two chunks
    (2, 16, 0, 16, 10, 16, 4, 5, 1, (2, 2, 2, 2), 0, 1, 'linear', 0, False),    # 5x5 output layer
    (1, 33, 0, 33, 8, 14, 17, 3, 1, (0, 2, 3, 0), 1, 1, 'tanh', 0, True),       # asymmetric halo, periodic rows, cin = 33
    (3, 12, 0, 12, 5, 6, 16, 3, 1, (1, 1, 1, 1), 0, 0, 'tanh', 0, True),        # tiny grid, fewest channels
]


@pytest.mark.parametrize('case', BF16_MFMA_CASES)
def test_conv2d_bf16_mfma_family(ops, case):
    """Layers with bf16-stored input on the bf16 matrix cores: bf16 x bf16 products are exact in fp32 and the sums are
    fp32, so against the oracle run on the bf16-ROUNDED weights the fp32 tolerance of the other families holds."""
    n, cin, c_off, c_tot, h, w, cout, k, dil, pads, mh, mw, act, src, out16 = case
    rng = np.random.default_rng(sum(pads) + cin + cout + h)
    xfull = np_ref.round_bf16(rng.standard_normal((n, c_tot, h, w))).astype(np.float32)
    wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    actc = {'tanh': ops.ACT_TANH, 'relu': ops.ACT_RELU, 'linear': ops.ACT_LINEAR}[act]
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mh, mw), actc, in_c_off=c_off, in_c_total=c_tot,
                       src_mode=src)
    w_ref, on16 = _weights_as_multiplied(ops, wt, (n, cin, h, w), cd, True, out16)
    assert on16, 'this geometry should select the bf16 matrix-core family'
    want = _conv_ref(xfull[:, c_off:c_off + cin], w_ref, b, dil, pads, mh, mw, act, src)
    out = torch.full(want.shape, float('nan'), dtype=torch.bfloat16 if out16 else torch.float32, device='cuda')
    ops.conv2d(dev(xfull).to(torch.bfloat16), dev(wt), dev(b), cd, out=out, x_channels=cin)
    got = out.to(torch.float32).cpu().numpy()
    if out16:
        assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 2e-6)
        assert np.mean(got == np_ref.round_bf16(want)) > 0.99
    else:
        _check_conv(ops, got, want, 'bf16 mfma')
    # the fp32 families on the same bf16 input differ only by the weight rounding (2^-9 relative per weight)
    prev = ops.set_bf16_mfma(False)
    try:
        out32 = torch.empty(want.shape, dtype=torch.float32, device='cuda')
        ops.conv2d(dev(xfull).to(torch.bfloat16), dev(wt), dev(b), cd, out=out32, x_channels=cin)
    finally:
        ops.set_bf16_mfma(prev)
    assert np.abs(out32.cpu().numpy() - want).max() < 2e-2 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('shape', [(2, 32, 20, 36, 64, 1), (1, 48, 19, 50, 32, 2), (2, 16, 9, 14, 40, 1)])
@pytest.mark.parametrize('out16', [True, False])
def test_conv2d_bf16_mfma_pooling_epilogue(ops, shape, out16):
    """MaxPooling2D(2) applied in the epilogue of the bf16 matrix-core kernel (odd output sizes drop the last row /
    column as Keras does)."""
    n, cin, h, w, cout, dil = shape
    rng = np.random.default_rng(sum(shape))
    x = np_ref.round_bf16(rng.standard_normal((n, cin, h, w))).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    pads = (dil, dil, dil, dil)
    cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(*pads, 0, 1), ops.ACT_TANH, out_pool=1)
    w_ref, on16 = _weights_as_multiplied(ops, wt, x.shape, cd, True, out16)
    assert on16
    want = np_ref.maxpool2(_conv_ref(x, w_ref, b, dil, pads, 0, 1, 'tanh', 0))
    out = torch.full(want.shape, float('nan'), dtype=torch.bfloat16 if out16 else torch.float32, device='cuda')
    ops.conv2d(dev(x).to(torch.bfloat16), dev(wt), dev(b), cd, out=out)
    got = out.to(torch.float32).cpu().numpy()
    if out16:
        assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 2e-6)
    else:
        _check_conv(ops, got, want, 'bf16 mfma pooled')


IN32_CASES = [
    # (n, cin, c_off, c_total, h, w, cout, k, dil, pads, mode_h, mode_w, act, src, pool, out16)
    (2, 6, 6, 12, 13, 20, 96, 3, 2, (2, 2, 2, 2), 0, 1, 'linear', 0, False, True),   # ConvLSTM2D input conv, step 1 of 2
    (2, 4, 0, 4, 16, 36, 32, 3, 2, (2, 2, 2, 2), 0, 1, 'tanh', 0, True, True),       # U-Net first layer + pooling
    (1, 8, 0, 8, 9, 14, 20, 5, 1, (2, 2, 2, 2), 0, 1, 'tanh', 0, False, False),      # 5x5, float32 out
    (2, 5, 0, 5, 7, 10, 33, 3, 1, (1, 1, 1, 1), 2, 0, 'relu', 1, False, True),       # up-sampled source, ragged
    (1, 20, 0, 20, 11, 18, 16, 3, 1, (1, 1, 1, 1), 0, 1, 'tanh', 0, False, True),    # two 16-channel chunks
]


@pytest.mark.parametrize('case', IN32_CASES)
def test_conv2d_bf16_mfma_on_float32_input(ops, case):
    """DLWP_COMPUTE_BF16: a float32-stored input (the model state in config 4) is rounded to bfloat16 by the loader and
    the layer runs on the bf16 matrix cores.  Oracle: the float64 convolution of the ROUNDED input and kernel."""
    n, cin, c_off, c_tot, h, w, cout, k, dil, pads, mh, mw, act, src, pool, out16 = case
    rng = np.random.default_rng(cin * 7 + cout + h)
    xfull = rng.standard_normal((n, c_tot, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    actc = {'tanh': ops.ACT_TANH, 'relu': ops.ACT_RELU, 'linear': ops.ACT_LINEAR}[act]
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mh, mw), actc, in_c_off=c_off, in_c_total=c_tot,
                       src_mode=src, out_pool=1 if pool else 0)
    w_ref, on16 = _weights_as_multiplied(ops, wt, (n, cin, h, w), cd, False, out16, compute_bf16=True)
    assert on16
    assert not _weights_as_multiplied(ops, wt, (n, cin, h, w), cd, False, out16)[1]      # only when the caller allows it
    want = _conv_ref(np_ref.round_bf16(xfull[:, c_off:c_off + cin]), w_ref, b, dil, pads, mh, mw, act, src)
    if pool:
        want = np_ref.maxpool2(want)
    out = torch.full(want.shape, float('nan'), dtype=torch.bfloat16 if out16 else torch.float32, device='cuda')
    ops.conv2d(dev(xfull), dev(wt), dev(b), cd, out=out, x_channels=cin, compute_bf16=True)
    got = out.to(torch.float32).cpu().numpy()
    if out16:
        assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 2e-6)
        assert np.mean(got == np_ref.round_bf16(want)) > 0.99
    else:
        _check_conv(ops, got, want, 'bf16 mfma, float32 in')
    # without the flag the same call keeps the float32 arithmetic on the unrounded input
    out2 = torch.empty(want.shape, dtype=torch.float32, device='cuda')
    ops.conv2d(dev(xfull), dev(wt), dev(b), cd, out=out2, x_channels=cin)
    exact = _conv_ref(xfull[:, c_off:c_off + cin], wt, b, dil, pads, mh, mw, act, src)
    _check_conv(ops, out2.cpu().numpy(), np_ref.maxpool2(exact) if pool else exact, 'fp32 families')


def test_conv2d_bf16_mfma_every_compiled_tile_configuration(ops):
    rng = np.random.default_rng(77)
    cfgs = ops.conv_configs()
    problems = {}
    seen = 0
    try:
        for i, (ks, dil, th, tw, waves, fa, bnf, ck, pool, lds, flags) in enumerate(cfgs):
            if pool < 2 or flags & 2:                         # (bit 1: cell-update instances, test_convlstm_cell_update_...)
                continue
            if flags & 24:                                    # (bits 3 / 4: octet-layout instances, test_gpu_bf16_octets.py)
                continue
            seen += 1
            in32 = pool == 3                                  # float32-stored input, rounded by the loader
            key = (ks, dil, in32, ck == 8)
            if key not in problems:
                n, cin, h, w, cout = 2, 52, 19, 50, 36        # ragged tiles, ragged chunks for CK = 16 / 32 / 48
                if ck == 8:                                   # tap-packed instances: at most one octet of input channels
                    cin = 6
                x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
                xr = np_ref.round_bf16(x).astype(np.float32)
                wt = np_ref.glorot_uniform((ks, ks, cin, cout), rng)
                b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
                p = dil * (ks - 1) // 2
                pads = (p, p, p, p)
                want = _conv_ref(xr, np_ref.round_bf16(wt), b, dil, pads, 0, 1, 'tanh', 0)
                problems[key] = (dev(x) if in32 else dev(xr).to(torch.bfloat16), dev(wt), dev(b), pads, want, cout)
            xd, wd, bd, pads, want, cout = problems[key]
            ops.force_conv_config(i)
            cd = ops.make_conv(cout, ks, ks, dil, ops.make_pad(*pads, 0, 1), ops.ACT_TANH)
            out = torch.empty(want.shape, dtype=torch.float32, device='cuda')
            ops.conv2d(xd, wd, bd, cd, out=out, compute_bf16=in32)
            _check_conv(ops, out.cpu().numpy(), want, 'config %d %r' % (i, cfgs[i]))
    finally:
        ops.force_conv_config(-1)
    assert seen >= 4


def test_conv2d_bf16_mfma_full_size_properties(ops):
    """BASELINE.json config 4's largest layer (180x360, 48 -> 32 channels): linearity in the input, longitude-shift
    equivariance (bit for bit: the family is translation invariant in even shifts... and in every shift) and batch
    invariance."""
    rng = np.random.default_rng(8)
    n, cin, h, w, cout = 2, 48, 180, 360, 32
    x = np_ref.round_bf16(0.5 * rng.standard_normal((n, cin, h, w))).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR)
    xd = dev(x).to(torch.bfloat16)
    y = ops.conv2d(xd, dev(wt), None, cd)
    assert y.dtype == torch.bfloat16
    yf = torch.empty(y.shape, dtype=torch.float32, device='cuda')
    ops.conv2d(xd, dev(wt), None, cd, out=yf)
    # shift by an odd and an even number of columns: same sums in the same order
    for s in (1, 46):
        ys = torch.empty_like(yf)
        ops.conv2d(torch.roll(xd, s, dims=3).contiguous(), dev(wt), None, cd, out=ys)
        assert torch.equal(ys, torch.roll(yf, s, dims=3))
    # batch invariance: sample 1 alone == sample 1 inside the batch
    y1 = torch.empty((1,) + tuple(yf.shape[1:]), dtype=torch.float32, device='cuda')
    ops.conv2d(xd[1:2].contiguous(), dev(wt), None, cd, out=y1)
    assert torch.equal(y1[0], yf[1])
    # linearity: conv(2x) == 2 conv(x) exactly (a power of two)
    y2 = torch.empty_like(yf)
    ops.conv2d((xd * 2).contiguous(), dev(wt), None, cd, out=y2)
    assert torch.equal(y2, 2 * yf)
    # against the oracle on a corner crop that includes the periodic seam and the zero pole rows
    want = _conv_ref(x[:1], np_ref.round_bf16(wt), np.zeros(cout, np.float32), 1, (1, 1, 1, 1), 0, 1, 'linear', 0)
    _check_conv(ops, yf[:1].cpu().numpy(), want, 'full size')


def test_maxpool2_bfloat16_is_exact(ops):
    rng = np.random.default_rng(77)
    for shape in ((3, 5, 8, 12), (2, 3, 9, 7)):
        x = np_ref.round_bf16(rng.standard_normal(shape)).astype(np.float32)
        got = ops.maxpool2(dev(x).to(torch.bfloat16))
        assert got.dtype == torch.bfloat16
        assert np.array_equal(got.to(torch.float32).cpu().numpy(), np_ref.maxpool2(x))


# ----------------------------------------------------------------------------------------------------------------- #
# MaxPooling2D(2) in the producing convolution's epilogue (inference plans)
# ----------------------------------------------------------------------------------------------------------------- #

@pytest.mark.parametrize('case', [
    # (n, cin, h, w, cout, k, dil, act)
    (2, 4, 16, 40, 32, 3, 2, 'tanh'),            # direct MFMA, the first U-Net layer (8x32 tiles: a wave = two rows)
    (2, 4, 19, 37, 20, 3, 2, 'tanh'),            # ragged: odd output size (floor), partial channel fragment
    (2, 32, 16, 40, 64, 3, 1, 'tanh'),           # Winograd: the lane's 2x2 tile is the pooling window
    (3, 24, 13, 27, 32, 3, 1, 'relu'),           # Winograd, odd sizes
    (2, 16, 12, 20, 32, 3, 1, 'linear'),
    (2, 48, 16, 72, 32, 3, 2, 'tanh'),           # Winograd, dilation 2: the window's outputs sit in four lanes -> max from LDS
    (3, 16, 13, 37, 64, 3, 2, 'relu'),           # ... odd sizes
])
@pytest.mark.parametrize('out16', [False, True])
def test_conv2d_with_pooling_epilogue(ops, case, out16):
    n, cin, h, w, cout, k, dil, act = case
    rng = np.random.default_rng(sum(case[:7]))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    p = dil * (k - 1) // 2
    pads = (p, p, p, p)
    want = np_ref.maxpool2(_conv_ref(x, wt, b, dil, pads, 0, 1, act, 0))
    actc = {'tanh': ops.ACT_TANH, 'relu': ops.ACT_RELU, 'linear': ops.ACT_LINEAR}[act]
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, 0, 1), actc, out_pool=True)
    assert ops.supports_out_pool((cin, h, w), ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, 0, 1), actc))
    ys = ops.conv_out_shape(ops.Shape4(n, cin, h, w), cd)
    assert (ys.h, ys.w) == (h // 2, w // 2)
    out = torch.full(want.shape, float('nan'), dtype=torch.bfloat16 if out16 else torch.float32, device='cuda')
    ops.conv2d(dev(x), dev(wt), dev(b), cd, out=out)
    got = out.to(torch.float32).cpu().numpy()
    if out16:
        assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 1e-6)
    else:
        _check_conv(ops, got, want, 'pooled epilogue')


def test_pooling_epilogue_is_refused_where_no_kernel_has_one(ops):
    cd = ops.make_conv(4, 7, 7, 1, ops.make_pad(3, 3, 3, 3, 0, 1), ops.ACT_LINEAR)       # 7x7: no MFMA instance at all
    assert not ops.supports_out_pool((8, 16, 40), cd)
    cd.out_pool = 1
    x = torch.zeros((1, 8, 16, 40), device='cuda')
    from dlwp_amd._lib import DlwpError
    with pytest.raises(DlwpError, match="pooling epilogue"):
        ops.conv2d(x, torch.zeros((7, 7, 8, 4), device='cuda'), None, cd)


# ----------------------------------------------------------------------------------------------------------------- #
# seeded random sweep over every kernel family / loader / epilogue combination
# ----------------------------------------------------------------------------------------------------------------- #

def test_conv2d_random_shapes_against_oracle(ops):
    """60 seeded random layer geometries (odd sizes, ragged channels, both dilations, all loaders, both halo modes,
    pooling epilogue where a kernel has one, mixed bf16 / fp32 storage) against the float64 oracle."""
    rng = np.random.default_rng(20240607)
    n_pool = n_wino = n_bf16 = 0
    for case in range(60):
        k = int(rng.choice([3, 3, 3, 5]))
        dil = int(rng.choice([1, 2])) if k == 3 else 1
        cin = int(rng.choice([1, 3, 4, 8, 16, 20, 24, 32, 40]))
        cout = int(rng.choice([2, 4, 12, 32, 36, 64, 96]))
        src = int(rng.choice([0, 0, 1, 2]))
        h, w = int(rng.integers(6, 30)), int(rng.integers(8, 50))
        if src == 2:
            h, w = h + 6, w + 8
        n = int(rng.integers(1, 4))
        mode_h, mode_w = int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2]))
        act = str(rng.choice(['tanh', 'linear', 'relu']))
        p = dil * (k - 1) // 2
        pads = (p, p, p, p)
        hh, ww = (h * 2, w * 2) if src == 1 else ((h // 2, w // 2) if src == 2 else (h, w))
        if (mode_h == 1 and p > hh) or (mode_w == 1 and p > ww):
            continue
        x = np_ref.round_bf16(rng.standard_normal((n, cin, h, w))).astype(np.float32)
        wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
        b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        actc = {'tanh': ops.ACT_TANH, 'relu': ops.ACT_RELU, 'linear': ops.ACT_LINEAR}[act]
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mode_h, mode_w), actc, src_mode=src)
        want = _conv_ref(x, wt, b, dil, pads, mode_h, mode_w, act, src)
        pool = bool(rng.integers(0, 2)) and ops.supports_out_pool((cin, h, w), cd) and min(want.shape[2:]) >= 2
        in16, out16 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if pool:
            cd.out_pool = 1
        w_ref, on16 = _weights_as_multiplied(ops, wt, x.shape, cd, in16, out16)
        n_bf16 += int(on16)
        if on16:
            want = _conv_ref(x, w_ref, b, dil, pads, mode_h, mode_w, act, src)
        if pool:
            want = np_ref.maxpool2(want)
            n_pool += 1
        xd = dev(x).to(torch.bfloat16) if in16 else dev(x)
        out = torch.full(want.shape, float('nan'), dtype=torch.bfloat16 if out16 else torch.float32, device='cuda')
        ops.conv2d(xd, dev(wt), dev(b), cd, out=out)
        got = out.to(torch.float32).cpu().numpy()
        what = 'case %d: n%d %d->%d k%d d%d src%d %dx%d modes %d/%d %s pool%d io %d/%d' % (
            case, n, cin, cout, k, dil, src, h, w, mode_h, mode_w, act, pool, in16, out16)
        if out16:
            assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 2e-6), what
        else:
            _check_conv(ops, got, want, what)
        n_wino += int(k == 3 and cin % 8 == 0 and cout % 32 == 0 and src != 2)
    # the sweep really reaches the pooled epilogues, the Winograd family and the bf16 matrix-core family
    assert n_pool >= 3 and n_wino >= 5 and n_bf16 >= 2, (n_pool, n_wino, n_bf16)


# ----------------------------------------------------------------------------------------------------------------- #
# Conv2D on an up-sampled tensor restated on its source: derived kernels and the depth-to-space interleave
# ----------------------------------------------------------------------------------------------------------------- #

@pytest.mark.parametrize('k,pt,pl', [(5, 2, 2), (3, 1, 1), (7, 3, 3), (5, 1, 2), (4, 1, 2)])
def test_phase_weights_match_the_oracle_restatement(ops, k, pt, pl):
    rng = np.random.default_rng(k * 10 + pt)
    w = rng.standard_normal((k, k, 6, 5)).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    w2_ref, b2_ref, (lo_h, hi_h, lo_w, hi_w) = np_ref.phase_weights(w, b, pt, pl)
    assert ops.phase_geometry(k, pt) == (hi_h - lo_h + 1, lo_h, hi_h) and ops.phase_geometry(k, pl) == (hi_w - lo_w + 1, lo_w, hi_w)
    w2, b2 = ops.phase_weights(dev(w), dev(b), pt, pl)
    assert tuple(w2.shape) == w2_ref.shape
    assert np.abs(host(w2) - w2_ref).max() < 1e-6 and np.array_equal(host(b2), b2_ref.astype(np.float32))
    w2n, b2n = ops.phase_weights(dev(w), None, pt, pl)
    assert b2n is None and torch.equal(w2n, w2)


def test_depth_to_space_is_exact(ops):
    rng = np.random.default_rng(4)
    for n, f, h, w in ((2, 4, 5, 7), (1, 3, 1, 1), (3, 1, 8, 6)):
        y = rng.standard_normal((n, 4 * f, h, w)).astype(np.float32)
        got = host(ops.depth_to_space2(dev(y), f))
        assert np.array_equal(got, np_ref.depth_to_space2(y, f))
        out = torch.full((n, f + 3, 2 * h, 2 * w), 7.0, device='cuda')
        ops.depth_to_space2(dev(y), f, out=out, c_off=2)
        o = host(out)
        assert np.array_equal(o[:, 2:2 + f], np_ref.depth_to_space2(y, f)) and np.all(o[:, :2] == 7.0) and np.all(o[:, 2 + f:] == 7.0)


@pytest.mark.parametrize('k,dil,pads,mh,mw', [(5, 1, (2, 2, 2, 2), 0, 1), (5, 1, (2, 2, 2, 2), 2, 0), (7, 1, (3, 3, 3, 3), 1, 1)])
def test_conv_on_upsampled_source_restated_path_equals_the_fused_upsampling_loader(ops, k, dil, pads, mh, mw):
    """The two ways the library can run UpSampling2D -> padding -> Conv2D: the loader that up-samples on the fly, and the
    phase kernels on the source + depth-to-space.  Same function (fp32 rounding apart), both against the oracle."""
    rng = np.random.default_rng(k)
    n, cin, h, w, cout = 2, 16, 9, 14, 4
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((k, k, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    want = _conv_ref(x, wt, b, dil, pads, mh, mw, 'tanh', 1)
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(*pads, mh, mw), ops.ACT_TANH, src_mode=1)
    _check_conv(ops, host(ops.conv2d(dev(x), dev(wt), dev(b), cd)), want, 'fused loader')
    w2, b2 = ops.phase_weights(dev(wt), dev(b), pads[0], pads[2])
    (k2h, lo_h, hi_h), (k2w, lo_w, hi_w) = ops.phase_geometry(k, pads[0]), ops.phase_geometry(k, pads[2])
    cd2 = ops.make_conv(4 * cout, k2h, k2w, 1, ops.make_pad(-lo_h, hi_h, -lo_w, hi_w, mh, mw), ops.ACT_TANH)
    got = host(ops.depth_to_space2(ops.conv2d(dev(x), w2, b2, cd2), cout))
    _check_conv(ops, got, want, 'restated')


@pytest.mark.parametrize('k,pt,pl', [(5, 2, 2), (3, 1, 1), (5, 1, 2)])
def test_phase_weights_adjoint_and_space_to_depth(ops, k, pt, pl):
    """dlwp_phase_weights_bwd is the transpose of the (linear) map w -> w2: <w2(w), g2> == <w, bwd(g2)> and bias likewise;
    dlwp_space_to_depth2 inverts dlwp_depth_to_space2 exactly."""
    rng = np.random.default_rng(k + pt)
    cin, cout = 5, 3
    w = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    w2, b2 = ops.phase_weights(dev(w), dev(b), pt, pl)
    g2 = rng.standard_normal(tuple(w2.shape)).astype(np.float32)
    gb2 = rng.standard_normal(4 * cout).astype(np.float32)
    dw = torch.empty((k, k, cin, cout), device='cuda')
    db = torch.empty(cout, device='cuda')
    ops.phase_weights_bwd(dev(g2), dev(gb2), dw, db, pt, pl)
    lhs = float((w2.double().cpu().numpy() * g2).sum()) + float((b2.double().cpu().numpy() * gb2).sum())
    rhs = float((w.astype(np.float64) * host(dw)).sum()) + float((b.astype(np.float64) * host(db)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    dw_acc = dw.clone()
    ops.phase_weights_bwd(dev(g2), dev(gb2), dw_acc, db.clone(), pt, pl, accumulate=True)
    assert torch.allclose(dw_acc, 2 * dw)
    y = rng.standard_normal((2, 4 * cout, 5, 7)).astype(np.float32)
    hi = ops.depth_to_space2(dev(y), cout)
    assert np.array_equal(host(ops.space_to_depth2(hi, cout)), y)


XLD_CASES = [
    # (n, cin, h, w of the STORED input, cout, mode_h, mode_w, src_mode, loader the launch must take)
    # (plain sources carry 16 channels: 5-8 channels of a plain source are direct-family layers since r6)
    (3, 16, 11, 23, 64, 0, 1, 1, 2),    # up-sampled 11 x 23 -> 22 x 46: ragged tiles both ways, zero rows / periodic columns
    (2, 24, 22, 45, 32, 1, 1, 1, 2),    # the U-Net's layer-4 source size, periodic both ways, three chunks
    (2, 8, 9, 20, 64, 2, 0, 1, 2),      # edge rows, zero columns
    (2, 16, 5, 16, 32, 0, 2, 1, 2),     # edge columns, one tile row
    (2, 8, 6, 10, 32, 4, 4, 1, 0),      # SYMMETRIC halo: does not commute with the replication -> element by element
    (3, 16, 13, 46, 32, 0, 1, 0, 1),    # plain source, even width: column pairs, ragged last column tile
    (2, 32, 44, 90, 64, 0, 1, 0, 1),    # the dominant layer's geometry (two 32-channel tiles)
    (2, 16, 10, 34, 32, 2, 0, 0, 1),     # zero columns, edge rows
    (2, 16, 12, 64, 32, 1, 1, 0, 1),     # whole tiles, periodic both ways
    (2, 16, 12, 45, 32, 0, 1, 0, 0),     # odd width: a pair would straddle the seam -> element by element
    (2, 16, 12, 20, 32, 0, 3, 0, 0),     # REFLECT columns -> element by element
    (2, 8, 7, 12, 32, 1, 1, 1, 2, (3, 1, 1, 3)),   # up-sampled, halos of 3 (top) and 3 (right): the window starts two source rows out
    (2, 8, 7, 12, 32, 2, 0, 1, 2, (1, 3, 3, 1)),   # ... 3 on the left under a zero halo, edge rows
    (2, 16, 10, 36, 32, 0, 1, 0, 1, (1, 1, 3, 3)),  # plain source, column halo of 3: pairs start three columns out
    (2, 16, 10, 36, 32, 0, 1, 0, 0, (1, 1, 2, 2)),  # even column halo: the first pair would start on an odd column -> element by element
]


@pytest.mark.parametrize('case', XLD_CASES)
def test_winograd_input_loaders_give_the_bits_of_the_element_wise_loader(ops, case):
    """DLWP_OPT_WINO_XLOADER (r5): the 8 x 32 Winograd instances fetch a plain source of even width as image-aligned column pairs
    (WinoCfg::PAIRX) and an up-sampled source at source resolution (WinoCfg::UPSQ).  Same values in the same patch positions: the
    launch must give the BITS of the element-wise loader -- on ragged tiles, every halo mode the loaders accept, several chunks --,
    dlwp_conv2d_launch_info must name the loader that ran, and the result is the float64 oracle's."""
    n, cin, h, w, cout, mh, mw, src, want_loader = case[:9]
    pads = case[9] if len(case) > 9 else (1, 1, 1, 1)
    rng = np.random.default_rng(7000 + XLD_CASES.index(case))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(*pads, mh, mw), ops.ACT_TANH, src_mode=src)
    cfgs = ops.conv_configs()
    xd, wd, bd = dev(x), dev(wt), dev(b)
    ran = 0
    try:
        for bnf in (2, 4):
            idx = [i for i, c in enumerate(cfgs) if c[:8] == (3, 1, 8, 32, 4, 0, bnf, 8) and not (c[10] & 1)]
            if not idx or cout % (16 * bnf):
                continue
            if bnf == 4 and src != 1:
                continue                               # (64-channel blocks exist for the 9-position variants only)
            ops.force_conv_config(idx[0])
            prev = ops.set_wino_xloader(0)
            try:
                info0 = ops.conv_launch_info(x.shape, cd)
                base = ops.conv2d(xd, wd, bd, cd)
                ops.set_wino_xloader(3)
                info = ops.conv_launch_info(x.shape, cd)
                got = ops.conv2d(xd, wd, bd, cd)
            finally:
                ops.set_wino_xloader(prev)
            assert info0[0][0] == idx[0] and info0[0][5] == 0, info0
            assert info[0][0] == idx[0] and info[0][5] == want_loader, (info, want_loader)
            assert torch.equal(got, base), (case, bnf, float((got - base).abs().max()))
            ran += 1
    finally:
        ops.force_conv_config(-1)
    assert ran >= 1
    want = _conv_ref(x, wt, b, 1, pads, mh, mw, 'tanh', src)
    _check_conv(ops, host(got), want, 'winograd loader %d' % want_loader)


EP_CASES = [
    # (n, cin, h, w of the STORED input, cout, mode_h, mode_w, src_mode, pooled epilogue, launch-info loader code at mask 7)
    # (16 channels and more: 5-8 channels of a plain source are direct-family layers since r6)
    (3, 16, 44, 90, 64, 0, 1, 0, False, 5),    # the U-Net's 44 x 90 layers: 5.5 tile rows, three column tiles (one pair + a single)
    (2, 32, 12, 64, 32, 1, 1, 0, True, 5),     # MaxPooling2D(2) in the epilogue, two column tiles, periodic rows
    (2, 16, 20, 70, 32, 0, 1, 0, False, 5),     # the map's right edge cuts the second tile's last pixel quad
    (2, 16, 11, 66, 32, 0, 0, 0, False, 5),     # three valid rows in the last tile, zero columns
    (2, 16, 12, 70, 32, 0, 1, 0, True, 5),      # pooled, pooled width 35: element stores at the edge
    (2, 16, 6, 33, 64, 0, 1, 1, False, 2),     # up-sampled source: stays on its source-resolution fetch (no pairs compiled)
    (2, 16, 12, 66, 32, 0, 2, 0, False, 0),     # edge columns: no column pairs, hence no edge pairs
    (2, 16, 44, 32, 32, 0, 1, 0, False, 1),     # one column tile: nothing to pair
    (2, 16, 48, 64, 32, 0, 1, 0, False, 1),     # whole tile rows
    (2, 16, 14, 64, 32, 0, 1, 0, False, 1),     # six valid rows in the last tile: more than half
]


@pytest.mark.parametrize('case', EP_CASES)
def test_winograd_edge_pairs_give_the_bits_of_the_plain_launch(ops, case):
    """DLWP_OPT_WINO_XLOADER bit 2 (WinoCfg::EP, r5): on a map whose last 8-row tile is at most half used the blocks of that tile
    row take the valid rows of two neighbouring column tiles each.  Per-lane offsets of the loader, the patch origin and the store
    phase change, the arithmetic does not: the launch must give the BITS of the launch without it (and the oracle's values), with
    fewer workgroups -- tiles_w (tiles_h - 1) + ceil(tiles_w / 2) per image and channel tile -- and only where it applies."""
    n, cin, h, w, cout, mh, mw, src, pool, code = case
    rng = np.random.default_rng(7300 + EP_CASES.index(case))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = np_ref.glorot_uniform((3, 3, cin, cout), rng)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, mh, mw), ops.ACT_TANH, src_mode=src, out_pool=pool)
    cfgs = ops.conv_configs()
    bnf = 4 if (src == 1 and cout % 64 == 0) else 2
    idx = [i for i, c in enumerate(cfgs) if c[:8] == (3, 1, 8, 32, 4, 0, bnf, 8) and not (c[10] & 1)][0]
    xd, wd, bd = dev(x), dev(wt), dev(b)
    ho, wo = (2 * h, 2 * w) if src == 1 else (h, w)
    th, tw = -(-ho // 8), -(-wo // 32)
    try:
        ops.force_conv_config(idx)
        prev = ops.set_wino_xloader(0)
        try:
            info0 = ops.conv_launch_info(x.shape, cd)
            base = ops.conv2d(xd, wd, bd, cd)
            ops.set_wino_xloader(7)
            info = ops.conv_launch_info(x.shape, cd)
            got = ops.conv2d(xd, wd, bd, cd)
        finally:
            ops.set_wino_xloader(prev)
    finally:
        ops.force_conv_config(-1)
    per_image = n * (cout // (16 * bnf))
    assert info0[0][0] == idx and info0[0][5] == 0 and info0[0][1] == per_image * th * tw, info0
    assert info[0][0] == idx and info[0][5] == code, (info, code)
    assert info[0][1] == per_image * ((tw * (th - 1) + (tw + 1) // 2) if code & 4 else th * tw), info
    assert torch.equal(got, base), (case, float((got - base).abs().max()))
    want = _conv_ref(x, wt, b, 1, (1, 1, 1, 1), mh, mw, 'tanh', src)
    _check_conv(ops, host(got), np_ref.maxpool2(want) if pool else want, 'edge pairs %d' % code)
