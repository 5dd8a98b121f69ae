"""dlwp_pair_begin / dlwp_pair_end (csrc/conv_pair.hip): a layer's weight gradient and its data gradient -- both read the layer's
pre-activation gradient, neither reads the other's output -- issued as ONE grid whose first blocks run the weight-gradient body and
whose other blocks run the data-gradient (forward-family Winograd) body.  Same bodies, so the bar is bit-equality with the two
separate launches; layers without a fused instance fall back to those launches inside the same calls.
Reference: the two gradient ops of a Conv2D in the Keras train step behind DLWP/model/models.py:188-228."""
import numpy as np
import pytest
import torch

from oracle import np_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


#         n  cin cout   h   w  src_mode stored  fused expected (the layers of the 88 x 180 U-Net at 8 samples)
CASES = [(8, 32, 16, 44, 90, 0, False, True),      # restated output layer: 32 -> 16 phase channels
         (8, 64, 32, 44, 90, 0, False, True),      # restated layer 5
         (8, 128, 64, 22, 45, 1, True, False),     # layer 4 (up-sampled source, the 2x2 sum in the data gradient's epilogue): no fused instance
         (8, 32, 64, 44, 90, 0, False, True),      # layer 2
         (3, 64, 128, 22, 45, 0, False, None),     # layer 3's shapes at a ragged batch: whatever the instances are, the same bits
         (8, 24, 40, 20, 36, 0, False, None),      # ragged channels
         (64, 64, 32, 44, 90, 0, False, False)]    # a full batch: more than two rounds of data-gradient workgroups -> two launches


@pytest.mark.parametrize('case', CASES)
def test_pair_launch_equals_the_two_separate_launches(case):
    from dlwp_amd import _lib, ops
    n, cin, cout, h, w, sm, stored, expect_fused = case
    rng = np.random.default_rng(77 + cin + cout)
    device = torch.device('cuda', torch.cuda.current_device())
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR, src_mode=sm)
    xs = _lib.Shape4(n, cin, h, w)
    ys = ops.conv_out_shape(xs, cd)
    wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
    x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    dz = dev(rng.standard_normal((n, cout, ys.h, ys.w)).astype(np.float32))
    prep = ops.conv2d_bwd_data_prepare(wt, cd, xs, stored=stored)
    hin, win = (2 * h, 2 * w) if sm == 1 else (h, w)
    dshape = (n, cin, h, w) if stored else (n, cin, hin, win)

    def run(paired):
        dw = torch.full((3, 3, cin, cout), float('nan'), device='cuda')
        dx = torch.full(dshape, float('nan'), device='cuda')
        if paired:
            ops.pair_begin(device)
        ops.conv2d_bwd_weight(x, dz, dw, cd, xs, ws_key=('pair-test', cin, cout))
        ops.conv2d_bwd_data(dz, wt, cd, xs, dx, prepared=prep, stored=stored)
        if paired:
            ops.pair_end(device)
        torch.cuda.synchronize()
        return dw, dx

    dw0, dx0 = run(False)
    before = ops.pair_fused_count(device)
    dw1, dx1 = run(True)
    fused = ops.pair_fused_count(device) - before
    assert torch.equal(dw0, dw1) and torch.equal(dx0, dx1), (case, fused)
    assert bool(torch.isfinite(dw1).all()) and bool(torch.isfinite(dx1).all())
    if expect_fused is not None:
        assert fused == (1 if expect_fused else 0), (case, fused)
    # the other order of the two calls, and the data gradient from the plain kernel (its flip runs at once, in front)
    ops.pair_begin(device)
    dx2 = torch.full(dshape, float('nan'), device='cuda')
    dw2 = torch.full((3, 3, cin, cout), float('nan'), device='cuda')
    if stored:
        assert ops.conv2d_bwd_data_stored(dz, wt, cd, xs, dx2)
    else:
        ops.conv2d_bwd_data(dz, wt, cd, xs, dx2)
    ops.conv2d_bwd_weight(x, dz, dw2, cd, xs, ws_key=('pair-test', cin, cout))
    ops.pair_end(device)
    assert torch.equal(dw0, dw2) and torch.equal(dx0, dx2)


def test_pair_mode_covers_only_what_it_says():
    """An empty pair, a pair with one call, a second weight gradient inside one pair (runs at once), a forward convolution
    inside a pair (never handed over), families without a fused instance, pair_end without begin, begin on an open pair (its
    launches go out, nothing is lost)."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(5)
    device = torch.device('cuda', torch.cuda.current_device())
    ops.pair_begin(device)
    ops.pair_end(device)
    with pytest.raises(_lib.DlwpError):
        ops.pair_end(device)
    ops.pair_begin(device)
    ops.pair_begin(device)          # a pair left open (an exception in the caller's step): flushed, the new one opens
    ops.pair_end(device)
    with pytest.raises(_lib.DlwpError):
        ops.pair_end(device)
    n, cin, cout, h, w = 8, 32, 64, 20, 36
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)
    xs = _lib.Shape4(n, cin, h, w)
    wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
    b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
    x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    dz = dev(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    y0 = ops.conv2d(x, wt, b, cd)
    dw0 = ops.conv2d_bwd_weight(x, dz, torch.empty((3, 3, cin, cout), device='cuda'), cd, xs, ws_key='pair-a')
    ops.pair_begin(device)
    y1 = ops.conv2d(x, wt, b, cd)                       # a forward convolution: issued at once
    dw1 = ops.conv2d_bwd_weight(x, dz, torch.empty((3, 3, cin, cout), device='cuda'), cd, xs, ws_key='pair-a')
    dw2 = ops.conv2d_bwd_weight(x, dz, torch.empty((3, 3, cin, cout), device='cuda'), cd, xs, ws_key='pair-b')   # second: at once
    ops.pair_end(device)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(dw0, dw1) and torch.equal(dw0, dw2)
    # an abandoned pair: its launches go out when the next pair opens
    ops.pair_begin(device)
    dw3 = ops.conv2d_bwd_weight(x, dz, torch.full((3, 3, cin, cout), float('nan'), device='cuda'), cd, xs, ws_key='pair-a')
    ops.pair_begin(device)
    ops.pair_end(device)
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw3)
    # 5x5 layer: neither call is of a family the mode hands over
    cd5 = ops.make_conv(4, 5, 5, 1, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_LINEAR)
    xs5 = _lib.Shape4(n, 32, h, w)
    w5 = dev(np_ref.glorot_uniform((5, 5, 32, 4), rng))
    dz5 = dev(rng.standard_normal((n, 4, h, w)).astype(np.float32))
    want_dw = ops.conv2d_bwd_weight(x, dz5, torch.empty((5, 5, 32, 4), device='cuda'), cd5, xs5, ws_key='pair-c')
    want_dx = ops.conv2d_bwd_data(dz5, w5, cd5, xs5, torch.empty((n, 32, h, w), device='cuda'))
    before = ops.pair_fused_count(device)
    ops.pair_begin(device)
    got_dw = ops.conv2d_bwd_weight(x, dz5, torch.empty((5, 5, 32, 4), device='cuda'), cd5, xs5, ws_key='pair-c')
    got_dx = ops.conv2d_bwd_data(dz5, w5, cd5, xs5, torch.empty((n, 32, h, w), device='cuda'))
    ops.pair_end(device)
    assert ops.pair_fused_count(device) == before
    assert torch.equal(want_dw, got_dw) and torch.equal(want_dx, got_dx)
