"""The folded training step (csrc/batch.hip, Trainer._fold_ok): weight preparations recorded and built by one launch,
final sums recorded and done by one launch, weight gradients on a second stream.  Same arithmetic as the one-launch-per-helper
step -- the prepared operands are bit-identical, the final sums differ by their (fixed) order of summation only -- so the
parity bar is the unfolded kernels' output, which the oracle tests of test_gpu_model.py pin (and which run folded by default).
Reference: the Keras train step behind DLWP/model/models.py:188-228."""
import numpy as np
import pytest
import torch

from oracle import np_ref
from tests.nets import unet_layers
from tests.test_gpu_model import _build, _torch_step, _weights_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


#            cin cout  k dil   h    w  mode_w src_mode stored
DGRAD_CASES = [(32, 64, 3, 1, 22, 45, 1, 0, False),      # Winograd data gradient (64 -> 32 channels)
               (64, 32, 3, 2, 24, 40, 1, 0, False),      # dilation 2
               (4, 32, 3, 2, 16, 24, 1, 0, False),       # direct family (4 output channels of the gradient: packed-N or direct)
               (32, 4, 5, 1, 16, 24, 1, 0, False),       # 5x5 output layer
               (64, 32, 3, 1, 11, 23, 1, 1, True),       # up-sampled source, 2x2 sum in the epilogue
               (128, 64, 3, 1, 11, 23, 0, 1, True),
               (16, 16, 3, 1, 10, 12, 1, 2, False),      # pooled source
               (24, 40, 3, 1, 9, 20, 2, 0, False)]       # edge halo: the fold-back route


def test_prepared_data_gradient_is_bit_identical_alone_and_batched():
    """dlwp_conv2d_bwd_data_prepared(prepare(w)) == dlwp_conv2d_bwd_data(w) bit for bit, whether each operand is built by
    its own call or all of them by ONE launch between prepare_begin / prepare_flush (together with forward operands)."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(41)
    n = 4
    built = []
    for (cin, cout, k, dil, h, w, mw, sm, stored) in DGRAD_CASES:
        p = dil * (k - 1) // 2
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, 0, mw), ops.ACT_LINEAR, src_mode=sm)
        xs = _lib.Shape4(n, cin, h, w)
        ys = ops.conv_out_shape(xs, cd)
        wt = dev(np_ref.glorot_uniform((k, k, cin, cout), rng))
        dz = dev(rng.standard_normal((n, cout, ys.h, ys.w)).astype(np.float32))
        hin, win = (2 * h, 2 * w) if sm == 1 else ((h // 2, w // 2) if sm == 2 else (h, w))
        shape = (n, cin, h, w) if stored else (n, cin, hin, win)
        want = torch.full(shape, float('nan'), device='cuda')
        if stored:
            assert ops.conv2d_bwd_data_stored(dz, wt, cd, xs, want)
        else:
            ops.conv2d_bwd_data(dz, wt, cd, xs, want)
        prep = ops.conv2d_bwd_data_prepare(wt, cd, xs, stored=stored)
        assert prep is not None
        got = torch.full(shape, float('nan'), device='cuda')
        ops.conv2d_bwd_data(dz, wt, cd, xs, got, prepared=prep, stored=stored)
        assert torch.equal(got, want), (cin, cout, k, dil, sm)
        built.append((cd, xs, wt, dz, want, prep, stored, shape))
    # all operands again in one launch, into fresh buffers, next to two forward preparations
    device = torch.device('cuda', torch.cuda.current_device())
    xf = dev(rng.standard_normal((n, 32, 22, 45)).astype(np.float32))
    wf = dev(np_ref.glorot_uniform((3, 3, 32, 64), rng))
    cf = ops.make_conv(64, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)
    uf_alone = ops.conv2d_prepare(xf, wf, cf)
    assert uf_alone is not None
    ops.prepare_begin(device)
    try:
        batched = [ops.conv2d_bwd_data_prepare(wt, cd, xs, stored=stored) for cd, xs, wt, _, _, _, stored, _ in built]
        uf = ops.conv2d_prepare(xf, wf, cf)
    finally:
        ops.prepare_flush(device)
    assert torch.equal(uf, uf_alone)
    for (cd, xs, wt, dz, want, prep, stored, shape), b in zip(built, batched):
        assert torch.equal(b, prep)
        got = torch.full(shape, float('nan'), device='cuda')
        ops.conv2d_bwd_data(dz, wt, cd, xs, got, prepared=b, stored=stored)
        assert torch.equal(got, want)


def test_deferred_final_sums_equal_the_immediate_ones():
    """Weight-gradient slab sums, bias-gradient and loss final sums recorded between reductions_begin / reductions_flush give
    the immediate kernels' results up to the order of a float32 sum; nothing is written before the flush."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(43)
    device = torch.device('cuda', torch.cuda.current_device())
    n = 8
    cases = []
    for j, (cin, cout, k, dil, h, w) in enumerate([(32, 64, 3, 1, 22, 45), (4, 32, 3, 2, 24, 36), (32, 16, 3, 1, 24, 36),
                                                    (32, 4, 5, 1, 16, 24), (128, 64, 3, 1, 22, 45)]):
        p = dil * (k - 1) // 2
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, 0, 1), ops.ACT_TANH)
        xs = _lib.Shape4(n, cin, h, w)
        x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        y = dev(np.tanh(rng.standard_normal((n, cout, h, w))).astype(np.float32))
        dy = dev(rng.standard_normal((n, cout, h, w)).astype(np.float32))
        dw = torch.empty((k, k, cin, cout), device='cuda')
        db = torch.empty(cout, device='cuda')
        dz = ops.act_bwd_bias_grad(y, dy, ops.ACT_TANH, db, cout)
        ops.conv2d_bwd_weight(x, dz, dw, cd, xs)
        cases.append((j, cd, xs, x, y, dy, dw, db))
    yp, yt = dev(rng.standard_normal((n, 4, 16, 24)).astype(np.float32)), dev(rng.standard_normal((n, 4, 16, 24)).astype(np.float32))
    out_now = torch.zeros(2, device='cuda')
    ops.mse_mae(yp, yt, out_now)
    torch.cuda.synchronize()
    outs = []
    out_def = torch.full((2,), -7.0, device='cuda')
    ops.reductions_begin(device)
    try:
        ops.mse_mae(yp, yt, out_def, ws_key=('t', 'mse'))
        for j, cd, xs, x, y, dy, dw, db in cases:
            dw2 = torch.full_like(dw, -7.0)
            db2 = torch.full_like(db, -7.0)
            dz = ops.act_bwd_bias_grad(y, dy, ops.ACT_TANH, db2, db.numel(), ws_key=('t', 'b', j))
            ops.conv2d_bwd_weight(x, dz, dw2, cd, xs, ws_key=('t', 'w', j))
            outs.append((dw2, db2))
        torch.cuda.synchronize()
        assert (out_def == -7.0).all() and all((a == -7.0).all() and (b == -7.0).all() for a, b in outs)   # not yet
    finally:
        ops.reductions_flush(device)
    torch.cuda.synchronize()
    assert torch.allclose(out_def, out_now, rtol=5e-6, atol=0)
    for (j, cd, xs, x, y, dy, dw, db), (dw2, db2) in zip(cases, outs):
        assert (dw2 - dw).abs().max().item() <= 2e-5 * dw.abs().max().item(), j
        assert (db2 - db).abs().max().item() <= 2e-5 * max(db.abs().max().item(), 1.0), j
    # accumulate: dw += second gradient, recorded after a first sum into the same tensor (flushes in between)
    j, cd, xs, x, y, dy, dw, db = cases[0]
    acc = torch.zeros_like(dw)
    ops.reductions_begin(device)
    try:
        dz = ops.act_bwd(y, dy, ops.ACT_TANH)
        ops.conv2d_bwd_weight(x, dz, acc, cd, xs, ws_key=('t', 'w', 'a'))
        ops.conv2d_bwd_weight(x, dz, acc, cd, xs, accumulate=True, ws_key=('t', 'w', 'b'))
    finally:
        ops.reductions_flush(device)
    assert (acc - 2 * dw).abs().max().item() <= 4e-5 * dw.abs().max().item()


@pytest.mark.parametrize('graph', [False, True])
def test_folded_step_equals_the_unfolded_step_and_the_autograd_oracle(monkeypatch, graph):
    """The whole step both ways from the same weights: gradients agree to float32 round-off of the final sums, and the folded
    ones meet the autograd bar of test_train_step_gradients_loss_and_adam_match_autograd_oracle; also as a captured hipGraph
    (the second stream forks and joins inside the capture)."""
    monkeypatch.setenv('DLWP_TRAIN_GRAPH', '1' if graph else '0')
    rng = np.random.default_rng(8)
    cs = (4, 16, 24)
    layers = unet_layers(cs)
    x = rng.standard_normal((6,) + cs).astype(np.float32)
    y = rng.standard_normal((6,) + cs).astype(np.float32)
    res = {}
    for fold in (True, False):
        d = _build(layers, time_dim=2)
        weights = _weights_of(d.model, np.random.default_rng(9))
        tr = d.model._trainer
        tr._fold = None if fold else False
        assert tr._fold_ok() == fold
        logs = [d.model.train_on_batch(x, y) for _ in range(5)]
        torch.cuda.synchronize()
        assert (len(tr._graphs) == 1) == graph
        res[fold] = (logs, tr.flat_grads.cpu().numpy().copy(), d.model.get_weights(), weights)
    for a, b in zip(res[True][0], res[False][0]):
        assert np.allclose(a, b, rtol=2e-5, atol=1e-7), (a, b)
    g1, g0 = res[True][1], res[False][1]
    assert np.abs(g1 - g0).max() <= 5e-6 * np.abs(g0).max()
    for a, b in zip(res[True][2], res[False][2]):
        assert np.abs(a - b).max() <= 5e-6
    # first step against the oracle
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, np.random.default_rng(9))
    loss_ref, mae_ref, grads_ref, _ = _torch_step(layers, weights, x, y)
    vals = d.model.train_on_batch(x, y)
    assert d.model._trainer._fold_ok()
    assert vals[0] == pytest.approx(loss_ref, rel=2e-5) and vals[1] == pytest.approx(mae_ref, rel=2e-5)
    off = 0
    tr = d.model._trainer
    for g_ref in grads_ref:
        g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
        off += g_ref.size
        assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape


def test_folded_step_launch_count():
    """What the fold is for: count the kernel launches of one step with the profiler-free method -- the handle's job tables
    are flushed once each, so the step's helpers shrink to two launches.  Checked through the recorded job counts."""
    from dlwp_amd import ops
    rng = np.random.default_rng(8)
    cs = (4, 16, 24)
    d = _build(unet_layers(cs), time_dim=2)
    tr = d.model._trainer
    assert tr._fold_ok()
    x = rng.standard_normal((6,) + cs).astype(np.float32)
    d.model.train_on_batch(x, x)
    prep = tr._prep_cache[6]
    n_conv = sum(1 for op in tr.plan.ops if op.kind == 'conv')
    assert len(prep['bwd']) == n_conv - 1                  # every layer but the first has a data gradient
    assert len(prep['fwd']) >= 4                           # the Winograd / packed-N layers of the forward
    assert ('cuda', torch.cuda.current_device(), ('wgrad', max(k for k in prep['bwd']))) in ops._workspaces


@pytest.mark.parametrize('n,f,h,w', [(3, 4, 8, 12), (5, 1, 7, 9), (8, 4, 44, 90)])
def test_loss_on_phase_channels_equals_the_separate_passes(n, f, h, w):
    """dlwp_mse_mae_phase == dlwp_depth_to_space2 -> dlwp_mse_mae -> dlwp_space_to_depth2 -> dlwp_bias_grad on the same data
    (sums in a different fixed order: float32 round-off), with and without the optional outputs."""
    from dlwp_amd import ops
    rng = np.random.default_rng(n * 100 + f)
    yph = dev(rng.standard_normal((n, 4 * f, h, w)).astype(np.float32))
    yt = dev(rng.standard_normal((n, f, 2 * h, 2 * w)).astype(np.float32))
    full = ops.depth_to_space2(yph, f)
    out_ref, dy = torch.zeros(2, device='cuda'), torch.empty_like(full)
    ops.mse_mae(full, yt, out_ref, dy, 0.7)
    dz_ref = ops.space_to_depth2(dy, f)
    db_ref = torch.empty(4 * f, device='cuda')
    ops.bias_grad(dz_ref, db_ref, 4 * f)
    out, dz, db = torch.zeros(2, device='cuda'), torch.full_like(yph, float('nan')), torch.empty(4 * f, device='cuda')
    ops.mse_mae_phase(yph, yt, out, dz, db, 0.7)
    assert torch.allclose(out, out_ref, rtol=2e-6, atol=0)
    assert torch.allclose(dz, dz_ref, rtol=1e-6, atol=0)
    assert torch.allclose(db, db_ref, rtol=1e-4, atol=1e-7)
    want = float(((full.double() - yt.double()) ** 2).mean())
    assert float(out[0]) == pytest.approx(want, rel=2e-6)
    out2 = torch.zeros(2, device='cuda')
    ops.mse_mae_phase(yph, yt, out2)                       # value only (test_on_batch)
    assert torch.equal(out2, out)


def test_step_with_the_loss_on_phase_channels_equals_the_step_with_separate_passes(monkeypatch):
    """The U-Net's restated 5x5 output layer: loss, gradient and bias gradient taken on its phase channels (default) against
    DLWP_PHASE_LOSS=0 (depth-to-space, loss, space-to-depth, bias gradient as four launches) -- the same numbers to float32
    round-off, eager and as a captured graph; evaluate() takes the same route."""
    rng = np.random.default_rng(18)
    cs = (4, 16, 24)
    layers = unet_layers(cs)
    x = rng.standard_normal((6,) + cs).astype(np.float32)
    y = rng.standard_normal((6,) + cs).astype(np.float32)
    res = {}
    for graph in ('0', '1'):
        for phase in ('1', '0'):
            monkeypatch.setenv('DLWP_TRAIN_GRAPH', graph)
            monkeypatch.setenv('DLWP_PHASE_LOSS', phase)
            d = _build(layers, time_dim=2)
            _weights_of(d.model, np.random.default_rng(9))
            tr = d.model._trainer
            logs = [d.model.train_on_batch(x, y) for _ in range(4)]
            logs.append(d.model.test_on_batch(x, y))
            torch.cuda.synchronize()
            assert bool(tr._phase_outputs()) == (phase == '1')
            res[(graph, phase)] = (logs, tr.flat_grads.cpu().numpy().copy(), d.model.get_weights())
    for graph in ('0', '1'):
        a, b = res[(graph, '1')], res[(graph, '0')]
        for la, lb in zip(a[0], b[0]):
            assert np.allclose(la, lb, rtol=2e-5, atol=1e-7), (la, lb)
        assert np.abs(a[1] - b[1]).max() <= 5e-6 * np.abs(b[1]).max()
        for wa, wb in zip(a[2], b[2]):
            assert np.abs(wa - wb).max() <= 5e-6


def test_copy_many_and_the_device_step_adam():
    """dlwp_copy_many: several tensors (16-byte aligned or not, multiples of 4 or not) in one launch; dlwp_adam_keras_dev: the
    update with the step number in device memory equals dlwp_adam_keras step for step, and advances the number itself."""
    from dlwp_amd import ops
    rng = np.random.default_rng(77)
    srcs = [dev(rng.standard_normal(s).astype(np.float32)) for s in [(8, 4, 10, 12), (1001,), (3, 5)]]
    srcs.append(dev(rng.standard_normal(260).astype(np.float32))[1:257])            # misaligned view
    dsts = [torch.full_like(s, float('nan')) for s in srcs]
    ops.copy_many(list(zip(srcs, dsts)))
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    n = 70001
    p0 = rng.standard_normal(n).astype(np.float32)
    pa, ma, va = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    pb, mb, vb = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    it_dev, ticket = torch.full((1,), 3, dtype=torch.int64, device='cuda'), torch.zeros(1, device='cuda')
    for step in range(3, 8):
        g = dev(rng.standard_normal(n).astype(np.float32))
        ops.adam_keras(pa, ma, va, g, step, lr=2e-3, decay=1e-3, grad_scale=0.5)
        ops.adam_keras_dev(pb, mb, vb, g, it_dev, ticket, lr=2e-3, decay=1e-3, grad_scale=0.5)
        assert int(it_dev) == step + 1
        assert torch.equal(ma, mb) and torch.equal(va, vb)
        assert torch.allclose(pa, pb, rtol=0, atol=1e-9)


@pytest.mark.parametrize('n,cin,cout,h,w,dil', [(3, 4, 32, 16, 24, 2), (2, 3, 36, 11, 21, 1), (4, 4, 32, 88, 180, 2), (2, 4, 16, 10, 36, 1)])
def test_weight_gradient_with_the_pooling_backward_in_its_loader(n, cin, cout, h, w, dil):
    """dlwp_conv2d_bwd_weight_pooled == dlwp_pool_act_bwd_bias_grad then dlwp_conv2d_bwd_weight (+ its bias gradient): the first
    layer under MaxPooling2D(2) without the gradient tensor in between.  Ties inside windows (the first maximum takes the
    gradient), odd rows / columns (no window: zero), ragged second cout tile, every activation."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(n * 7 + cout)
    x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1), ops.ACT_TANH)
    xs = _lib.Shape4(n, cin, h, w)
    assert ops.conv2d_bwd_weight_pooled_supported(xs, cd)
    big = ops.make_conv(cout, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1), ops.ACT_TANH)
    assert not ops.conv2d_bwd_weight_pooled_supported(_lib.Shape4(n, 8, h, w), big)          # more than 4 input channels
    for act in (ops.ACT_TANH, ops.ACT_RELU, ops.ACT_LINEAR):
        yv = np.tanh(rng.standard_normal((n, cout, h, w))).astype(np.float32)
        yv[rng.random(yv.shape) < 0.3] = 0.25                      # plenty of ties inside windows
        y = dev(yv)
        dp = dev(rng.standard_normal((n, cout, h // 2, w // 2)).astype(np.float32))
        db_ref = torch.empty(cout, device='cuda')
        dz = ops.pool_act_bwd_bias_grad(y, dp, act, db_ref)
        dw_ref = torch.empty((3, 3, cin, cout), device='cuda')
        ops.conv2d_bwd_weight(x, dz, dw_ref, cd, xs)
        dw, db = torch.full_like(dw_ref, float('nan')), torch.full_like(db_ref, float('nan'))
        ops.conv2d_bwd_weight_pooled(x, y, dp, dw, db, cd, xs, act)
        scale = max(1.0, float(dw_ref.abs().max()))
        assert float((dw - dw_ref).abs().max()) <= 2e-5 * scale, (act, float((dw - dw_ref).abs().max()))
        assert torch.allclose(db, db_ref, rtol=1e-4, atol=1e-4 * max(1.0, float(db_ref.abs().max())))
        dw2 = torch.full_like(dw_ref, float('nan'))
        ops.conv2d_bwd_weight_pooled(x, y, dp, dw2, None, cd, xs, act)          # without a bias
        assert torch.equal(dw2, dw)


def test_step_with_the_first_layers_pooling_backward_in_its_weight_gradient(monkeypatch):
    """The U-Net's first layer (4 fields in, MaxPooling2D(2) behind it): DLWP_WGRAD_POOLED=0 keeps the separate
    dlwp_pool_act_bwd_bias_grad launch -- the same step to float32 round-off, eager and captured."""
    rng = np.random.default_rng(19)
    cs = (4, 16, 24)
    layers = unet_layers(cs)
    x = rng.standard_normal((6,) + cs).astype(np.float32)
    y = rng.standard_normal((6,) + cs).astype(np.float32)
    res = {}
    for graph in ('0', '1'):
        for fused in ('1', '0'):
            monkeypatch.setenv('DLWP_TRAIN_GRAPH', graph)
            monkeypatch.setenv('DLWP_WGRAD_POOLED', fused)
            d = _build(layers, time_dim=2)
            _weights_of(d.model, np.random.default_rng(9))
            logs = [d.model.train_on_batch(x, y) for _ in range(4)]
            torch.cuda.synchronize()
            res[(graph, fused)] = (logs, d.model._trainer.flat_grads.cpu().numpy().copy(), d.model.get_weights())
    for graph in ('0', '1'):
        a, b = res[(graph, '1')], res[(graph, '0')]
        for la, lb in zip(a[0], b[0]):
            assert np.allclose(la, lb, rtol=2e-5, atol=1e-7), (la, lb)
        assert np.abs(a[1] - b[1]).max() <= 5e-6 * np.abs(b[1]).max()
        for wa, wb in zip(a[2], b[2]):
            assert np.abs(wa - wb).max() <= 5e-6


def test_convolution_that_stores_its_pooled_image_as_well():
    """dlwp_conv2d_fwd_pool2 == dlwp_conv2d_fwd followed by dlwp_maxpool2_fwd, both tensors bit for bit (the activations are
    monotonic: the maximum of the activated values is the activated maximum); a layer whose kernel has no such epilogue (Winograd)
    reports unsupported and writes nothing."""
    from dlwp_amd import ops
    rng = np.random.default_rng(31)
    for (n, cin, cout, h, w, dil, act) in [(3, 4, 32, 16, 24, 2, ops.ACT_TANH), (2, 4, 32, 88, 180, 2, ops.ACT_TANH),
                                           (2, 3, 20, 11, 21, 1, ops.ACT_RELU), (2, 4, 32, 10, 36, 2, ops.ACT_LINEAR)]:
        x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
        b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
        cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1), act)
        y_ref = ops.conv2d(x, wt, b, cd)
        p_ref = ops.maxpool2(y_ref)
        y = torch.full_like(y_ref, float('nan'))
        p = torch.full_like(p_ref, float('nan'))
        assert ops.conv2d(x, wt, b, cd, out=y, out_pool2=p) is not None, (cin, cout, h, w)
        assert torch.equal(y, y_ref) and torch.equal(p, p_ref), (cin, cout, h, w)
    # Winograd layers: the 8 x 32 / 32-channel instance has the second staging area (config 3's second layer at its grid, ragged
    # right edge: 90 = 2 x 32 + 26 columns), with prepared filters or not; where the heuristic takes another instance: None
    for (n, cin, cout, h, w) in [(8, 32, 64, 44, 90), (64, 32, 64, 44, 90), (4, 64, 32, 22, 46)]:
        x = dev(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
        b = dev((0.1 * rng.standard_normal(cout)).astype(np.float32))
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_TANH)
        y_ref = ops.conv2d(x, wt, b, cd)
        p_ref = ops.maxpool2(y_ref)
        for prep in (None, ops.conv2d_prepare(x, wt, cd)):
            y = torch.full_like(y_ref, float('nan'))
            p = torch.full_like(p_ref, float('nan'))
            if ops.conv2d(x, wt, b, cd, out=y, prepared=prep, out_pool2=p) is None:
                assert (n, h) != (64, 44), 'config 3, layer 2 at 64 samples must store both'
                assert torch.isnan(y).all() and torch.isnan(p).all()
                continue
            assert torch.equal(y, y_ref) and torch.equal(p, p_ref), (n, cin, cout, h, w, prep is not None)
    x = dev(rng.standard_normal((2, 32, 16, 24)).astype(np.float32))
    wt = dev(np_ref.glorot_uniform((5, 5, 32, 8), rng))
    cd = ops.make_conv(8, 5, 5, 1, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH)
    y = torch.full((2, 8, 16, 24), float('nan'), device='cuda')
    p = torch.full((2, 8, 8, 12), float('nan'), device='cuda')
    if ops.conv2d(x, wt, None, cd, out=y, out_pool2=p) is None:               # (a layer without such an epilogue writes nothing)
        assert torch.isnan(y).all() and torch.isnan(p).all()
    else:
        y_ref = ops.conv2d(x, wt, None, cd)
        assert torch.equal(y, y_ref) and torch.equal(p, ops.maxpool2(y_ref))


@pytest.mark.parametrize('n,cin,cout,h,w,act', [(4, 64, 32, 44, 90, 'tanh'), (8, 32, 16, 44, 90, 'tanh'), (64, 32, 16, 44, 90, 'tanh'),
                                               (4, 64, 64, 16, 40, 'relu'), (4, 96, 32, 11, 37, 'tanh')])
def test_data_gradient_with_the_activation_backward_in_its_store_phase(n, cin, cout, h, w, act):
    """dlwp_conv2d_bwd_data_act == dlwp_conv2d_bwd_data followed by dlwp_act_bwd_bias_grad on the layer's input (the activation
    output of the layer in front): the gradient bit for bit (the same products in the same order, one multiply later), the bias
    gradient to float32 round-off; prepared operand or not; ragged right edge (90, 37 columns)."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(cin + cout + h)
    a = ops.ACT_TANH if act == 'tanh' else ops.ACT_RELU
    xv = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    x = dev(np.tanh(xv) if act == 'tanh' else np.maximum(xv, 0.0))
    dz = dev(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    wt = dev(np_ref.glorot_uniform((3, 3, cin, cout), rng))
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR)
    xs = _lib.Shape4(n, cin, h, w)
    dx_ref = torch.empty((n, cin, h, w), device='cuda')
    ops.conv2d_bwd_data(dz, wt, cd, xs, dx_ref)
    db_ref = torch.empty(cin, device='cuda')
    ops.act_bwd_bias_grad(x, dx_ref, a, db_ref, cin, out=dx_ref)
    dx = torch.full_like(dx_ref, float('nan'))
    db = torch.full_like(db_ref, float('nan'))
    if not ops.conv2d_bwd_data_act(dz, wt, cd, xs, dx, x, a, db):
        assert (n, cin, h) != (4, 64, 44), 'the config-3 decoder shape must run fused'
        assert torch.isnan(dx).all()
        pytest.skip('the heuristic put this gradient on another instance: the caller keeps the two launches')
    assert torch.equal(dx, dx_ref)
    assert torch.allclose(db, db_ref, rtol=1e-4, atol=1e-4 * max(1.0, float(db_ref.abs().max())))
    prep = ops.conv2d_bwd_data_prepare(wt, cd, xs)
    dx2 = torch.full_like(dx_ref, float('nan'))
    assert ops.conv2d_bwd_data_act(dz, wt, cd, xs, dx2, x, a, None, prepared=prep)
    assert torch.equal(dx2, dx_ref)


def test_data_gradient_with_activation_backward_reports_unsupported_layers():
    """5x5 kernels, few channels and dilation 2 do not run on the Winograd instance with the fused store phase: False, nothing
    written."""
    from dlwp_amd import _lib, ops
    rng = np.random.default_rng(3)
    for (cin, cout, k, dil) in [(4, 32, 3, 2), (32, 4, 5, 1), (20, 32, 3, 1)]:
        p = dil * (k - 1) // 2
        x = dev(np.tanh(rng.standard_normal((2, cin, 16, 40))).astype(np.float32))
        dz = dev(rng.standard_normal((2, cout, 16, 40)).astype(np.float32))
        wt = dev(np_ref.glorot_uniform((k, k, cin, cout), rng))
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, 0, 1), ops.ACT_LINEAR)
        dx = torch.full((2, cin, 16, 40), float('nan'), device='cuda')
        assert not ops.conv2d_bwd_data_act(dz, wt, cd, _lib.Shape4(2, cin, 16, 40), dx, x, ops.ACT_TANH, None)
        assert torch.isnan(dx).all()


def test_step_with_the_activation_backward_in_the_data_gradients_equals_the_step_without(monkeypatch):
    """Config 3 at its full grid (the fused store phase exists on the 8 x 32 Winograd instance only, which small maps do not
    take): DLWP_DGRAD_ACT=0 keeps dlwp_conv2d_bwd_data + dlwp_act_bwd_bias_grad -- the same step to float32 round-off, eager and
    captured; and the planner found the two decoder layers."""
    from tests.nets import unet_layers as ul
    rng = np.random.default_rng(23)
    cs = (4, 88, 180)
    layers = ul(cs)
    x = rng.standard_normal((8,) + cs).astype(np.float32)
    y = rng.standard_normal((8,) + cs).astype(np.float32)
    res = {}
    for graph in ('0', '1'):
        for fused in ('1', '0'):
            monkeypatch.setenv('DLWP_TRAIN_GRAPH', graph)
            monkeypatch.setenv('DLWP_DGRAD_ACT', fused)
            d = _build(layers, time_dim=2)
            _weights_of(d.model, np.random.default_rng(9))
            tr = d.model._trainer
            logs = [d.model.train_on_batch(x, y) for _ in range(3)]
            torch.cuda.synchronize()
            assert (len(tr._dgrad_act_ops()) >= 2) == (fused == '1')
            res[(graph, fused)] = (logs, tr.flat_grads.cpu().numpy().copy(), d.model.get_weights())
    for graph in ('0', '1'):
        a, b = res[(graph, '1')], res[(graph, '0')]
        for la, lb in zip(a[0], b[0]):
            assert np.allclose(la, lb, rtol=2e-5, atol=1e-7), (la, lb)
        assert np.abs(a[1] - b[1]).max() <= 5e-6 * np.abs(b[1]).max()
        for wa, wb in zip(a[2], b[2]):
            assert np.abs(wa - wb).max() <= 5e-6
