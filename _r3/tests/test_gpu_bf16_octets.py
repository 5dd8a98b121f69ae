"""The octet layout DLWP_BF16_O8 = (N, C/8, H, W, 8) bfloat16 of the bf16 matrix-core convolutions (csrc/conv_fwd_bf16_kernel.h:
IN8 loaders, SW epilogues; BASELINE.json config 4).  It is a STORAGE choice between the layers: the same products are summed
in the same order, so every result must equal the NCHW-bfloat16 instances' bit for bit after re-ordering -- which the oracle
tests of test_gpu_model.py / test_gpu_configs.py pin against float64 (and which run in octets by default)."""
import numpy as np
import pytest
import torch

from oracle import np_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def to_o8(t):
    """(n, C, h, w) -> the same values in octet order, still shaped (n, C, h, w) (C % 8 == 0)"""
    n, c, h, w = t.shape
    return t.view(n, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous().view(n, c, h, w)


def from_o8(t):
    n, c, h, w = t.shape
    return t.view(n, c // 8, h, w, 8).permute(0, 1, 4, 2, 3).contiguous().view(n, c, h, w)


def test_octet_helpers_round_trip():
    x = torch.arange(2 * 16 * 3 * 4, dtype=torch.float32, device='cuda').view(2, 16, 3, 4).bfloat16()
    assert torch.equal(from_o8(to_o8(x)), x)
    o = to_o8(x).view(2, 2, 3, 4, 8)
    assert torch.equal(o[1, 1, 2, 3], x[1, 8:16, 2, 3])


#        cin cout k dil   h    w  mode_w src pool  in_window        out_window
CASES = [(32, 64, 3, 1, 22, 46, 1, 0, False, None, None),
         (48, 32, 3, 2, 24, 40, 1, 0, True, None, None),            # config 4's first U-Net layer: dilation 2 + pooling epilogue
         (32, 64, 3, 1, 18, 36, 1, 0, True, None, None),
         (64, 128, 3, 1, 11, 22, 0, 0, False, None, None),          # zero column halo
         (128, 64, 3, 1, 11, 22, 1, 1, False, None, None),          # up-sampled source
         (24, 96, 3, 1, 20, 40, 0, 0, False, (0, 48), None),        # a 24-channel window (3 octets) of a 48-channel buffer
         (24, 32, 3, 1, 20, 40, 1, 0, False, (24, 48), (32, 96)),   # windows on both sides
         (16, 32, 3, 1, 9, 70, 1, 0, False, None, None)]            # 4 x 64 tiles territory, ragged rows


@pytest.mark.parametrize('case', CASES)
def test_octet_convolutions_equal_the_nchw_bf16_instances_bit_for_bit(case):
    from dlwp_amd import ops
    cin, cout, k, dil, h, w, mw, sm, pool, iw, ow = case
    rng = np.random.default_rng(cin * 1000 + cout + h)
    n = 3
    p = dil * (k - 1) // 2
    c_in_tot = iw[1] if iw else cin
    c_out_tot = ow[1] if ow else cout
    cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, 0, mw), ops.ACT_TANH, in_c_off=iw[0] if iw else 0,
                       in_c_total=c_in_tot if iw else 0, out_c_off=ow[0] if ow else 0, out_c_total=c_out_tot if ow else 0,
                       src_mode=sm, out_pool=pool)
    x = torch.from_numpy(rng.standard_normal((n, c_in_tot, h, w)).astype(np.float32)).cuda().bfloat16()
    wt = torch.from_numpy(np_ref.glorot_uniform((k, k, cin, cout), rng)).cuda()
    b = torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32)).cuda()
    ho, wo = (2 * h, 2 * w) if sm == 1 else (h, w)
    if pool:
        ho, wo = ho // 2, wo // 2

    def fresh(dtype):
        return torch.full((n, c_out_tot, ho, wo), 3.0, device='cuda').to(dtype)
    want16 = ops.conv2d(x, wt, b, cd, out=fresh(torch.bfloat16), x_channels=cin)            # NCHW bf16 -> NCHW bf16
    want32 = ops.conv2d(x, wt, b, cd, out=fresh(torch.float32), x_channels=cin)             # NCHW bf16 -> float32
    xo = to_o8(x)
    got = ops.conv2d(xo, wt, b, cd, out=to_o8(fresh(torch.bfloat16)), x_channels=cin, in_o8=True, out_o8=True)
    assert torch.equal(from_o8(got), want16), 'O8 -> O8'
    got32 = ops.conv2d(xo, wt, b, cd, out=fresh(torch.float32), x_channels=cin, in_o8=True)
    assert torch.equal(got32, want32), 'O8 -> float32'
    got16 = ops.conv2d(xo, wt, b, cd, out=fresh(torch.bfloat16), x_channels=cin, in_o8=True)
    assert torch.equal(got16, want16), 'O8 -> NCHW bf16'
    # prepared weights (the rollout graph's route) give the same launch
    prep = ops.conv2d_prepare(xo, wt, cd, out_dtype=torch.bfloat16, x_channels=cin, in_o8=True, out_o8=True)
    assert prep is not None
    got_p = ops.conv2d(xo, wt, b, cd, out=to_o8(fresh(torch.bfloat16)), x_channels=cin, prepared=prep, in_o8=True, out_o8=True)
    assert torch.equal(got_p, got)


@pytest.mark.parametrize('dil,h,w', [(2, 20, 40), (1, 12, 36)])
def test_float32_state_into_octets(dil, h, w):
    """The layers that read the float32 model state (rounded by the loader, DLWP_COMPUTE_BF16) and write octets: config 4's
    ConvLSTM2D input convolution of a later step."""
    from dlwp_amd import ops
    rng = np.random.default_rng(7)
    n, cin, cout = 3, 6, 96
    cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1), ops.ACT_LINEAR, in_c_off=6, in_c_total=12)
    x = torch.from_numpy(rng.standard_normal((n, 12, h, w)).astype(np.float32)).cuda()
    wt = torch.from_numpy(np_ref.glorot_uniform((3, 3, cin, cout), rng)).cuda()
    b = torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32)).cuda()
    want = ops.conv2d(x, wt, b, cd, out=torch.empty((n, cout, h, w), device='cuda', dtype=torch.bfloat16), x_channels=cin,
                      compute_bf16=True)
    got = ops.conv2d(x, wt, b, cd, out=torch.empty((n, cout, h, w), device='cuda', dtype=torch.bfloat16), x_channels=cin,
                     compute_bf16=True, out_o8=True)
    assert torch.equal(from_o8(got), want)


def to_o8_f32(t):
    return to_o8(t)          # the same re-ordering on float32 (the cell state: (n, F/8, h, w, 8) float32)


@pytest.mark.parametrize('first', [True, False])
def test_convlstm_step_in_octets_equals_the_nchw_step(first):
    """dlwp_convlstm_conv_fwd with an octet output: h in octets, z_add read in octets, the float32 cell state in octets --
    the same cell update as the NCHW instance, bit for bit.  first: the input convolution of step 1 (float32 state in, no
    z_add / c_prev); otherwise the recurrent convolution of a later step (h in octets, z_add, c_prev)."""
    from dlwp_amd import ops
    rng = np.random.default_rng(11 if first else 12)
    n, f, h, w = 3, 24, 20, 40
    if first:
        cin, dil = 6, 2
        x = torch.from_numpy(rng.standard_normal((n, 12, h, w)).astype(np.float32)).cuda()
        in_off, in_tot = 0, 12
    else:
        cin, dil = f, 1
        x = torch.from_numpy(rng.standard_normal((n, 2 * f, h, w)).astype(np.float32)).cuda().bfloat16()
        in_off, in_tot = 0, 2 * f
    out_off = 0 if first else f
    cd = ops.make_conv(4 * f, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1 if first else 0), ops.ACT_TANH, in_c_off=in_off,
                       in_c_total=in_tot, out_c_off=out_off, out_c_total=2 * f, lstm_f=f, lstm_rec_act=0)
    wt = torch.from_numpy(np_ref.glorot_uniform((3, 3, cin, 4 * f), rng)).cuda()
    b = torch.from_numpy((0.1 * rng.standard_normal(4 * f)).astype(np.float32)).cuda()
    z = None if first else torch.from_numpy(rng.standard_normal((n, 4 * f, h, w)).astype(np.float32)).cuda().bfloat16()
    cp = None if first else torch.from_numpy(rng.standard_normal((n, f, h, w)).astype(np.float32)).cuda()
    hseq = torch.from_numpy(rng.standard_normal((n, 2 * f, h, w)).astype(np.float32)).cuda().bfloat16()

    h_ref, c_ref = hseq.clone(), torch.empty((n, f, h, w), device='cuda')
    ops.convlstm_conv(x, wt, b, cd, h_ref, c_ref, z_add=z, c_prev=cp, x_channels=cin, compute_bf16=first)
    h_o8, c_o8 = to_o8(hseq), torch.empty((n, f, h, w), device='cuda')
    ops.convlstm_conv(x if first else to_o8(x), wt, b, cd, h_o8, c_o8, z_add=None if z is None else to_o8(z),
                      c_prev=None if cp is None else to_o8_f32(cp), x_channels=cin, compute_bf16=first, in_o8=not first,
                      out_o8=True)
    assert torch.equal(from_o8(h_o8), h_ref)          # the step's window AND the untouched other half of the h sequence
    assert torch.equal(from_o8(c_o8), c_ref)


def test_recurrent_model_forecasts_identically_with_and_without_octets(monkeypatch):
    """The config-4 stack (ConvLSTM2D front end + U-Net, bfloat16 between the layers) at a small grid: the hipGraph rollout with
    the octet layout (default) and with DLWP_BF16_O8=0 returns the same series, and the planner really chose octets."""
    from dlwp_amd.model import DLWPNeuralNet
    from tests.nets import lstm_unet_layers
    rng = np.random.default_rng(5)
    cs = (2, 6, 24, 40)
    x = rng.standard_normal((4,) + cs).astype(np.float32)
    outs = {}
    # (the whole-step ConvLSTM2D launch exists in octets only and rounds one tensor less: compared on its own below)
    monkeypatch.setenv('DLWP_LSTM_STEP', '0')
    for o8 in ('1', '0'):
        monkeypatch.setenv('DLWP_BF16_O8', o8)
        np.random.seed(3)
        d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
        d.build_model(lstm_unet_layers(cs), loss='mse', optimizer='adam')
        d.model.set_activation_dtype('bfloat16')
        ex = d.model.executor
        assert bool(ex._oct) == (o8 == '1')
        if o8 == '1':
            assert len(ex._oct) >= 5 and all(b in ex._bf16 for b in ex._oct)
        outs[o8] = (d.predict(x), d.predict_timeseries(x, 4))
    assert np.array_equal(outs['1'][0], outs['0'][0])
    assert np.array_equal(outs['1'][1], outs['0'][1])
    # the default plan: later ConvLSTM2D steps as one launch -- the same forecast up to the rounding of the pre-activation tensor
    # that is no longer stored
    monkeypatch.setenv('DLWP_LSTM_STEP', '1')
    monkeypatch.setenv('DLWP_BF16_O8', '1')
    np.random.seed(3)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(lstm_unet_layers(cs), loss='mse', optimizer='adam')
    d.model.set_activation_dtype('bfloat16')
    assert any(op.kind == 'conv' and op.src2 is not None for op in d.model.infer_plan.ops)
    one = d.predict(x)
    assert np.abs(one - outs['1'][0]).max() < 2e-2 * max(1.0, np.abs(one).max())
    assert np.array_equal(np.asarray(d.predict_timeseries(x, 2, keep_time_dim=True))[0].reshape(one.shape), one)   # graph == eager


def test_every_compiled_octet_instance_against_the_nchw_result():
    """Every registered octet instance (dlwp_conv2d_config_flags bits 3 / 4), forced: the NCHW instances' result up to the
    order of the float32 sums (instances differ in their channel chunk) -- one bfloat16 ulp on stored outputs."""
    from dlwp_amd import ops
    rng = np.random.default_rng(21)
    cfgs = ops.conv_configs()
    seen = 0
    problems = {}
    try:
        for i, (ks, dil, th, tw, waves, fa, bnf, ck, pool, lds, flags) in enumerate(cfgs):
            if not flags & 24 or flags & 2:
                continue
            in8, sw, in32 = bool(flags & 8), bool(flags & 16), pool == 3
            key = (ks, dil, in32)
            if key not in problems:
                ops.force_conv_config(-1)                                     # (the reference: the heuristic's NCHW instance)
                n, cin, h, w, cout = 2, (8 if in32 else 56), 19, 50, 40       # ragged tiles, ragged chunks for CK = 16 / 32;
                #                                                               8 float32 channels: the tap-packed instances too
                x = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32)).cuda()
                x = x if in32 else x.bfloat16()
                wt = torch.from_numpy(np_ref.glorot_uniform((ks, ks, cin, cout), rng)).cuda()
                b = torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32)).cuda()
                p = dil * (ks - 1) // 2
                cd = ops.make_conv(cout, ks, ks, dil, ops.make_pad(p, p, p, p, 0, 1), ops.ACT_TANH)
                want = ops.conv2d(x, wt, b, cd, out=torch.empty((n, cout, h, w), device='cuda'), compute_bf16=in32)
                problems[key] = (x, wt, b, cd, want)
            x, wt, b, cd, want = problems[key]
            ops.force_conv_config(i)
            out = torch.empty(want.shape, device='cuda', dtype=torch.bfloat16 if sw else torch.float32)
            got = ops.conv2d(to_o8(x) if in8 else x, wt, b, cd, out=out, compute_bf16=in32, in_o8=in8, out_o8=sw)
            got = from_o8(got).float() if sw else got
            tol = 8e-3 if sw else 2e-5                     # tanh outputs below 1: one bf16 ulp is 2^-8 = 3.9e-3
            assert (got - want).abs().max().item() <= tol, 'config %d %r' % (i, cfgs[i])
            seen += 1
    finally:
        ops.force_conv_config(-1)
    assert seen >= 8


@pytest.mark.parametrize('f,cx,h,w', [(24, 6, 20, 40), (16, 4, 18, 36), (16, 8, 9, 72)])
def test_whole_convlstm_step_in_one_launch_equals_the_two_launch_step(f, cx, h, w):
    """dlwp_convlstm_step_fwd (recurrent + input convolution + cell update in one launch) against the two launches it replaces
    -- the input convolution into a stored bfloat16 z, then dlwp_convlstm_conv_fwd with z_add -- up to that tensor's rounding;
    and against a float64 restatement of the cell (np_ref.conv_lstm2d arithmetic) with bf16-rounded inputs and kernels."""
    from dlwp_amd import ops
    rng = np.random.default_rng(100 + f)
    n = 3
    x = torch.from_numpy(rng.standard_normal((n, 2 * cx, h, w)).astype(np.float32)).cuda()
    hseq = torch.from_numpy((0.5 * rng.standard_normal((n, 2 * f, h, w))).astype(np.float32)).cuda().bfloat16()
    cp = torch.from_numpy(rng.standard_normal((n, f, h, w)).astype(np.float32)).cuda()
    w_h = torch.from_numpy(np_ref.glorot_uniform((3, 3, f, 4 * f), rng)).cuda()
    w_x = torch.from_numpy(np_ref.glorot_uniform((3, 3, cx, 4 * f), rng)).cuda()
    b = torch.from_numpy((0.1 * rng.standard_normal(4 * f)).astype(np.float32)).cuda()
    cd_h = ops.make_conv(4 * f, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 0), ops.ACT_TANH, in_c_off=0, in_c_total=2 * f, out_c_off=f,
                         out_c_total=2 * f, lstm_f=f, lstm_rec_act=0)
    cd_x = ops.make_conv(4 * f, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_LINEAR, in_c_off=cx, in_c_total=2 * cx)
    assert ops.convlstm_step_supported((f, h, w), cd_h, (cx, h, w), cd_x)
    # one launch
    h1, c1 = to_o8(hseq), torch.empty((n, f, h, w), device='cuda')
    ops.convlstm_step(h1, x, w_h, w_x, b, cd_h, cd_x, to_o8(cp), c1, cx)
    h1, c1 = from_o8(h1), from_o8(c1)
    # prepared weights: the same launch
    prep = ops.convlstm_step_prepare(to_o8(hseq), x, w_h, w_x, cd_h, cd_x, cx)
    h1p, c1p = to_o8(hseq), torch.empty((n, f, h, w), device='cuda')
    ops.convlstm_step(h1p, x, w_h, w_x, b, cd_h, cd_x, to_o8(cp), c1p, cx, prepared=prep)
    assert torch.equal(from_o8(h1p), h1) and torch.equal(from_o8(c1p), c1)
    # two launches (NCHW instances), z stored as bfloat16
    z = ops.conv2d(x, w_x, b, cd_x, out=torch.empty((n, 4 * f, h, w), device='cuda', dtype=torch.bfloat16), x_channels=cx,
                   compute_bf16=True)
    h2, c2 = hseq.clone(), torch.empty((n, f, h, w), device='cuda')
    ops.convlstm_conv(hseq, w_h, None, cd_h, h2, c2, z_add=z, c_prev=cp, x_channels=f)
    assert torch.equal(h1[:, :f], hseq[:, :f])                      # the step's window only
    assert (c1 - c2).abs().max().item() < 2e-2 and (h1.float() - h2.float()).abs().max().item() < 2e-2
    # float64 cell arithmetic on the rounded operands
    r = np_ref.round_bf16
    xt = np.pad(r(x[:, cx:].cpu().numpy()).astype(np.float64), ((0, 0), (0, 0), (2, 2), (0, 0)))
    xt = np.concatenate([xt[..., -2:], xt, xt[..., :2]], axis=-1)
    hp = np.pad(hseq[:, :f].float().cpu().numpy().astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    zz = np_ref.conv2d(xt, r(w_x.cpu().numpy()), b.cpu().numpy(), 2, 'linear') + \
        np_ref.conv2d(hp, r(w_h.cpu().numpy()), None, 1, 'linear')
    zi, zf, zc, zo = zz[:, :f], zz[:, f:2 * f], zz[:, 2 * f:3 * f], zz[:, 3 * f:]
    c_want = np_ref.hard_sigmoid(zf) * cp.cpu().numpy() + np_ref.hard_sigmoid(zi) * np.tanh(zc)
    h_want = np_ref.hard_sigmoid(zo) * np.tanh(c_want)
    assert np.abs(c1.cpu().numpy() - c_want).max() < 3e-5 * max(1.0, np.abs(c_want).max())
    assert np.abs(h1[:, f:].float().cpu().numpy() - h_want).max() < 4.1e-3
