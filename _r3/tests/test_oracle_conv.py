"""Cross-checks inside the oracle for the UNPINNED arithmetic (Conv2D / pooling / Adam): float64 numpy direct sum
vs torch-CPU float32 ops vs torch autograd, plus the circular-convolution identity for the periodic axis."""
import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref

UNET = (
    ('PeriodicPadding2D', ((0, 2),), {'data_format': 'channels_first'}),
    ('ZeroPadding2D', ((2, 0),), {'data_format': 'channels_first'}),
    ('Conv2D', (8, 3), {'dilation_rate': 2, 'padding': 'valid', 'activation': 'tanh', 'data_format': 'channels_first'}),
    ('MaxPooling2D', (2,), {'data_format': 'channels_first'}),
    ('PeriodicPadding2D', ((0, 1),), {'data_format': 'channels_first'}),
    ('ZeroPadding2D', ((1, 0),), {'data_format': 'channels_first'}),
    ('Conv2D', (16, 3), {'dilation_rate': 1, 'padding': 'valid', 'activation': 'tanh', 'data_format': 'channels_first'}),
    ('UpSampling2D', (2,), {'data_format': 'channels_first'}),
    ('PeriodicPadding2D', ((0, 2),), {'data_format': 'channels_first'}),
    ('ZeroPadding2D', ((2, 0),), {'data_format': 'channels_first'}),
    ('Conv2D', (4, 5), {'padding': 'valid', 'activation': 'linear', 'data_format': 'channels_first'}),
)


@pytest.mark.parametrize('k,d', [(3, 1), (3, 2), (5, 1)])
def test_conv2d_numpy_vs_torch(k, d):
    rng = np.random.default_rng(k * 10 + d)
    x = rng.standard_normal((2, 5, 12, 14)).astype(np.float32)
    w = np_ref.glorot_uniform((k, k, 5, 7), rng)
    b = rng.standard_normal(7).astype(np.float32)
    y = np_ref.conv2d(x, w, b, d, 'tanh')
    (wt, bt), = torch_ref.to_torch_weights([(w, b)], dtype=torch.float64)
    yt = torch.tanh(torch.nn.functional.conv2d(torch.tensor(x, dtype=torch.float64), wt, bt, dilation=d)).numpy()
    assert y.shape == yt.shape
    assert np.abs(y - yt).max() < 1e-13


def test_conv2d_grads_vs_autograd():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 9, 10))
    w = rng.standard_normal((3, 3, 3, 4))
    dz = rng.standard_normal((2, 4, 5, 6))
    dx, dw, db = np_ref.conv2d_grads(x, w, dz, dilation=2)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(np.transpose(w, (3, 2, 0, 1)).copy(), requires_grad=True)
    bt = torch.zeros(4, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xt, wt, bt, dilation=2)
    y.backward(torch.tensor(dz))
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-12
    assert np.abs(dw - wt.grad.numpy().transpose(2, 3, 1, 0)).max() < 1e-12
    assert np.abs(db - bt.grad.numpy()).max() < 1e-12


def test_pool_upsample_and_grads():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 3, 7, 10))
    xt = torch.tensor(x, requires_grad=True)
    yt = torch.nn.functional.max_pool2d(xt, 2)
    assert np.array_equal(np_ref.maxpool2(x), yt.detach().numpy())
    dy = rng.standard_normal(yt.shape)
    yt.backward(torch.tensor(dy))
    assert np.allclose(np_ref.maxpool2_grad(x, dy), xt.grad.numpy())
    u = np_ref.upsample2(x)
    ut = torch.nn.functional.interpolate(torch.tensor(x), scale_factor=2, mode='nearest').numpy()
    assert np.array_equal(u, ut)
    du = rng.standard_normal(u.shape)
    assert np.isclose((np_ref.upsample2_grad(du) * x).sum(), (du * u).sum())


def test_periodic_conv_is_circular_convolution():
    """periodic-pad o valid-conv along longitude == circular cross-correlation (FFT identity)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 1, 1, 24))
    w = rng.standard_normal((1, 5, 1, 1))
    y = np_ref.conv2d(np_ref.periodic_padding2d(x, (0, 2)), w)[0, 0, 0]
    k = np.zeros(24)
    for v in range(5):
        k[(v - 2) % 24] = w[0, v, 0, 0]
    want = np.real(np.fft.ifft(np.fft.fft(x[0, 0, 0]) * np.conj(np.fft.fft(k))))
    assert np.abs(y - want).max() < 1e-12


def test_layer_stack_numpy_vs_torch_and_shift_equivariance():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 4, 8, 12)).astype(np.float32)
    weights = np_ref.init_weights(UNET, 4, rng)
    weights = [(w, rng.standard_normal(b.shape).astype(np.float32) * 0.1) for w, b in weights]
    y64 = np_ref.run_layers(UNET, x, weights)
    y32 = torch_ref.run_layers(UNET, torch.from_numpy(x), torch_ref.to_torch_weights(weights)).numpy()
    assert y64.shape == (2, 4, 8, 12)
    assert np.abs(y64 - y32).max() < 2e-5
    # the whole periodic stack commutes with a longitude shift by a multiple of the pooling factor
    ys = np_ref.run_layers(UNET, np.roll(x, 4, axis=-1), weights)
    assert np.abs(ys - np.roll(y64, 4, axis=-1)).max() < 1e-12


def test_adam_keras_form():
    rng = np.random.default_rng(7)
    p = rng.standard_normal(50)
    m = np.zeros(50)
    v = np.zeros(50)
    pt, mt, vt = (torch.tensor(a, dtype=torch.float32) for a in (p, m, v))
    for it in range(5):
        g = rng.standard_normal(50)
        p, m, v = np_ref.adam_keras_step(p, m, v, g, it)
        torch_ref.adam_keras_step(pt, mt, vt, torch.tensor(g, dtype=torch.float32), it)
    assert np.abs(p - pt.numpy()).max() < 1e-6
    # first step of Adam moves every weight by ~lr regardless of gradient scale
    p1, _, _ = np_ref.adam_keras_step(np.zeros(3), np.zeros(3), np.zeros(3), np.array([1e-3, 1., 1e3]), 0)
    assert np.allclose(p1, -1e-3, rtol=5e-3)      # eps=1e-7 shows at |g|=1e-3


def test_conv_lstm2d_numpy_and_torch_restatements_agree():
    """ConvLSTM2D parity is unpinned (Keras is absent): the two independent restatements -- numpy direct sums and
    torch-CPU F.conv2d -- must at least agree with each other, through the reference's recurrent front end
    (examples/train.py:144-157)."""
    import torch
    from oracle import torch_ref
    from tests.nets import lstm_unet_layers
    rng = np.random.default_rng(5)
    cs = (3, 2, 8, 12)
    layers = lstm_unet_layers(cs, widths=(8, 8, 8, 8, 8))[:4]          # pads + ConvLSTM2D + Reshape
    (k, r, b), = np_ref.init_weights(layers, cs[1], rng)
    assert k.shape == (3, 3, 2, 32) and r.shape == (3, 3, 8, 32) and np.all(b[8:16] == 1) and b.sum() == 8
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.standard_normal((2,) + cs)
    a = np_ref.run_layers(layers, x, [(k, r, b)])
    t = torch_ref.run_layers(layers, torch.tensor(x), torch_ref.to_torch_weights([(k, r, b)], dtype=torch.float64)).numpy()
    assert a.shape == (2, 3 * 8, 8, 12) and np.abs(a - t).max() < 1e-12
    # first step by hand: h_{-1} = c_{-1} = 0  ->  c_0 = hs(z_i) tanh(z_c), h_0 = hs(z_o) tanh(c_0)
    xp = np_ref.zero_padding3d(np_ref.periodic_padding3d(x, (0, 0, 2)), (0, 2, 0))
    z = np_ref.conv2d(xp[:, 0], k, b, 2, 'linear')
    c0 = np_ref.hard_sigmoid(z[:, :8]) * np.tanh(z[:, 16:24])
    assert np.abs(a[:, :8] - np_ref.hard_sigmoid(z[:, 24:]) * np.tanh(c0)).max() < 1e-12


def test_conv_on_upsampled_tensor_equals_its_restatement_on_the_source():
    """np_ref.phase_weights / depth_to_space2 (what the product's inference plan does with the decoder layers) against the
    definition: conv(pad(upsample2(x)), w) for 5x5 / 3x3 / 7x7 kernels, symmetric and asymmetric halos, every halo mode;
    and the dilation-2 identity conv_d2(pad_2p(upsample2(x))) == upsample2(conv_d1(pad_p(x)))."""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 5, 6, 8))
    for k, pads in ((5, (2, 2, 2, 2)), (3, (1, 1, 1, 1)), (7, (3, 3, 3, 3)), (5, (1, 3, 2, 2)), (4, (1, 2, 2, 1))):
        for mh, mw in ((0, 1), (1, 1), (2, 0), (0, 2)):
            w = rng.standard_normal((k, k, 5, 3))
            b = rng.standard_normal(3)
            want = np_ref.conv2d(np_ref.pad2d_modes(np_ref.upsample2(x), pads, mh, mw), w, b, 1, 'tanh')
            if want.shape[2:] != (12, 16):
                continue                                   # only 'same' geometries are restated
            w2, b2, (lo_h, hi_h, lo_w, hi_w) = np_ref.phase_weights(w, b, pads[0], pads[2])
            y = np_ref.conv2d(np_ref.pad2d_modes(x, (-lo_h, hi_h, -lo_w, hi_w), mh, mw), w2, b2, 1, 'tanh')
            got = np_ref.depth_to_space2(y, 3)
            assert np.abs(got - want).max() < 1e-12, (k, pads, mh, mw)
    w = rng.standard_normal((3, 3, 5, 4))
    for mh, mw in ((0, 1), (1, 1), (2, 2)):
        want = np_ref.conv2d(np_ref.pad2d_modes(np_ref.upsample2(x), (2, 2, 2, 2), mh, mw), w, None, 2, 'tanh')
        got = np_ref.upsample2(np_ref.conv2d(np_ref.pad2d_modes(x, (1, 1, 1, 1), mh, mw), w, None, 1, 'tanh'))
        assert np.abs(got - want).max() < 1e-12
