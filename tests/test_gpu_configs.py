"""Every BASELINE.json configuration AT ITS OWN SIZE on the GPU, through the reference API, against the oracle (one sample:
the oracle finishes in seconds) plus the size-independent properties (hipGraph rollout == eager loop bit for bit, member /
batch invariance).  The small-grid tests of test_gpu_model.py cover the arithmetic in breadth; these cover the tile
choices, grid sizes and buffer sizes the benchmarked configurations actually run with.

  cfg1  2.5-degree 73 x 144, Z500 x 2 input steps, 2 x (PeriodicPadding2D + ZeroPadding2D + Conv2D 5x5)  (examples/train.py)
  cfg2  88 x 180 x 4 U-Net: tests/test_gpu_model.py::test_full_size_unet_longitude_shift_equivariance_and_spot_parity
  cfg3  the same U-Net, batch 64 train step: gradient parity at full size here; the 2-rank run in test_gpu_parallel.py
  cfg4  1-degree 180 x 360, 6 variables x 2 steps, ConvLSTM2D front end + U-Net, float32 and bfloat16 storage
  cfg5  1-degree 180 x 360 x 12 U-Net, 4 members per GPU, 40-forward (80-step, 20-day) hipGraph rollout
"""
import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref
from tests.nets import cnn2_layers, lstm_unet_layers, unet_layers
from tests.test_gpu_model import FWD_TOL, _bf16_lstm_parts, _bf16_weight_indices, _build, _lstm_weights, _rel, _weights_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def test_cfg1_two_layer_cnn_at_73x144():
    rng = np.random.default_rng(101)
    cs = (2, 73, 144)
    layers = cnn2_layers(cs, hidden=32)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((3,) + cs).astype(np.float32)
    got = d.predict(x)
    want = np_ref.run_layers(layers, x[:1], weights)                  # float64 direct sums
    assert _rel(got[:1], want) < FWD_TOL
    want32 = torch_ref.run_layers(layers, torch.from_numpy(x), torch_ref.to_torch_weights(weights)).numpy()
    assert _rel(got, want32) < FWD_TOL
    # 4-forward rollout (2 days): graph == host loop, bit for bit
    kept = d.predict_timeseries(x, 8, keep_time_dim=True)
    p, ser = x, []
    for _ in range(4):
        p = d.predict(p)
        ser.append(p)
    assert np.array_equal(kept.reshape((4, 3) + cs), np.stack(ser))
    # one train step: loss and gradients against torch autograd in float64
    y = rng.standard_normal((3,) + cs).astype(np.float32)
    tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
    out = torch_ref.run_layers(layers, torch.tensor(x, dtype=torch.float64), tw)
    loss = ((out - torch.tensor(y, dtype=torch.float64)) ** 2).mean()
    loss.backward()
    vals = d.model.train_on_batch(x, y)
    assert vals[0] == pytest.approx(float(loss), rel=2e-5)
    tr = d.model._trainer
    off = 0
    for w, b in tw:
        for g_ref in (w.grad.numpy().transpose(2, 3, 1, 0), b.grad.numpy()):
            g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
            off += g_ref.size
            assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6)


def test_cfg3_train_step_gradients_at_88x180_batch_8():
    """The per-GPU share of config 3 (global batch 64 over 8 GPUs = 8 samples per rank) at full grid size."""
    rng = np.random.default_rng(103)
    cs = (4, 88, 180)
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    x = rng.standard_normal((8,) + cs).astype(np.float32)
    y = rng.standard_normal((8,) + cs).astype(np.float32)
    tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
    out = torch_ref.run_layers(layers, torch.tensor(x, dtype=torch.float64), tw)
    loss = ((out - torch.tensor(y, dtype=torch.float64)) ** 2).mean()
    loss.backward()
    vals = d.model.train_on_batch(x, y)
    assert vals[0] == pytest.approx(float(loss), rel=2e-5)
    tr = d.model._trainer
    off = 0
    for w, b in tw:
        for g_ref in (w.grad.numpy().transpose(2, 3, 1, 0), b.grad.numpy()):
            g = tr.flat_grads[off:off + g_ref.size].cpu().numpy().reshape(g_ref.shape)
            off += g_ref.size
            assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape


def test_cfg4_recurrent_stack_at_180x360_float32_and_bfloat16():
    from dlwp_amd.model import DLWPNeuralNet
    rng = np.random.default_rng(104)
    cs = (2, 6, 180, 360)                        # (input time steps, variables, lat, lon)
    layers = lstm_unet_layers(cs)
    np.random.seed(11)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(layers, loss='mse', optimizer='adam')
    assert d.model.count_params() == 234092
    weights = _lstm_weights(d.model, rng)
    x = rng.standard_normal((2,) + cs).astype(np.float32)
    got = d.predict(x)
    want32 = torch_ref.run_layers(layers, torch.from_numpy(x), torch_ref.to_torch_weights(weights)).numpy()
    assert _rel(got, want32) < FWD_TOL
    want = np_ref.run_layers(layers, x[:1], weights)
    assert _rel(got[:1], want) < FWD_TOL
    assert np.array_equal(d.predict(x[1:2]), got[1:2])                 # batch invariance at this size
    # bfloat16 storage between the layers (the mode BASELINE.json names for this config) against the float64 oracle run
    # with the same roundings.  A stored value that sits on a rounding boundary may land one bf16 ulp (2^-8 relative) apart
    # in fp32 and fp64 accumulation, and that difference travels through the following layers: over the 1.5 M outputs of
    # this size the worst element is a little over one ulp (the 16 x 24 tests stay under 4e-3); the mean error shows that
    # the same roundings are applied in the same places
    d.model.set_activation_dtype('bfloat16')
    on16 = _bf16_weight_indices(d.model, 1)
    parts = _bf16_lstm_parts(d.model, 1)
    got16 = d.predict(x[:1])
    fused = any(op.kind == 'conv' and op.lstm_f for op in d.model.infer_plan.ops)   # cell update in the convolutions' epilogues
    assert fused
    if any(op.kind == 'conv' and op.src2 is not None for op in d.model.infer_plan.ops):
        fused = 'step'              # ... and every later step as ONE launch (dlwp_convlstm_step_fwd): nothing stored in between
    want16 = np_ref.run_layers(layers, x[:1], weights, bf16_activations=True, bf16_weights=on16, bf16_lstm=parts,
                               lstm_fused=fused)
    assert _rel(got16, want16) < 1e-2
    assert float(np.abs(got16 - want16).mean()) < 1e-3 * max(1.0, float(np.abs(want16).max()))
    # and the rollout graph replays exactly that forward
    series = d.predict_timeseries(x[:1], 2, keep_time_dim=True)
    assert np.array_equal(np.asarray(series)[0].reshape(got16.shape), got16)


def test_cfg5_ensemble_rollout_at_180x360x12_four_members_forty_forwards():
    from dlwp_amd.parallel import shard_bounds
    rng = np.random.default_rng(105)
    cs = (12, 180, 360)
    layers = unet_layers(cs)
    d = _build(layers, time_dim=2)
    weights = _weights_of(d.model, rng)
    base = rng.standard_normal((1,) + cs).astype(np.float32)
    members = (base + 0.01 * rng.standard_normal((4,) + cs)).astype(np.float32)      # perturbed initial conditions
    net = d.model
    forwards = 40                                                      # 80 six-hour steps = 20 days
    x = torch.from_numpy(members).to(net.device)
    series = net.rollout_on_device(x, forwards).clone()                # (40, 4, 12, 180, 360) in HBM
    assert tuple(series.shape) == (forwards, 4) + cs
    assert bool(torch.isfinite(series).all())
    # (1) the one-launch hipGraph rollout == the eager loop over the same forward, bit for bit, at every step
    p = x
    for t in range(forwards):
        p = net.predict_on_device(p)
        assert torch.equal(p, series[t]), 'graph and eager forward differ at forward %d' % t
    # (2) members are independent: a one-member and a two-member rollout equal their rows of the 4-member rollout, so
    #     the 8-GPU sharding (4 members per GPU out of 32) cannot change any member's forecast
    for lo, hi in (shard_bounds(4, 0, 4), shard_bounds(4, 1, 2)):
        part = net.rollout_on_device(x[lo:hi].contiguous(), forwards)
        assert torch.equal(part, series[:, lo:hi])
    # (3) through the API the reference's validation scripts call: numpy in, numpy out, merged time axis
    ts = d.predict_timeseries(members, 2 * forwards)
    assert ts.shape == (2 * forwards, 4, 6, 180, 360) and ts.dtype == np.float32
    assert np.array_equal(ts, np_ref._merge_time(series.cpu().numpy(), forwards, 4, 2, cs, False))
    # (4) parity of the forward at this size: member 0's first forward against the float64 oracle and torch-CPU
    first = series[0, :1].cpu().numpy()
    want32 = torch_ref.run_layers(layers, torch.from_numpy(members[:1]), torch_ref.to_torch_weights(weights)).numpy()
    assert _rel(first, want32) < FWD_TOL
    want = np_ref.run_layers(layers, members[:1], weights)
    assert _rel(first, want) < FWD_TOL
