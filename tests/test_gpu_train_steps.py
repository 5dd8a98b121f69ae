"""The replay forms of the training step AT THE SIZES THE BENCHMARK AND fit_generator RUN THEM (VERDICT r4 'missing' 4 / weak a):
config 3's U-Net on the 88 x 180 grid, 12 and 64 samples, under every form of dlwp_train_step_launch -- 'graph' (one hipGraph),
'lanes' (launch by launch, weight gradients on the recorded side streams: what `train_cfg3` and fit_generator take above 10
samples), 'branches' (one hipGraph with the lanes as branches) and 'lanes' on the library's own streams -- against the eager step
(launch by launch from Python, the form the oracle tests of test_gpu_configs.py / test_gpu_model.py pin), and the first step's
gradients against torch autograd in float64.  And fit_generator over a shuffling DataGenerator through the DeviceLoader against
the same batches through train_on_batch.
Reference: keras Model.train_on_batch / fit_generator as DLWPNeuralNet drives them (DLWP/model/models.py:188-228,
examples/train.py:262-263)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import torch_ref
from tests.nets import unet_layers
from tests.test_gpu_model import _build, _weights_of

pytestmark = pytest.mark.gpu

CS = (4, 88, 180)
STEPS = 5                      # steps 1-2 eager (graph_after = 2), step 3 recorded (+ validated by a replay), steps 4-5 replayed


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _batches(n, seed):
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal((n,) + CS).astype(np.float32), rng.standard_normal((n,) + CS).astype(np.float32))
            for _ in range(STEPS)]


def _run(monkeypatch, batches, form, lanes=None):
    """STEPS train_on_batch calls on a fresh model under one step form; returns what the step leaves behind"""
    for k in ('DLWP_TRAIN_GRAPH', 'DLWP_TRAIN_STEP', 'DLWP_TRAIN_LANES'):
        monkeypatch.delenv(k, raising=False)
    if form == 'eager':
        monkeypatch.setenv('DLWP_TRAIN_GRAPH', '0')
    elif form != 'auto':
        monkeypatch.setenv('DLWP_TRAIN_STEP', form)
    if lanes:
        monkeypatch.setenv('DLWP_TRAIN_LANES', lanes)
    d = _build(unet_layers(CS), time_dim=2, seed=7)
    _weights_of(d.model, np.random.default_rng(70))
    tr = d.model._trainer
    logs, grads1 = [], None
    for i, (x, y) in enumerate(batches):
        logs.append(d.model.train_on_batch(x, y))
        if i == 0:
            torch.cuda.synchronize()
            grads1 = tr.flat_grads.cpu().numpy().copy()
    torch.cuda.synchronize()
    info = None
    if form != 'eager':
        n = batches[0][0].shape[0]
        ents = [e for k, e in tr._graphs.items() if k[0] == n and e.get('step') is not None]
        assert len(ents) == 1, 'the step of %d samples was not recorded (refused: %r)' % (n, tr._no_tape)
        from dlwp_amd import _lib
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib.dlwp_train_step_info(ents[0]['step'].h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        info = {'launches': a.value, 'lanes': b.value, 'waits': c.value, 'form': tr._graph_ok(n)}
    return {'weights': [w.copy() for w in d.model.get_weights()], 'logs': logs, 'grads1': grads1,
            'adam': [s.cpu().numpy().copy() for s in tr.opt_state], 'iterations': d.model.optimizer.iterations, 'info': info,
            'last_grads': tr.flat_grads.cpu().numpy().copy()}


def _assert_same(got, ref, what):
    assert got['iterations'] == ref['iterations'] == STEPS
    for a, b in zip(got['logs'], ref['logs']):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7), (what, a, b)
    for a, b in zip(got['weights'], ref['weights']):
        assert np.abs(a - b).max() <= 2e-7 * STEPS, what
    for a, b in zip(got['adam'], ref['adam']):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), what
    assert np.abs(got['last_grads'] - ref['last_grads']).max() <= 1e-5 * np.abs(ref['last_grads']).max(), what


@pytest.mark.parametrize('n', [12, 64])
def test_every_replay_form_of_the_config3_step_equals_the_eager_step(monkeypatch, n):
    batches = _batches(n, 300 + n)
    ref = _run(monkeypatch, batches, 'eager')
    assert ref['info'] is None
    for form, lanes in (('graph', None), ('lanes', None), ('branches', None), ('lanes', 'own'), ('auto', None)):
        got = _run(monkeypatch, batches, form, lanes)
        info = got['info']
        # the form that actually ran: what the trainer asks the library for at this size, on a step object with side lanes
        assert info['form'] == ('lanes' if form == 'auto' else form), info          # auto: above 10 samples of this grid -> 'lanes'
        assert info['launches'] >= 20 and info['lanes'] >= 2 and info['waits'] >= 2, info
        _assert_same(got, ref, (n, form, lanes))
        # (the first two steps of every run are eager: identical bits)
        assert np.array_equal(got['grads1'], ref['grads1'])
    if n == 12:
        # ... and the eager step's gradients are the oracle's: torch autograd on the unfused float64 graph (as test_gpu_configs.py at 8)
        d = _build(unet_layers(CS), time_dim=2, seed=7)
        weights = _weights_of(d.model, np.random.default_rng(70))
        x, y = batches[0]
        tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
        out = torch_ref.run_layers(unet_layers(CS), torch.tensor(x, dtype=torch.float64), tw)
        loss = ((out - torch.tensor(y, dtype=torch.float64)) ** 2).mean()
        loss.backward()
        assert ref['logs'][0][0] == pytest.approx(float(loss), rel=2e-5)
        off = 0
        for w, b in tw:
            for g_ref in (w.grad.numpy().transpose(2, 3, 1, 0), b.grad.numpy()):
                g = ref['grads1'][off:off + g_ref.size].reshape(g_ref.shape)
                off += g_ref.size
                assert np.abs(g - g_ref).max() <= 2e-4 * max(np.abs(g_ref).max(), 1e-6), g_ref.shape


def test_fit_generator_through_the_device_loader_equals_train_on_batch_on_the_same_batches(monkeypatch):
    """fit_generator(DataGenerator(shuffle=True)) -- DeviceLoader: host gather into pinned slots, copy stream, the 'lanes' step it
    picks at 12 samples of this grid -- leaves the weights that train_on_batch leaves on the SAME shuffled batches, eagerly."""
    from dlwp_amd.model import ArrayDataset, DataGenerator
    for k in ('DLWP_TRAIN_GRAPH', 'DLWP_TRAIN_STEP', 'DLWP_TRAIN_LANES'):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(404)
    n, bs = 72, 12
    P = rng.standard_normal((n, 2, 2, 88, 180)).astype(np.float32)
    T = rng.standard_normal((n, 2, 2, 88, 180)).astype(np.float32)

    def model():
        d = _build(unet_layers(CS), time_dim=2, seed=9)
        _weights_of(d.model, np.random.default_rng(90))
        return d
    d1 = model()
    np.random.seed(77)
    gen = DataGenerator(d1, ArrayDataset(P, T), batch_size=bs, shuffle=True)
    order = gen._indices.copy()
    assert not np.array_equal(order, np.arange(n))
    assert gen.batch_sources() is not None                      # plain row gathers: the native host-gather path
    d1.fit_generator(gen, epochs=1, verbose=0)
    tr1 = d1.model._trainer
    assert tr1._graph_ok(bs) == 'lanes'
    assert any(k[0] == bs and e.get('step') is not None for k, e in tr1._graphs.items())     # ... and it was replayed
    assert d1.model.optimizer.iterations == n // bs
    # the same batches, step by step, eagerly
    monkeypatch.setenv('DLWP_TRAIN_GRAPH', '0')
    d2 = model()
    for i in range(n // bs):
        rows = order[i * bs:(i + 1) * bs]
        d2.model.train_on_batch(P[rows].reshape((bs,) + CS), T[rows].reshape((bs,) + CS))
    torch.cuda.synchronize()
    for a, b in zip(d1.model.get_weights(), d2.model.get_weights()):
        assert np.abs(a - b).max() <= 2e-7 * (n // bs)
    assert d1.evaluate(P[:8].reshape((8,) + CS), T[:8].reshape((8,) + CS), verbose=0) == \
        pytest.approx(d2.evaluate(P[:8].reshape((8,) + CS), T[:8].reshape((8,) + CS), verbose=0), rel=1e-5)


def test_a_step_with_a_launch_the_tape_does_not_carry_is_refused_and_stays_eager(monkeypatch):
    """a replay must never silently miss device work.  An entry point without a tape record (here
    dlwp_series_merge_time, called from inside the step through a hooked _forward_backward) marks the tape foreign:
    dlwp_train_step_create refuses, the shape stays eager and keeps giving the eager results.  (dlwp_copy_many, r5's example,
    records itself since r6 -- ADVICE r5 -- and a step that copies replays: the second half of this test.)"""
    from dlwp_amd import ops
    monkeypatch.setenv('DLWP_TRAIN_GRAPH', '1')
    cs = (4, 16, 24)
    rng = np.random.default_rng(5)
    layers = unet_layers(cs, widths=(8, 16, 16, 16, 8))
    xs = [rng.standard_normal((6,) + cs).astype(np.float32) for _ in range(5)]
    ys = [rng.standard_normal((6,) + cs).astype(np.float32) for _ in range(5)]

    def run(hook):
        d = _build(layers, time_dim=2, seed=3)
        tr = d.model._trainer
        if hook:
            inner = tr._forward_backward
            scratch = [torch.ones((1, 2, 4, 8, 8), device='cuda'), torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')]

            def fb(x, y, scale):
                if hook == 'untaped':
                    ops.series_merge_time(scratch[0], 2)          # a launch through the library that the tape does not record
                else:
                    ops.copy_many([(scratch[1], scratch[2])])     # records itself (dlwp_tape_push): the step still replays
                return inner(x, y, scale)
            tr._forward_backward = fb
        logs = [d.model.train_on_batch(x, y) for x, y in zip(xs, ys)]
        return d, tr, logs
    d0, tr0, logs0 = run(False)
    d1, tr1, logs1 = run('untaped')
    d2, tr2, logs2 = run('copy_many')
    assert any(e.get('step') is not None for e in tr0._graphs.values())
    assert not tr1._graphs and len(tr1._no_tape) == 1                      # refused, remembered, eager from then on
    assert any(e.get('step') is not None for e in tr2._graphs.values()) and not tr2._no_tape
    for a, b in zip(logs0, logs2):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
    for a, b in zip(logs0, logs1):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
    for a, b in zip(d0.model.get_weights(), d1.model.get_weights()):
        assert np.abs(a - b).max() <= 2e-7 * len(xs)
