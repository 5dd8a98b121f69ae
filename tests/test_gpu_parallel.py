"""The data-parallel PRODUCT path on hardware: two ranks share the one GPU of the test box (DLWP_SHARE_GPUS=1, 'gloo'
collectives on device tensors -- RCCL refuses two ranks on one device) and run the real Trainer through
build_model(gpus=2): replica alignment at compile, shard-aware feeding (each rank gathers / uploads only its rows), the
single flat-buffer all-reduce with the loss table in its tail, ragged shards.  The result must equal the single-process
step on the whole batch (reference semantics: keras.utils.multi_gpu_model splits one batch inside one process,
DLWP/model/models.py:104-109).  Plus the library's own RCCL communicator (dlwp_comm_*) at world size 1."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.nets import unet_layers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


CS = (4, 16, 24)


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n,) + CS).astype(np.float32)
    y = (0.5 * x + 0.25 * np.roll(x, 1, axis=-1) + 0.1 * rng.standard_normal((n,) + CS)).astype(np.float32)
    return x, y


def _series(n=23):
    rng = np.random.default_rng(5)
    P = rng.standard_normal((n, 2, 2, 16, 24)).astype(np.float32)
    T = (0.5 * P + 0.25 * np.roll(P, 1, axis=-1)).astype(np.float32)
    return P, T


def _scenario(d, mode, n_global, shuffle_seed=77):
    """The same training calls on one process or on every rank of a group; returns the reported values.  shuffle_seed
    seeds THIS process' numpy stream right before anything shuffles: rank 0 and the single process share it, the other
    ranks get another one and must still cut the same batches (index broadcast)."""
    from dlwp_amd.model import ArrayDataset, DataGenerator
    logs = []
    if mode in ('batch', 'batch6'):
        x, y = _data(n_global)
        for _ in range(3 if mode == 'batch' else 6):
            logs.append(d.model.train_on_batch(x, y))
    elif mode == 'fit':
        x, y = _data(n_global)
        np.random.seed(shuffle_seed)
        h = d.fit(x, y, batch_size=8, epochs=2, verbose=0, shuffle=True)
        logs.append([h.history['loss'][-1], h.history['mean_absolute_error'][-1]])
    elif mode == 'generator':
        P, T = _series(n_global)
        np.random.seed(shuffle_seed)
        gen = DataGenerator(d, ArrayDataset(P, T), batch_size=8, shuffle=True)
        h = d.fit_generator(gen, epochs=2, verbose=0)
        logs.append([h.history['loss'][-1], h.history['mean_absolute_error'][-1]])
    return logs


def _worker(rank, world, port, mode, n_global, ret, extra_env=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DLWP_SHARE_GPUS='1', DLWP_DIST_BACKEND='gloo')
    os.environ.update(extra_env or {})
    from dlwp_amd import parallel
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.training import Adam
    parallel.init()
    np.random.seed(1000 + 17 * rank)            # DIFFERENT initial weights and shuffle streams per rank
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers(CS, widths=(8, 16, 16, 16, 8)), loss='mse', optimizer=Adam(lr=1e-3), metrics=['mae'],
                  gpus=world)
    w0 = [w.copy() for w in d.model.get_weights()]       # after compile: must already be rank 0's on every rank
    uploaded = []
    tr = d.model._trainer
    orig = tr._to_device

    def spy(a):
        t = orig(a)
        uploaded.append(int(t.shape[0]))
        return t
    tr._to_device = spy
    logs = _scenario(d, mode, n_global, 77 if rank == 0 else 4242 + rank)
    torch.cuda.synchronize()
    ret[rank] = {'w0': w0, 'w1': d.model.get_weights(), 'logs': logs, 'iters': d.model.optimizer.iterations,
                 'max_rows': max(uploaded) if uploaded else 0, 'graphs': len(tr._graphs), 'rccl_abi': tr.dp.uses_rccl_abi(),
                 'device': torch.cuda.current_device(), 'oneshot': bool(getattr(tr.dp, '_xchg', None)),
                 'oneshot_timed_out': tr.dp.oneshot_timed_out()}
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run_group(mode, n_global, world=2, extra_env=None):
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, mode, n_global, ret, extra_env)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            if p.is_alive():
                p.kill()
                pytest.fail('data-parallel worker timed out')
            assert p.exitcode == 0, 'data-parallel worker failed'
        return dict(ret)


def _check_against_single_process(res, mode, n_global):
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.training import Adam
    # replicas were aligned on rank 0's initial weights at compile, and stay identical
    for a, b in zip(res[0]['w0'], res[1]['w0']):
        assert np.array_equal(a, b)
    for a, b in zip(res[0]['w1'], res[1]['w1']):
        assert np.array_equal(a, b)
    assert res[0]['logs'] == res[1]['logs']
    # each rank uploaded only its shard: never more than ceil(batch / 2) rows at a time
    per_batch = n_global if mode == 'batch' else 8
    assert 0 < res[0]['max_rows'] <= -(-per_batch // 2) and res[1]['max_rows'] <= -(-per_batch // 2)
    # the single-process run from the same initial weights
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers(CS, widths=(8, 16, 16, 16, 8)), loss='mse', optimizer=Adam(lr=1e-3), metrics=['mae'])
    d.model.set_weights(res[0]['w0'])
    logs = _scenario(d, mode, n_global)
    assert d.model.optimizer.iterations == res[0]['iters']
    steps = res[0]['iters']
    for got, want in zip(res[0]['logs'], logs):
        assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (got, want)
    for a, b in zip(res[0]['w1'], d.model.get_weights()):
        # identical mathematics, different summation split (two half-batch gradients vs one): fp32 round-off per step
        assert np.abs(a - b).max() <= 1e-6 * steps, np.abs(a - b).max()


@pytest.mark.parametrize('mode,n_global', [('batch', 8), ('batch', 7), ('fit', 23), ('generator', 23)])
def test_two_rank_product_training_equals_the_single_process_run(mode, n_global):
    _check_against_single_process(_run_group(mode, n_global), mode, n_global)


def test_two_rank_training_with_the_captured_step_equals_the_single_process_run():
    """DLWP_TRAIN_GRAPH=1 under data parallelism: forward + loss + backward of a rank's shard replay as one hipGraph, the
    all-reduce and the optimizer launch stay OUTSIDE the capture (Trainer._capture_step) on the same stream -- no collective is
    ever captured -- and the result is still the single-process (eager) one."""
    res = _run_group('batch6', 8, extra_env={'DLWP_TRAIN_GRAPH': '1'})
    assert res[0]['graphs'] == 1 and res[1]['graphs'] == 1          # the step really was captured on both ranks
    _check_against_single_process(res, 'batch6', 8)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank (this box has one)')
@pytest.mark.parametrize('mode,n_global', [('batch', 8), ('batch', 7), ('fit', 23)])
def test_two_gpu_rccl_training_equals_the_single_process_run(mode, n_global):
    """The transport the product uses on a multi-GPU node: backend 'nccl', one GPU per rank, gradients summed by
    dlwp_allreduce_sum_f32 and replicas aligned by dlwp_broadcast_f32 (csrc/comm.hip).  Skipped on the one-GPU test box."""
    res = _run_group(mode, n_global, extra_env={'DLWP_SHARE_GPUS': '0', 'DLWP_DIST_BACKEND': 'nccl'})
    assert res[0]['rccl_abi'] and res[1]['rccl_abi'] and res[0]['device'] != res[1]['device']
    _check_against_single_process(res, mode, n_global)


def test_bench_gpus_2_as_a_plain_script_prints_one_line_with_the_collective_sub_records():
    """`python bench.py --gpus 2` the way the driver starts it (no torchrun environment): bench.py spawns its ranks, here two
    on the one GPU over gloo, and rank 0 prints ONE JSON line whose sub-records include the data-parallel training step
    (train_cfg3) and the sharded ensemble (ensemble_cfg5)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(DLWP_SHARE_GPUS='1', DLWP_DIST_BACKEND='gloo')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--members', '16', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['finite'] and rec['value'] > 0
    assert rec['config']['members_total'] == 32 and rec['scaling'] == 'weak'
    sub = rec['sub_records']
    assert 'error_collective' not in sub and 'error_local' not in sub, sub
    tr, ens = sub['train_cfg3'], sub['ensemble_cfg5']
    assert tr['global_batch'] == 64 and tr['batch_per_gpu'] == 32 and tr['value'] > 0 and np.isfinite(tr['loss'])
    assert ens['total_members'] == 32 and ens['members_per_gpu'] == 16 and ens['finite'] and 0 < ens['frac'] < 1
    assert 0 < sub['members_1']['frac'] < sub['members_8']['frac'] < 1
    # r5: both transports of the step's exchange were run on the same buffer -- torch.distributed / RCCL and the
    # library's one-shot all-reduce (two processes on the one GPU here) --, the sums are exact and equal, both latencies are there,
    # the one-shot region sits in uncached (or fine-grained) memory, no launch timed out; and every rank reports its stream probes
    ex = tr['exchange_check']
    assert 'error' not in ex, ex
    assert ex['default_sum_exact'] and ex['oneshot_sum_exact'] and ex['equal_sums'] and not ex['oneshot_timed_out'], ex
    assert ex['oneshot_ms'] > 0 and ex['default_ms'] > 0 and 'gloo' in ex['default_transport'] and ex['oneshot_region']['memory'] in ('uncached', 'fine-grained', 'plain')
    assert ex['oneshot_region']['blocks'] >= 1
    probes = rec['stream_probes']
    assert isinstance(probes, list) and len(probes) == 2 and all(isinstance(p_, list) for p_ in probes), probes
    assert any('rejected' in e for p_ in probes for e in p_), probes
    # the same total under --scaling strong
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--members', '16', '--scaling', 'strong', '--no-cpu-baseline', '--no-extras'], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert rec['scaling'] == 'strong' and rec['config']['members_total'] == 16 and rec['config']['members_per_gpu'] == 8


def test_rccl_communicator_of_the_c_abi_world_one():
    """dlwp_comm_* bind RCCL (the instance torch already loaded) and run the collectives on a stream.  A one-rank
    communicator is all a single-GPU box allows; sum and broadcast over one rank are the identity."""
    from dlwp_amd import _lib
    lib = _lib.lib
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.dlwp_comm_unique_id(None, ctypes.byref(nbytes)))
    assert nbytes.value == 128
    uid = (ctypes.c_char * nbytes.value)()
    _lib.check(lib.dlwp_comm_unique_id(uid, ctypes.byref(nbytes)))
    comm = ctypes.c_void_p()
    _lib.check(lib.dlwp_comm_init_rank(ctypes.byref(comm), 0, 1, 0, uid, nbytes.value))
    try:
        w, r, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.dlwp_comm_info(comm, ctypes.byref(w), ctypes.byref(r), ctypes.byref(v)))
        assert (w.value, r.value) == (1, 0) and v.value >= 20000
        g = torch.randn(188996 + 7, device='cuda')
        want = g.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            st = ctypes.c_void_p(s.cuda_stream)
            _lib.check(lib.dlwp_allreduce_sum_f32(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), st))
            _lib.check(lib.dlwp_broadcast_f32(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), 0, st))
        s.synchronize()
        assert torch.equal(g, want)
        assert lib.dlwp_broadcast_f32(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), 3, None) == _lib.EINVAL
    finally:
        _lib.check(lib.dlwp_comm_destroy(comm))


@pytest.mark.parametrize('mode,n_global', [('batch6', 8), ('batch', 7), ('generator', 23)])
def test_two_rank_training_through_the_one_shot_exchange_equals_the_single_process_run(mode, n_global):
    """DLWP_ALLREDUCE=oneshot (csrc/xchg.hip): the step's exchange through the library's own one-shot
    all-reduce -- each rank publishes its flat gradient buffer in memory the peer has mapped through hipIpcMemHandle (here: two
    processes of the one GPU), raises a flag, reads both buffers, sums them in rank order and applies the Keras-form Adam update in
    the same kernel.  The replicas stay bit-identical (same order of the sum on every rank) and follow the single-process run to
    float32 round-off; ragged shards (7 = 4 + 3, batches of 23) included; no wait timed out."""
    res = _run_group(mode, n_global, extra_env={'DLWP_ALLREDUCE': 'oneshot'})
    assert res[0]['oneshot'] and res[1]['oneshot'], 'the one-shot exchange was not taken'
    assert not res[0]['oneshot_timed_out'] and not res[1]['oneshot_timed_out']
    _check_against_single_process(res, mode, n_global)


def _xchg_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DLWP_SHARE_GPUS='1', DLWP_DIST_BACKEND='gloo', DLWP_ALLREDUCE='oneshot')
    from dlwp_amd import parallel
    parallel.init()
    dp = parallel.DataParallel()
    n = 189024                                            # the config-2 U-Net's exchange: 188 996 parameters + 7 loss values, padded
    outs = []
    for step in range(5):                                  # both payload parities, flags that keep counting
        g = torch.Generator().manual_seed(100 * step + rank)
        flat = torch.randn(n, generator=g).cuda()
        dp.oneshot_all_reduce_(flat)
        outs.append(flat.cpu().numpy())
    torch.cuda.synchronize()
    info = dp.oneshot_info()
    timed_out = dp.oneshot_timed_out()
    # r5: a buffer far above what r4's one-workgroup-per-1024-floats grid could keep resident (~2 M floats: every launch
    # timed out beyond that) -- the persistent grid walks it; a new size means a new region (created collectively)
    big_n = 6 * 1024 * 1024
    big_ok = []
    for step in range(3):
        flat = (torch.arange(big_n, dtype=torch.float32) % 509.0 + float(10 * step + rank)).cuda()
        dp.oneshot_all_reduce_(flat)
        want = (torch.arange(big_n, dtype=torch.float32) % 509.0) * 2 + float(20 * step + 1)
        big_ok.append(bool(torch.equal(flat.cpu(), want)))
    ret[rank] = {'outs': outs, 'timed_out': timed_out, 'info': info, 'big_ok': big_ok, 'big_timed_out': dp.oneshot_timed_out(),
                 'big_info': dp.oneshot_info()}
    torch.distributed.barrier()
    dp.close()
    torch.distributed.destroy_process_group()


def test_one_shot_all_reduce_sums_in_rank_order_on_every_rank():
    """dlwp_xchg_allreduce_sum_f32 alone, five exchanges of the config-2 buffer between two processes: every rank ends with
    rank 0's values + rank 1's, the same bits on both"""
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_xchg_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            if p.is_alive():
                p.kill()
                pytest.fail('one-shot exchange worker timed out')
            assert p.exitcode == 0
        res = dict(ret)
    assert not res[0]['timed_out'] and not res[1]['timed_out']
    for r in range(2):
        assert res[r]['info']['memory'] in ('uncached', 'fine-grained', 'plain') and 1 <= res[r]['info']['blocks'] <= 256, res[r]['info']
        assert res[r]['big_ok'] == [True, True, True] and not res[r]['big_timed_out'], (res[r]['big_ok'], res[r]['big_timed_out'])
        assert res[r]['big_info']['blocks'] >= 128          # one workgroup per CU, whatever the size
    print('one-shot exchange region memory:', res[0]['info'], res[0]['big_info'])
    for step in range(5):
        want = (torch.randn(189024, generator=torch.Generator().manual_seed(100 * step)) +
                torch.randn(189024, generator=torch.Generator().manual_seed(100 * step + 1))).numpy()
        assert np.array_equal(res[0]['outs'][step], res[1]['outs'][step])
        assert np.array_equal(res[0]['outs'][step], want)


def _xchg_timeout_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DLWP_SHARE_GPUS='1', DLWP_DIST_BACKEND='gloo', DLWP_ALLREDUCE='oneshot')
    import time
    from dlwp_amd import parallel
    parallel.init()
    dp = parallel.DataParallel()
    n, n_par = 4096, 4000
    flat = torch.ones(n).cuda()
    dp.xchg(flat)                                  # (collective: both ranks create and connect their regions)
    if rank == 0:                                  # ... and then only rank 0 shows up for the step
        p, m, v = torch.full((n_par,), 3.0).cuda(), torch.zeros(n_par).cuda(), torch.zeros(n_par).cuda()
        t0 = time.time()
        dp.oneshot_adam_(flat, n_par, p, m, v, 1e-3, 0.9, 0.999, 1e-7, 0.0, 0, 0.5)
        torch.cuda.synchronize()
        took = time.time() - t0
        raised = False
        try:
            dp.oneshot_check()
        except RuntimeError:
            raised = True
        ret[0] = {'took': took, 'timed_out': dp.oneshot_timed_out(), 'raised': raised,
                  'p_untouched': bool((p == 3.0).all()), 'm_untouched': bool((m == 0).all()),
                  'tail_nan': bool(torch.isnan(flat[n_par:]).all()), 'grads_kept': bool((flat[:n_par] == 1.0).all())}
    else:
        time.sleep(0.5)
    torch.distributed.barrier()
    dp.close()
    torch.distributed.destroy_process_group()


def test_one_shot_exchange_that_waits_in_vain_gives_up_leaves_the_parameters_alone_and_is_reported():
    """a peer that never arrives.  The launch waits 2 s (bounded: never a hung GPU), then leaves p / m / v untouched,
    writes NaN over the loss table behind the parameters, and dlwp_xchg_status / DataParallel.oneshot_check report it -- the
    trainer calls oneshot_check wherever it reads a loss, so training never continues on a half-applied step."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_xchg_timeout_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            if p.is_alive():
                p.kill()
                pytest.fail('worker hung')
            assert p.exitcode == 0
        res = dict(ret)[0]
    assert 1.5 < res['took'] < 10.0, res
    assert res['timed_out'] and res['raised'] and res['p_untouched'] and res['m_untouched'] and res['tail_nan'] and res['grads_kept'], res


@pytest.mark.parametrize('mode,n_global,functional', [('batch', 7, False), ('fit', 23, False), ('generator', 23, False),
                                                      ('batch', 8, True)])
def test_a_plain_script_that_asks_for_two_gpus_trains_like_the_single_process_run(tmp_path, mode, n_global, functional):
    """VERDICT r5 missing 2 / SURVEY 8b: build_model(..., gpus=2) from a plain `python script.py` -- the reference's
    keras.utils.multi_gpu_model is a single-process call (DLWP/model/models.py:104-109) -- starts the second rank itself
    (dlwp_amd/worker.py), mirrors the training calls to it, shards predict_timeseries over the ranks and gathers on rank 0.  The
    result must equal the single-process run from the same initial weights."""
    import subprocess
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.training import Adam
    out = str(tmp_path / 'driver.npz')
    env = dict(os.environ, DLWP_SHARE_GPUS='1', DLWP_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'scripts', 'driver_mode.py'), mode, str(n_global), out] +
                       (['functional'] if functional else []), env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = np.load(out)
    w0 = [got['w0_%d' % i] for i in range(int(got['n_w']))]
    w1 = [got['w1_%d' % i] for i in range(int(got['n_w']))]
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers(CS, widths=(8, 16, 16, 16, 8)), loss='mse', optimizer=Adam(lr=1e-3), metrics=['mae'])
    d.model.set_weights(w0)
    logs = _scenario(d, mode, n_global)
    steps = int(got['iters']) - 1                       # (the script took one more step after its scenario)
    assert d.model.optimizer.iterations == steps
    for a, b in zip(got['logs'], np.asarray(logs, dtype=np.float64)):
        assert np.allclose(a, b, rtol=2e-5, atol=1e-6), (a, b)
    for a, b in zip(w1, d.model.get_weights()):
        assert np.abs(a - b).max() <= 1e-6 * steps, np.abs(a - b).max()
    # the sharded forecast: members are independent, so the joined series is the one-process series bit for bit
    d.model.set_weights(w1)
    x, _ = _data(5, seed=9)
    assert np.array_equal(got['series'], d.predict_timeseries(x, 4))
    d2 = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d2.build_model(unet_layers(CS, widths=(8, 16, 16, 16, 8)), loss='mse', optimizer=Adam(lr=1e-3), metrics=['mae'])
    d2.model.set_weights(w0)
    assert np.allclose(got['again'], d2.model.train_on_batch(*_data(8)), rtol=2e-5, atol=1e-6)
