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
                 'device': torch.cuda.current_device()}
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
