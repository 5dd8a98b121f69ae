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
    if mode == 'batch':
        x, y = _data(n_global)
        for _ in range(3):
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


def _worker(rank, world, port, mode, n_global, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DLWP_SHARE_GPUS='1', DLWP_DIST_BACKEND='gloo')
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
                 'max_rows': max(uploaded) if uploaded else 0}
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run_group(mode, n_global, world=2):
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, mode, n_global, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            if p.is_alive():
                p.kill()
                pytest.fail('data-parallel worker timed out')
            assert p.exitcode == 0, 'data-parallel worker failed'
        return dict(ret)


@pytest.mark.parametrize('mode,n_global', [('batch', 8), ('batch', 7), ('fit', 23), ('generator', 23)])
def test_two_rank_product_training_equals_the_single_process_run(mode, n_global):
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.training import Adam
    res = _run_group(mode, n_global)
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
