"""The N > 1 path on CPU: world_size-2 `gloo` process groups exercise dlwp_amd.parallel (row sharding, the single flat
gradient all-reduce, loss averaging, parameter broadcast) with exactly the weighting the Trainer uses, and check that a
data-parallel step on shards equals the single-process step on the whole batch (oracle arithmetic on the CPU, since the
HIP kernels need a GPU).  Also the host half of the loader."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_global, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dlwp_amd import parallel
    from oracle import np_ref, torch_ref
    from tests.nets import cnn2_layers
    r, w, _ = parallel.init(backend='gloo')
    assert (r, w) == (rank, world)
    dp = parallel.DataParallel()
    torch.set_num_threads(1)

    # every rank builds the same global batch (same seed), as Trainer.train_on_batch expects
    rng = np.random.default_rng(0)
    cs = (2, 6, 8)
    layers = cnn2_layers(cs, hidden=4)
    weights = np_ref.init_weights(layers, 2, np.random.RandomState(0))
    x = rng.standard_normal((n_global,) + cs)
    y = rng.standard_normal((n_global,) + cs)

    def grads_of(xb, yb, weight):
        tw = torch_ref.to_torch_weights(weights, dtype=torch.float64, requires_grad=True)
        out = torch_ref.run_layers(layers, torch.tensor(xb), tw)
        loss = ((out - torch.tensor(yb)) ** 2).mean()
        (loss * weight).backward()
        flat = torch.cat([p.grad.reshape(-1) for wb in tw for p in wb])
        return flat, float(loss.detach())

    lo, hi = dp.shard(n_global)
    scale = (hi - lo) * dp.world / float(n_global)           # Trainer.train_on_batch's ragged-shard weighting
    flat, local_loss = grads_of(x[lo:hi], y[lo:hi], scale)
    dp.all_reduce_sum_(flat)
    flat = flat / dp.world                                   # the grad_scale handed to the Adam kernel
    loss = dp.mean_loss(torch.tensor([local_loss * scale], dtype=torch.float64))
    full, full_loss = grads_of(x, y, 1.0)
    b = torch.full((3,), float(rank))
    dp.broadcast_(b, src=0)
    ret[rank] = {'shard': (lo, hi), 'err': float((flat - full).abs().max()), 'gmax': float(full.abs().max()),
                 'loss_err': abs(float(loss[0]) - full_loss), 'bcast': b.tolist()}
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('n_global', [8, 7])
def test_gloo_world2_dp_step_equals_single_process_step(n_global):
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0, 'worker failed / timed out'
        res = dict(ret)
    assert res[0]['shard'][0] == 0 and res[0]['shard'][1] == res[1]['shard'][0] and res[1]['shard'][1] == n_global
    for r in range(world):
        assert res[r]['err'] <= 1e-12 * max(1.0, res[r]['gmax']), res[r]
        assert res[r]['loss_err'] < 1e-12
        assert res[r]['bcast'] == [0.0, 0.0, 0.0]


def test_shard_bounds_partition_rows_exactly():
    from dlwp_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 31, 32, 256):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_attach_refuses_a_single_process():
    from dlwp_amd import parallel
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        pytest.skip('a process group is active')
    os.environ.pop('WORLD_SIZE', None)
    with pytest.raises(RuntimeError, match='one process per GPU'):
        parallel.attach(object(), 8)


def test_device_loader_host_path_preserves_order_and_content():
    """DeviceLoader on the CPU device runs the same worker-thread / slot-recycling logic without pinned memory."""
    from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPNeuralNet
    from dlwp_amd.model.generators import DeviceLoader
    rng = np.random.default_rng(0)
    P = rng.standard_normal((23, 2, 2, 5, 6)).astype(np.float32)
    T = rng.standard_normal((23, 2, 2, 5, 6)).astype(np.float32)
    d = DLWPNeuralNet(is_convolutional=True, time_dim=2, scaler_type=None, scale_targets=False)
    gen = DataGenerator(d, ArrayDataset(P, T), batch_size=4)
    order = [5, 0, 3, 1, 2, 4]
    got = [(X.clone().numpy(), y.clone().numpy()) for X, y in DeviceLoader(gen, torch.device('cpu'), order=order)]
    assert len(got) == len(order)
    for (X, y), i in zip(got, order):
        Xr, yr = gen[i]
        assert np.array_equal(X, Xr) and np.array_equal(y, yr)

    class Boom(object):
        def __len__(self):
            return 3

        def __getitem__(self, i):
            if i == 1:
                raise KeyError('bad batch')
            return gen[i]
    with pytest.raises(KeyError):
        list(DeviceLoader(Boom(), torch.device('cpu')))
