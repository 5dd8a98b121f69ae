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
    # the epoch's shuffle: every rank draws its own permutation, all must leave with rank 0's
    perm = dp.broadcast_indices(np.random.RandomState(100 + rank).permutation(11))
    ret[rank] = {'shard': (lo, hi), 'err': float((flat - full).abs().max()), 'gmax': float(full.abs().max()),
                 'loss_err': abs(float(loss[0]) - full_loss), 'bcast': b.tolist(), 'perm': perm.tolist(),
                 'rccl_abi': dp.uses_rccl_abi()}
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
        assert res[r]['perm'] == np.random.RandomState(100).permutation(11).tolist()
        assert res[r]['rccl_abi'] is False          # gloo group: collectives stay in torch.distributed


def test_shard_bounds_partition_rows_exactly():
    from dlwp_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 31, 32, 256):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_attach_from_a_single_process_needs_the_model_specification():
    """build_model(gpus=n) from a plain process starts the other ranks itself (parallel.spawn) and sends them the model's
    specification; a bare attach() without one has nothing to send and says what to do."""
    from dlwp_amd import parallel
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        pytest.skip('a process group is active')
    os.environ.pop('WORLD_SIZE', None)
    assert parallel.needs_spawn()
    with pytest.raises(RuntimeError, match='one process per GPU'):
        parallel.attach(object(), 8)


def test_driver_commands_travel_with_arrays_in_shared_files_and_object_references():
    """parallel.dumps / loads: what rank 0 of the driver mode sends its workers -- large arrays through files the receiver maps,
    the wrapper / network objects as references to the receiver's own (a generator holds its DLWP model), the rest pickled."""
    from dlwp_amd import parallel
    from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPNeuralNet
    rng = np.random.default_rng(0)
    P = rng.standard_normal((40, 2, 2, 16, 24)).astype(np.float32)          # 245 KB: a file
    T = rng.standard_normal((40, 2, 2, 16, 24)).astype(np.float32)
    small = np.arange(5, dtype=np.int64)
    mine = DLWPNeuralNet(is_convolutional=True, time_dim=2, scaler_type=None, scale_targets=False)
    theirs = DLWPNeuralNet(is_convolutional=True, time_dim=2, scaler_type=None, scale_targets=False)
    np.random.seed(3)
    gen = DataGenerator(mine, ArrayDataset(P, T), batch_size=8, shuffle=True)
    files = []
    blob = parallel.dumps(('call', 'fit_generator', (gen,), {'epochs': 2, 'idx': small, 't': torch.arange(4.)}), wrapper=mine,
                          files=files)
    try:
        assert len(files) == 2 and all(os.path.exists(f) for f in files) and len(blob) < 20000
        cmd = parallel.loads(blob, wrapper=theirs)
        g2 = cmd[2][0]
        assert cmd[:2] == ('call', 'fit_generator') and cmd[3]['epochs'] == 2 and np.array_equal(cmd[3]['idx'], small)
        assert np.array_equal(cmd[3]['t'], np.arange(4, dtype=np.float32))
        assert g2.model is theirs and g2 is not gen
        assert np.array_equal(g2._indices, gen._indices)                     # the shuffle of the sender
        for i in range(len(gen)):
            a, b = gen[i], g2[i]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        for f in files:
            os.unlink(f)


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


def test_device_loader_rank_shards_partition_every_global_batch():
    """The data-parallel feed: rank r of w gathers ONLY rows shard_bounds(n_batch, r, w) of global batch i, in the
    generator's own (shuffled) sample order; the shards of all ranks concatenate to the batch a single process sees."""
    from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPNeuralNet
    from dlwp_amd.model.generators import DeviceLoader
    from dlwp_amd.parallel import shard_bounds
    rng = np.random.default_rng(3)
    P = rng.standard_normal((23, 2, 2, 5, 6)).astype(np.float32)
    T = rng.standard_normal((23, 2, 2, 5, 6)).astype(np.float32)
    d = DLWPNeuralNet(is_convolutional=True, time_dim=2, scaler_type=None, scale_targets=False)
    np.random.seed(5)
    gen = DataGenerator(d, ArrayDataset(P, T), batch_size=8, shuffle=True)
    gathered = []
    orig = gen.generate

    def spy(samples, *a, **k):
        gathered.append(len(samples))
        return orig(samples, *a, **k)
    for world in (2, 3):
        parts = []
        for rank in range(world):
            gen.generate = spy
            rows = [(X.clone().numpy(), y.clone().numpy(), n)
                    for X, y, n in DeviceLoader(gen, torch.device('cpu'), shard=(rank, world)).iter_batches()]
            gen.generate = orig
            parts.append(rows)
        # no rank ever gathered more than its share of a batch (float32 row gathers go to the library's host threads and never
        # call generate(): the rows they copied are checked below)
        assert all(g <= -(-8 // world) for g in gathered)
        del gathered[:]
        for i in range(len(gen)):
            Xf, yf = gen[i]
            assert all(parts[r][i][2] == Xf.shape[0] for r in range(world))
            assert np.array_equal(np.concatenate([parts[r][i][0] for r in range(world)]), Xf)
            assert np.array_equal(np.concatenate([parts[r][i][1] for r in range(world)]), yf)
            for r in range(world):
                lo, hi = shard_bounds(Xf.shape[0], r, world)
                assert parts[r][i][0].shape[0] == hi - lo


def test_device_loader_list_targets_and_empty_shards():
    """Multi-output generators (SeriesDataGenerator(sequence=K)) hand a LIST of target arrays: each is staged on its own
    and comes out as a list.  A rank whose shard of a short batch is empty gets zero-row tensors of the right shape."""
    from dlwp_amd.model.generators import DeviceLoader

    class Seq(object):
        _batch_size = 4

        def __init__(self, n):
            self._indices = np.arange(n)[::-1].copy()
            self.data = np.arange(n * 6, dtype=np.float32).reshape(n, 2, 3)

        def __len__(self):
            return -(-len(self._indices) // self._batch_size)

        def generate(self, samples):
            assert len(samples) > 0
            x = self.data[np.asarray(samples)]
            return x, [x + 1, x * 2, x - 3]

        def __getitem__(self, i):
            return self.generate(self._indices[i * 4:(i + 1) * 4])
    gen = Seq(9)                                      # batches of 4, 4, 1
    whole = list(DeviceLoader(gen, torch.device('cpu')))
    assert len(whole) == 3 and isinstance(whole[0][1], list) and len(whole[0][1]) == 3
    for world in (2, 3):
        for i in range(3):
            Xs, ys = [], [[], [], []]
            for rank in range(world):
                X, y, n = list(DeviceLoader(gen, torch.device('cpu'), order=[i], shard=(rank, world)).iter_batches())[0]
                assert n == gen[i][0].shape[0] and isinstance(y, list) and len(y) == 3
                assert tuple(X.shape[1:]) == (2, 3) and all(tuple(t.shape) == tuple(X.shape) for t in y)
                Xs.append(X.clone().numpy())
                for k in range(3):
                    ys[k].append(y[k].clone().numpy())
            assert np.array_equal(np.concatenate(Xs), gen[i][0])
            for k in range(3):
                assert np.array_equal(np.concatenate(ys[k]), gen[i][1][k])


def test_bench_starts_its_own_ranks_when_launched_as_a_plain_script():
    """`python bench.py --gpus 2` with no torchrun environment (how the driver starts the multi-GPU bench) must spawn its two
    ranks itself.  Without a GPU every rank stops at 'needs an MI355X' -- AFTER the gloo process group of world size 2 came
    up -- and the parent hands the failure on as its exit code instead of hanging or printing a line."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip('the GPU box runs the real thing (tests/test_gpu_parallel.py)')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['DLWP_DIST_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--no-extras', '--no-cpu-baseline', '--launch-timeout', '240'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode not in (0, 124), (p.returncode, p.stderr[-400:])
    assert p.stderr.count('needs an MI355X') == 2, p.stderr[-800:]
    assert 'launch with torch.distributed.run' not in p.stderr
    assert not any(l.startswith('{') for l in p.stdout.splitlines()), p.stdout[-400:]
