"""
Data parallelism: one process per GPU.  Replaces keras.utils.multi_gpu_model (reference DLWP/model/models.py:104-109,
365-372), which replicates the graph inside one process, keeps the weights on the CPU and moves them over PCIe every step.

  training : every rank holds the full (tiny, <1 MB) weight set and trains on ITS ROWS of the global batch -- contiguous
             row shards, exactly how multi_gpu_model's get_slice cuts a batch; the global batch is n_gpu x the per-GPU
             batch (reference Azure/train_tf.py:163-164).  The ONE flat fp32 gradient buffer (with the loss table riding
             in its tail) is summed with a single all-reduce per step: ~756 KB for the config-2 U-Net, latency-bound, so
             one collective and no bucketing.  On the GPU box that all-reduce is the library's own RCCL communicator
             (include/dlwp_hip.h: dlwp_comm_*, dlwp_allreduce_sum_f32) enqueued on the training stream right behind the
             last weight-gradient kernel; torch.distributed only carries the rendezvous (the 128-byte unique id) and the
             host-side index broadcasts.  With the 'gloo' backend (CPU tests, two ranks sharing one GPU) the same calls go
             through torch.distributed.
  feeding  : every rank gathers and uploads only its own rows (DeviceLoader(shard=...), Trainer.fit): the batch
             index list is rank 0's, broadcast once per epoch, so shuffles agree without relying on seeds.
  inference: ensemble members / samples are independent (reference models.py:277-293 is elementwise over the sample
             axis): shard_bounds() gives each rank its members and NO collective is issued during the rollout.

Two ways in, both one process per GPU underneath ("one process per GPU or one process with 8 handles -- both must
work"):
  launched : `python -m torch.distributed.run --nproc-per-node N script.py` -- every rank runs the script (SPMD).
  driver   : a PLAIN `python script.py` that calls build_model(..., gpus=N), as the reference's scripts do
             (examples/train_generator.py:243, Azure/train_func.py:101 -> keras.utils.multi_gpu_model inside one process):
             attach() starts the N - 1 other ranks itself (`python -m dlwp_amd.worker`), sends them the model's SPECIFICATION --
             layer triples or the saved functional graph, compile arguments; never the user's script -- and from then on
             MIRRORS the calls that hold a collective (compile, set_weights, train_on_batch, fit, fit_generator) to them:
             arguments travel pickled, arrays through files in /dev/shm that the workers map, the wrapper / network objects
             inside generators are replaced by the worker's own.  predict_timeseries shards the members over the ranks and
             gathers the series on rank 0.  Callbacks and validation run on rank 0 only; its stop_training flag is agreed at
             the end of every epoch.
"""
import ctypes
import os

import numpy as np
import torch


def shard_bounds(n, rank, world):
    """Contiguous row shard [lo, hi) of n rows for `rank` of `world` (remainder spread over the first ranks)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallel(object):
    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; call dlwp_amd.parallel.init() first')
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self._comm = None           # dlwp_comm_t (RCCL through the C ABI) once a device buffer asked for it
        self._comm_device = None
        #: driver mode (a plain script that asked for gpus=n, parallel.spawn): rank 0 alone runs callbacks, so its stop_training
        #: flag is agreed at the end of every epoch
        self.mirror = os.environ.get('DLWP_WORKER') == '1'

    def shard(self, n):
        return shard_bounds(n, self.rank, self.world)

    # -- the library's RCCL communicator ------------------------------------------------------------------------------ #
    def uses_rccl_abi(self, t=None):
        """True when device collectives go through dlwp_comm_* (backend 'nccl' on a GPU; DLWP_RCCL_ABI=0 keeps them in
        torch.distributed)."""
        if self.backend != 'nccl' or os.environ.get('DLWP_RCCL_ABI', '1') == '0':
            return False
        return t is None or t.is_cuda

    def comm(self, device):
        """dlwp_comm_t of this rank on `device`; created collectively at the first call (every rank must make it)."""
        if self._comm is None:
            from . import _lib
            idx = device.index if device.index is not None else torch.cuda.current_device()
            nbytes = ctypes.c_size_t(0)
            _lib.check(_lib.lib.dlwp_comm_unique_id(None, ctypes.byref(nbytes)))
            uid = (ctypes.c_char * nbytes.value)()
            if self.rank == 0:
                _lib.check(_lib.lib.dlwp_comm_unique_id(uid, ctypes.byref(nbytes)))
            box = [bytes(uid)]
            self.dist.broadcast_object_list(box, src=0, group=self.group)       # rendezvous only: 128 bytes
            raw = (ctypes.c_char * nbytes.value).from_buffer_copy(box[0])
            out = ctypes.c_void_p()
            _lib.check(_lib.lib.dlwp_comm_init_rank(ctypes.byref(out), int(idx), self.world, self.rank, raw, nbytes.value))
            self._comm, self._comm_device = out, idx
        return self._comm

    def close(self):
        if self._comm is not None:
            from . import _lib
            _lib.lib.dlwp_comm_destroy(self._comm)
            self._comm = None
        if getattr(self, '_xchg', None):
            from . import _lib
            _lib.lib.dlwp_xchg_destroy(self._xchg[0])
            self._xchg = None

    # -- the library's own one-shot exchange (csrc/xchg.hip), DLWP_ALLREDUCE=oneshot ------------------------------------- #
    def wants_oneshot(self, flat):
        """the step's exchange through dlwp_xchg_* (every rank publishes its buffer in peer-mapped memory, reads all of them, sums
        in rank order and updates its parameters in the same kernel) instead of RCCL / torch.distributed.  Opt-in
        (DLWP_ALLREDUCE=oneshot): RCCL stays the default until a multi-GPU run has been through it."""
        return os.environ.get('DLWP_ALLREDUCE', '') == 'oneshot' and flat.is_cuda and flat.numel() % 4 == 0 and self.world <= 16

    def xchg(self, flat):
        """dlwp_xchg_t for buffers of flat.numel() floats; created collectively at the first call (the 64-byte IPC handles travel
        through torch.distributed's host channel, like RCCL's unique id).  Failure is collective too: a rank whose region cannot be
        created, exported or mapped still takes part in every host-side exchange of this call, and ALL ranks raise together -- no
        rank is left waiting in a collective for one that gave up (r5: the first run between real GPUs must not hang)."""
        ent = getattr(self, '_xchg', None)
        if ent is None or ent[1] != flat.numel():
            from . import _lib
            if ent is not None:
                _lib.lib.dlwp_xchg_destroy(ent[0])
                self._xchg = None
            idx = flat.device.index if flat.device.index is not None else torch.cuda.current_device()
            hd = (ctypes.c_char * 64)()
            out = ctypes.c_void_p()
            err = None
            if _lib.lib.dlwp_xchg_create(_lib.handle(idx), self.world, self.rank, flat.numel(), hd, ctypes.byref(out)) != _lib.OK:
                err = _lib.lib.dlwp_last_error().decode('utf-8', 'replace')
                out = ctypes.c_void_p()
            got = [None] * self.world
            self.dist.all_gather_object(got, (err, bytes(hd)), group=self.group)
            if err is None and all(g[0] is None for g in got):
                blob = (ctypes.c_char * (64 * self.world)).from_buffer_copy(b''.join(g[1] for g in got))
                if _lib.lib.dlwp_xchg_connect(out, blob) != _lib.OK:
                    err = _lib.lib.dlwp_last_error().decode('utf-8', 'replace')
            errs = [None] * self.world
            self.dist.all_gather_object(errs, err if err is not None else next((g[0] for g in got if g[0] is not None), None),
                                        group=self.group)          # (also the barrier: every region is mapped everywhere
            bad = [(r, e) for r, e in enumerate(errs) if e is not None]          #  before the first flag is written)
            if bad:
                if out:
                    _lib.lib.dlwp_xchg_destroy(out)
                raise RuntimeError('dlwp_xchg: the one-shot exchange could not be set up (rank %d: %s)' % bad[0])
            ent = self._xchg = (out, flat.numel())
        return ent[0]

    def oneshot_all_reduce_(self, flat):
        from . import _lib
        _lib.check(_lib.lib.dlwp_xchg_allreduce_sum_f32(self.xchg(flat), ctypes.c_void_p(flat.data_ptr()), flat.numel(), self._stream(flat)))
        return flat

    def oneshot_adam_(self, flat, n_params, p, m, v, lr, beta_1, beta_2, epsilon, decay, iteration, grad_scale):
        """flat <- the sums over the ranks (in rank order); p, m, v <- the Keras-form Adam step on grad_scale * flat[:n_params]"""
        from . import _lib
        _lib.check(_lib.lib.dlwp_xchg_allreduce_adam(self.xchg(flat), ctypes.c_void_p(flat.data_ptr()), int(n_params), flat.numel(),
                                                     ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(m.data_ptr()),
                                                     ctypes.c_void_p(v.data_ptr()), float(lr), float(beta_1), float(beta_2),
                                                     float(epsilon), float(decay), int(iteration), float(grad_scale), self._stream(flat)))

    def oneshot_timed_out(self):
        from . import _lib
        ent = getattr(self, '_xchg', None)
        if ent is None:
            return False
        t = ctypes.c_int(0)
        _lib.check(_lib.lib.dlwp_xchg_status(ent[0], ctypes.byref(t)))
        return bool(t.value)

    def oneshot_check(self):
        """Raises when a launch of the one-shot exchange gave up waiting for a peer (2 s): that launch left the parameters alone
        and wrote NaN over the loss table, and the exchange is dead from then on -- training must not continue on it.
        Synchronises the device: the trainer calls it wherever it reads a loss back anyway (every train_on_batch, once per epoch
        of fit / fit_generator)."""
        if getattr(self, '_xchg', None) is not None and self.oneshot_timed_out():
            raise RuntimeError('dlwp_xchg: rank %d waited 2 s for a peer of the one-shot all-reduce (DLWP_ALLREDUCE=oneshot); the step '
                               'was NOT applied.  A rank died, or the ranks do not call the step collectively.' % self.rank)

    def oneshot_info(self):
        """{'memory': 'uncached' | 'fine-grained' | 'plain', 'blocks': persistent grid size} of the exchange, or None"""
        from . import _lib
        ent = getattr(self, '_xchg', None)
        if ent is None:
            return None
        k, b = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib.dlwp_xchg_info(ent[0], ctypes.byref(k), ctypes.byref(b)))
        return {'memory': {2: 'uncached', 1: 'fine-grained', 0: 'plain'}.get(k.value, '?'), 'blocks': b.value}

    # -- collectives on flat float32 device buffers ---------------------------------------------------------------------- #
    @staticmethod
    def _stream(t):
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def all_reduce_sum_(self, flat):
        if self.uses_rccl_abi(flat):
            from . import _lib
            if flat.dtype != torch.float32 or not flat.is_contiguous():
                raise ValueError('the RCCL all-reduce takes a contiguous float32 buffer')
            _lib.check(_lib.lib.dlwp_allreduce_sum_f32(self.comm(flat.device), ctypes.c_void_p(flat.data_ptr()),
                                                       flat.numel(), self._stream(flat)))
        else:
            self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        return flat

    def broadcast_(self, flat, src=0):
        if self.uses_rccl_abi(flat) and flat.dtype == torch.float32 and flat.is_contiguous():
            from . import _lib
            _lib.check(_lib.lib.dlwp_broadcast_f32(self.comm(flat.device), ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                                   int(src), self._stream(flat)))
        else:
            self.dist.broadcast(flat, src=src, group=self.group)
        return flat

    def mean_loss(self, vals):
        vals = vals.clone()
        self.all_reduce_sum_(vals)
        return vals / self.world

    # -- host-side agreement (index lists, flags) ------------------------------------------------------------------------ #
    def broadcast_indices(self, idx, src=0):
        """Rank `src`'s integer index array on every rank (one small object broadcast; the shuffle of an epoch)."""
        box = [np.ascontiguousarray(idx, dtype=np.int64) if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def barrier(self):
        self.dist.barrier(group=self.group)


# --------------------------------------------------------------------------------------------------------------------- #
# driver mode: a single script process that owns N - 1 worker ranks
# --------------------------------------------------------------------------------------------------------------------- #

#: arrays of at least this many bytes travel through /dev/shm files instead of the pickle
SHM_MIN_BYTES = 1 << 16


def _shm_dir():
    return '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else __import__('tempfile').gettempdir()


def dumps(obj, wrapper=None, net=None, files=None, prefix='dlwp'):
    """pickle `obj` for the other ranks: the DLWP wrapper / its network become references to the receiver's own, arrays (numpy
    or torch) of SHM_MIN_BYTES or more are written to files the receiver maps (their paths are appended to `files`)."""
    import io
    import pickle

    class P(pickle.Pickler):
        def persistent_id(self, o):
            if wrapper is not None and o is wrapper:
                return ('wrapper',)
            if net is not None and o is net:
                return ('net',)
            if isinstance(o, torch.Tensor):
                o = o.detach().cpu().numpy()
                if o.nbytes < SHM_MIN_BYTES:
                    return ('array', o)
            if isinstance(o, np.ndarray) and o.nbytes >= SHM_MIN_BYTES and o.dtype != object and files is not None:
                path = os.path.join(_shm_dir(), '%s_%d_%d.bin' % (prefix, os.getpid(), len(files) + dumps.counter))
                dumps.counter += 1
                a = np.ascontiguousarray(o)
                with open(path, 'wb') as f:
                    f.write(a.data)
                files.append(path)
                return ('file', path, a.dtype.str, a.shape)
            return None
    buf = io.BytesIO()
    P(buf, protocol=4).dump(obj)
    return buf.getvalue()


dumps.counter = 0


def loads(data, wrapper=None, net=None):
    import io
    import pickle

    class U(pickle.Unpickler):
        def persistent_load(self, pid):
            if pid[0] == 'wrapper':
                return wrapper
            if pid[0] == 'net':
                return net if net is not None else getattr(wrapper, 'model', None)
            if pid[0] == 'array':
                return pid[1]
            if pid[0] == 'file':
                _, path, dt, shape = pid
                if int(np.prod(shape)) == 0:
                    return np.empty(shape, dtype=np.dtype(dt))
                return np.memmap(path, dtype=np.dtype(dt), mode='r', shape=tuple(shape))
            raise pickle.UnpicklingError('unknown persistent id %r' % (pid,))
    return U(io.BytesIO(data)).load()


class Driver(object):
    """Rank 0 of a group it started itself.  send() broadcasts one command to the worker loop (dlwp_amd/worker.py); call() mirrors a
    method call: the workers run it on their replica with the same arguments while rank 0 runs it locally."""

    def __init__(self, dp, procs):
        self.dp, self.procs, self.in_call, self.closed = dp, procs, False, False
        self.wrapper = None

    @property
    def world(self):
        return self.dp.world

    def check_workers(self):
        dead = [(r + 1, p.poll()) for r, p in enumerate(self.procs) if p.poll() is not None]
        if dead and not self.closed:
            raise RuntimeError('data-parallel worker rank %d exited with status %r' % dead[0])

    def send(self, cmd, net=None, files=None):
        self.check_workers()
        blob = dumps(cmd, wrapper=self.wrapper, net=net, files=files)
        self.dp.dist.broadcast_object_list([blob], src=0, group=self.dp.group)

    def call(self, net, name, args, kwargs, fn):
        """mirror net.<name>(*args, **kwargs): the workers get the call (callbacks / validation / verbosity stay here), then rank 0
        runs fn -- the collectives inside pair up."""
        wkw = {k: v for k, v in kwargs.items() if k not in ('callbacks', 'validation_data', 'validation_steps', 'verbose')}
        if name in ('fit', 'fit_generator'):
            wkw['verbose'] = 0
        files = []
        self.in_call = True
        try:
            self.send(('call', name, args, wkw), net=net, files=files)
            out = fn(net, *args, **kwargs)
            self.dp.barrier()                # the workers are through the call: their maps of the argument files are no longer needed
            return out
        finally:
            self.in_call = False
            for f in files:
                try:
                    os.unlink(f)
                except OSError:
                    pass

    def sharded_rollout(self, wrapper, predictors, time_steps, kwargs):
        """predict_timeseries with the members sharded over the ranks (no collective in the rollout; reference models.py:277-293 is
        elementwise over the sample axis): every rank forecasts its contiguous rows and leaves the series in a /dev/shm file, rank 0
        joins them along the sample axis."""
        predictors = np.ascontiguousarray(predictors, dtype=np.float32)
        files = []
        self.in_call = True
        try:
            self.send(('rollout', predictors, int(time_steps), kwargs), net=wrapper.model, files=files)
            lo, hi = self.dp.shard(predictors.shape[0])
            mine = wrapper.predict_timeseries(predictors[lo:hi], time_steps, **kwargs) if hi > lo else None
            parts = [None] * self.world
            self.dp.dist.gather_object(None, parts, dst=0, group=self.dp.group)
            series = []
            for r, part in enumerate(parts):
                if r == 0:
                    if mine is not None:
                        series.append(np.asarray(mine))
                    continue
                if part is None:
                    continue
                if isinstance(part, str):
                    raise RuntimeError('data-parallel worker rank %d failed in predict_timeseries: %s' % (r, part))
                path, dt, shape = part
                files.append(path)
                series.append(np.fromfile(path, dtype=np.dtype(dt)).reshape(shape))
            return np.concatenate(series, axis=1) if len(series) > 1 else series[0]
        finally:
            self.in_call = False
            for f in files:
                try:
                    os.unlink(f)
                except OSError:
                    pass

    def close(self):
        if self.closed:
            return
        self.closed = True
        try:
            if all(p.poll() is None for p in self.procs):
                blob = dumps(('stop',))
                self.dp.dist.broadcast_object_list([blob], src=0, group=self.dp.group)
        except Exception:  # noqa: BLE001
            pass
        for p in self.procs:
            try:
                p.wait(timeout=20)
            except Exception:  # noqa: BLE001
                p.kill()
        try:
            if self.dp.dist.is_initialized():
                self.dp.dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


_driver = [None]


def unmirrored(net):
    """context: calls on `net` inside are NOT mirrored to the workers (rank 0's own compile during build_model: the workers
    compile as part of their build command)"""
    import contextlib

    @contextlib.contextmanager
    def ctx():
        drv = getattr(net, '_driver', None)
        prev = drv.in_call if drv is not None else None
        if drv is not None:
            drv.in_call = True
        try:
            yield
        finally:
            if drv is not None:
                drv.in_call = prev
    return ctx()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn(gpus):
    """Start ranks 1 .. gpus - 1 (`python -m dlwp_amd.worker`) and make this process rank 0 of their group.  Returns the Driver."""
    import atexit
    import subprocess
    import sys
    if _driver[0] is not None and not _driver[0].closed:
        if _driver[0].world != gpus:
            raise RuntimeError('this process already drives %d ranks; gpus=%d asked' % (_driver[0].world, gpus))
        return _driver[0]
    if torch.cuda.is_available() and torch.cuda.device_count() < gpus and os.environ.get('DLWP_SHARE_GPUS') != '1':
        raise RuntimeError('gpus=%d requested but only %d device(s) are visible' % (gpus, torch.cuda.device_count()))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), WORLD_SIZE=str(gpus), DLWP_WORKER='1',
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env['PYTHONPATH'] = root + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
    procs = []
    for r in range(1, gpus):
        procs.append(subprocess.Popen([sys.executable, '-m', 'dlwp_amd.worker'], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      cwd=root))
    os.environ.update(MASTER_ADDR=env['MASTER_ADDR'], MASTER_PORT=env['MASTER_PORT'], WORLD_SIZE=str(gpus), RANK='0',
                      LOCAL_RANK=os.environ.get('LOCAL_RANK', '0'), DLWP_DRIVER='1')
    try:
        init()
    except Exception:
        for p in procs:
            p.kill()
        raise
    dp = DataParallel()
    dp.mirror = True
    drv = _driver[0] = Driver(dp, procs)
    atexit.register(drv.close)
    return drv


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and
    bind this process to its GPU.  Returns (rank, world, local_rank).  No-op when already initialised."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if local >= ndev and os.environ.get('DLWP_SHARE_GPUS') == '1':
            local %= ndev          # testing only: several ranks on one GPU (needs DLWP_DIST_BACKEND=gloo, RCCL refuses)
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = backend or os.environ.get('DLWP_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        kwargs = {}
        if backend == 'nccl':
            kwargs['device_id'] = torch.device('cuda', local)
        if os.environ.get('DLWP_DP_TIMEOUT') or os.environ.get('DLWP_WORKER') == '1' or os.environ.get('DLWP_DRIVER') == '1':
            # driver mode: a collective whose peer died must not hold the script for the backend's default (10-30 minutes)
            import datetime
            kwargs['timeout'] = datetime.timedelta(seconds=int(os.environ.get('DLWP_DP_TIMEOUT', '600')))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local


def needs_spawn():
    """does attach() need a model specification for worker ranks -- a plain single process (no group, no launcher environment:
    attach starts the workers) or a process that already drives some?"""
    import torch.distributed as dist
    if _driver[0] is not None and not _driver[0].closed:
        return True
    if os.environ.get('DLWP_WORKER') == '1' or (dist.is_available() and dist.is_initialized()):
        return False
    return int(os.environ.get('WORLD_SIZE', '1')) <= 1


def attach(model, gpus, spec=None, wrapper=None):
    """build_model(gpus=n): make `model` data parallel.  Launched one process per GPU (a process group exists / WORLD_SIZE is set):
    over that group.  From a plain single process: this process becomes rank 0 and starts the other n - 1 ranks itself (spawn);
    `spec` -- what the workers need to build the same model -- is sent to them: ('sequential', wrapper arguments, layer triples,
    compile arguments) or ('functional', wrapper arguments, path of the saved graph, compile arguments)."""
    import torch.distributed as dist
    drv = None
    if os.environ.get('DLWP_WORKER') == '1' or (dist.is_available() and dist.is_initialized() and _driver[0] is None):
        pass                                    # a launched rank (or a worker of a driver): the group is there
    elif _driver[0] is not None and not _driver[0].closed:
        drv = _driver[0]
    elif int(os.environ.get('WORLD_SIZE', '1')) > 1:
        init()
    else:
        if spec is None:
            raise RuntimeError('gpus=%d requested from a single process: build the model through DLWPNeuralNet / DLWPFunctional '
                               'build_model (which starts the other ranks), or launch one process per GPU, e.g. '
                               '`python -m torch.distributed.run --nproc-per-node %d script.py`' % (gpus, gpus))
        drv = spawn(gpus)
    dp = drv.dp if drv is not None else DataParallel()
    if dp.world != gpus:
        raise RuntimeError('gpus=%d but the process group has %d ranks' % (gpus, dp.world))
    model._dp = dp
    if drv is not None:
        drv.wrapper = wrapper
        model._driver = drv
        drv.send(('build', spec))
    tr = getattr(model, '_trainer', None)
    if tr is not None:              # attached after compile: the trainer picks the group up and aligns the replicas
        tr.dp = dp
        tr.sync_parameters()
    return dp


def sync_parameters(model):
    """Broadcast rank 0's flat parameter buffer (and optimizer state) so every replica is identical.  Model.compile and
    the first training step after set_weights / load do this themselves; kept for scripts that edit weights in place."""
    tr = getattr(model, '_trainer', None)
    if tr is not None:
        tr.sync_parameters()
