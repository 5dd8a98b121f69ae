"""
Data parallelism: one process per GPU under torch.distributed (backend 'nccl' == RCCL over xGMI on the GPU box, 'gloo'
in the CPU tests).  Replaces keras.utils.multi_gpu_model (reference DLWP/model/models.py:104-109, 365-372), which
replicates the graph inside one process, keeps the weights on the CPU and moves them over PCIe every step.

  training : every rank holds the full (tiny, <1 MB) weight set, trains on its row shard of the global batch, and the
             ONE flat fp32 gradient buffer is summed with a single all-reduce per step (the message is ~756 KB for the
             config-2 U-Net: latency-bound, so one collective, no bucketing).
  inference: ensemble members / samples are independent (reference models.py:277-293 is elementwise over the sample
             axis): shard_rows() gives each rank its members and NO collective is issued during the rollout.
"""
import os

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

    def shard(self, n):
        return shard_bounds(n, self.rank, self.world)

    def all_reduce_sum_(self, flat):
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        return flat

    def mean_loss(self, vals):
        vals = vals.clone()
        self.dist.all_reduce(vals, op=self.dist.ReduceOp.SUM, group=self.group)
        return vals / self.world

    def broadcast_(self, flat, src=0):
        self.dist.broadcast(flat, src=src, group=self.group)
        return flat


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
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local


def attach(model, gpus):
    """build_model(gpus=n): make `model` data parallel over the current process group.  If the script was not launched
    with one process per GPU (no process group, WORLD_SIZE unset) this raises instead of silently training on one GPU."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            init()
        else:
            raise RuntimeError('gpus=%d requested but this is a single process: launch one process per GPU, e.g. '
                               '`python -m torch.distributed.run --nproc-per-node %d script.py`' % (gpus, gpus))
    dp = DataParallel()
    if dp.world != gpus:
        raise RuntimeError('gpus=%d but the process group has %d ranks' % (gpus, dp.world))
    model._dp = dp
    return dp


def sync_parameters(model):
    """Broadcast rank 0's flat parameter buffer so every replica starts identical."""
    dp = getattr(model, '_dp', None)
    tr = getattr(model, '_trainer', None)
    if dp is not None and tr is not None and dp.world > 1:
        dp.broadcast_(tr.flat_params)
