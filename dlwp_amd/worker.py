"""
A worker rank of the driver mode of dlwp_amd.parallel: `python -m dlwp_amd.worker`, started by parallel.spawn() when a plain
single-process script asks for build_model(..., gpus=N) -- the way the reference's scripts ask keras.utils.multi_gpu_model for
N GPUs inside one process (DLWP/model/models.py:104-109; examples/train_generator.py:243, Azure/train_func.py:101).

The worker never sees the user's script.  It joins the process group (RANK / WORLD_SIZE / MASTER_* from the driver), then
serves commands rank 0 broadcasts:
    ('build', spec)                      build the same model from its specification (layer triples | saved functional graph,
                                         wrapper arguments, compile arguments); compile aligns the replicas on rank 0's weights
    ('call', name, args, kwargs)         run net.<name>(*args, **kwargs) -- the calls that hold a collective (compile, set_weights,
                                         train_on_batch, fit, fit_generator); a barrier closes the call
    ('rollout', predictors, steps, kw)   forecast this rank's rows of the members, leave the series in a /dev/shm file
    ('stop',)
A failure is fatal for the worker (traceback on stderr, exit status 1): rank 0 notices the dead process at its next command.
"""
import os
import sys
import traceback

import numpy as np


def build(spec):
    kind, wargs, body, compile_kwargs = spec
    from . import parallel
    from .model import DLWPFunctional, DLWPNeuralNet
    world = int(os.environ['WORLD_SIZE'])
    if kind == 'sequential':
        d = DLWPNeuralNet(**wargs)
        d.build_model(body, gpus=world, **compile_kwargs)
    elif kind == 'functional':
        from . import serialization
        d = DLWPFunctional(**wargs)
        net = serialization.load_model_file(body)
        d.build_model(net, gpus=world, **compile_kwargs)
    else:
        raise ValueError('unknown model specification %r' % (kind,))
    return d


def main():
    from . import parallel
    rank, world, _ = parallel.init()
    import torch
    import torch.distributed as dist
    wrapper = None
    try:
        while True:
            box = [None]
            dist.broadcast_object_list(box, src=0)
            cmd = parallel.loads(box[0], wrapper=wrapper)
            if cmd[0] == 'stop':
                break
            if cmd[0] == 'build':
                wrapper = build(cmd[1])
            elif cmd[0] == 'call':
                _, name, args, kwargs = cmd
                getattr(wrapper.model, name)(*args, **kwargs)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                dist.barrier()
            elif cmd[0] == 'rollout':
                _, predictors, steps, kwargs = cmd
                part = None
                try:
                    lo, hi = parallel.shard_bounds(predictors.shape[0], rank, world)
                    if hi > lo:
                        out = np.ascontiguousarray(wrapper.predict_timeseries(np.asarray(predictors[lo:hi]), steps, **kwargs))
                        path = os.path.join(parallel._shm_dir(), 'dlwp_series_%d_%d.bin' % (os.getpid(), rank))
                        out.tofile(path)
                        part = (path, out.dtype.str, out.shape)
                except Exception as e:  # noqa: BLE001  (rank 0 waits in gather_object: it must get an answer)
                    traceback.print_exc()
                    part = 'rank %d: %r' % (rank, e)
                dist.gather_object(part, None, dst=0)
            else:
                raise ValueError('unknown command %r' % (cmd[0],))
    except BaseException:  # noqa: BLE001
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == '__main__':
    main()
