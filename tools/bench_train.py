#!/usr/bin/env python3
"""Training-step throughput (BASELINE.json configs[2]): the 88x180x4 sequential U-Net, fp32, batch 64 per GPU, 'mse' loss,
'mae' metric, Adam; under torch.distributed the flat gradient buffer is all-reduced once per step (RCCL).
    python tools/bench_train.py [--batch 64] [--steps 20] [--warmup 3]
    python -m torch.distributed.run --nproc-per-node 8 tools/bench_train.py --gpus 8"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='samples per GPU per step')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=15)
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--channels', type=int, default=4)
    a = ap.parse_args()
    from dlwp_amd import parallel
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    rank, world, local = parallel.init()
    grid = tuple(int(v) for v in a.grid.split('x'))
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((a.channels,) + grid), loss='mse', optimizer='adam', metrics=['mae'], gpus=world)
    dev = d.model.device
    g = torch.Generator().manual_seed(0)
    n_global = a.batch * world
    x = torch.randn((n_global, a.channels) + grid, generator=g).to(dev)
    y = torch.randn((n_global, a.channels) + grid, generator=g).to(dev)
    tr = d.model._trainer
    for _ in range(a.warmup):
        tr.train_on_batch(x, y, return_device=True)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        lv, _ = tr.train_on_batch(x, y, return_device=True)
    t_host = time.perf_counter() - t0          # the host's share: launches issued, nothing waited for
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    flops = 3.0 * d.model.plan.conv_flops_per_sample() * n_global * a.steps   # fwd + dgrad + wgrad (L1 dgrad skipped)
    if rank == 0:
        print(json.dumps({'metric': 'training samples/s (cfg3: 88x180x4 U-Net, fp32, Adam, mse)', 'value': n_global * a.steps / dt,
                          'unit': 'samples/s', 'n_gpus': world, 'batch_per_gpu': a.batch, 'steps': a.steps,
                          'ms_per_step': 1e3 * dt / a.steps, 'host_ms_per_step': 1e3 * t_host / a.steps,
                          'captured_graph': bool(tr._graphs), 'approx_tflops_per_gpu': flops / dt / 1e12 / world,
                          'loss': [float(v) for v in lv.detach().cpu().numpy().ravel()]}))


if __name__ == '__main__':
    main()
