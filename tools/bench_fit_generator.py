#!/usr/bin/env python3
"""BASELINE.json configs[2] as the reference drives it: `fit_generator(DataGenerator(batch), ...)` (examples/train.py:262-263,
DLWP/model/models.py:216-228) -- host gather of every shuffled batch, H2D upload, training step -- against the same step on
device-resident tensors.  Prints one JSON line.
    python tools/bench_fit_generator.py [--batch 64] [--samples 2560] [--epochs 3]"""
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
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--samples', type=int, default=2560)
    ap.add_argument('--epochs', type=int, default=3, help='timed epochs (one more runs first, untimed)')
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--channels', type=int, default=4)
    a = ap.parse_args()
    from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    grid = tuple(int(v) for v in a.grid.split('x'))
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((a.channels,) + grid), loss='mse', optimizer='adam', metrics=['mae'])
    rng = np.random.default_rng(0)
    n = a.samples - a.samples % a.batch
    P = rng.standard_normal((n, 2, a.channels // 2) + grid, dtype=np.float32)
    T = rng.standard_normal((n, 2, a.channels // 2) + grid, dtype=np.float32)
    gen = DataGenerator(d, ArrayDataset(P, T), batch_size=a.batch, shuffle=True)
    tr = d.model._trainer
    steps = len(gen)
    d.fit_generator(gen, epochs=1, verbose=0)                       # warm-up: buffers, the recorded step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.fit_generator(gen, epochs=a.epochs, verbose=0)
    torch.cuda.synchronize()
    fed = (time.perf_counter() - t0) / (a.epochs * steps)
    # the same step on device-resident tensors, no loader
    dev = d.model.device
    x = torch.from_numpy(P[:a.batch].reshape((a.batch, a.channels) + grid)).to(dev)
    y = torch.from_numpy(T[:a.batch].reshape((a.batch, a.channels) + grid)).to(dev)
    for _ in range(10):
        tr.train_on_shard(x, y, a.batch, return_device=True)
    torch.cuda.synchronize()
    k = max(50, a.epochs * steps)
    t0 = time.perf_counter()
    for _ in range(k):
        tr.train_on_shard(x, y, a.batch, return_device=True)
    host = (time.perf_counter() - t0) / k                           # launches issued, nothing waited for
    torch.cuda.synchronize()
    res = (time.perf_counter() - t0) / k
    form = tr._graph_ok(a.batch)
    print(json.dumps({'metric': 'fit_generator step (cfg3: 88x180x4 U-Net, DataGenerator feed, H2D included)', 'batch': a.batch,
                      'steps_per_epoch': steps, 'loader_fed_ms': round(1e3 * fed, 4), 'device_resident_ms': round(1e3 * res, 4),
                      'loader_over_resident': round(fed / res, 3), 'host_ms_per_step': round(1e3 * host, 4),
                      'step_form': form or 'python', 'native_gather': gen.batch_sources() is not None,
                      'samples_per_s_loader_fed': round(a.batch / fed, 1),
                      'h2d_mb_per_step': round(2 * a.batch * a.channels * grid[0] * grid[1] * 4 / 1e6, 2)}))


if __name__ == '__main__':
    main()
