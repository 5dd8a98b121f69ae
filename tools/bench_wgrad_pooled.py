#!/usr/bin/env python3
"""dlwp_conv2d_bwd_weight_pooled on the first layer of config 3 (4 -> 32, 3x3 dilation 2, 88 x 180, MaxPooling2D(2) behind it) --
the launch the training step takes for that layer -- timed alone with HIP events.  python tools/bench_wgrad_pooled.py [--batch 64]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    from dlwp_amd import _lib, ops
    n, cin, cout, h, w, dil = a.batch, 4, 32, 88, 180, 2
    rng = np.random.default_rng(1)
    x = torch.tensor(rng.standard_normal((n, cin, h, w)).astype(np.float32), device='cuda')
    y = torch.tensor(np.tanh(rng.standard_normal((n, cout, h, w))).astype(np.float32), device='cuda')
    dp = torch.tensor(rng.standard_normal((n, cout, h // 2, w // 2)).astype(np.float32), device='cuda')
    cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1), ops.ACT_TANH)
    xs = _lib.Shape4(n, cin, h, w)
    dw = torch.empty((3, 3, cin, cout), device='cuda')
    db = torch.empty(cout, device='cuda')
    for _ in range(5):
        ops.conv2d_bwd_weight_pooled(x, y, dp, dw, db, cd, xs, ops.ACT_TANH)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.conv2d_bwd_weight_pooled(x, y, dp, dw, db, cd, xs, ops.ACT_TANH)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    gb = (y.numel() + dp.numel() + x.numel()) * 4 / 1e9
    print(json.dumps({'ms': ms, 'batch': n, 'algorithmic_gbs': gb / ms * 1e3, 'note': 'weight gradient + slab sums + bias gradient'}))


if __name__ == '__main__':
    main()
