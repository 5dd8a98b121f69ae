#!/usr/bin/env python3
"""HBM GB/s of the standalone halo / pooling / up-sampling kernels on the shapes of the config-2 U-Net (what the reference
executes as separate TF ops: DLWP/custom.py:202-204 + ZeroPadding2D + MaxPooling2D + UpSampling2D).
Algorithmic bytes = bytes(in) + bytes(out).  GPU only."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_amd import ops  # noqa: E402


def timeit(fn, iters, repeats=3):
    """ms per call: the best of `repeats` event-bracketed runs of `iters` calls (r6: one run of 10 calls read 0.32 ms on a 0.20 ms
    kernel once -- a first touch of freshly allocated memory or a clock ramp inside the only window; the counters of the same
    launch said 5.5 TB/s)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        best = ms if best is None else min(best, ms)
    return best


def measure(n=256, iters=20, pads_only=False):
    """[{kernel, shape, halo, ms, gbs}] -- HIP-event time and ALGORITHMIC GB/s (bytes in + bytes out) per kernel and shape"""
    rows = []
    # the six composite halos of one forward: (channels, H, W, k)
    for c, h, w, k in ((4, 88, 180, 2), (32, 44, 90, 1), (64, 22, 45, 1), (128, 44, 90, 1), (64, 88, 180, 2), (32, 88, 180, 2)):
        x = torch.randn((n, c, h, w), device='cuda')
        p = ops.make_pad(k, k, k, k, ops.PAD_ZERO, ops.PAD_WRAP)
        y = torch.empty((n, c, h + 2 * k, w + 2 * k), device='cuda')
        ms = timeit(lambda: ops.pad2d(x, p, out=y), iters)
        nbytes = 4.0 * (x.numel() + y.numel())
        rows.append({'kernel': 'pad2d_fwd', 'shape': [n, c, h, w], 'halo': k, 'ms': ms, 'gbs': nbytes / ms / 1e6})
        dy = torch.randn_like(y)
        ms = timeit(lambda: ops.pad2d_bwd(dy, (n, c, h, w), p), iters)
        rows.append({'kernel': 'pad2d_bwd', 'shape': [n, c, h, w], 'halo': k, 'ms': ms, 'gbs': nbytes / ms / 1e6})
    if pads_only:
        return rows
    for c, h, w in ((32, 88, 180), (64, 44, 90)):
        x = torch.randn((n, c, h, w), device='cuda')
        y = torch.empty((n, c, h // 2, w // 2), device='cuda')
        ms = timeit(lambda: ops.maxpool2(x, out=y), iters)
        rows.append({'kernel': 'maxpool2_fwd', 'shape': [n, c, h, w], 'ms': ms, 'gbs': 4.0 * (x.numel() + y.numel()) / ms / 1e6})
    for c, h, w in ((128, 22, 45), (64, 44, 90)):
        x = torch.randn((n, c, h, w), device='cuda')
        y = torch.empty((n, c, 2 * h, 2 * w), device='cuda')
        ms = timeit(lambda: ops.upsample2(x, out=y), iters)
        rows.append({'kernel': 'upsample2_fwd', 'shape': [n, c, h, w], 'ms': ms, 'gbs': 4.0 * (x.numel() + y.numel()) / ms / 1e6})
    s = torch.randn((28, n, 4, 88, 180), device='cuda')
    ms = timeit(lambda: ops.series_merge_time(s, 2), max(2, iters // 4))
    rows.append({'kernel': 'series_merge_time', 'shape': list(s.shape), 'ms': ms, 'gbs': 8.0 * s.numel() / ms / 1e6})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    rows = measure(a.batch, a.iters)
    for r in rows:
        print('%-18s %-24s halo %s : %8.3f ms  %7.1f GB/s  (%.0f%% of 8 TB/s)' %
              (r['kernel'], r['shape'], r.get('halo', '-'), r['ms'], r['gbs'], r['gbs'] / 80.0))
    if a.out:
        json.dump(rows, open(a.out, 'w'))


if __name__ == '__main__':
    main()
