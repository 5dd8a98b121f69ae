#!/usr/bin/env python3
"""Sweep every compiled MFMA tile configuration over the conv layers of a network and print per-layer timings.
GPU only.  Usage: python tools/tune_conv.py [--batch 64] [--grid 88x180] [--cin 4] [--iters 20]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_amd import ops  # noqa: E402


def unet_layers(cin, h, w):
    # (name, cin, cout, k, dil, src_mode, stored_h, stored_w)  -- Azure/train_tf.py:208-268 after fusion
    return [
        ('L1', cin, 32, 3, 2, 0, h, w),
        ('L2', 32, 64, 3, 1, 2, h, w),
        ('L3', 64, 128, 3, 1, 2, h // 2, w // 2),
        ('L4', 128, 64, 3, 1, 1, h // 4, w // 4),
        ('L5', 64, 32, 3, 2, 1, h // 2, w // 2),
        ('L6', 32, cin, 5, 1, 0, h, w),
        # the pooled layers as the default plan runs them: plain source written by dlwp_maxpool2_fwd
        ('L2p', 32, 64, 3, 1, 0, h // 2, w // 2),
        ('L3p', 64, 128, 3, 1, 0, h // 4, w // 4),
        # ... and as the inference plan runs them: MaxPooling2D(2) in the epilogue (name ends with 'o')
        ('L1o', cin, 32, 3, 2, 0, h, w),
        ('L2o', 32, 64, 3, 1, 0, h // 2, w // 2),
        # the decoder layers as both plans restate them on their low-resolution source
        ('L5r', 64, 32, 3, 1, 0, h // 2, w // 2),
        ('L6r', 32, 4 * cin, 3, 1, 0, h // 2, w // 2),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--cin', type=int, default=4)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--layers', default='')
    ap.add_argument('--out', default='')
    ap.add_argument('--cfg', type=int, default=-2, help='only time this configuration index (-1 = heuristic only)')
    a = ap.parse_args()
    h, w = (int(v) for v in a.grid.split('x'))
    cfgs = ops.conv_configs()
    rng = np.random.default_rng(0)
    results = {}
    for name, cin, cout, k, dil, src, sh, sw in unet_layers(a.cin, h, w):
        if a.layers and name not in a.layers.split(','):
            continue
        x = torch.from_numpy(rng.standard_normal((a.batch, cin, sh, sw)).astype(np.float32)).cuda()
        wt = torch.from_numpy((rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)).cuda()
        b = torch.zeros(cout, device='cuda')
        p = dil * (k - 1) // 2
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, ops.PAD_ZERO, ops.PAD_WRAP), ops.ACT_TANH,
                           src_mode=src, out_pool=name.endswith('o'))
        ys = ops.conv_out_shape(ops.Shape4(a.batch, cin, sh, sw), cd)
        out = torch.empty((a.batch, cout, ys.h, ys.w), device='cuda')
        flops = 2.0 * a.batch * ys.h * ys.w * cout * cin * k * k * (4 if name.endswith('o') else 1)
        rows = []
        for i, c in enumerate(cfgs):
            if (c[0], c[1]) != (k, dil) or (c[8] == 1) != (src == 2) or c[8] >= 2 or (c[6] < 0 and cout > 16 // (-c[6])):
                continue
            if a.cfg != -2 and i != a.cfg:
                continue
            ops.force_conv_config(i)
            try:
                for _ in range(5):
                    ops.conv2d(x, wt, b, cd, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv2d(x, wt, b, cd, out=out)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
            except Exception as ex:  # noqa: BLE001
                print('  cfg %d failed: %s' % (i, ex))
                continue
            rows.append((ms, i, c))
        ops.force_conv_config(-1)
        # heuristic choice
        for _ in range(3):
            ops.conv2d(x, wt, b, cd, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv2d(x, wt, b, cd, out=out)
        e1.record()
        torch.cuda.synchronize()
        hms = e0.elapsed_time(e1) / a.iters
        rows.sort()
        print('%s  %d->%d k%d d%d src%d out %dx%d  batch %d  %.1f MFLOP/sample   heuristic: %.3f ms = %.1f TF' %
              (name, cin, cout, k, dil, src, ys.h, ys.w, a.batch, flops / a.batch / 1e6, hms, flops / hms / 1e9))
        for ms, i, c in rows:
            print('   cfg %2d th=%2d tw=%2d waves=%d fa=%d bnf=%d ck=%2d pool=%d lds=%6d : %8.3f ms  %7.1f TF' %
                  (i, c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], ms, flops / ms / 1e9))
        results[name] = {'flops': flops, 'heuristic_ms': hms, 'rows': [(ms, i) + tuple(c) for ms, i, c in rows]}
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(results, f)


if __name__ == '__main__':
    main()
