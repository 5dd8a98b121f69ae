#!/usr/bin/env python3
"""Sweep the compiled weight-gradient tile configurations over the U-Net layers.  GPU only."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_amd import _lib, ops  # noqa: E402
from tools.tune_conv import unet_layers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--cin', type=int, default=4)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--layers', default='')
    a = ap.parse_args()
    h, w = (int(v) for v in a.grid.split('x'))
    cfgs = ops.wgrad_configs()
    rng = np.random.default_rng(0)
    for name, cin, cout, k, dil, src, sh, sw in unet_layers(a.cin, h, w):
        if a.layers and name not in a.layers.split(','):
            continue
        x = torch.from_numpy(rng.standard_normal((a.batch, cin, sh, sw)).astype(np.float32)).cuda()
        p = dil * (k - 1) // 2
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, ops.PAD_ZERO, ops.PAD_WRAP), ops.ACT_TANH, src_mode=src)
        xs = _lib.Shape4(a.batch, cin, sh, sw)
        ys = ops.conv_out_shape(xs, cd)
        dz = torch.from_numpy(rng.standard_normal((a.batch, cout, ys.h, ys.w)).astype(np.float32)).cuda()
        dw = torch.empty((k, k, cin, cout), device='cuda')
        flops = 2.0 * a.batch * ys.h * ys.w * cout * cin * k * k
        rows = []
        for i, c in [(-1, None)] + list(enumerate(cfgs)):
            if c is not None and ((c[0], c[1]) != (k, dil) or (c[4] < 0 and cout > -c[4])):
                continue
            ops.force_wgrad_config(i)
            try:
                for _ in range(2):
                    ops.conv2d_bwd_weight(x, dz, dw, cd, xs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv2d_bwd_weight(x, dz, dw, cd, xs)
                e1.record()
                torch.cuda.synchronize()
                rows.append((e0.elapsed_time(e1) / a.iters, i, c))
            except Exception as ex:  # noqa: BLE001
                print('   cfg %d failed: %s' % (i, ex))
        ops.force_wgrad_config(-1)
        rows.sort(key=lambda r: r[0])
        print('%s wgrad %d->%d k%d d%d src%d out %dx%d batch %d  (%.1f GFLOP)' % (name, cin, cout, k, dil, src, ys.h, ys.w,
                                                                             a.batch, flops / 1e9))
        for ms, i, c in rows[:14]:
            tag = ('heuristic -> cfg %d' % _lib.lib.dlwp_conv2d_wgrad_pick_config(_lib.handle(0), xs, cd)) if c is None else 'th=%d tw=%d nt=%d waves=%d lds=%d' % (c[2], c[3], c[4], c[5], c[6])
            print('   cfg %3d %-40s : %7.3f ms  %6.1f TF' % (i, tag, ms, flops / ms / 1e9))


if __name__ == '__main__':
    main()
