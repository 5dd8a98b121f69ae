#!/usr/bin/env python3
"""Layer 1 of the config-2 U-Net alone (4 -> 32, 3x3 dilation 2, periodic + zero halo, tanh): with the MaxPooling2D(2) epilogue
of the product's inference plan on the closed 88 x 180 grid, and unpooled at the nominal 91 x 180 -- time, executed matrix
FLOP/s, algorithmic HBM bytes / s.  usage: python tools/bench_layer1.py [--members 256]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    from dlwp_amd import _lib, ops
    from oracle import np_ref
    rng = np.random.default_rng(0)
    w = torch.from_numpy(np_ref.glorot_uniform((3, 3, 4, 32), rng)).cuda()
    b = torch.zeros(32, device='cuda')
    out = {}
    for name, hw, pool in (('pooled_88x180', (88, 180), True), ('unpooled_91x180', (91, 180), False)):
        cd = ops.make_conv(32, 3, 3, 2, ops.make_pad(2, 2, 2, 2, _lib.PAD_ZERO, _lib.PAD_WRAP), _lib.ACT_TANH, out_pool=pool)
        x = torch.randn((a.members, 4) + hw, device='cuda')
        oh, ow = (hw[0] // 2, hw[1] // 2) if pool else hw
        y = torch.empty((a.members, 32, oh, ow), device='cuda')
        for _ in range(3):
            ops.conv2d(x, w, b, cd, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv2d(x, w, b, cd, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        info = ops.conv_launch_info((a.members, 4) + hw, cd, None, 0)
        nb = 4.0 * (x.numel() + y.numel())
        out[name] = {'ms': round(ms, 4), 'executed_tflops': round(sum(i[3] for i in info) / ms / 1e9, 1), 'gbs': round(nb / ms / 1e6, 1),
                     'hbm_frac': round(nb / ms / 1e6 / 8000.0, 3), 'config': info[0][0]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
