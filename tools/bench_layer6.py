#!/usr/bin/env python3
"""The restated output layer of the config-2 U-Net alone (32 channels at 44 x 90 -> 4 fields x 4 phases, 3x3, stored depth-to-space
into 88 x 180): the general position-split Winograd instance against its streaming form (csrc/conv_fwd_wino2s.hip) --
time, executed matrix FLOP/s, algorithmic HBM bytes / s.  usage: python tools/bench_layer6.py [--members 256]"""
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
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--modes', default='0,1')
    a = ap.parse_args()
    from dlwp_amd import _lib, ops
    from oracle import np_ref
    rng = np.random.default_rng(0)
    w = torch.from_numpy(np_ref.glorot_uniform((3, 3, 32, 16), rng)).cuda()
    b = torch.zeros(16, device='cuda')
    cd = ops.make_conv(16, 3, 3, 1, ops.make_pad(1, 1, 1, 1, _lib.PAD_ZERO, _lib.PAD_WRAP), _lib.ACT_LINEAR, out_d2s=True)
    x = torch.randn((a.members, 32, 44, 90), device='cuda')
    y = torch.empty((a.members, 4, 88, 180), device='cuda')
    out = {}
    for mode in [int(m) for m in a.modes.split(',')]:
        prev = ops.set_few_stream(mode)
        try:
            prep = ops.conv2d_prepare(x, w, cd)
            for _ in range(3):
                ops.conv2d(x, w, b, cd, out=y, prepared=prep)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv2d(x, w, b, cd, out=y, prepared=prep)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            info = ops.conv_launch_info(tuple(x.shape), cd, None, 0)
        finally:
            ops.set_few_stream(prev)
        nb = 4.0 * (x.numel() + y.numel())
        out['few_stream_%d' % mode] = {'ms': round(ms, 4), 'executed_tflops': round(sum(i[3] for i in info) / ms / 1e9, 1),
                                       'mfma_frac': round(sum(i[3] for i in info) / ms / 1e9 / 157.3, 3),
                                       'gbs': round(nb / ms / 1e6, 1), 'config': info[0][0], 'grid': info[0][1]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
