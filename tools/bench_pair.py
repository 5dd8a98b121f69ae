#!/usr/bin/env python3
"""A layer's weight gradient + data gradient at 8 samples (the layers of the 88 x 180 U-Net): two launches back to back against ONE
launch (dlwp_pair_begin / dlwp_pair_end, csrc/conv_pair.hip).  usage: python tools/bench_pair.py [--batch 8]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    from dlwp_amd import _lib, ops
    from oracle import np_ref
    rng = np.random.default_rng(0)
    device = torch.device('cuda', torch.cuda.current_device())
    n = a.batch
    out = {}
    for name, cin, cout, h, w, sm, stored in (('layer6_restated', 32, 16, 44, 90, 0, False), ('layer5_restated', 64, 32, 44, 90, 0, False),
                                              ('layer4', 128, 64, 22, 45, 1, True), ('layer3', 64, 128, 22, 45, 0, False),
                                              ('layer2', 32, 64, 44, 90, 0, False)):
        cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 0, 1), ops.ACT_LINEAR, src_mode=sm)
        xs = _lib.Shape4(n, cin, h, w)
        ys = ops.conv_out_shape(xs, cd)
        wt = torch.from_numpy(np_ref.glorot_uniform((3, 3, cin, cout), rng)).cuda()
        x = torch.randn((n, cin, h, w), device='cuda')
        dz = torch.randn((n, cout, ys.h, ys.w), device='cuda')
        prep = ops.conv2d_bwd_data_prepare(wt, cd, xs, stored=stored)
        dw = torch.empty((3, 3, cin, cout), device='cuda')
        dx = torch.empty((n, cin, h, w) if stored or sm == 0 else (n, cin, 2 * h, 2 * w), device='cuda')
        rec = {}
        for paired in (False, True):
            def step():
                if paired:
                    ops.pair_begin(device)
                ops.conv2d_bwd_weight(x, dz, dw, cd, xs, ws_key=('bench-pair', name))
                ops.conv2d_bwd_data(dz, wt, cd, xs, dx, prepared=prep, stored=stored)
                if paired:
                    ops.pair_end(device)
            # the launches captured in a graph, as the training step replays them: no host gaps in the measurement
            # (each step = the pair + the weight gradient's slab sum, in both variants)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(10):
                        step()
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            rec['pair_us' if paired else 'separate_us'] = round(1e3 * e0.elapsed_time(e1) / a.iters / 10, 2)
        before = ops.pair_fused_count(device)
        ops.pair_begin(device)
        ops.conv2d_bwd_weight(x, dz, dw, cd, xs, ws_key=('bench-pair', name))
        ops.conv2d_bwd_data(dz, wt, cd, xs, dx, prepared=prep, stored=stored)
        ops.pair_end(device)
        rec['fused'] = ops.pair_fused_count(device) - before
        out[name] = rec
    print(json.dumps(out))


if __name__ == '__main__':
    main()
