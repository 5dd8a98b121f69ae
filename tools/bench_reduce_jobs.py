#!/usr/bin/env python3
"""The batched final sums of a training step (csrc/batch.hip: reduce_jobs_kernel) alone: the six weight gradients of the config-2
U-Net are computed with their final sums deferred, then the ONE flush launch is timed; beside it the per-layer route (weight
gradient kernel + its own reduce_slabs launches) minus the deferred kernels.  usage: python tools/bench_reduce_jobs.py [--batch 64]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--fill', type=int, default=0, help='DLWP_OPT_WGRAD_FILL (0: the default)')
    a = ap.parse_args()
    from dlwp_amd import _lib, ops
    dev = torch.device('cuda', 0)
    if a.fill:
        _lib.set_option(_lib.OPT_WGRAD_FILL, a.fill)
    n = a.batch
    layers = [(4, 32, 2, 88, 180), (32, 64, 1, 44, 90), (64, 128, 1, 22, 45), (128, 64, 1, 44, 90), (64, 32, 2, 88, 180), (32, 16, 1, 88, 180)]
    items = []
    for j, (cin, cout, dil, h, w) in enumerate(layers):
        cd = ops.make_conv(cout, 3, 3, dil, ops.make_pad(dil, dil, dil, dil, 0, 1), 0)
        xs = _lib.Shape4(n, cin, h, w)
        x = torch.randn((n, cin, h, w), device=dev)
        dz = torch.randn((n, cout, h, w), device=dev)
        dw = torch.empty((3, 3, cin, cout), device=dev)
        need = ops.conv_bwd_workspace_bytes(0, xs, cd, 1)
        items.append((j, cd, xs, x, dz, dw, need))
    print('slab bytes per layer (MB):', [round(it[6] / 1e6, 1) for it in items], 'total', round(sum(it[6] for it in items) / 1e6, 1))

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def immediate():
        for j, cd, xs, x, dz, dw, _ in items:
            ops.conv2d_bwd_weight(x, dz, dw, cd, xs, ws_key=('r', j))

    def deferred(flush=True):
        ops.reductions_begin(dev)
        for j, cd, xs, x, dz, dw, _ in items:
            ops.conv2d_bwd_weight(x, dz, dw, cd, xs, ws_key=('r', j))
        if flush:
            ops.reductions_flush(dev)
        else:
            _lib.lib.dlwp_reductions_begin(_lib.handle(0))      # drop the recorded jobs
            _lib.lib.dlwp_reductions_flush(_lib.handle(0), None)
    t_imm, t_def, t_none = timed(immediate), timed(deferred), timed(lambda: deferred(False))
    print('six weight gradients, ms: own reduce launches %.4f | one batched flush %.4f | no final sums %.4f' % (t_imm, t_def, t_none))
    print('final sums alone, ms: per-layer launches %.4f | batched %.4f' % (t_imm - t_none, t_def - t_none))


if __name__ == '__main__':
    main()
