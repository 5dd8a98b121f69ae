#!/usr/bin/env python3
"""Time the bf16 matrix-core convolution family (conv_fwd_bf16_kernel.h) against the fp32 families on the layers of
BASELINE.json config 4 (1 deg grid, bf16 activation storage).  GPU only.
Usage: python tools/bench_bf16_conv.py [--batch 8] [--grid 180x360] [--iters 20] [--out file.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_amd import ops  # noqa: E402


def layers(h, w):
    # (name, cin, cout, k, dil, src_mode, stored_h, stored_w, zero_cols)
    return [
        ('lstm_in 6->96 d2 f32in', 6, 96, 3, 2, 0, h, w, False),
        ('lstm_rec 24->96', 24, 96, 3, 1, 0, h, w, True),
        ('conv2d_1 48->32 d2', 48, 32, 3, 2, 0, h, w, False),
        ('conv2d_2 32->64', 32, 64, 3, 1, 0, h // 2, w // 2, False),
        ('conv2d_3 64->128', 64, 128, 3, 1, 0, h // 4, w // 4, False),
        ('conv2d_4 128->64 up', 128, 64, 3, 1, 1, h // 4, w // 4, False),
        ('conv2d_5 64->32 up d2', 64, 32, 3, 2, 1, h // 2, w // 2, False),
        ('conv2d_6 32->12 5x5', 32, 12, 5, 1, 0, h, w, False),
    ]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--grid', default='180x360')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    h, w = (int(v) for v in a.grid.split('x'))
    cfgs = ops.conv_configs()
    rng = np.random.default_rng(0)
    res = {}
    for name, cin, cout, k, dil, src, sh, sw, zc in layers(h, w):
        f32in = 'f32in' in name
        x = torch.from_numpy(rng.standard_normal((a.batch, cin, sh, sw)).astype(np.float32)).cuda()
        if not f32in:
            x = x.to(torch.bfloat16)
        wt = torch.from_numpy((rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)).cuda()
        b = torch.zeros(cout, device='cuda')
        p = dil * (k - 1) // 2
        cd = ops.make_conv(cout, k, k, dil, ops.make_pad(p, p, p, p, ops.PAD_ZERO, ops.PAD_ZERO if zc else ops.PAD_WRAP),
                           ops.ACT_TANH, src_mode=src)
        ys = ops.conv_out_shape(ops.Shape4(a.batch, cin, sh, sw), cd)
        out = torch.empty((a.batch, cout, ys.h, ys.w), device='cuda', dtype=torch.bfloat16)
        flops = 2.0 * a.batch * ys.h * ys.w * cout * cin * k * k
        byts = 2.0 * (x.numel() + out.numel())
        row = {}
        prev = ops.set_bf16_mfma(False)
        row['fp32 families'] = timed(lambda: ops.conv2d(x, wt, b, cd, out=out, compute_bf16=f32in), a.iters)
        ops.set_bf16_mfma(True)
        row['bf16 heuristic'] = timed(lambda: ops.conv2d(x, wt, b, cd, out=out, compute_bf16=f32in), a.iters)
        for i, c in enumerate(cfgs):
            if c[8] != (3 if f32in else 2) or (c[0], c[1]) != (k, dil) or c[10] & 2:     # (bit 1: cell-update instances)
                continue
            ops.force_conv_config(i)
            try:
                row['cfg%d %r' % (i, c[2:8])] = timed(lambda: ops.conv2d(x, wt, b, cd, out=out, compute_bf16=f32in), a.iters)
            except Exception as e:      # noqa: BLE001
                row['cfg%d' % i] = str(e)
            finally:
                ops.force_conv_config(-1)
        ops.set_bf16_mfma(prev)
        res[name] = {'ms': row, 'gflop': flops / 1e9, 'io_mb': byts / 1e6}
        best = min(v for v in row.values() if isinstance(v, float))
        print('%-24s %7.2f GFLOP %6.1f MB io | ' % (name, flops / 1e9, byts / 1e6) +
              ' | '.join('%s %.4f' % (kk, v) if isinstance(v, float) else '%s ERR' % kk for kk, v in row.items()))
        print('    best %.4f ms = %.1f TFLOP/s, %.0f GB/s of in+out' % (best, flops / best / 1e9, byts / best / 1e6))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
