#!/usr/bin/env python3
"""RowConnected2D (reference DLWP/custom.py:695-896; examples/train_functional.py:191-196) on one MI355X: forward at the
rollout's member count, data / weight gradient at the training batch, with the matrix-core work each launch EXECUTES
(padding and the packed-column expansion included) against the dense fp32 MFMA peak, and the plain Conv2D the layer
replaces timed beside it.
    python tools/bench_rowconv.py [--grid 88x180] [--fields 4] [--members 256] [--batch 64]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MFMA_F32_PEAK = 157.3     # TFLOP/s, MI355X_MICROARCH.md


def timed(fn, min_ms=60.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    reps = max(5, int(min_ms / max(e0.elapsed_time(e1), 1e-3)))
    for _ in range(reps):        # clocks up
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--fields', type=int, default=4)
    ap.add_argument('--cin', type=int, default=32)
    ap.add_argument('--members', type=int, default=256)
    ap.add_argument('--batch', type=int, default=64)
    a = ap.parse_args()
    from dlwp_amd import ops
    from dlwp_amd._lib import Shape4
    h, w = (int(v) for v in a.grid.split('x'))
    cin, cout = a.cin, a.fields
    cd = ops.make_conv(cout, 5, 5, 1, ops.make_pad(2, 2, 2, 2, ops.PAD_ZERO, ops.PAD_WRAP), ops.ACT_LINEAR)
    rng = np.random.default_rng(0)
    k = torch.from_numpy(rng.standard_normal((h, 5, 5, cin, cout)).astype(np.float32) * 0.05).cuda()
    b = torch.zeros((h, 1, cout), device='cuda')
    out = {'layer': 'RowConnected2D(%d, 5) on (%d, %d, %d) behind PeriodicPadding2D((0, 2)) + ZeroPadding2D((2, 0))' %
                    (cout, cin, h, w), 'rows': []}
    alg = lambda n: 2.0 * n * h * w * cout * cin * 25      # noqa: E731

    # executed matrix work: forward packs P = 16 / cout_p pixels per instruction row and widens the k axis to kw + P - 1
    cout_p = 16 if cout > 8 else 1 << max(0, int(np.ceil(np.log2(cout))))
    P = 16 // cout_p
    fwd_exec = lambda n: 2.0 * n * h * 16 * int(np.ceil(w / (16.0 * P))) * 16 * (5 * (5 + P - 1) * cin) * \
        int(np.ceil(cout / 16.0))      # noqa: E731
    n = a.members
    x = torch.randn((n, cin, h, w), device='cuda')
    y = torch.empty((n, cout, h, w), device='cuda')
    ms = timed(lambda: ops.rowconv2d(x, k, b, cd, out=y))
    out['rows'].append({'pass': 'forward', 'samples': n, 'ms': round(ms, 4), 'algorithmic_tflops': round(alg(n) / ms / 1e9, 1),
                        'executed_tflops': round(fwd_exec(n) / ms / 1e9, 1),
                        'mfma_frac': round(fwd_exec(n) / ms / 1e9 / MFMA_F32_PEAK, 3),
                        'gbs': round((x.numel() + y.numel()) * 4 / ms / 1e6, 1)})
    # the shared-filter Conv2D the layer replaces (the model's own last layer), as the rollout runs it
    kc = torch.from_numpy(rng.standard_normal((5, 5, cin, cout)).astype(np.float32) * 0.05).cuda()
    bc = torch.zeros(cout, device='cuda')
    ms_c = timed(lambda: ops.conv2d(x, kc, bc, cd, out=y))
    out['rows'].append({'pass': 'forward, Conv2D with shared filters (same geometry)', 'samples': n, 'ms': round(ms_c, 4),
                        'algorithmic_tflops': round(alg(n) / ms_c / 1e9, 1)})
    n = a.batch
    x = torch.randn((n, cin, h, w), device='cuda')
    dz = torch.randn((n, cout, h, w), device='cuda')
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(k), torch.empty_like(b)
    xs = Shape4(n, cin, h, w)
    ms = timed(lambda: ops.rowconv2d_bwd_data(dz, k, cd, xs, dx))
    co4 = (cout + 3) // 4 * 4
    # (periodic columns + zero rows: the kernel computes the stored h x w grid directly, dz wraps while it is staged)
    ex = 2.0 * n * h * int(np.ceil(w / 16.0)) * 16 * int(np.ceil(cin / 16.0)) * 16 * 25 * co4
    out['rows'].append({'pass': 'data gradient (halo adjoint included)', 'samples': n, 'ms': round(ms, 4),
                        'algorithmic_tflops': round(alg(n) / ms / 1e9, 1), 'executed_tflops': round(ex / ms / 1e9, 1),
                        'mfma_frac': round(ex / ms / 1e9 / MFMA_F32_PEAK, 3)})
    ms = timed(lambda: ops.rowconv2d_bwd_weight(x, dz, dw, db, cd, xs))
    ex = 2.0 * h * 5 * (5 * int(np.ceil(cin / 16.0)) * 16) * 16 * int(np.ceil(cout / 16.0)) * n * int(np.ceil(w / 4.0)) * 4
    out['rows'].append({'pass': 'weight + bias gradient', 'samples': n, 'ms': round(ms, 4),
                        'algorithmic_tflops': round(alg(n) / ms / 1e9, 1), 'executed_tflops': round(ex / ms / 1e9, 1),
                        'mfma_frac': round(ex / ms / 1e9 / MFMA_F32_PEAK, 3)})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
