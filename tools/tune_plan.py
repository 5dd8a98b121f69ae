#!/usr/bin/env python3
"""Sweep every compiled forward tile configuration over the convolution launches of a model's INFERENCE PLAN as the rollout
really issues them (restated decoder layers, pooling epilogues, interleaved phase stores), at a given member count: the
heuristic's pick (dlwp_conv2d_pick_config) against the best forced configuration per launch.
    python tools/tune_plan.py [--grid 88x180] [--channels 4] [--members 4] [--iters 30]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters):
    for _ in range(max(5, iters)):
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
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--channels', type=int, default=4)
    ap.add_argument('--members', type=int, default=4)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--recurrent-bf16', action='store_true',
                    help='BASELINE config 4: ConvLSTM2D front end + U-Net on (2, channels / 2, H, W), bfloat16 activation storage')
    a = ap.parse_args()
    from dlwp_amd import ops
    from dlwp_amd._lib import DlwpError
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import lstm_unet_layers, unet_layers
    h, w = (int(v) for v in a.grid.split('x'))
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=a.recurrent_bf16, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(lstm_unet_layers((2, a.channels // 2, h, w)) if a.recurrent_bf16 else unet_layers((a.channels, h, w)),
                  loss='mse', optimizer='adam')
    net = d.model
    if a.recurrent_bf16:
        net.set_activation_dtype('bfloat16')
    ex, plan = net.executor, net.infer_plan
    n = a.members
    x = torch.randn((n,) + plan._in_store, device=net.device)
    outs = ex.run(x)
    bufs = ex.scratch(n)

    def res(i):
        return bufs[i] if i >= 0 else (x if i == -1 else outs[-2 - i])
    cfgs = ops.conv_configs()
    report = []
    for op, desc in zip(plan.ops, ex._descriptors()):
        if op.kind != 'conv':
            continue
        kern, bias = ex.conv_weights(op)
        src, dst = res(op.src), res(op.dst)
        c16 = op.dst in ex._bf16 and op.src not in ex._bf16      # float32 state rounded by the loader (Executor.run)

        def fn():
            if op.lstm_f:
                za, cp, co = op.aux
                ops.convlstm_conv(src, kern, bias, desc, dst, res(co), z_add=res(za) if za is not None else None,
                                  c_prev=res(cp) if cp is not None else None, x_channels=op.xs[0], compute_bf16=c16)
            else:
                ops.conv2d(src, kern, bias, desc, out=dst, x_channels=op.xs[0], compute_bf16=c16)
        ops.force_conv_config(-1)
        base = timed(fn, a.iters)
        pick = ops.conv_launch_info((n,) + tuple(op.xs), desc, ex._conv_dtype(op), net.device.index or 0)
        rows = []
        for i in range(len(cfgs)):
            ops.force_conv_config(i)
            try:
                fn()
                torch.cuda.synchronize()
            except (DlwpError, RuntimeError):
                continue
            rows.append((timed(fn, a.iters), i))
        ops.force_conv_config(-1)
        rows.sort()
        best = rows[0] if rows else (base, -1)
        report.append({'layer': op.layer.name, 'xs': op.xs, 'heuristic_cfg': [p[0] for p in pick], 'heuristic_ms': round(base, 4),
                       'best_cfg': best[1], 'best_ms': round(best[0], 4), 'gain': round(base / best[0], 3),
                       'best_info': cfgs[best[1]] if best[1] >= 0 else None,
                       'top3': [(round(t, 4), i) for t, i in rows[:3]]})
    print(json.dumps({'grid': a.grid, 'channels': a.channels, 'members': n,
                      'sum_heuristic_ms': round(sum(r['heuristic_ms'] for r in report), 4),
                      'sum_best_ms': round(sum(r['best_ms'] for r in report), 4), 'launches': report}))


if __name__ == '__main__':
    main()
