#!/usr/bin/env python3
"""Split-K sweep (DLWP_OPT_SPLITK, csrc/conv_fwd_k3d1s.hip): every Winograd launch of a model's INFERENCE PLAN at a given member
count, timed unsplit, under the library's rule, and with every forced split count -- on the heuristic's tile configuration and on
every other 32 / 64-channel Winograd instance that has a split variant.  The rule of conv_fwd.hip:plan_splitk is fitted to this.
    python tools/sweep_splitk.py [--grid 88x180] [--channels 4] [--members 8] [--iters 200]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--channels', type=int, default=4)
    ap.add_argument('--members', type=int, default=8)
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--splits', default='2,3,4,6,8,16')
    ap.add_argument('--all-configs', action='store_true', help='also force every other Winograd instance with a split variant')
    a = ap.parse_args()
    from dlwp_amd import ops
    from dlwp_amd._lib import DlwpError
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    h, w = (int(v) for v in a.grid.split('x'))
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((a.channels, h, w)), loss='mse', optimizer='adam')
    net = d.model
    ex, plan = net.executor, net.infer_plan
    n = a.members
    x = torch.randn((n,) + plan._in_store, device=net.device)
    ops.set_splitk(0)
    outs = ex.run(x)
    bufs = ex.scratch(n)

    def res(i):
        return bufs[i] if i >= 0 else (x if i == -1 else outs[-2 - i])
    cfgs = ops.conv_configs()
    splits = [int(v) for v in a.splits.split(',')]
    report = []
    for op, desc in zip(plan.ops, ex._descriptors()):
        if op.kind != 'conv':
            continue
        kern, bias = ex.conv_weights(op)
        src, dst = res(op.src), res(op.dst)
        xs = (n,) + tuple(op.xs)
        prepared = ops.conv2d_prepare(src, kern, desc, x_channels=op.xs[0])      # as the rollout graph: prepared once

        def fn():
            ops.conv2d(src, kern, bias, desc, out=dst, x_channels=op.xs[0], prepared=prepared)
        ops.set_splitk(0)
        ops.force_conv_config(-1)
        pick = ops.conv_launch_info(xs, desc)
        base = timed(fn, a.iters)
        row = {'layer': op.layer.name, 'xs': op.xs, 'cout': desc.cout, 'cfg': pick[0][0], 'grid': pick[0][1], 'unsplit_us': round(1e3 * base, 2)}
        ops.set_splitk(1)
        row['rule_S'] = ops.conv_split_count(xs, desc)
        row['rule_us'] = round(1e3 * timed(fn, a.iters), 2) if row['rule_S'] > 1 else row['unsplit_us']
        forced = {}
        if pick[0][0] >= 0 and cfgs[pick[0][0]][10] & 32:
            for s in splits:
                ops.set_splitk(s)
                eff = ops.conv_split_count(xs, desc)
                if eff < 2 or str(eff) in forced:
                    continue
                forced[str(eff)] = round(1e3 * timed(fn, a.iters), 2)
        row['forced_us'] = forced
        if a.all_configs:
            other = {}
            for i, c in enumerate(cfgs):
                if not (c[5] == 0 and c[10] & 32) or i == pick[0][0]:
                    continue
                ops.force_conv_config(i)
                ops.set_splitk(0)
                try:
                    prep_i = ops.conv2d_prepare(src, kern, desc, x_channels=op.xs[0])
                    fi = (lambda p=prep_i: ops.conv2d(src, kern, bias, desc, out=dst, x_channels=op.xs[0], prepared=p))
                    fi()
                    torch.cuda.synchronize()
                except (DlwpError, RuntimeError):
                    continue
                ent = {'1': round(1e3 * timed(fi, a.iters), 2)}
                for s in splits:
                    ops.set_splitk(s)
                    eff = ops.conv_split_count(xs, desc)
                    if eff < 2 or str(eff) in ent:
                        continue
                    ent[str(eff)] = round(1e3 * timed(fi, a.iters), 2)
                other['%d:%r' % (i, c[2:7])] = ent
            ops.force_conv_config(-1)
            row['other_configs_us'] = other
        ops.set_splitk(1)
        best = min([row['unsplit_us']] + list(forced.values()))
        row['best_us'] = best
        report.append(row)
    print(json.dumps({'grid': a.grid, 'channels': a.channels, 'members': n,
                      'sum_unsplit_us': round(sum(r['unsplit_us'] for r in report), 2),
                      'sum_rule_us': round(sum(r['rule_us'] for r in report), 2),
                      'sum_best_us': round(sum(r['best_us'] for r in report), 2), 'launches': report}))


if __name__ == '__main__':
    main()
