#!/usr/bin/env python3
"""Config 4 (1-degree recurrent stack, bfloat16 between the layers, octet layout): every compiled bf16 instance that can run
each convolution launch of the inference plan, forced and timed (HIP events on the launch stream), next to the heuristic's
pick.  usage: python tools/tune_cfg4_octets.py [--members 8]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=8)
    ap.add_argument('--iters', type=int, default=30)
    a = ap.parse_args()
    from dlwp_amd import ops
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import lstm_unet_layers
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(lstm_unet_layers((2, 6, 180, 360)), loss='mse', optimizer='adam')
    net = d.model
    net.set_activation_dtype('bfloat16')
    ex, plan = net.executor, net.infer_plan
    x = torch.randn((a.members,) + plan._in_store, device=net.device)
    outs = ex.run(x)
    bufs = ex.scratch(a.members)

    def res(i):
        return bufs[i] if i >= 0 else (x if i == -1 else outs[-2 - i])
    cfgs = ops.conv_configs()

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters
    report = []
    for op, desc in zip(plan.ops, ex._descriptors()):
        if op.kind != 'conv':
            continue
        src, dst = res(op.src), res(op.dst)
        kern, bias = ex.conv_weights(op)
        c16 = op.dst in ex._bf16 and op.src not in ex._bf16
        in8, sw = op.src in ex._oct, op.dst in ex._oct
        if op.src2 is not None:
            continue                                  # whole-step launches: two instances only (forced through the same option)
        if op.lstm_f:
            za, cp, co = op.aux
            fn = lambda: ops.convlstm_conv(src, kern, bias, desc, dst, res(co), z_add=res(za) if za is not None else None,  # noqa: E731
                                           c_prev=res(cp) if cp is not None else None, x_channels=op.xs[0], compute_bf16=c16,
                                           in_o8=in8, out_o8=sw)
        else:
            fn = lambda: ops.conv2d(src, kern, bias, desc, out=dst, x_channels=op.xs[0], compute_bf16=c16, in_o8=in8, out_o8=sw)  # noqa: E731
        base = timed(fn)
        _, (kh, kw), dil = op.conv_geometry
        rows = []
        for i, c in enumerate(cfgs):
            ks, cdil, th, tw, waves, fa, bnf, ck, pool, lds, flags = c
            if pool < 2 or ks != kh or cdil != dil[0]:
                continue
            if bool(flags & 8) != in8 or bool(flags & 16) != sw or bool(flags & 2) != bool(op.lstm_f) or (pool == 3) != (src.dtype == torch.float32):
                continue
            ops.force_conv_config(i)
            try:
                rows.append((round(timed(fn), 4), i, (th, tw, waves, fa, bnf, ck)))
            except Exception as e:  # noqa: BLE001
                rows.append((None, i, str(e)[-60:]))
            finally:
                ops.force_conv_config(-1)
        report.append({'layer': op.layer.name, 'cin': op.xs[0], 'k': kh, 'dil': dil[0], 'lstm': bool(op.lstm_f), 'heuristic_ms': round(base, 4),
                       'forced': sorted([r for r in rows if r[0] is not None])})
        print(report[-1])
    print(json.dumps(report))


if __name__ == '__main__':
    main()
