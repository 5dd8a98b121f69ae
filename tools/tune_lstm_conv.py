#!/usr/bin/env python3
"""Time every compiled instance of the fused ConvLSTM2D step convolution (dlwp_convlstm_conv_fwd) on config-4 shapes.
GPU only.  Usage: python tools/tune_lstm_conv.py [--members 8] [--grid 180x360] [--filters 24]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=8)
    ap.add_argument('--grid', default='180x360')
    ap.add_argument('--filters', type=int, default=24)
    a = ap.parse_args()
    h, w = (int(v) for v in a.grid.split('x'))
    n, f = a.members, a.filters
    cfgs = ops.conv_configs()
    for first in (True, False):
        cin, dil, halo = (6, 2, (2, 2, 2, 2, 0, 1)) if first else (f, 1, (1, 1, 1, 1, 0, 0))
        x = torch.randn(n, cin, h, w, device='cuda')
        if not first:
            x = x.to(torch.bfloat16)
        wt = torch.randn(3, 3, cin, 4 * f, device='cuda') * 0.05
        b = torch.zeros(4 * f, device='cuda')
        cd = ops.make_conv(4 * f, 3, 3, dil, ops.make_pad(*halo), ops.ACT_TANH, out_c_off=0, out_c_total=2 * f, lstm_f=f)
        hb = torch.zeros(n, 2 * f, h, w, device='cuda', dtype=torch.bfloat16)
        c = torch.empty(n, f, h, w, device='cuda')
        za = None if first else torch.randn(n, 4 * f, h, w, device='cuda').to(torch.bfloat16)
        cp = None if first else torch.randn(n, f, h, w, device='cuda')
        call = lambda: ops.convlstm_conv(x, wt, b, cd, hb, c, z_add=za, c_prev=cp, compute_bf16=first)  # noqa: E731
        for i in [-1] + list(range(len(cfgs))):
            ops.force_conv_config(i)
            try:
                for _ in range(3):
                    call()
            except Exception:  # noqa: BLE001  (instance does not match the layer)
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            print('input conv + cell update' if first else 'recurrent conv + cell update', 'heuristic' if i < 0 else 'cfg %d %r' % (i, cfgs[i][:9]),
                  '%.4f ms' % (e0.elapsed_time(e1) / 10))
        ops.force_conv_config(-1)


if __name__ == '__main__':
    main()
