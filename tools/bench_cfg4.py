#!/usr/bin/env python3
"""BASELINE.json configs[3]: 1-degree grid (180x360 after the pole crop), 6 variables x 2 input steps, the recurrent stack of
examples/train.py (ConvLSTM2D front end + U-Net), bfloat16 activation storage, 1 GPU.  Reports the rollout rate and, per
launch class, the time and the fraction of the roofline that bounds it: HBM for the element-wise kernels (ConvLSTM gate
update, pooling), MFMA for the convolutions.  (There is no separate padding kernel to time: every halo is fused into a
convolution's loader.)
    python tools/bench_cfg4.py [--members 8] [--forwards 4] [--activation-dtype bfloat16]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBS, MFMA_F32_PEAK, MFMA_BF16_PEAK = 8000.0, 157.3, 2500.0     # MI355X_MICROARCH.md, dense


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=8)
    ap.add_argument('--forwards', type=int, default=4)
    ap.add_argument('--grid', default='180x360')
    ap.add_argument('--variables', type=int, default=6)
    ap.add_argument('--activation-dtype', default='bfloat16', choices=['float32', 'bfloat16'])
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--no-bf16-mfma', action='store_true', help='keep bf16-stored layers on the fp32 kernel families')
    a = ap.parse_args()
    from dlwp_amd import ops
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import lstm_unet_layers
    if a.no_bf16_mfma:
        ops.set_bf16_mfma(False)
    h, w = (int(v) for v in a.grid.split('x'))
    cs = (2, a.variables, h, w)
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(lstm_unet_layers(cs), loss='mse', optimizer='adam')
    net = d.model
    net.set_activation_dtype(a.activation_dtype)
    dev = net.device
    x = torch.randn((a.members,) + net.infer_plan._in_store, device=dev)
    for _ in range(1 + a.iters):      # capture + >= 50 ms of work: after host-side model building the clocks are down
        series = net.rollout_on_device(x, a.forwards)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        series = net.rollout_on_device(x, a.forwards)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    finite = bool(torch.isfinite(series[-1]).all().item())

    # per-launch timing, HIP events on the launch stream
    ex, plan = net.executor, net.infer_plan
    outs = ex.run(x)
    bufs = ex.scratch(a.members)
    xin = x.reshape((a.members,) + plan._in_store)

    def res(i):
        return bufs[i] if i >= 0 else (xin if i == -1 else outs[-2 - i])
    rows = []
    for op, desc in zip(plan.ops, ex._descriptors()):
        src, dst = res(op.src), res(op.dst)
        if op.kind == 'conv' and op.src2 is not None:                # a whole ConvLSTM2D step in one launch
            kern, bias = ex.conv_weights(op)
            za, cp, co = op.aux
            il = op.src2['layer']
            d2 = ex._descriptor2(op)
            fn = lambda: ops.convlstm_step(dst, res(op.src2['buf']), kern, il.kernel, il.bias, desc, d2, res(cp), res(co),  # noqa: E731
                                           op.src2['xs'][0])
        elif op.kind == 'conv' and op.lstm_f:                        # cell update in the convolution's epilogue
            c16 = op.dst in ex._bf16 and op.src not in ex._bf16
            kern, bias = ex.conv_weights(op)
            za, cp, co = op.aux
            fn = lambda: ops.convlstm_conv(src, kern, bias, desc, dst, res(co), z_add=res(za) if za is not None else None,  # noqa: E731
                                           c_prev=res(cp) if cp is not None else None, x_channels=op.xs[0], compute_bf16=c16,
                                           in_o8=op.src in ex._oct, out_o8=op.dst in ex._oct)
        elif op.kind == 'conv':
            c16 = op.dst in ex._bf16 and op.src not in ex._bf16      # as Executor.run: float32 state rounded by the loader
            kern, bias = ex.conv_weights(op)
            fn = lambda: ops.conv2d(src, kern, bias, desc, out=dst, x_channels=op.xs[0],  # noqa: E731
                                    compute_bf16=c16, in_o8=op.src in ex._oct, out_o8=op.dst in ex._oct)
        elif op.kind == 'lstm':
            zh, cp, co = op.aux
            fn = lambda: ops.convlstm_gates(src, res(zh) if zh is not None else None, res(cp) if cp is not None else None,  # noqa: E731
                                            res(co), dst, op.xs[0], h_c_off=op.out_c_off, act=op.act, rec_act=op.rec_act)
        elif op.kind == 'maxpool':
            fn = lambda: ops.maxpool2(src, out=dst)  # noqa: E731
        elif op.kind == 'copy':
            fn = lambda: ops.copy_channels(src, dst, op.xs[0], op.in_c_off, op.out_c_off)  # noqa: E731
        else:
            continue
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        row = {'op': op.kind, 'ms': round(ms, 4)}
        if op.kind == 'conv':
            kh, kw = op.layer.kernel_size
            co_, ho, wo = getattr(op, 'conv_out_shape', None) or op.out_shape
            fl = 2.0 * ho * wo * co_ * op.xs[0] * kh * kw * a.members
            if op.alg_flops is not None:       # decoder layer restated on its low-resolution source: algorithmic FLOPs
                fl = float(op.alg_flops) * a.members
            on16 = any(op.layer is l16 for l16 in ex.bf16_weight_layers(a.members))
            # executed = what the matrix cores issue (padding included; Winograd's 16 / 9 multiplies per 2x2 outputs)
            info = ops.conv_launch_info((a.members,) + tuple(op.xs), desc, ex._conv_dtype(op), dev.index or 0)
            ex_fl = sum(i[3] for i in info)
            peak = MFMA_BF16_PEAK if on16 else MFMA_F32_PEAK
            nb = float(src.element_size()) * a.members * op.xs[0] * op.xs[1] * op.xs[2] + \
                float(dst.element_size()) * a.members * int(np.prod(op.out_shape))
            if op.lstm_f:      # + the stored pre-activations it reads (bf16), c_prev in, c out (float32)
                za, cp, co = op.aux
                hw_ = op.out_shape[1] * op.out_shape[2]
                nb += a.members * hw_ * op.lstm_f * ((8.0 if za is not None else 0.0) + (4.0 if cp is not None else 0.0) + 4.0)
                row['cell_update'] = ('whole step in one launch (dlwp_convlstm_step_fwd)' if op.src2 is not None else
                                      'in the epilogue (dlwp_convlstm_conv_fwd)')
                if op.src2 is not None:      # + the float32 state channels the input convolution reads
                    nb += 4.0 * a.members * op.src2['xs'][0] * hw_
                co_ = 4 * op.lstm_f
                fl = 2.0 * ho * wo * co_ * op.xs[0] * kh * kw * a.members
            row['storage'] = '%s -> %s' % ('octets' if op.src in ex._oct else str(src.dtype).replace('torch.', ''),
                                           'octets' if op.dst in ex._oct else str(dst.dtype).replace('torch.', ''))
            row.update(layer=op.layer.name, cin=op.xs[0], cout=co_, k=kh, algorithmic_tflops=round(fl / ms / 1e9, 1),
                       executed_tflops=round(ex_fl / ms / 1e9, 1), family='bf16 mfma' if on16 else 'fp32 mfma',
                       frac_of_matrix_peak=round(ex_fl / ms / 1e9 / peak, 3), matrix_peak_tflops=peak,
                       gbs=round(nb / ms / 1e6, 1), frac_of_hbm_peak=round(nb / ms / 1e6 / HBM_PEAK_GBS, 3), bound='mfma')
        else:
            f = op.xs[0]
            hw = op.xs[1] * op.xs[2]
            if op.kind == 'lstm':
                zh, cp, co = op.aux
                zsz, hsz = src.element_size(), dst.element_size()      # zx / zh and h storage; the cell state is float32
                nb = a.members * hw * f * ((8 if zh is not None else 4) * zsz + (1 if cp is not None else 0) * 4.0 +
                                           4.0 + hsz)
            else:
                nb = float(src.numel() * src.element_size() + dst.numel() * dst.element_size()) if op.kind == 'maxpool' \
                    else 2.0 * a.members * f * hw * 4.0
            row.update(gbs=round(nb / ms / 1e6, 1), frac_of_hbm_peak=round(nb / ms / 1e6 / HBM_PEAK_GBS, 3), bound='hbm')
        rows.append(row)
    t_conv = sum(r['ms'] for r in rows if r['bound'] == 'mfma')
    t_hbm = sum(r['ms'] for r in rows if r['bound'] == 'hbm')
    out = {'config': 'cfg4: %dx%d, %d variables x 2 steps, ConvLSTM2D front end + U-Net (%d params), %s activation storage, '
                     '%s, %d members, %d-forward rollout' % (
               h, w, a.variables, net.count_params(), a.activation_dtype,
               'fp32 arithmetic' if (a.no_bf16_mfma or a.activation_dtype == 'float32') else
               'bf16 matrix cores (fp32 accumulation) on the layers with bf16-stored input', a.members, a.forwards),
           'six_hour_steps_per_s': a.members * a.forwards * 2 / dt, 'ms_per_forward': 1e3 * dt / a.forwards,
           'finite': finite, 'launches_per_forward': len(plan.ops),
           'split_ms_per_forward': {'convolutions (MFMA-bound)': round(t_conv, 4), 'element-wise (HBM-bound)': round(t_hbm, 4)},
           'launches': rows}
    print(json.dumps(out))


if __name__ == '__main__':
    if os.environ.get('DLWP_BENCH_SIDE_STREAM') == '1':      # the whole run on a stream of the caller's own instead of the null stream
        with torch.cuda.stream(torch.cuda.Stream()):
            main()
    else:
        main()
