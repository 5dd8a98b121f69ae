#!/usr/bin/env python3
"""
bench.py -- BASELINE.json's metric: 6-h forecast steps/s on the 2-degree 4-channel state, predict_timeseries rollout.

Workload (BASELINE.json configs[1]): the sequential PeriodicPadding2D U-Net of
Azure/train_tf.py:208-268 (4 -> 32 -> 64 -> 128 -> 64 -> 32 -> 4, 188 996 parameters, glorot_uniform seed 1234, zero biases)
on the closed 88 x 180 grid (nominal 91 x 180 does not close under two 2x poolings), float32, built through
DLWPNeuralNet.build_model from the reference's own (name, args, kwargs) triples; a 14-day rollout = 28 forwards x
time_dim 2 = 56 six-hour steps per member.  One "step" of this benchmark = ONE such rollout of all members on this GPU =
one hipGraph launch (168 fused conv kernels).  Members are sharded across ranks, no collective (weak scaling).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints one JSON line on rank 0.  `value` = members x 56 x K x N / wall  [6-h forecast steps / s], state resident in HBM.
`roofline`: the dominant kernel of the forward (largest share of time), timed live with HIP events on the launch stream;
`cpu_baseline`: the unfused torch-CPU restatement of the reference graph + its host rollout loop (oracle/torch_ref.py),
timed on this node's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0


def build_model(grid, cin, seed=1234):
    from dlwp_amd.model import DLWPNeuralNet
    from tests.nets import unet_layers
    np.random.seed(seed)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((cin,) + grid), loss='mse', optimizer='adam', metrics=['mae'], gpus=1)
    return d


def measured_traffic(cfg_tuple, members, ups=False):
    """HBM bytes per launch of a conv tile configuration from the committed rocprofv3 --pmc passes
    (profiles/*hbm_traffic_b256.json: FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE, separate passes, 256 members),
    scaled to `members`.  None when that configuration was not profiled."""
    import glob
    ks, dil, th, tw, waves, fa, bnf, ck, pool = cfg_tuple[:9]
    if fa == 0:
        key = 'WinoCfg<%d, %d, %d, %d, %d, %d, false, %s>' % (dil, th, tw, waves, bnf, ck, 'true' if ups else 'false')
    elif bnf < 0:
        key = 'PackCfg<%d, %d, %d, %d, %d, %d, %d, %d>' % (ks, dil, th, tw, waves, fa, ck, -bnf)
    else:
        key = 'ConvCfg<%d, %d, %d, %d, %d, %d, %d, %d, %s>' % (ks, dil, th, tw, waves, fa, bnf, ck, 'true' if pool else 'false')
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*hbm_traffic_b256.json')), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for name, e in d.items():
            if key in name and 'hbm_read_bytes' in e and 'hbm_write_bytes' in e:
                return (e['hbm_read_bytes'] + e['hbm_write_bytes']) * members / 256.0, os.path.basename(f)
    return None, None


def time_layers(model, members, iters=5):
    """Per-launch duration of every kernel of one forward, HIP events on the stream the kernels are launched on."""
    from dlwp_amd import ops
    ex = model.executor
    plan = ex.plan                                      # the inference plan (pooling in the producers' epilogues)
    x = torch.randn((members,) + plan._in_store, device=model.device)
    outs = ex.run(x)                                    # fills every scratch buffer with realistic data
    bufs = ex.scratch(members)
    rows = []
    for op, d in zip(plan.ops, ex._descriptors()):
        if op.kind not in ('conv', 'maxpool'):
            continue
        src = x if op.src == -1 else (bufs[op.src] if op.src >= 0 else outs[-2 - op.src])
        dst = bufs[op.dst] if op.dst >= 0 else outs[-2 - op.dst]
        if op.kind == 'maxpool':        # the pooling in front of a Winograd layer: an HBM-bound pass
            for _ in range(2):
                ops.maxpool2(src, out=dst)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.maxpool2(src, out=dst)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            nb = float(src.numel() * src.element_size() + dst.numel() * dst.element_size())
            rows.append({'layer': 'maxpool2 %dx%d' % (op.xs[1], op.xs[2]), 'kind': 'hbm', 'ms': ms, 'gbs': nb / ms / 1e6,
                         'flops': 0.0, 'bytes': nb})
            continue
        lay = op.layer
        kern, bias = ex.conv_weights(op)       # the layer's, or the phase-summed kernels of a restated decoder layer
        # weights prepared once, as in the rollout graph (dlwp_conv2d_prepare): the events bracket the conv kernel only
        prep = ops.conv2d_prepare(src, kern, d, out_dtype=dst.dtype, x_channels=op.xs[0])
        for _ in range(2):
            ops.conv2d(src, kern, bias, d, out=dst, x_channels=op.xs[0], prepared=prep)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.conv2d(src, kern, bias, d, out=dst, x_channels=op.xs[0], prepared=prep)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        _, (kh, kw), dil_run = op.conv_geometry
        co, ho, wo = getattr(op, 'conv_out_shape', None) or op.out_shape     # FLOPs: what the convolution computes
        flops = 2.0 * ho * wo * co * op.xs[0] * kh * kw * members
        executed = flops
        if op.alg_flops is not None:           # restated layer: ALGORITHMIC FLOPs are those of the reference's layer
            flops = float(op.alg_flops) * members
        co, ho, wo = op.out_shape                                             # bytes: what it stores
        nbytes = (float(src.element_size()) * members * op.xs[0] * op.xs[1] * op.xs[2] +
                  float(dst.element_size()) * members * co * ho * wo + 4.0 * kh * kw * op.xs[0] * co)
        from dlwp_amd import _lib
        import ctypes
        pick = _lib.lib.dlwp_conv2d_pick_config(_lib.handle(model.device.index or 0),
                                                _lib.Shape4(members, op.xs[0], op.xs[1], op.xs[2]), ctypes.byref(d))
        cfg = ops.conv_configs()[pick] if pick >= 0 else None
        rows.append({'layer': lay.name, 'cin': op.xs[0], 'cout': co, 'k': kh, 'dil': dil_run[0], 'tile_cfg': cfg,
                     'out': [ho, wo], 'ms': ms, 'tflops': flops / ms / 1e9, 'gbs': nbytes / ms / 1e6,
                     'flops': flops, 'bytes': nbytes})
        # Winograd on an up-sampled source with odd halos leaves out the 7 identically-zero positions (WinoCfg::UPS)
        rows[-1]['wino_multiplies_per_tile'] = 9 if (kh == 3 and op.src_mode == 1 and tuple(dil_run) == (1, 1) and
                                                     op.halo.top % 2 == 1 and op.halo.left % 2 == 1) else 16
        if op.alg_flops is not None:
            rows[-1]['restated'] = ('on the low-resolution source of the UpSampling2D in front (dlwp_amd/plan.py): '
                                    'executes %.3f of the algorithmic multiplies' % (executed / flops))
    return rows


def cpu_baseline(grid, cin, forwards, weights, budget_s=12.0):
    """The reference's CPU path as restated in oracle/torch_ref.py: unfused pad-copy / zero-pad-copy / conv / bias /
    tanh / pool / upsample in torch-CPU float32 + the host rollout loop with a full state copy per step."""
    from oracle import torch_ref
    from tests.nets import unet_layers
    layers = unet_layers((cin,) + grid)
    tw = torch_ref.to_torch_weights(weights)
    rng = np.random.default_rng(0)
    members = 8
    x = rng.standard_normal((members, cin) + grid).astype(np.float32)
    ncpu = os.cpu_count() or 1
    t0 = time.time()
    torch_ref.rollout_host_loop(layers, tw, x, 1)       # warm-up (thread pool, oneDNN primitive creation)
    t1 = time.time()
    # give the CPU its best thread count: more threads than this small problem can feed only add overhead
    best = None
    for nt in sorted({min(ncpu, v) for v in (8, 16, 32, 64, 128, ncpu)}):
        torch.set_num_threads(nt)
        torch_ref.rollout_host_loop(layers, tw, x, 1)
        ts = time.time()
        torch_ref.rollout_host_loop(layers, tw, x, 1)
        per = time.time() - ts
        if best is None or per < best[0]:
            best = (per, nt)
        if time.time() - t1 > 0.6 * budget_s:
            break
    per_fwd, nt = best
    torch.set_num_threads(nt)
    # bounded sample: whole rollouts (or a truncated one if a single rollout would exceed the budget) for ~budget_s
    n_fwd = int(max(1, min(forwards, budget_s / max(per_fwd, 1e-3))))
    reps = 0
    t2 = time.time()
    while True:
        torch_ref.rollout_host_loop(layers, tw, x, n_fwd)
        reps += 1
        if time.time() - t2 >= budget_s or reps >= 64:
            break
    dt = time.time() - t2
    steps = members * n_fwd * 2 * reps
    name = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    name = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': steps / dt, 'unit': '6-h forecast steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d x (%d members x %d of %d forwards) of the same U-Net rollout, torch-CPU fp32 unfused restatement '
                      '(oracle/torch_ref.py), %.1f s' % (reps, members, n_fwd, forwards, dt),
            'cpu': name, 'first_call_s': t1 - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--members', type=int, default=256, help='ensemble members / initial conditions PER GPU')
    ap.add_argument('--forwards', type=int, default=28, help='model applications per rollout (28 = 14 days)')
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--channels', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--activation-dtype', default='float32', choices=['float32', 'bfloat16'],
                    help="storage of the tensors between the layers (BASELINE config 4 uses bfloat16; the headline "
                         "config 2 is float32)")
    a = ap.parse_args()
    grid = tuple(int(v) for v in a.grid.split('x'))

    from dlwp_amd import parallel
    rank, world, local = parallel.init()
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' %
                         (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; there is no CPU fallback for the product path')
    dev = torch.device('cuda', local)

    d = build_model(grid, a.channels)
    net = d.model
    net.set_activation_dtype(a.activation_dtype)
    flops_fwd = net.plan.conv_flops_per_sample()
    bytes_fwd = net.infer_plan.algorithmic_bytes_per_sample()
    weights_np = [(w, b) for w, b in zip(net.get_weights()[0::2], net.get_weights()[1::2])]

    # members: one base state + 0.01 * N(0,1) perturbations, different per rank
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    base = torch.randn((1, a.channels) + grid, generator=torch.Generator().manual_seed(0))
    state0 = (base + 0.01 * torch.randn((a.members, a.channels) + grid, generator=g)).to(dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def rollout():
        return net.rollout_on_device(state0, a.forwards)

    series = rollout()                           # set-up: captures the hipGraph (never inside the timed region)
    for _ in range(a.warmup):
        series = rollout()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        series = rollout()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(series[-1]).all().item())

    six_hour_steps = a.members * world * a.forwards * 2 * a.steps
    value = six_hour_steps / dt
    fwd_per_s = a.members * world * a.forwards * a.steps / dt
    out = {
        'metric': '6-h forecast steps/sec on 91x180x4-chan state (closed grid 88x180), predict_timeseries rollout',
        'value': value, 'unit': '6-h forecast steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if a.activation_dtype == 'float32' else 'f32 arithmetic, bf16 activation storage',
        'data': 'synthetic',
        'config': {'workload': 'cfg2: 2-deg %dx%d x%d-chan sequential PeriodicPadding2D U-Net (188996 params), fp32, '
                               '%d-forward (14-day) predict_timeseries rollout as one hipGraph, %d members per GPU'
                               % (grid[0], grid[1], a.channels, a.forwards, a.members),
                   'members_per_gpu': a.members, 'forwards_per_rollout': a.forwards, 'time_dim': 2,
                   'grid': list(grid), 'channels': a.channels, 'launches_per_forward': net.infer_plan.n_launches,
                   'launch_note': 'plan operations; a Winograd layer on a 22x45 map takes a second (16-wide) launch for its ragged last column tile at chip-filling batches',
                   'parallelism': 'members sharded over %d GPU(s), no collective' % world,
                   'inference_plan': ('Winograd F(2x2,3x3) on the 3x3 layers; the decoder layers that read an up-sampled '
                                      'tensor are restated on the low-resolution tensor (same function, DESIGN.md 5.7; '
                                      'DLWP_RESTATE_UPSAMPLED=0 runs the reference formulation); FLOP figures are '
                                      'algorithmic (the reference graph, SURVEY.md 8d)')},
        'forwards_per_s': fwd_per_s,
        'conv_mflop_per_forward_per_member': flops_fwd / 1e6,
        'forward': {'achieved_tflops': fwd_per_s * flops_fwd / 1e12 / world,
                    'mfma_util_frac': fwd_per_s * flops_fwd / 1e12 / world / PEAK_F32_MFMA_TFLOPS,
                    'algorithmic_gbs': fwd_per_s * bytes_fwd / 1e9 / world, 'per_gpu': True},
        'finite': finite,
    }
    if rank == 0:
        rows = time_layers(net, a.members)
        tot = sum(r['ms'] for r in rows)
        dom = max((r for r in rows if r.get('kind') != 'hbm'), key=lambda r: r['ms'])
        wino = bool(dom.get('tile_cfg')) and dom['tile_cfg'][5] == 0
        out['roofline'] = {'bound': 'mfma', 'kernel': '%s (%s: %d->%d, %dx%d dil %d, %dx%d)' %
                           ('conv2d_fwd_wino_f32' if wino else 'conv2d_fwd_mfma_f32',
                            dom['layer'], dom['cin'], dom['cout'], dom['k'], dom['k'], dom['dil'], dom['out'][0], dom['out'][1]),
                           'achieved': dom['tflops'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': dom['tflops'] / PEAK_F32_MFMA_TFLOPS, 'traffic': None, 'traffic_unit': 'bytes per launch',
                           'algorithmic_bytes_per_launch': dom['bytes'],
                           'launch_ms': dom['ms'], 'algorithmic_flops_per_launch': dom['flops'],
                           'share_of_forward_time': dom['ms'] / tot}
        if wino:
            # Winograd F(2x2,3x3): the kernel executes 16 multiplies per 2x2 outputs and input channel where the
            # algorithmic (direct) count is 36 -- `achieved`/`frac` are algorithmic FLOPs as the contract asks and may
            # exceed the dense peak; `executed_frac` is what the matrix cores actually issue (tile padding included)
            th_, tw_ = dom['tile_cfg'][2], dom['tile_cfg'][3]
            ho_, wo_ = dom['out']
            pad = (-(-ho_ // th_) * th_) * (-(-wo_ // tw_) * tw_) / float(ho_ * wo_)
            mult = dom.get('wino_multiplies_per_tile', 16)
            out['roofline']['algorithm'] = ('winograd F(2x2,3x3): %d multiplies per 2x2 output tile and channel pair where '
                                            'the algorithmic count is 36%s' %
                                            (mult, ' (up-sampled source: 7 of the 16 positions are identically zero)'
                                             if mult == 9 else ''))
            out['roofline']['executed_frac'] = dom['tflops'] * mult / 36.0 * pad / PEAK_F32_MFMA_TFLOPS
        if dom.get('tile_cfg'):
            tr, src = measured_traffic(dom['tile_cfg'], a.members, ups=dom.get('wino_multiplies_per_tile') == 9)
            out['roofline']['traffic'] = tr
            out['roofline']['traffic_source'] = src
        out['layers'] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k not in ('flops', 'bytes')}
                         for r in rows]
        out['forward']['sum_of_kernel_ms'] = tot
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(grid, a.channels, a.forwards, weights_np)
        print(json.dumps(out))
    barrier()


if __name__ == '__main__':
    main()
