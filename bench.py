#!/usr/bin/env python3
"""
bench.py -- BASELINE.json's metric: 6-h forecast steps/s on the 2-degree 4-channel state, predict_timeseries rollout.

Workload (BASELINE.json configs[1]): the sequential PeriodicPadding2D U-Net of
Azure/train_tf.py:208-268 (4 -> 32 -> 64 -> 128 -> 64 -> 32 -> 4, 188 996 parameters, glorot_uniform seed 1234, zero biases)
on the closed 88 x 180 grid (nominal 91 x 180 does not close under two 2x poolings), float32, built through
DLWPNeuralNet.build_model from the reference's own (name, args, kwargs) triples; a 14-day rollout = 28 forwards x
time_dim 2 = 56 six-hour steps per member.  One "step" of this benchmark = ONE such rollout of all members on this GPU =
one hipGraph launch.  Members are sharded across ranks, no collective (weak scaling).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.
  value      members x 56 x K x N / wall  [6-h forecast steps / s], state resident in HBM.
  roofline   the kernel with the largest share of the forward's time (all its launches), timed live with HIP events on the
             launch stream.  `achieved` / `frac` are the matrix-core FLOPs the kernel EXECUTES (dlwp_conv2d_launch_info:
             the padded GEMM volume of its MFMA instructions == SQ_INSTS_MFMA x 2048 of a rocprofv3 --pmc pass) over the
             fp32 MFMA peak -- never above 1.  `algorithmic_tflops` is the reference layer's direct-convolution FLOP count
              over the same time and `algorithmic_speedup` their ratio: what Winograd F(2x2,3x3) and the
             restated decoder layers save.  `traffic`: HBM bytes per launch from the committed rocprofv3 --pmc summary of
             THIS kernel source (null when the summary was taken from other source: profiles/*hbm_traffic*.json carries a
             hash of dlwp_amd/csrc).
  sub_records   the same API at the other operating points of SURVEY.md 8d and the verdict: members 1 and 8, the
             host-visible (numpy in / numpy out, PCIe-inclusive) rollout, layer 1 alone at the nominal 91 x 180, the
             config-3 training step (global batch 64, data parallel over the ranks with the RCCL all-reduce of the C ABI)
             and the config-5 ensemble (1 degree, 12 channels, 32 members IN TOTAL over the ranks: strong scaling).
  cpu_baseline  the unfused torch-CPU restatement of the reference graph + its host rollout loop (oracle/torch_ref.py),
             timed on this node's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import glob
import json
import os
import re
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # v_mfma_f32_16x16x32_bf16 dense
PEAK_HBM_GBS = 8000.0


def build_model(grid, cin, seed=1234, gpus=1, lr=None):
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    from dlwp_amd.training import Adam
    np.random.seed(seed)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((cin,) + grid), loss='mse', optimizer='adam' if lr is None else Adam(lr=lr),
                  metrics=['mae'], gpus=gpus)
    return d


# --------------------------------------------------------------------------------------------------------------------- #
# kernel-level timing and the roofline
# --------------------------------------------------------------------------------------------------------------------- #

def kernel_source_hash():
    """a PMC summary in profiles/ is only quoted for the kernel source it was measured on"""
    from dlwp_amd import _lib
    return _lib.kernel_source_hash()


def config_symbol(cfg, ups=False, x_loader=0):
    """kernel symbol of a tile configuration tuple (dlwp_conv2d_config_info), as rocprofv3 prints it"""
    ks, dil, th, tw, waves, fa, bnf, ck, pool = cfg[:9]
    if fa == 0:
        split = len(cfg) > 10 and (cfg[10] & 1) and not ups
        if split:          # the 16-position case of these entries: positions split over two waves per tile fragment
            return 'conv2d_fwd_wino2_f32<WinoSplitCfg<%d, %d, %d, %d, %d, %d, false, false> >' % (dil, th, tw, waves, bnf,
                                                                                                 ck)
        # <..., IN16, UPS, DACT, POOL2, SPLITK, PAIRX, UPSQ, EP>: the inference plan uses neither the training epilogue nor the second
        # output, and launches that fill the chip are never split; the last three = the input loader and the edge pairs
        # (dlwp_launch_info.x_loader: 1 / 2, + 4)
        return 'conv2d_fwd_wino_f32<WinoCfg<%d, %d, %d, %d, %d, %d, false, %s, false, false, false, %s, %s, %s> >' % (
            dil, th, tw, waves, bnf, ck, 'true' if ups else 'false', 'true' if (x_loader & 3) == 1 else 'false',
            'true' if (x_loader & 3) == 2 else 'false', 'true' if x_loader & 4 else 'false')
    if bnf < 0:
        return 'conv2d_fwd_packn_f32<PackCfg<%d, %d, %d, %d, %d, %d, %d, %d> >' % (ks, dil, th, tw, waves, fa, ck, -bnf)
    if pool >= 2:
        return 'conv2d_fwd_mfma_bf16<...%dx%d, %d waves>' % (th, tw, waves)
    return 'conv2d_fwd_mfma_f32<ConvCfg<%d, %d, %d, %d, %d, %d, %d, %d, %s> >' % (ks, dil, th, tw, waves, fa, bnf, ck,
                                                                               'true' if pool else 'false')


def kernel_family(symbol):
    """A Winograd instance's symbol without its last three template arguments (the input loader and the edge pairs, r5:
    WinoCfg::PAIRX / UPSQ / EP): those variants of one tile configuration run the same loop on the same bits and count as ONE
    kernel in the roofline -- a prefix of every variant's full name, so the lookups below find (and average over) all of them."""
    m = re.match(r'^(conv2d_fwd_wino_f32<WinoCfg<(?:[^,<>]+, ){10}[^,<>]+), (?:true|false), (?:true|false), (?:true|false)> >$', symbol)
    return m.group(1) if m else symbol


def measured_traffic(symbol, members):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/*hbm_traffic*.json: FETCH_SIZE
    x2 [gfx950 correction] + WRITE_SIZE, separate passes), scaled to `members`.  Only a summary measured on the current
    kernel source counts (its `_meta.source_sha`); otherwise (None, reason)."""
    sha = kernel_source_hash()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*hbm_traffic*.json')), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        meta = d.get('_meta', {})
        hits = [e for name, e in d.items() if name != '_meta' and symbol in name and 'hbm_read_bytes' in e and 'hbm_write_bytes' in e]
        if not hits:
            continue
        if meta.get('source_sha') != sha:
            stale = stale or os.path.basename(f)
            continue
        scale = members / float(meta.get('members', 256))
        n = sum(float(e.get('launches', 1)) for e in hits)        # (several instances of one family: mean over all their launches)
        return sum((e['hbm_read_bytes'] + e['hbm_write_bytes']) * float(e.get('launches', 1)) for e in hits) / n * scale, \
            os.path.basename(f)
    return None, ('stale: %s was measured on other kernel source' % stale) if stale else 'no PMC summary for this kernel'


def rocprof_launch_ms(symbol, members):
    """Average duration of `symbol` in the newest committed `rocprofv3 --kernel-trace --stats` summary
    (profiles/*kernel_stats.csv) of the bench command, with the file's name and whether it was taken on the kernel source
    this library was built from (tools/profile_bench.sh writes <tag>_kernel_stats.meta.json; older summaries are matched
    through the <tag>_mfma_busy.json of the same tag).  bench.py never runs rocprofv3 itself: this is a lookup."""
    import csv
    sha = kernel_source_hash()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*kernel_stats.csv')), reverse=True):
        base = os.path.basename(f)
        if any(t in base for t in ('train', 'cfg4', 'cfg5', 'rowconv', 'pad')):
            continue
        tag = base[:base.index('kernel_stats')]
        meta = None
        for mf in (f[:-4] + '.meta.json', os.path.join(ROOT, 'profiles', tag + 'mfma_busy.json')):
            try:
                meta = json.load(open(mf))
                meta = meta.get('_meta', meta)
                break
            except (OSError, ValueError):
                continue
        if meta is None or int(meta.get('members', 256)) != members:
            continue
        same = meta.get('source_sha') == sha
        try:
            hits = [r for r in csv.DictReader(open(f)) if symbol in r['Name']]
            if hits:                         # (several instances of one family: mean over all their calls)
                calls = sum(int(r['Calls']) for r in hits)
                ent = {'file': base, 'avg_ms': sum(float(r['TotalDurationNs']) for r in hits) / calls / 1e6, 'calls': calls,
                       'same_source': same}
                if len(hits) > 1:
                    ent['instances'] = {r['Name']: {'avg_ms': float(r['AverageNs']) / 1e6, 'calls': int(r['Calls'])} for r in hits}
                if same:
                    return ent
                best = best or ent
        except (OSError, KeyError, ValueError):
            continue
    return best


def pmc_mfma_crosscheck(rows, members):
    """Analytic executed-MFMA counts against SQ_INSTS_MFMA of the newest profiles/*mfma_busy.json measured on this source
    (per kernel symbol: mean over its launches in one forward).  Returns {symbol: {analytic, counter, busy_frac}}."""
    sha = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*mfma_busy*.json')), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        meta = d.get('_meta', {})
        if meta.get('source_sha') != sha or int(meta.get('members', 256)) != members:
            continue
        out = {'file': os.path.basename(f)}
        by = {}
        for r in rows:
            for sym, fl in r.get('launch_flops', []):
                by.setdefault(sym, []).append(fl / 2048.0)
        for sym, counts in by.items():
            for name, e in d.items():
                if name != '_meta' and sym in name and 'SQ_INSTS_MFMA' in e:
                    ent = {'analytic_mfma_per_launch': sum(counts) / len(counts), 'SQ_INSTS_MFMA': e['SQ_INSTS_MFMA']}
                    if e.get('GRBM_GUI_ACTIVE'):
                        # busy cycles are summed over 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                        ent['mfma_pipe_busy_frac'] = e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / \
                            (e['GRBM_GUI_ACTIVE'] / 8.0)
                    out[sym] = ent
        return out
    return None


def _time_calls(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_layers(model, members, iters=5):
    """Per-launch duration of every kernel of one forward, HIP events on the stream the kernels are launched on.  Two
    figures per launch: `ms` -- events between the launches of an eager forward (a layer's input was just written by the
    layer in front and partly sits in the 256 MB Infinity Cache: the situation inside the rollout graph, and what rocprofv3
    reports for the graph's kernels), and `ms_isolated` -- the layer alone, back to back (its working set comes from HBM
    every time)."""
    from dlwp_amd import ops
    ex = model.executor
    plan = ex.plan                                      # the inference plan (pooling in the producers' epilogues)
    x = torch.randn((members,) + plan._in_store, device=model.device)
    outs = ex.run(x)                                    # fills every scratch buffer with realistic data
    bufs = ex.scratch(members)
    cfgs = ops.conv_configs()
    rows, seq = [], []
    for op, d in zip(plan.ops, ex._descriptors()):
        if op.kind not in ('conv', 'maxpool', 'd2s'):
            continue
        src = x if op.src == -1 else (bufs[op.src] if op.src >= 0 else outs[-2 - op.src])
        dst = bufs[op.dst] if op.dst >= 0 else outs[-2 - op.dst]
        if op.kind in ('maxpool', 'd2s'):        # HBM-bound passes left in the forward
            if op.kind == 'maxpool':
                fn = lambda src=src, dst=dst: ops.maxpool2(src, out=dst)  # noqa: E731
                name = 'maxpool2 %dx%d' % (op.xs[1], op.xs[2])
                nb = float(src.numel() * src.element_size() + dst.numel() * dst.element_size())
            else:
                fn = lambda src=src, dst=dst, op=op: ops.depth_to_space2(src, op.xs[0], out=dst, c_off=op.out_c_off)  # noqa: E731
                name = 'depth_to_space2 %dx%d' % (op.xs[1], op.xs[2])
                nb = 2.0 * src.numel() * src.element_size()
            ms = _time_calls(fn, iters)
            rows.append({'layer': name, 'kind': 'hbm', 'ms_isolated': ms, 'flops': 0.0, 'bytes': nb, 'executed_flops': 0.0})
            seq.append(fn)
            continue
        lay = op.layer
        kern, bias = ex.conv_weights(op)       # the layer's, or the phase-summed kernels of a restated decoder layer
        # weights prepared once, as in the rollout graph (dlwp_conv2d_prepare): the events bracket the conv kernel only
        prep = ops.conv2d_prepare(src, kern, d, out_dtype=dst.dtype, x_channels=op.xs[0])
        fn = lambda src=src, kern=kern, bias=bias, d=d, dst=dst, op=op, prep=prep: ops.conv2d(  # noqa: E731
            src, kern, bias, d, out=dst, x_channels=op.xs[0], prepared=prep)
        ms = _time_calls(fn, iters)
        seq.append(fn)
        _, (kh, kw), dil_run = op.conv_geometry
        co, ho, wo = getattr(op, 'conv_out_shape', None) or op.out_shape     # FLOPs: what the convolution computes
        flops = 2.0 * ho * wo * co * op.xs[0] * kh * kw * members
        if op.alg_flops is not None:           # restated layer: ALGORITHMIC FLOPs are those of the reference's layer
            flops = float(op.alg_flops) * members
        co, ho, wo = op.out_shape                                             # bytes: what it stores
        nbytes = (float(src.element_size()) * members * op.xs[0] * op.xs[1] * op.xs[2] +
                  float(dst.element_size()) * members * co * ho * wo + 4.0 * kh * kw * op.xs[0] * co)
        ups = (kh == 3 and op.src_mode == 1 and tuple(dil_run) == (1, 1) and op.halo.top % 2 == 1 and
               op.halo.left % 2 == 1)
        info = ops.conv_launch_info((members, op.xs[0], op.xs[1], op.xs[2]), d, ex._conv_dtype(op),
                                    model.device.index or 0)
        def symbol_of(i):
            if i[0] >= 0:
                return config_symbol(cfgs[i[0]], ups, i[5] if len(i) > 5 else 0)
            if i[0] == -2:      # dlwp_conv2d_launch_info: the few-channel streaming kernel (csrc/conv_fwd_few.hip), <DIL, ACT, QUAD>
                return 'conv2d_fwd_few_f32<%d, ' % dil_run[0]
            if i[0] == -3:      # ... the streaming position-split Winograd kernel (csrc/conv_fwd_wino2s.hip), <chunks, depth-to-space>
                return 'conv2d_fwd_wino2s_f32<%d, ' % ((op.xs[0] + 7) // 8)
            return 'conv2d_fwd_direct_f32'
        launch_flops = [(symbol_of(i), i[3]) for i in info]
        cfg = cfgs[info[0][0]] if info and info[0][0] >= 0 else None
        # USEFUL matrix work: what the instance's GEMMs would multiply without tile / channel padding -- for a Winograd instance
        # 16 (9 on an up-sampled source with odd halos) GEMMs of (2x2 output tiles) x cout x cin; otherwise the direct sum
        cco, cho, cwo = getattr(op, 'conv_out_shape', None) or op.out_shape
        if cfg is not None and cfg[0] == 3 and cfg[5] == 0:
            useful = 2.0 * (9 if ups else 16) * ((cho + 1) // 2) * ((cwo + 1) // 2) * cco * op.xs[0] * members
        else:
            useful = 2.0 * cho * cwo * cco * op.xs[0] * kh * kw * members
        rows.append({'layer': lay.name, 'cin': op.xs[0], 'cout': co, 'k': kh, 'dil': dil_run[0], 'tile_cfg': cfg,
                     'kernel': launch_flops[0][0], 'launches': len(info), 'out': [ho, wo], 'ms_isolated': ms,
                     'flops': flops, 'bytes': nbytes,
                     'executed_flops': sum(i[3] for i in info), 'useful_flops': min(useful, sum(i[3] for i in info)),
                     'launch_flops': launch_flops,
                     'bf16_matrix': bool(info and info[0][4])})
        if op.alg_flops is not None:
            rows[-1]['restated'] = 'on the low-resolution source of the UpSampling2D in front (DESIGN.md 5.7)'
    # the launches in forward order, an event in front of each
    n = len(seq)
    for _ in range(2):
        for fn in seq:
            fn()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] for _ in range(iters)]
    for it in range(iters):
        for k, fn in enumerate(seq):
            evs[it][k].record()
            fn()
        evs[it][n].record()
    torch.cuda.synchronize()
    for k, r in enumerate(rows):
        ms = sum(evs[it][k].elapsed_time(evs[it][k + 1]) for it in range(iters)) / iters
        r['ms'] = ms
        r['gbs'] = r['bytes'] / ms / 1e6
        if r.get('kind') != 'hbm':
            r['algorithmic_tflops'] = r['flops'] / ms / 1e9
            r['executed_tflops'] = r['executed_flops'] / ms / 1e9
    return rows


def executed_per_forward(net, members):
    """Matrix-core FLOPs ONE forward of `members` members executes (dlwp_conv2d_launch_info of every convolution launch of
    the inference plan: MFMA instructions x their FLOPs, tile and channel padding included), split by matrix-core type.
    No kernel is run.  Returns (fp32-MFMA FLOPs, bf16-MFMA FLOPs)."""
    from dlwp_amd import ops
    ex = net.executor
    f32 = b16 = 0.0
    for op, d in zip(ex.plan.ops, ex._descriptors()):
        if op.kind != 'conv':
            continue
        for i in ops.conv_launch_info((members, op.xs[0], op.xs[1], op.xs[2]), d, ex._conv_dtype(op), net.device.index or 0):
            if i[4]:
                b16 += i[3]
            else:
                f32 += i[3]
    return f32, b16


def operating_point(net, members, forwards_per_s_per_gpu):
    """`frac` of a sub-record: executed matrix-core FLOP/s of this GPU over the dense peak of the cores they run on, and
    the fused-forward algorithmic bytes (sum over launches of in + out + weights) over the HBM peak."""
    f32, b16 = executed_per_forward(net, members)
    itemsize = 2 if getattr(net, 'activation_dtype', 'float32') == 'bfloat16' else 4
    nbytes = float(net.infer_plan.algorithmic_bytes_per_sample(itemsize)) * members
    fwd = forwards_per_s_per_gpu / float(members)               # forwards of the whole member batch per second
    out = {'executed_tflops': (f32 + b16) * fwd / 1e12,
           'frac': f32 * fwd / 1e12 / PEAK_F32_MFMA_TFLOPS + b16 * fwd / 1e12 / PEAK_BF16_MFMA_TFLOPS,
           'frac_definition': 'time share the executed MFMA work needs at the dense peak of its matrix cores '
                              '(fp32 157.3 TFLOP/s%s)' % (', bf16 2500 TFLOP/s' if b16 else ''),
           'algorithmic_gbs': nbytes * fwd / 1e9, 'hbm_frac': nbytes * fwd / 1e9 / PEAK_HBM_GBS}
    if b16:
        out['bf16_tflops'] = b16 * fwd / 1e12
        out['bf16_mfma_frac'] = b16 * fwd / 1e12 / PEAK_BF16_MFMA_TFLOPS
    return out


def roofline_of(rows, members):
    """The kernel (all launches of one symbol inside a forward) with the largest share of the forward's time."""
    tot = sum(r['ms'] for r in rows)
    groups = {}
    for r in rows:
        if r.get('kind') == 'hbm':
            continue
        groups.setdefault(kernel_family(r['kernel']), []).append(r)
    sym, rs = max(groups.items(), key=lambda kv: sum(r['ms'] for r in kv[1]))
    variants = sorted({r['kernel'] for r in rs})
    ms = sum(r['ms'] for r in rs)
    n_launch = len(rs)
    executed = sum(r['executed_flops'] for r in rs)
    algorithmic = sum(r['flops'] for r in rs)
    peak = PEAK_BF16_MFMA_TFLOPS if rs[0]['bf16_matrix'] else PEAK_F32_MFMA_TFLOPS
    achieved = executed / ms / 1e9
    ms_iso = sum(r['ms_isolated'] for r in rs)
    out = {'bound': 'mfma', 'kernel': sym,
           'layers': ['%s (%d->%d, %dx%d dil %d, out %dx%d)' % (r['layer'], r['cin'], r['cout'], r['k'], r['k'], r['dil'],
                                                                 r['out'][0], r['out'][1]) for r in rs],
           'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
           #  executed minus multiplied padding: a change that removes padded work lowers `frac` and raises this
           'useful_frac': sum(r.get('useful_flops', r['executed_flops']) for r in rs) / ms / 1e9 / peak,
           'useful_definition': 'the same launches priced on the matrix work WITHOUT tile / channel padding (Winograd: positions x '
                                '2x2 output tiles x cout x cin); frac - useful_frac is multiplied padding.  Judge kernel changes by '
                                'steps/s and useful_frac, not by frac',
           'definition': 'matrix-core FLOPs the kernel executes (MFMA instructions x 2048, tile and channel padding '
                         'included) / HIP-event time of its launches inside a forward / dense fp32 MFMA peak; `traffic` and '
                         '`rocprof` are LOOKUPS in the committed rocprofv3 summaries under profiles/ (counters cannot be read '
                         'inside this run), quoted only with the hash of the kernel source they were measured on',
           'algorithmic_tflops': algorithmic / ms / 1e9, 'algorithmic_speedup': algorithmic / executed,
           'launches_per_forward': n_launch, 'launch_ms': ms / n_launch,
           'launch_ms_isolated': ms_iso / n_launch, 'frac_isolated': executed / ms_iso / 1e9 / peak,
           'executed_flops_per_launch': executed / n_launch, 'algorithmic_flops_per_launch': algorithmic / n_launch,
           'algorithmic_bytes_per_launch': sum(r['bytes'] for r in rs) / n_launch,
           'share_of_forward_time': ms / tot, 'traffic_unit': 'bytes per launch'}
    if len(variants) == 1:
        out['kernel'] = variants[0]
    else:
        # `kernel` stays a symbol rocprofv3 prints (the variant with the largest share); the figures are the family's
        out['kernel'] = max(variants, key=lambda v: sum(r['ms'] for r in rs if r['kernel'] == v))
        out['kernel_family'] = sym + ', *, *, *> >'
        out['instances'] = {v: [r['layer'] for r in rs if r['kernel'] == v] for v in variants}
        out['instances_note'] = ('one tile configuration, one loop, the same bits; the last three template arguments pick the input '
                                 'loader and the edge pairs (DLWP_OPT_WINO_XLOADER: image-aligned column pairs on even widths, '
                                 'element by element otherwise; the half-used last tile row of a 44-row map paired up); `rocprof` '
                                 'and `traffic` are means over all their launches')
    if any(r['launches'] > 1 for r in rs):
        out['note'] = ('layers on a 45-column map hand their ragged last column tile to a 16-wide instance in a second '
                       'launch; its time and FLOPs are inside these figures')
    tr, src = measured_traffic(sym, members)
    out['traffic'], out['traffic_source'] = tr, src
    rp = rocprof_launch_ms(sym, members)
    if rp:
        out['rocprof'] = dict(rp, frac_at_rocprof_avg=executed / n_launch / rp['avg_ms'] / 1e9 / peak,
                              launch_ms_over_rocprof=(ms / n_launch) / rp['avg_ms'],
                              note='HIP events between the launches of an eager forward include the launch gap; rocprofv3 '
                                   'times the kernel alone inside the rollout graph')
    return out


# --------------------------------------------------------------------------------------------------------------------- #
# CPU baseline
# --------------------------------------------------------------------------------------------------------------------- #

def cpu_baseline(grid, cin, forwards, weights, budget_s=18.0, warm_s=1.0, sweep_s=3.0):
    """The reference's CPU path as restated in oracle/torch_ref.py: unfused pad-copy / zero-pad-copy / conv / bias /
    tanh / pool / upsample in torch-CPU float32 + the host rollout loop with a full state copy per step."""
    from oracle import torch_ref
    from dlwp_amd.presets import unet_layers
    layers = unet_layers((cin,) + grid)
    tw = torch_ref.to_torch_weights(weights)
    rng = np.random.default_rng(0)
    members = 8
    x = rng.standard_normal((members, cin) + grid).astype(np.float32)
    ncpu = os.cpu_count() or 1
    t0 = time.time()
    torch_ref.rollout_host_loop(layers, tw, x, 1)       # first call (thread pool, oneDNN primitive creation)
    t1 = time.time()

    def window(seconds):
        """(steps/s, forwards, seconds) of whole one-forward calls for at least `seconds`"""
        n, ts = 0, time.time()
        while True:
            torch_ref.rollout_host_loop(layers, tw, x, 1)
            n += 1
            el = time.time() - ts
            if el >= seconds:
                return members * 2 * n / el, n, el

    # the sweep is pinned to {8, 16, 32} threads (the counts that ever won on these hosts: more threads than
    # this small problem can feed only add overhead), every count gets a 1-s warm-up and a window of >= 3 s, every count's rate is
    # in the record, the best count is then measured once more over a longer window and THAT is the value
    sweep = {}
    for nt in sorted({min(ncpu, v) for v in (8, 16, 32)}):
        torch.set_num_threads(nt)
        window(warm_s)
        sweep[nt] = window(sweep_s)[0]
    nt = max(sweep, key=sweep.get)
    torch.set_num_threads(nt)
    window(0.5 * warm_s)
    rate, reps, dt = window(max(sweep_s, budget_s - len(sweep) * (warm_s + sweep_s)))
    steps = members * 2 * reps
    n_fwd = 1
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
            'sweep': {str(k): v for k, v in sorted(sweep.items())}, 'sweep_note': 'steps/s per thread count, 1 s warm-up + >= 3 s each',
            'cpu': name, 'first_call_s': t1 - t0}


# --------------------------------------------------------------------------------------------------------------------- #
# sub-records
# --------------------------------------------------------------------------------------------------------------------- #

def _sync_time(fn, reps, barrier=None, world=1, dev=None):
    """wall time of `reps` calls bracketed by synchronize (+ barrier) on both sides; MAX over ranks"""
    torch.cuda.synchronize()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _warm_until_steady(fn, batch=5, min_s=0.15, max_s=2.0):
    """Run `fn` in batches until two consecutive batches take the same time within 3 % (and at least `min_s` of work has
    passed): short sub-records start right after host-side model building, with the GPU clocks down, and the ramp back up
    takes anything from 30 ms to a few hundred (measured: the config-4 sub-record read 16-21 k steps/s in two of five runs
    behind a fixed 45 ms warm-up, 44-46 k otherwise)."""
    t_start = time.perf_counter()
    prev = None
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(batch):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        now = time.perf_counter() - t_start
        if (prev is not None and now >= min_s and abs(dt - prev) <= 0.03 * prev) or now >= max_s:
            return
        prev = dt


def sub_small_batches(net, grid, cin, forwards):
    """latency end of the same rollout: 1 and 8 members per GPU (N = 1 and N = 8)"""
    out = {}
    for m in (1, 8):
        s0 = torch.randn((m, cin) + grid, device=net.device)
        net.rollout_on_device(s0, forwards)                       # (graph capture)
        _warm_until_steady(lambda: net.rollout_on_device(s0, forwards))
        reps = 20
        dt = _sync_time(lambda: net.rollout_on_device(s0, forwards), reps)
        out['members_%d' % m] = {'value': m * forwards * 2 * reps / dt, 'unit': '6-h forecast steps/s',
                                 'ms_per_rollout': 1e3 * dt / reps, 'ms_per_forward': 1e3 * dt / reps / forwards}
        out['members_%d' % m].update(operating_point(net, m, m * forwards * reps / dt))
    return out


def sub_host_visible(d, grid, cin, members, forwards):
    """the API exactly as the reference exposes it: numpy in, numpy out (PCIe both ways, pinned result array)"""
    x = np.random.default_rng(0).standard_normal((members, cin) + grid).astype(np.float32)
    out = d.predict_timeseries(x, 2 * forwards)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = d.predict_timeseries(x, 2 * forwards)
        times.append(time.perf_counter() - t0)
    return {'value': members * forwards * 2 / min(times), 'unit': '6-h forecast steps/s', 'members': members,
            'series_bytes': int(out.nbytes), 's_per_call': min(times),
            'note': 'DLWPNeuralNet.predict_timeseries(numpy) -> numpy: one hipGraph per model call, every finished forecast slot '
                    'leaves for ONE recycled pinned result array (time first, as the reference returns it) on a copy stream while the '
                    'next call runs; the first call in member chunks under the upload (DESIGN.md 5.15).  Link-bound: the series / '
                    's_per_call is the rate the PCIe link delivers beside the rollout',
            'series_gbs': out.nbytes / min(times) / 1e9}


def sub_timeseries_estimator(grid, members, forwards, plain_steps_per_s):
    """The rollout examples/validate.py:191-205 runs: SeriesDataGenerator(add_insolation=True) ->
    TimeSeriesEstimator.predict -- inputs 2 time steps x (2 variables + insolation) = 6 channels, outputs 2 x 2 = 4 channels, so
    the forecast is NOT the next input: between two model calls the rows shift to the later start time, the insolation of the rows
    past the data is refreshed and the predicted channels are scattered into the state (csrc/feedback.hip), all `forwards` calls
    in ONE hipGraph.  value = the device-resident loop (as the headline); api = TimeSeriesEstimator.predict -> LabeledArray
    (host generator, insolation table, upload, D2H of the series); host_loop = the reference's form of the same loop (one
    model.predict round trip + numpy re-indexing per step, DLWP/model/extensions.py:206-240) around the same device forward."""
    from dlwp_amd.model import DLWPNeuralNet, SeriesDataGenerator, SeriesDataset, TimeSeriesEstimator
    from dlwp_amd.presets import unet_layers
    rng = np.random.default_rng(3)
    n_t = members + 3                                   # samples = n_t - input steps - output steps + 1
    dates = (np.datetime64('2010-01-01T00') + np.arange(n_t) * np.timedelta64(6, 'h')).astype('datetime64[s]')
    series = rng.standard_normal((n_t, 2, 1) + grid).astype(np.float32)
    ds = SeriesDataset(series, {'sample': dates, 'variable': np.array(['z', 'tau']), 'level': np.array([500]),
                                'lat': np.linspace(88., -88., grid[0]), 'lon': np.arange(0., 360., 360. / grid[1])},
                       ('sample', 'variable', 'level', 'lat', 'lon'))
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((6,) + grid, cout=4), loss='mse', optimizer='adam', metrics=['mae'])
    gen = SeriesDataGenerator(d, ds, input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=64)
    est = TimeSeriesEstimator(d, gen)
    steps = 2 * forwards
    est.predict(steps, return_device=True)                        # (capture)
    graph = next(iter(d.model._rollouts_fed.values()))[0][0]
    _warm_until_steady(graph.launch)
    reps = 5
    dt = _sync_time(graph.launch, reps)
    value = members * steps * reps / dt
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = est.predict(steps)
        times.append(time.perf_counter() - t0)
    rec = {'value': value, 'unit': '6-h forecast steps/s', 'samples': members, 'model_calls': forwards,
           'ms_per_rollout': 1e3 * dt / reps, 'vs_plain_rollout': value / plain_steps_per_s,
           'channels': '6 in (2 steps x (z500, tau300-700, insolation)) -> 4 out', 'finite': bool(np.isfinite(out.values).all()),
           'api': {'value': members * steps / min(times), 's_per_call': min(times), 'series_bytes': int(out.values.nbytes),
                   'note': 'TimeSeriesEstimator.predict(steps) -> LabeledArray: generator gather + insolation table on the host, '
                           'upload, the graph, D2H of the series'}}
    # the two HBM-bound launches this path adds (csrc/feedback.hip), alone: ALGORITHMIC bytes (one state in + one state out; one
    # call's output in + out) / HIP-event time, as the padding kernels are reported (pad_hbm)
    from dlwp_amd import _lib, ops
    import ctypes
    dev = d.model.device
    old_s, out_s = torch.randn((members, 6) + grid, device=dev), torch.randn((members, 4) + grid, device=dev)
    new_s, sol_s = torch.empty_like(old_s), torch.randn((2, 2) + grid, device=dev)
    src_map = [0, 1, 2, 3, 4, 5]
    for m in range(2):
        for j in range(2):
            src_map[m * 3 + j] = -1 - (m * 2 + j)
    fb = ops.make_feedback(members, 6, 4, grid[0] * grid[1], src_map, shift=2, tail=2, sol=[-1, -1, 0, -1, -1, 1], sol_planes=2)
    ms_fb = _time_calls(lambda: ops.state_feedback(old_s, out_s, fb, sol=sol_s, new_state=new_s), iters=20, warm=5)
    arranged = torch.empty_like(out_s)
    perm = (ctypes.c_int * 2)(1, 0)
    hnd, st = _lib.handle(dev.index or 0), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ms_ar = _time_calls(lambda: _lib.check(_lib.lib.dlwp_series_arrange(hnd, ctypes.c_void_p(out_s.data_ptr()),
                                                                       ctypes.c_void_p(arranged.data_ptr()), members, 2, 2,
                                                                       grid[0] * grid[1], 2, perm, 1, _lib.F32, st)), iters=20, warm=5)
    by_fb, by_ar = 2.0 * new_s.numel() * 4, 2.0 * out_s.numel() * 4
    rec['hbm_kernels'] = {'state_feedback': {'ms': ms_fb, 'gbs': by_fb / ms_fb / 1e6, 'hbm_frac': by_fb / ms_fb / 1e6 / PEAK_HBM_GBS,
                                             'algorithmic_bytes': by_fb},
                          'series_arrange': {'ms': ms_ar, 'gbs': by_ar / ms_ar / 1e6, 'hbm_frac': by_ar / ms_ar / 1e6 / PEAK_HBM_GBS,
                                             'algorithmic_bytes': by_ar},
                          'unit': 'GB/s, algorithmic bytes (in + out) / HIP-event time; bound: HBM (8 TB/s)'}
    os.environ['DLWP_ESTIMATOR_HOST'] = '1'
    try:
        t0 = time.perf_counter()
        host = est.predict(steps)
        rec['host_loop'] = {'value': members * steps / (time.perf_counter() - t0),
                            'note': 'the reference-form loop: model.predict round trip + numpy re-indexing per step',
                            'bit_identical': bool(np.array_equal(host.values, out.values, equal_nan=True))}
    finally:
        del os.environ['DLWP_ESTIMATOR_HOST']
    for old in d.model._rollouts_fed.values():
        for g_ in old[0]:
            g_.close()
    d.model._rollouts_fed.clear()
    return rec


def sub_layer1_nominal(net, members):
    """layer 1 (4 -> 32, 3x3 dilation 2, periodic + zero halo, tanh) alone at the NOMINAL 91 x 180 grid"""
    from dlwp_amd import _lib, ops
    lay = [l for l in net.layers if hasattr(l, 'kernel')][0]
    cd = ops.make_conv(32, 3, 3, 2, ops.make_pad(2, 2, 2, 2, _lib.PAD_ZERO, _lib.PAD_WRAP), _lib.ACT_TANH)
    x = torch.randn((members, 4, 91, 180), device=net.device)
    y = torch.empty((members, 32, 91, 180), device=net.device)
    ms = _time_calls(lambda: ops.conv2d(x, lay.kernel, lay.bias, cd, out=y), 10)
    info = ops.conv_launch_info((members, 4, 91, 180), cd, None, net.device.index or 0)
    ex = sum(i[3] for i in info)
    alg = 2.0 * 91 * 180 * 32 * 4 * 9 * members
    nb = 4.0 * members * (4 + 32) * 91 * 180
    return {'ms': ms, 'members': members, 'executed_tflops': ex / ms / 1e9, 'mfma_frac': ex / ms / 1e9 / PEAK_F32_MFMA_TFLOPS,
            'algorithmic_tflops': alg / ms / 1e9, 'algorithmic_gbs': nb / ms / 1e6, 'hbm_frac': nb / ms / 1e6 / PEAK_HBM_GBS,
            'bound': 'hbm (AI 16 F/B < ridge 20)', 'note': 'full 91x180 output written (no pooling epilogue)'}


def sub_cfg4(members=8, forwards=4):
    """BASELINE config 4: 1-degree 180 x 360, 6 variables x 2 input steps, ConvLSTM2D front end + U-Net, bfloat16 storage
    between the layers (bf16 matrix cores, ConvLSTM2D cell update in the convolutions' epilogues), rollout as one hipGraph;
    per-launch split and counters: tools/bench_cfg4.py / tools/profile_cfg4.sh."""
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import lstm_unet_layers
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(lstm_unet_layers((2, 6, 180, 360)), loss='mse', optimizer='adam')
    net = d.model
    net.set_activation_dtype('bfloat16')
    x = torch.randn((members,) + net.infer_plan._in_store, device=net.device)
    # On a stream of the caller's own: the rollout's two member chains are a FORKED graph, which the library launches directly on a
    # real stream and through a stream of its own (two cross-queue hops, ~40 us of this 1 ms rollout) on torch's null stream
    # (csrc/rollout.hip: dlwp_rollout_launch; profiles/r5_cfg4_stream_ab.txt).  `null_stream` = the same loop on the null stream.
    side = torch.cuda.Stream(net.device)
    side.wait_stream(torch.cuda.current_stream(net.device))
    with torch.cuda.stream(side):
        ser = net.rollout_on_device(x, forwards)
        _warm_until_steady(lambda: net.rollout_on_device(x, forwards), batch=10)   # the model was built on the host, the GPU sat idle
        reps = 30
        dt = _sync_time(lambda: net.rollout_on_device(x, forwards), reps)
    torch.cuda.current_stream(net.device).wait_stream(side)
    dt0 = _sync_time(lambda: net.rollout_on_device(x, forwards), reps)
    #  `value` is what a caller gets BY DEFAULT -- torch's null stream; the caller-stream figure rides beside it
    rec = {'value': members * forwards * 2 * reps / dt0, 'unit': '6-h forecast steps/s', 'members': members, 'forwards': forwards,
           'ms_per_forward': 1e3 * dt0 / reps / forwards, 'dtype': 'bf16 storage and matrix cores, f32 accumulation and cell state',
           'launch_stream': "torch's null stream (the default; a forked graph hops through a stream of the library's own there)",
           'caller_stream': {'value': members * forwards * 2 * reps / dt, 'ms_per_forward': 1e3 * dt / reps / forwards,
                             'note': "launched on a torch.cuda.Stream of the caller's own: the direct launch"},
           'launches_per_forward': net.infer_plan.n_launches, 'finite': bool(torch.isfinite(ser[-1]).all().item())}
    rec.update(operating_point(net, members, members * forwards * reps / dt0))
    return rec


def sub_row_connected(grid, cin, members, forwards):
    """The same U-Net with the reference's `latitude_dependent` output layer (DLWP.custom.RowConnected2D, per-latitude filters:
    examples/train_functional.py:53, 191-196; csrc/rowconv.hip), same rollout.  Per-pass timings: tools/bench_rowconv.py."""
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((cin,) + tuple(grid), latitude_dependent=True), loss='mse', optimizer='adam')
    net = d.model
    ws = net.get_weights()
    net.set_weights([w * 0.5 if w.ndim > 3 else w for w in ws])      # keep a 28-forward rollout finite
    x = torch.randn((members, cin) + tuple(grid), device=net.device)
    ser = net.rollout_on_device(x, forwards)
    _warm_until_steady(lambda: net.rollout_on_device(x, forwards), batch=1)
    reps = 3
    dt = _sync_time(lambda: net.rollout_on_device(x, forwards), reps)
    return {'value': members * forwards * 2 * reps / dt, 'unit': '6-h forecast steps/s', 'members': members,
            'ms_per_forward': 1e3 * dt / reps / forwards, 'launches_per_forward': net.infer_plan.n_launches,
            'output_layer': 'RowConnected2D(%d, 5): %d parameters of %d' % (cin, net.layers[-1].count_params(), net.count_params()),
            'finite': bool(torch.isfinite(ser[-1]).all().item())}


def _train_exec_frac(batch):
    """Executed matrix-core FLOPs of one training step of `batch` samples per GPU, from the committed rocprofv3 --pmc SQ_INSTS_MFMA
    pass of tools/bench_train.py (profiles/r*_train_mfma_b<batch>.json) -- quoted only when it was taken on THIS kernel source."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, 'profiles', 'r*_train_mfma_b%d.json' % batch)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if d.get('_meta', {}).get('source_sha') == kernel_source_hash():
            return d.get('mfma_per_step')
    return None


def _time_all_reduce(tr, world, barrier, dev, reps=50):
    """ms per all-reduce of the step's flat exchange buffer (gradients + loss table), alone on the stream"""
    if world <= 1:
        return None
    for _ in range(5):
        tr.dp.all_reduce_sum_(tr._flat_exchange)
    dt = _sync_time(lambda: tr.dp.all_reduce_sum_(tr._flat_exchange), reps, barrier, world, dev)
    tr._flat_exchange.zero_()
    return 1e3 * dt / reps


def sub_pad_hbm(members):
    """north_star: 'rocprof HBM GB/s reported for the padding kernels'.  The standalone PeriodicPadding2D + ZeroPadding2D kernel
    (dlwp_pad2d_fwd / _bwd, csrc/halo.hip; DLWP/custom.py:197-204 + keras ZeroPadding2D) on the six composite halos of one
    config-2 forward at this member count: ALGORITHMIC bytes (in + out) / HIP-event time, as a fraction of the 8 TB/s HBM peak.
    (The product's forward never launches these -- every halo is fused into a convolution's loader; they serve layer stacks
    nothing fuses and the TFPadding2D / FillPadding2D modes.)  `counters`: HBM bytes from the rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of tools/profile_pads.sh on THIS kernel source (profiles/r*_pad_pool_hbm.json), when there is one."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
    import bench_pad
    rows = bench_pad.measure(members, 10, pads_only=True)
    out = {'rows': [{'kernel': r['kernel'], 'shape': r['shape'], 'halo': r['halo'], 'ms': round(r['ms'], 4), 'gbs': round(r['gbs'], 1),
                     'hbm_frac': round(r['gbs'] / PEAK_HBM_GBS, 3)} for r in rows],
           'unit': 'GB/s, algorithmic bytes (in + out) / HIP-event time', 'peak_gbs': PEAK_HBM_GBS}
    out['min_hbm_frac'] = min(r['hbm_frac'] for r in out['rows'])
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pad_pool_hbm.json')), reverse=True)
        profs = [json.load(open(f)) for f in cands]
        prof = next((q for q in profs if q.get('_meta', {}).get('source_sha') == kernel_source_hash()), profs[0] if profs else {})
        if prof.get('_meta', {}).get('source_sha') == kernel_source_hash():
            out['counters'] = [{'kernel': r['kernel'], 'shape': r['shape'], 'traffic_over_algorithmic': round(r['traffic_over_algorithmic'], 3),
                                'hbm_gbs_from_counters': round(r['hbm_gbs_from_counters'], 1)}
                               for r in prof['rows'] if r['kernel'].startswith('pad2d') and 'traffic_over_algorithmic' in r]
        else:
            out['counters'] = 'no profiles/r*_pad_pool_hbm.json of this kernel source'
    except Exception:  # noqa: BLE001
        out['counters'] = None
    return out


def _compare_exchanges(tr, world, rank, barrier, dev, reps=50, crash_line=None, rec=None):
    """Both transports of the step's exchange on the SAME buffer, once for the sums and `reps` times for the latency: RCCL (the
    default; torch.distributed where the group is not on RCCL) and the library's own one-shot all-reduce (csrc/xchg.hip, DLWP_ALLREDUCE=oneshot: peer-mapped uncached regions,
    rank-order sum).  The first multi-GPU run of this script thereby yields the comparison -- equal sums, both latencies -- instead
    of a hang or a silently wrong sum: a one-shot launch that waits 2 s for a peer gives up, dlwp_xchg_status reports it, and the
    record says so (the training records above always ran on RCCL).  Collective: every rank calls it."""
    out = {}
    try:
        n = int(tr._flat_exchange.numel())
        # small integers: every partial sum is exact in float32, whatever the order a transport adds in
        base = (torch.arange(n, device=dev, dtype=torch.float32) % 251.0) + float(rank + 1)
        a = base.clone()
        tr.dp.all_reduce_sum_(a)
        torch.cuda.synchronize()
        want = ((torch.arange(n, device=dev, dtype=torch.float32) % 251.0) * world + world * (world + 1) / 2.0)
        out['default_transport'] = ('RCCL via dlwp_allreduce_sum_f32 (C ABI)' if tr.dp.uses_rccl_abi() else
                                    'torch.distributed (%s)' % tr.dp.backend)
        out['default_sum_exact'] = bool(torch.equal(a, want))
        prev = os.environ.get('DLWP_ALLREDUCE')
        os.environ['DLWP_ALLREDUCE'] = 'oneshot'
        # the one-shot exchange has never run between two DEVICES: should a GPU memory fault abort this process, the line as it
        # stands (this record marked) still reaches stdout (dlwp_set_crash_message)
        if crash_line is not None and rec is not None:
            rec['exchange_check'] = dict(out, oneshot='the process died inside the one-shot exchange (line written by the crash handler)')
            crash_line(True, rec)
        try:
            if not tr.dp.wants_oneshot(base):
                out['oneshot'] = 'not available for this buffer / world size'
                return out
            b = base.clone()
            tr.dp.oneshot_all_reduce_(b)
            torch.cuda.synchronize()
            # (the ranks agree on the outcome before anyone goes on: a rank that timed out must not leave the others launching
            #  exchanges it no longer takes part in)
            flag = torch.tensor([1.0 if tr.dp.oneshot_timed_out() else 0.0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if float(flag.item()) > 0.0:
                out['oneshot'] = 'TIMED OUT waiting for a peer (dlwp_xchg_status) on at least one rank: RCCL stays the transport'
                return out
            out['oneshot_sum_exact'] = bool(torch.equal(b, want))
            out['equal_sums'] = bool(torch.equal(a, b))
            out['oneshot_region'] = tr.dp.oneshot_info()
            for _ in range(5):
                tr.dp.oneshot_all_reduce_(b)
            dt = _sync_time(lambda: tr.dp.oneshot_all_reduce_(b), reps, barrier, world, dev)
            out['oneshot_ms'] = 1e3 * dt / reps
            out['oneshot_timed_out'] = bool(tr.dp.oneshot_timed_out())
            for _ in range(5):
                tr.dp.all_reduce_sum_(a)
            dt = _sync_time(lambda: tr.dp.all_reduce_sum_(a), reps, barrier, world, dev)
            out['default_ms'] = 1e3 * dt / reps
        finally:
            if crash_line is not None:
                crash_line(False)
            if prev is None:
                os.environ.pop('DLWP_ALLREDUCE', None)
            else:
                os.environ['DLWP_ALLREDUCE'] = prev
    except Exception as e:  # noqa: BLE001  (the comparison must never take the bench line down)
        out['error'] = repr(e)[:300]
    return out


def sub_train(grid, cin, world, rank, barrier, dev, per_gpu_batch=64, steps=20, warmup=40, share_of=8, crash_line=None):
    """BASELINE config 3: the same U-Net, training ('mse', Adam), data parallel over the ranks -- each rank trains on its rows of
    the global batch, one all-reduce of the flat gradient buffer per step (RCCL through dlwp_allreduce_sum_f32).  Both conventions
    at N > 1: 'strong' -- the global batch stays 64, every rank takes 64 / N rows -- and 'weak' -- the reference's: the batch grows
    with the GPU count, 64 per GPU (Azure/train_tf.py:163-164: batch_size = n_gpu * batch_size)."""
    from dlwp_amd.parallel import shard_bounds
    d = build_model(grid, cin, gpus=world, lr=1e-4)
    tr = d.model._trainer
    flops_s = 3.0 * d.model.plan.conv_flops_per_sample()

    def run(global_batch, label):
        lo, hi = shard_bounds(global_batch, rank, world)
        g = torch.Generator().manual_seed(0)
        x = torch.randn((global_batch, cin) + grid, generator=g)[lo:hi].to(dev)
        y = torch.randn((global_batch, cin) + grid, generator=g)[lo:hi].to(dev)
        for _ in range(warmup):
            tr.train_on_shard(x, y, global_batch, return_device=True)
        dt = _sync_time(lambda: tr.train_on_shard(x, y, global_batch, return_device=True), steps, barrier, world, dev)
        lv, _ = tr.train_on_shard(x, y, global_batch, return_device=True)
        flops = flops_s * global_batch * steps
        rec = {'value': global_batch * steps / dt, 'unit': 'samples/s', 'scaling': label, 'global_batch': global_batch,
               'batch_per_gpu': hi - lo, 'ms_per_step': 1e3 * dt / steps, 'steps': steps,
               'step_form': tr._graph_ok(hi - lo) or 'python',
               'algorithmic_tflops_per_gpu': flops / dt / 1e12 / world,
               'algorithmic_frac': flops / dt / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 'loss': float(lv[0, 1].item())}
        mf = _train_exec_frac(hi - lo)
        if mf:
            rec['executed_frac'] = mf * 2048.0 * steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS
            rec['executed_mfma_per_step'] = mf
        return rec, x, y

    rec, x, y = run(per_gpu_batch, 'strong')
    rec['frac_definition'] = ('algorithmic_frac: ALGORITHMIC FLOPs of the step (3 x the direct-convolution count of the forward: '
                              'forward, data and weight gradients) / wall / 157.3 TFLOP/s -- the Winograd forward and weight-gradient '
                              'kernels execute fewer multiplies than that count; executed_frac: SQ_INSTS_MFMA x 2048 of the step '
                              '(rocprofv3 --pmc pass of tools/bench_train.py on this kernel source, profiles/r*_train_mfma_b*.json) '
                              '/ wall / 157.3 TFLOP/s')
    rec['step_forms'] = ("'graph' / 'lanes' / 'branches': the step replayed by the library (dlwp_train_step_launch: one hipGraph / "
                         "launch by launch over the recorded lanes / one hipGraph with the lanes as branches); 'python': launch by "
                         "launch from the interpreter")
    rec['all_reduce'] = ('none (single rank)' if world == 1 else
                         ('RCCL via dlwp_allreduce_sum_f32 (C ABI), %d floats incl. the loss table'
                          % tr._flat_exchange.numel() if tr.dp.uses_rccl_abi() else 'torch.distributed (%s)' % tr.dp.backend))
    ar_ms = _time_all_reduce(tr, world, barrier, dev)
    if ar_ms is not None:
        rec['all_reduce_ms_measured'] = ar_ms
        rec['all_reduce_bytes'] = int(tr._flat_exchange.numel()) * 4
        rec['exchange_check'] = _compare_exchanges(tr, world, rank, barrier, dev, crash_line=crash_line, rec=rec)
    if world > 1:
        weak, _, _ = run(per_gpu_batch * world, 'weak')
        weak['convention'] = 'reference: batch_size = n_gpu * batch_size (Azure/train_tf.py:163-164)'
        rec['weak'] = weak
    if world == 1 and share_of:
        # the share of one of `share_of` GPUs of the same global batch, on this one GPU (no exchange): the strong-scaling
        # projection adds an ASSUMED 40 us for the one 756 KB all-reduce over xGMI (no multi-GPU box to measure it on; the N > 1
        # runs of this record print all_reduce_ms_measured)
        nb = max(1, per_gpu_batch // share_of)
        xs, ys = x[:nb].contiguous(), y[:nb].contiguous()
        for _ in range(warmup):
            tr.train_on_shard(xs, ys, nb, return_device=True)
        dts = _sync_time(lambda: tr.train_on_shard(xs, ys, nb, return_device=True), steps)
        ms8 = 1e3 * dts / steps
        sh = {'batch': nb, 'ms_per_step': ms8, 'value': nb * steps / dts, 'unit': 'samples/s',
              'step_form': tr._graph_ok(nb) or 'python', 'input': 'device-resident tensors (no loader)',
              'algorithmic_frac': flops_s * nb * steps / dts / 1e12 / PEAK_F32_MFMA_TFLOPS,
              'assumed_all_reduce_ms': 0.04, 'all_reduce': 'ASSUMED, not measured (single GPU)',
              'projected_speedup_%d_gpus' % share_of: rec['ms_per_step'] / (ms8 + 0.04)}
        mf = _train_exec_frac(nb)
        if mf:
            sh['executed_frac'] = mf * 2048.0 * steps / dts / 1e12 / PEAK_F32_MFMA_TFLOPS
        rec['share_of_%d_gpus' % share_of] = sh
    return rec


def sub_train_loader_fed(grid, cin, dev, batches=(64, 8), samples=2560, epochs=3):
    """BASELINE config 3 the way the reference drives it: fit_generator(DataGenerator(batch, shuffle=True), ...)
    (examples/train.py:262-263, DLWP/model/models.py:216-228) over a host-resident float32 training set -- every step's rows cross
    the link (H2D included), against the same step on device-resident tensors.  Single rank."""
    from dlwp_amd.model import ArrayDataset, DataGenerator
    out = {}
    for b in batches:
        d = build_model(grid, cin, lr=1e-4)
        tr = d.model._trainer
        rng = np.random.default_rng(0)
        n = samples - samples % b          # the same training set for every batch size: 40 steps per epoch at 64, 320 at 8
        P = rng.standard_normal((n, 2, cin // 2) + tuple(grid), dtype=np.float32)
        T = rng.standard_normal((n, 2, cin // 2) + tuple(grid), dtype=np.float32)
        gen = DataGenerator(d, ArrayDataset(P, T), batch_size=b, shuffle=True)
        steps = len(gen)
        d.fit_generator(gen, epochs=1, verbose=0)                   # warm-up: buffers, page-locking, the recorded step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.fit_generator(gen, epochs=epochs, verbose=0)
        torch.cuda.synchronize()
        fed = (time.perf_counter() - t0) / (epochs * steps)
        x = torch.from_numpy(P[:b].reshape((b, cin) + tuple(grid))).to(dev)
        y = torch.from_numpy(T[:b].reshape((b, cin) + tuple(grid))).to(dev)
        for _ in range(10):
            tr.train_on_shard(x, y, b, return_device=True)
        torch.cuda.synchronize()
        k = 60
        t0 = time.perf_counter()
        for _ in range(k):
            tr.train_on_shard(x, y, b, return_device=True)
        host = (time.perf_counter() - t0) / k                       # launches issued, nothing waited for
        torch.cuda.synchronize()
        res = (time.perf_counter() - t0) / k
        out['batch_%d' % b] = {
            'loader_fed_ms_per_step': 1e3 * fed, 'device_resident_ms_per_step': 1e3 * res, 'loader_over_resident': fed / res,
            'host_ms_per_step': 1e3 * host, 'step_form': tr._graph_ok(b) or 'python', 'samples_per_s': b / fed,
            'h2d_mb_per_step': 2 * b * cin * grid[0] * grid[1] * 4 / 1e6,
            'feed': ('host gather into pinned buffers (dlwp_host_gather_rows) + H2D copy'
                     if gen.batch_sources() is not None else 'generator batches + H2D copy'),
            'steps_timed': epochs * steps}
        del d, tr, gen, P, T
        torch.cuda.empty_cache()
    out['definition'] = ('fit_generator over DataGenerator(shuffle=True) on a host-resident float32 set, wall / step over whole epochs '
                         '(callbacks, the epoch-end loss read-back included); device_resident: train_on_shard on tensors already in '
                         'HBM; host_ms_per_step: time to ISSUE a device-resident step')
    return out


def sub_cfg5(world, rank, barrier, dev, total_members=32, forwards=40, share_of=8):
    """BASELINE config 5: 1-degree 180 x 360 x 12-channel U-Net, a 32-member perturbed-IC ensemble IN TOTAL, 40 forwards
    (80 six-hour steps) as one hipGraph per rank; members sharded over the ranks (4 per GPU at 8).  Strong scaling."""
    from dlwp_amd.parallel import shard_bounds
    grid, cin = (180, 360), 12
    d = build_model(grid, cin)
    net = d.model
    lo, hi = shard_bounds(total_members, rank, world)
    base = torch.randn((1, cin) + grid, generator=torch.Generator().manual_seed(0))
    pert = torch.randn((total_members, cin) + grid, generator=torch.Generator().manual_seed(1))
    s0 = (base + 0.01 * pert)[lo:hi].contiguous().to(dev)
    ser = net.rollout_on_device(s0, forwards)
    _warm_until_steady(lambda: net.rollout_on_device(s0, forwards), batch=1, min_s=0.1)   # (per rank: no collective in a rollout)
    dt = _sync_time(lambda: net.rollout_on_device(s0, forwards), 3, barrier, world, dev)
    flops = net.plan.conv_flops_per_sample()
    rec = {'value': total_members * forwards * 2 * 3 / dt, 'unit': '6-h forecast steps/s', 'scaling': 'strong',
           'total_members': total_members, 'members_per_gpu': hi - lo, 'forwards': forwards,
           'ms_per_rollout': 1e3 * dt / 3, 'finite': bool(torch.isfinite(ser[-1]).all().item()),
           'algorithmic_tflops_per_gpu': total_members * forwards * 3 * flops / dt / 1e12 / world}
    rec.update(operating_point(net, hi - lo, (hi - lo) * forwards * 3 / dt))
    if world == 1 and share_of:
        # the 8-GPU share of the same ensemble on this one GPU: what strong scaling over `share_of` GPUs can reach at best
        m = max(1, total_members // share_of)
        s1 = s0[:m].contiguous()
        net.rollout_on_device(s1, forwards)
        _warm_until_steady(lambda: net.rollout_on_device(s1, forwards), batch=1, min_s=0.1)
        dts = _sync_time(lambda: net.rollout_on_device(s1, forwards), 3)
        sh = {'members': m, 'value': m * forwards * 2 * 3 / dts, 'unit': '6-h forecast steps/s', 'ms_per_rollout': 1e3 * dts / 3}
        sh.update(operating_point(net, m, m * forwards * 3 / dts))
        sh['projected_speedup_%d_gpus' % share_of] = share_of * sh['value'] / rec['value']
        rec['share_of_%d_gpus' % share_of] = sh
    if world == 1:
        # the same ensemble through the API (numpy in, numpy out): 4 GB of series for 31 ms of rollout -- the link decides
        x = s0.cpu().numpy()
        d.predict_timeseries(x, 2 * forwards)
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            out = d.predict_timeseries(x, 2 * forwards)
            times.append(time.perf_counter() - t0)
        rec['host_visible'] = {'value': total_members * forwards * 2 / min(times), 'unit': '6-h forecast steps/s',
                               'series_bytes': int(out.nbytes), 's_per_call': min(times), 'series_gbs': out.nbytes / min(times) / 1e9,
                               'note': 'predict_timeseries(numpy) -> numpy, streamed return (DESIGN.md 5.15); r4 returned this series '
                                       'through one pageable copy after the rollout'}
        del out
    return rec


# --------------------------------------------------------------------------------------------------------------------- #
# self-launch: `python bench.py --gpus N` with no torchrun environment starts its own N ranks
# --------------------------------------------------------------------------------------------------------------------- #

def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, timeout_s):
    """One process per GPU (reference: keras.utils.multi_gpu_model replicates inside one process,
    DLWP/model/models.py:104-109).  The parent only starts the ranks with the torchrun environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT), forwards rank 0's stdout -- the ONE JSON line -- and returns the worst exit
    code.  A rank that dies takes the others (its exact PIDs) with it so that nothing waits in a collective forever."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DLWP_BENCH_CHILD='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() - t0 > timeout_s:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            if not failed:
                sys.stderr.write('bench.py: ranks did not finish within %.0f s\n' % timeout_s)
                return 124
            break
        time.sleep(0.05)
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--members', type=int, default=256, help='ensemble members / initial conditions PER GPU (weak scaling) '
                                                             'or IN TOTAL over the GPUs (--scaling strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: --members per GPU (the default headline); strong: --members in total, sharded over the ranks')
    ap.add_argument('--launch-timeout', type=float, default=1500.0, help='self-launched ranks are stopped after this many seconds')
    ap.add_argument('--forwards', type=int, default=28, help='model applications per rollout (28 = 14 days)')
    ap.add_argument('--grid', default='88x180')
    ap.add_argument('--channels', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the sub-records (members 1/8, host-visible, layer 1, '
                                                             'training, config 5)')
    ap.add_argument('--extras-timeout', type=float, default=240.0,
                    help='N > 1: seconds the collective sub-records may take before the line is printed without them')
    ap.add_argument('--activation-dtype', default='float32', choices=['float32', 'bfloat16'],
                    help="storage of the tensors between the layers (BASELINE config 4 uses bfloat16; the headline "
                         "config 2 is float32)")
    a = ap.parse_args()
    grid = tuple(int(v) for v in a.grid.split('x'))

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (under torch.distributed.run the environment is there)
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:], a.launch_timeout))

    from dlwp_amd import parallel
    rank, world, local = parallel.init()
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or without a '
                         'torchrun environment (bench.py then starts its own ranks)' % (a.gpus, world, a.gpus))
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
    if a.scaling == 'strong':          # --members in total: this rank's contiguous shard (no collective in a rollout)
        lo, hi = parallel.shard_bounds(a.members, rank, world)
        m_local, m_total = hi - lo, a.members
        if m_local == 0:
            raise SystemExit('--scaling strong needs at least one member per rank (%d members, %d ranks)' % (a.members, world))
    else:
        m_local, m_total = a.members, a.members * world
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    base = torch.randn((1, a.channels) + grid, generator=torch.Generator().manual_seed(0))
    state0 = (base + 0.01 * torch.randn((m_local, a.channels) + grid, generator=g)).to(dev)

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

    six_hour_steps = m_total * a.forwards * 2 * a.steps
    value = six_hour_steps / dt
    fwd_per_s = m_total * a.forwards * a.steps / dt
    out = {
        'metric': '6-h forecast steps/sec on 91x180x4-chan state (closed grid 88x180), predict_timeseries rollout',
        'value': value, 'unit': '6-h forecast steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': a.scaling, 'vs_baseline': None,
        'dtype': 'f32' if a.activation_dtype == 'float32' else 'f32 arithmetic, bf16 activation storage',
        'data': 'synthetic',
        'config': {'workload': 'cfg2: 2-deg %dx%d x%d-chan sequential PeriodicPadding2D U-Net (188996 params), fp32, '
                               '%d-forward (14-day) predict_timeseries rollout as one hipGraph, %d members per GPU%s'
                               % (grid[0], grid[1], a.channels, a.forwards, m_local,
                                  ' (%d in total, strong scaling)' % m_total if a.scaling == 'strong' else ''),
                   'members_per_gpu': m_local, 'members_total': m_total, 'forwards_per_rollout': a.forwards, 'time_dim': 2,
                   'grid': list(grid), 'channels': a.channels, 'launches_per_forward': net.infer_plan.n_launches,
                   'parallelism': 'members sharded over %d GPU(s), no collective' % world,
                   'inference_plan': ('Winograd F(2x2,3x3) on the 3x3 layers; the decoder layers that read an up-sampled '
                                      'tensor are restated on the low-resolution tensor (same function, DESIGN.md 5.7; '
                                      'DLWP_RESTATE_UPSAMPLED=0 runs the reference formulation)')},
        'forwards_per_s': fwd_per_s,
        'conv_mflop_per_forward_per_member': flops_fwd / 1e6,
        'finite': finite,
    }
    state = {'printed': False}
    lock = threading.Lock()

    def emit():
        with lock:
            if not state['printed'] and rank == 0:
                print(json.dumps(out))
                sys.stdout.flush()
            state['printed'] = True

    if rank == 0:
        rows = time_layers(net, m_local)
        executed_fwd = sum(r['executed_flops'] for r in rows) / m_local            # per member per forward
        per_gpu_fwd_per_s = fwd_per_s / world
        out['forward'] = {
            'executed_tflops': per_gpu_fwd_per_s * executed_fwd / 1e12,
            'mfma_util_frac': per_gpu_fwd_per_s * executed_fwd / 1e12 / PEAK_F32_MFMA_TFLOPS,
            'algorithmic_tflops': per_gpu_fwd_per_s * flops_fwd / 1e12,
            'algorithmic_speedup': flops_fwd / executed_fwd,
            'algorithmic_gbs': per_gpu_fwd_per_s * bytes_fwd / 1e9, 'hbm_frac': per_gpu_fwd_per_s * bytes_fwd / 1e9 / PEAK_HBM_GBS,
            'per_gpu': True, 'sum_of_kernel_ms': sum(r['ms'] for r in rows),
            'ms_per_forward_in_graph': 1e3 * dt / a.steps / a.forwards,
            'definition': 'executed = matrix-core FLOPs the kernels issue (MFMA x 2048, padding included); mfma_util_frac = '
                          'executed / wall / 157.3 TFLOP/s; algorithmic = direct-convolution FLOPs of the reference graph '
                          '(1597.7 MFLOP per forward per member)'}
        out['roofline'] = roofline_of(rows, m_local)
        cc = pmc_mfma_crosscheck(rows, m_local)
        if cc:
            out['roofline']['pmc_crosscheck'] = cc
        out['layers'] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                          if k not in ('flops', 'bytes', 'executed_flops', 'useful_flops', 'launch_flops', 'bf16_matrix')} for r in rows]
    if not a.no_extras:
        sub = {}
        if rank == 0:
            try:
                sub.update(sub_small_batches(net, grid, a.channels, a.forwards))
                sub['host_visible'] = sub_host_visible(d, grid, a.channels, m_local, a.forwards)
                if grid == (88, 180) and a.channels == 4:
                    sub['layer1_at_91x180'] = sub_layer1_nominal(net, m_local)
                if world == 1:
                    sub['timeseries_estimator'] = sub_timeseries_estimator(grid, m_local, a.forwards, value)
                if world == 1:
                    sub['pad_hbm'] = sub_pad_hbm(m_local)
                    sub['recurrent_cfg4_bf16'] = sub_cfg4()
                    sub['recurrent_cfg4_bf16_m32'] = sub_cfg4(members=32)     # a member count that fills the chip (config 5's)
                    sub['row_connected_output_layer'] = sub_row_connected(grid, a.channels, m_local, a.forwards)
            except Exception as e:  # noqa: BLE001  (a sub-record must never cost the headline line)
                sub['error_local'] = repr(e)
        out['sub_records'] = sub
        # the collective sub-records: every rank takes part; a watchdog prints the line without them if they stall
        if world > 1:
            def watchdog():
                time.sleep(a.extras_timeout)
                sub['error_collective'] = 'timed out after %.0f s' % a.extras_timeout
                emit()
                os._exit(0)
            threading.Thread(target=watchdog, daemon=True).start()
        try:
            del series
            net.__dict__.pop('_rollouts', None)
            torch.cuda.empty_cache()
            def crash_line(on, rec=None):
                """(rank 0) park / clear the line as it stands -- with the training record measured so far -- for the library's
                crash handler (dlwp_set_crash_message)"""
                if rank != 0:
                    return
                from dlwp_amd import _lib
                if on:
                    _lib.lib.dlwp_set_crash_message(json.dumps(dict(out, sub_records=dict(sub, train_cfg3=rec))).encode('utf-8'))
                else:
                    _lib.lib.dlwp_set_crash_message(None)
            sub['train_cfg3'] = sub_train(grid if grid == (88, 180) else (88, 180), 4, world, rank, barrier, dev, crash_line=crash_line)
            if world == 1:
                sub['train_cfg3_loader_fed'] = sub_train_loader_fed((88, 180), 4, dev)
                # the strong-scaling projection the way the reference drives training (fit_generator: H2D included), beside the
                # device-resident one
                lf, sh = sub['train_cfg3_loader_fed'], sub['train_cfg3'].get('share_of_8_gpus')
                if sh and 'batch_64' in lf and 'batch_8' in lf:
                    sh['loader_fed'] = {'ms_per_step_64': lf['batch_64']['loader_fed_ms_per_step'],
                                        'ms_per_step_8': lf['batch_8']['loader_fed_ms_per_step'],
                                        'projected_speedup_8_gpus': lf['batch_64']['loader_fed_ms_per_step'] /
                                        (lf['batch_8']['loader_fed_ms_per_step'] + sh['assumed_all_reduce_ms']),
                                        'all_reduce': 'ASSUMED %.2f ms, not measured (single GPU)' % sh['assumed_all_reduce_ms']}
            sub['ensemble_cfg5'] = sub_cfg5(world, rank, barrier, dev)
        except Exception as e:  # noqa: BLE001
            sub['error_collective'] = repr(e)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(grid, a.channels, a.forwards, weights_np)
    # what the stream probes of every rank found (util.distinct_streams: side streams of the training step, the loader's and the
    # rollout's copy streams): candidates passed over because they shared a hardware queue, streams that had to share one anyway
    from dlwp_amd import util as _util
    probes = list(_util.probe_log)
    if world > 1 and 'error_collective' not in out.get('sub_records', {}):
        try:
            box = [None] * world
            torch.distributed.all_gather_object(box, probes)
            probes = box
        except Exception as e:  # noqa: BLE001
            probes = {'error': repr(e)[:200], 'rank0': probes}
    out['stream_probes'] = probes
    emit()
    if world > 1 and 'error_collective' in out.get('sub_records', {}):
        os._exit(0)                  # a rank may be gone: no barrier to wait in
    barrier()


if __name__ == '__main__':
    main()
