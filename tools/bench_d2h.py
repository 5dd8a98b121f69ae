#!/usr/bin/env python3
"""Device -> page-locked host transfer forms, alone and beside a running rollout (what is the ceiling, and which
form reaches it under compute).  GPU only.

  1-D copy-engine transfers (hipMemcpyAsync) of 256 / 65 / 32 / 8 MB pieces, one stream and two streams alternating;
  the strided 2-D copy r4 used (hipMemcpy2DAsync: a member chunk's rows of a (T, N, ...) series);
  dlwp_store2d_to_host (a kernel storing into the mapped result array) at 4 ... 128 workgroups, contiguous and transposing;
  the best of each BESIDE the 256-member rollout graph on the main stream: its rate, and what it costs the rollout.

    python tools/bench_d2h.py > gpurun_out/d2h.json"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from dlwp_amd import _lib, util  # noqa: E402


def timed(fn, streams, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main = torch.cuda.current_stream()
        e0.record(main)
        for s in streams:
            s.wait_stream(main)
        fn()
        for s in streams:
            main.wait_stream(s)
        e1.record(main)
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        best = t if best is None or t < best else best
    return best


def main():
    dev = torch.device('cuda', 0)
    h = _lib.handle(0)
    total = 1 << 30                                      # 1 GiB moved per measurement
    src = torch.randn(total // 4, device=dev)
    host = torch.empty(total // 4, dtype=torch.float32, pin_memory=True)
    host.zero_()
    down = util.d2h_streams(dev)
    out = {'bytes': total, 'rows': []}

    def rec(name, t, **kw):
        r = dict(form=name, ms=round(1e3 * t, 3), GBs=round(total / t / 1e9, 2), **kw)
        out['rows'].append(r)
        print(json.dumps(r), file=sys.stderr)

    # ---- 1-D copy-engine transfers
    for piece_mb in (256, 64, 32, 8):
        q = (piece_mb << 20) // 4
        for ns in (1, 2):
            def run():
                for i, lo in enumerate(range(0, total // 4, q)):
                    with torch.cuda.stream(down[i % ns]):
                        host[lo:lo + q].copy_(src[lo:lo + q], non_blocking=True)
            rec('dma_1d', timed(run, down[:ns]), piece_mb=piece_mb, streams=ns)
    # ---- the strided 2-D copy of r4: rows of 8 MB at a pitch of 32 MB (4 member chunks side by side)
    rows, width = 32, 8 << 20
    def run2d():
        for c in range(4):
            _lib.check(_lib.lib.dlwp_copy2d_d2h_async(ctypes.c_void_p(host.data_ptr() + c * width), 4 * width,
                                                      ctypes.c_void_p(src.data_ptr() + c * rows * width), width, rows,
                                                      ctypes.c_void_p(down[0].cuda_stream)))
    rec('dma_2d_strided', timed(run2d, down[:1]), rows=rows, width_mb=8)
    # ---- the store kernel: contiguous 64 MB pieces, and the transposing form (rows of 126 KB, source pitch twice that)
    for blocks in (4, 8, 16, 32, 64, 128):
        q = 64 << 20
        def runk():
            for i, lo in enumerate(range(0, total, q)):
                _lib.check(_lib.lib.dlwp_store2d_to_host(h, ctypes.c_void_p(host.data_ptr() + lo), q, ctypes.c_void_p(src.data_ptr() + lo), q,
                                                         q, 1, blocks, ctypes.c_void_p(down[0].cuda_stream)))
        rec('store_kernel', timed(runk, down[:1]), blocks=blocks, piece_mb=64)
    run_b = 2 * 88 * 180 * 4
    n = 256
    for blocks in (8, 16, 32, 64):
        def runt():
            lo = 0
            i = 0
            while lo + 2 * n * run_b <= total:
                for j in range(2):
                    _lib.check(_lib.lib.dlwp_store2d_to_host(h, ctypes.c_void_p(host.data_ptr() + lo + j * n * run_b), run_b,
                                                             ctypes.c_void_p(src.data_ptr() + lo + j * run_b), 2 * run_b, run_b, n, blocks,
                                                             ctypes.c_void_p(down[i % 2].cuda_stream)))
                lo += 2 * n * run_b
                i += 1
        rec('store_kernel_transposing', timed(runt, down), blocks=blocks, streams=2)
    # ---- beside the rollout: 256 members, 28 calls on the main stream; the transfer forms on the copy streams
    d = build_model((88, 180), 4)
    x = torch.randn(256, 4, 88, 180, device=dev)
    d.predict_timeseries(x, 56, return_device=True)
    torch.cuda.synchronize()

    def rollout():
        d.predict_timeseries(x, 56, return_device=True)
    t_alone = timed(rollout, [])
    out['rollout_alone_ms'] = round(1e3 * t_alone, 3)
    print('rollout alone %.3f ms' % (1e3 * t_alone), file=sys.stderr)

    def beside(name, issue, streams, **kw):
        def run():
            rollout()
            issue()
        t = timed(run, streams)
        r = dict(form=name + '_beside_rollout', ms=round(1e3 * t, 3), rollout_alone_ms=round(1e3 * t_alone, 3),
                 GBs_if_copy_bound=round(total / t / 1e9, 2), **kw)
        out['rows'].append(r)
        print(json.dumps(r), file=sys.stderr)
    q = (64 << 20) // 4

    def dma2():
        for i, lo in enumerate(range(0, total // 4, q)):
            with torch.cuda.stream(down[i % 2]):
                host[lo:lo + q].copy_(src[lo:lo + q], non_blocking=True)
    beside('dma_1d_64mb_2streams', dma2, down)
    for blocks in (8, 16, 32):
        def k2(blocks=blocks):
            for i, lo in enumerate(range(0, total, 64 << 20)):
                _lib.check(_lib.lib.dlwp_store2d_to_host(h, ctypes.c_void_p(host.data_ptr() + lo), 64 << 20, ctypes.c_void_p(src.data_ptr() + lo),
                                                         64 << 20, 64 << 20, 1, blocks, ctypes.c_void_p(down[i % 2].cuda_stream)))
        beside('store_kernel_64mb_2streams', k2, down, blocks=blocks)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
