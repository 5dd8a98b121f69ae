#!/usr/bin/env python3
"""The stages of the DataGenerator feed on this box, each alone: host row gather (dlwp_host_gather_rows, 1 / 8 threads) and the
pinned H2D copy (torch).  (r4 also measured a pull path -- a kernel reading the page-locked training set over the link, 51 GB/s
alone, slower inside a training loop: profiles/r4_feed_stages.json, r4_loader_timeline.txt -- removed in r5.)
    python tools/bench_feed_stages.py"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dlwp_amd import _lib
    dev = torch.device('cuda', 0)
    h = _lib.handle(0)
    row = 4 * 88 * 180
    n_src = 2560
    src = np.random.default_rng(0).standard_normal((n_src, row), dtype=np.float32)
    out = {'row_bytes': row * 4, 'cpus': os.cpu_count()}
    for n in (8, 64, 256):
        rows = np.random.default_rng(1).permutation(n_src)[:n].astype(np.int64)
        nbytes = n * row * 4
        pinned = torch.empty(n * row, dtype=torch.float32, pin_memory=True)
        dst = torch.empty(n * row, dtype=torch.float32, device=dev)
        rec = {}
        for thr in (1, 8):
            for _ in range(3):
                _lib.lib.dlwp_host_gather_rows(ctypes.c_void_p(pinned.data_ptr()), ctypes.c_void_p(src.ctypes.data),
                                               rows.ctypes.data_as(ctypes.c_void_p), n, row * 4, n_src, thr)
            t0 = time.perf_counter()
            for _ in range(10):
                _lib.lib.dlwp_host_gather_rows(ctypes.c_void_p(pinned.data_ptr()), ctypes.c_void_p(src.ctypes.data),
                                               rows.ctypes.data_as(ctypes.c_void_p), n, row * 4, n_src, thr)
            dt = (time.perf_counter() - t0) / 10
            rec['host_gather_%dthr' % thr] = {'ms': round(1e3 * dt, 3), 'GBs': round(nbytes / dt / 1e9, 1)}
        for _ in range(3):
            dst.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        rec['pinned_h2d_copy'] = {'ms': round(1e3 * dt, 3), 'GBs': round(nbytes / dt / 1e9, 1)}
        out['rows_%d' % n] = rec
    print(json.dumps(out))


if __name__ == '__main__':
    main()
