#!/usr/bin/env python3
"""The stages of the DataGenerator feed on this box, each alone: host row gather (dlwp_host_gather_rows, 1 / 8 threads), pinned H2D
copy (torch), and the pull path (dlwp_gather_rows_h2d: a kernel reading the page-locked training set over the link).
    python tools/bench_pull.py"""
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
    dptr = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = _lib.lib.dlwp_host_register(ctypes.c_void_p(src.ctypes.data), src.nbytes, ctypes.byref(dptr))
    out['register_s'] = round(time.perf_counter() - t0, 3)
    out['register_rc'] = rc
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
        if rc == 0:
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                _lib.check(_lib.lib.dlwp_gather_rows_h2d(h, ctypes.c_void_p(dst.data_ptr()), dptr, rows.ctypes.data_as(ctypes.c_void_p),
                                                         n, row * 4, n_src, ctypes.c_void_p(st)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                _lib.check(_lib.lib.dlwp_gather_rows_h2d(h, ctypes.c_void_p(dst.data_ptr()), dptr, rows.ctypes.data_as(ctypes.c_void_p),
                                                         n, row * 4, n_src, ctypes.c_void_p(st)))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            ok = bool(np.array_equal(dst.cpu().numpy().reshape(n, row), src[rows]))
            rec['pull_kernel'] = {'ms': round(1e3 * dt, 3), 'GBs': round(nbytes / dt / 1e9, 1), 'equal': ok}
        out['rows_%d' % n] = rec
    print(json.dumps(out))


if __name__ == '__main__':
    main()
