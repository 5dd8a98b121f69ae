"""bench.py's cpu_baseline leg (the oracle's torch-CPU restatement timed beside the GPU line) on a tiny grid: the record carries the
thread-count sweep (every count's rate in the line, the best one re-measured over a longer window)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_reports_its_thread_sweep():
    import torch
    import bench
    from oracle import np_ref
    from dlwp_amd.presets import unet_layers
    grid, cin = (8, 12), 4
    weights = np_ref.init_weights(unet_layers((cin,) + grid), cin, np.random.default_rng(0))
    before = torch.get_num_threads()
    try:
        rec = bench.cpu_baseline(grid, cin, 4, weights, budget_s=0.6, warm_s=0.05, sweep_s=0.15)
    finally:
        torch.set_num_threads(before)
    assert rec['kind'] == 'port' and rec['value'] > 0 and rec['unit'] == '6-h forecast steps/s'
    counts = sorted(int(k) for k in rec['sweep'])
    ncpu = os.cpu_count() or 1
    assert counts == sorted({min(ncpu, v) for v in (8, 16, 32)}) and rec['cores'] in counts
    assert all(v > 0 for v in rec['sweep'].values())
    assert rec['sweep'][str(rec['cores'])] == max(rec['sweep'].values())
