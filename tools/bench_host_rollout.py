#!/usr/bin/env python3
"""Host-visible rate of DLWPNeuralNet.predict_timeseries (numpy in, numpy out) vs the device-resident rollout.  GPU only."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=256)
    ap.add_argument('--steps', type=int, default=56)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    d = build_model((88, 180), 4)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((a.members, 4, 88, 180)).astype(np.float32)
    out = d.predict_timeseries(x, a.steps)          # graph capture, buffers
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        out = d.predict_timeseries(x, a.steps)
        times.append(time.perf_counter() - t0)
    xd = torch.from_numpy(x).cuda()
    d.predict_timeseries(xd, a.steps, return_device=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        d.predict_timeseries(xd, a.steps, return_device=True)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / a.reps
    nbytes = out.nbytes
    print(json.dumps({'members': a.members, 'six_hour_steps': a.steps, 'series_bytes': nbytes,
                      'host_visible_s': min(times), 'host_visible_all_s': times, 'device_resident_s': dev,
                      'host_visible_steps_per_s': a.members * a.steps / min(times),
                      'device_steps_per_s': a.members * a.steps / dev,
                      'effective_d2h_gbs': nbytes / max(min(times) - dev, 1e-9) / 1e9}))


if __name__ == '__main__':
    main()
