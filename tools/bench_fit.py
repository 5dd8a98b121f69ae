#!/usr/bin/env python3
"""Host-visible rate of DLWPNeuralNet.fit(x, y) on numpy arrays (config-3 shapes) with the training set resident in HBM
vs gathered on the host and uploaded per batch.  GPU only."""
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
    ap.add_argument('--samples', type=int, default=2560)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--epochs', type=int, default=2)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((a.samples, 4, 88, 180)).astype(np.float32)
    y = rng.standard_normal((a.samples, 4, 88, 180)).astype(np.float32)
    out = {'samples': a.samples, 'batch': a.batch, 'epochs': a.epochs}
    for name, frac in (('resident', 0.5), ('per_batch_upload', 0.0)):
        d = build_model((88, 180), 4)
        tr = d.model._trainer
        tr.resident_fraction = frac
        d.fit(x[:a.batch * 2], y[:a.batch * 2], batch_size=a.batch, epochs=1, verbose=0)      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.fit(x, y, batch_size=a.batch, epochs=a.epochs, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = a.epochs * (-(-a.samples // a.batch))
        out[name] = {'seconds': dt, 'ms_per_step': 1e3 * dt / steps, 'samples_per_s': a.epochs * a.samples / dt}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
