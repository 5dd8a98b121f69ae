#!/usr/bin/env python3
"""Where the HOST spends a loader-fed training step: cProfile of `fit_generator(DataGenerator(batch, shuffle=True))` (consumer thread:
generator protocol, DeviceLoader slot handling and H2D enqueue, the recorded step's launch, the epoch accumulator, callbacks).  At 8
samples of the 88 x 180 grid the device needs 0.36 ms per step: a host loop that takes longer IS the step time.
usage: python tools/profile_fit_generator_host.py [--batch 8] [--samples 2048]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--samples', type=int, default=2048)
    ap.add_argument('--lines', type=int, default=30)
    a = ap.parse_args()
    from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((4, 88, 180)), loss='mse', optimizer='adam', metrics=['mae'])
    rng = np.random.default_rng(0)
    n = a.samples - a.samples % a.batch
    P = rng.standard_normal((n, 2, 2, 88, 180), dtype=np.float32)
    T = rng.standard_normal((n, 2, 2, 88, 180), dtype=np.float32)
    gen = DataGenerator(d, ArrayDataset(P, T), batch_size=a.batch, shuffle=True)
    d.fit_generator(gen, epochs=1, verbose=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.fit_generator(gen, epochs=1, verbose=0)
    torch.cuda.synchronize()
    print('loader-fed, unprofiled: %.4f ms per step' % (1e3 * (time.perf_counter() - t0) / len(gen)))
    pr = cProfile.Profile()
    pr.enable()
    d.fit_generator(gen, epochs=1, verbose=0)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    print('steps', len(gen))
    st.sort_stats('tottime').print_stats(a.lines)


if __name__ == '__main__':
    main()
