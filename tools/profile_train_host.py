#!/usr/bin/env python3
"""Where the HOST spends a training step (cProfile of Trainer.train_on_shard, eager): the step is host-bound below ~32 samples of
the 88 x 180 grid.  usage: python tools/profile_train_host.py [--batch 8]"""
import argparse
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DLWP_TRAIN_GRAPH', '0')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    a = ap.parse_args()
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    np.random.seed(1234)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers((4, 88, 180)), loss='mse', optimizer='adam', metrics=['mae'])
    x = torch.randn((a.batch, 4, 88, 180), device=d.model.device)
    tr = d.model._trainer
    for _ in range(10):
        tr.train_on_batch(x, x, return_device=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        tr.train_on_batch(x, x, return_device=True)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
