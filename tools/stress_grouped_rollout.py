#!/usr/bin/env python3
"""Reproducer hunt (r4): the full GPU suite died inside hipGraphLaunch of a grouped (forked) rollout graph in 4 of 17 runs once
rollouts allocated / freed uncached memory for split-K regions.  One process: N x [single-chain rollout with split launches:
create, launch, destroy] + [grouped rollout: create, launch, compare, destroy].  DLWP_UNCACHED_FREE=1 = the hipFree-per-rollout form.
    python tools/stress_grouped_rollout.py [iterations]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from dlwp_amd import ops
    from dlwp_amd.model import DLWPNeuralNet
    from dlwp_amd.presets import unet_layers
    cs = (4, 16, 24)
    np.random.seed(0)
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    d.build_model(unet_layers(cs), loss='mse', optimizer='adam')
    net = d.model
    x = torch.randn((6,) + cs, device=net.device)
    ops.set_splitk(0)
    want = net.rollout_on_device(x, 5, graph_cache=False).clone()
    ops.set_splitk(1)
    for it in range(n_it):
        a = net.rollout_on_device(x, 5, graph_cache=False).clone()          # split launches: an uncached region comes and goes
        for groups in (2, 3, 6):
            s0 = torch.empty_like(x)
            ser = torch.empty_like(want)
            g = net.executor.make_rollout(s0, ser, 5, groups=groups)
            s0.copy_(x)
            g.launch()
            torch.cuda.synchronize()
            assert torch.equal(ser, want), (it, groups)
            g.close()
        if it % 20 == 0:
            print('iteration', it, float((a - want).abs().max()), flush=True)
    print('done', n_it)


if __name__ == '__main__':
    main()
