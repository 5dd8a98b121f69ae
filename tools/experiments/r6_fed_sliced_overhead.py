"""fed rollout: one graph of 28 calls vs 28 one-call graphs launched back to back (no copies): what the slicing itself costs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dlwp_amd.model import DLWPNeuralNet
from dlwp_amd.presets import unet_layers
grid, n, calls = (88, 180), 256, 28
d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
d.build_model(unet_layers((6,) + grid, cout=4), loss='mse', optimizer='adam')
net = d.model
src = [0, 1, 2, 3, 4, 5]
for m in range(2):
    for j in range(2):
        src[m * 3 + j] = -1 - (m * 2 + j)
sol = np.zeros((calls - 1, 2, 2) + grid, np.float32)
sol_map = [-1, -1, 0, -1, -1, 1]
x = torch.randn((n, 6) + grid, device='cuda')
for sliced in (False, True):
    ent = net._fed_entry(n, calls, src, 2, 2, sol, sol_map, None, sliced=sliced)
    net._fed_inputs(ent, calls, x, sol, None)
    def run():
        for g in ent[0]:
            g.launch()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print('sliced' if sliced else 'one graph', '%.2f ms per rollout' % (1e3 * (time.perf_counter() - t0) / 5))
