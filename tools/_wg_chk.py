import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dlwp_amd import _lib, ops
from oracle import np_ref
rng = np.random.default_rng(11)
cfgs = ops.wgrad_configs()
for (cin, cout, h, w, n) in [(33, 40, 12, 20, 3), (64, 64, 12, 40, 2), (32, 32, 8, 32, 1)]:
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    xp = np_ref.pad2d_modes(np.asarray(x, np.float64), (1, 1, 1, 1), 1, 1)
    dz = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    _, dw_ref, _ = np_ref.conv2d_grads(xp, np.zeros((3, 3, cin, cout)), dz, 1)
    cd = ops.make_conv(cout, 3, 3, 1, ops.make_pad(1, 1, 1, 1, 1, 1), ops.ACT_LINEAR)
    xs = _lib.Shape4(n, cin, h, w)
    print('case', cin, cout, h, w, 'pick', _lib.lib.dlwp_conv2d_wgrad_pick_config(_lib.handle(0), xs, ctypes.byref(cd)))
    for i, c in enumerate(cfgs):
        if i < 55 or (c[0], c[1]) != (3, 1):
            continue
        ops.force_wgrad_config(i)
        dwd = torch.empty((3, 3, cin, cout), dtype=torch.float32, device='cuda')
        ops.conv2d_bwd_weight(torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda(), dwd, cd, xs)
        g = dwd.cpu().numpy()
        err = np.abs(g - dw_ref)
        print('  cfg', i, c[2:6], 'max err %.3g' % err.max(), 'nan', int(np.isnan(g).sum()), 'bad taps', sorted(set(np.argwhere(err > 1e-3)[:, 0] * 3 + np.argwhere(err > 1e-3)[:, 1]))[:9],
              'bad ci', sorted(set(np.argwhere(err > 1e-3)[:, 2]))[:6], 'bad co', sorted(set(np.argwhere(err > 1e-3)[:, 3]))[:6])
    ops.force_wgrad_config(-1)
