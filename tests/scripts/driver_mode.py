"""A PLAIN single-process script (no launcher) that asks for two GPUs the way the reference's scripts do --
build_model(..., gpus=2), examples/train_generator.py:243 -- run by tests/test_gpu_parallel.py with DLWP_SHARE_GPUS=1.
usage: driver_mode.py MODE N_GLOBAL OUT.npz [functional]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    mode, n_global, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    functional = len(sys.argv) > 4 and sys.argv[4] == 'functional'
    import torch
    from dlwp_amd.model import DLWPFunctional, DLWPNeuralNet
    from dlwp_amd.training import Adam
    from tests.nets import unet_layers
    from tests.test_gpu_parallel import CS, _data, _scenario
    np.random.seed(1000)
    layers = unet_layers(CS, widths=(8, 16, 16, 16, 8))
    if functional:
        from dlwp_amd.engine import Sequential
        from dlwp_amd import layers as L, util
        net = Sequential()
        for name, args, kwargs in layers:
            try:
                cls = util.get_from_class('dlwp_amd.layers', name)
            except (ImportError, AttributeError):
                cls = util.get_from_class('dlwp_amd.custom', name)
            net.add(cls(*args, **(kwargs or {})))
        d = DLWPFunctional(is_convolutional=True, is_recurrent=False, time_dim=2)
        d.build_model(net, loss='mse', optimizer=Adam(lr=1e-3), metrics=['mae'], gpus=2)
    else:
        d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
        d.build_model(layers, loss='mse', optimizer=Adam(lr=1e-3), metrics=['mae'], gpus=2)
    assert d.model._dp.world == 2 and d.model._driver is not None
    w0 = [w.copy() for w in d.model.get_weights()]
    logs = _scenario(d, mode, n_global)
    torch.cuda.synchronize()
    w1 = d.model.get_weights()
    x, _ = _data(5, seed=9)
    series = d.predict_timeseries(x, 4)                 # members sharded 3 + 2 over the ranks, gathered here
    d.model.set_weights(w0)                             # mirrored: the replicas re-align at the next step ...
    again = d.model.train_on_batch(*_data(8))           # ... which is a collective one
    np.savez(out, logs=np.asarray(logs, dtype=np.float64), iters=d.model.optimizer.iterations, series=series,
             again=np.asarray(again, dtype=np.float64), n_w=len(w0), **{'w0_%d' % i: w for i, w in enumerate(w0)},
             **{'w1_%d' % i: w for i, w in enumerate(w1)})


if __name__ == '__main__':
    main()
