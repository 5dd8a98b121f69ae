"""TimeSeriesEstimator.predict -> LabeledArray, wall time of the call (best of 5) and of the device loop alone."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dlwp_amd.model import DLWPNeuralNet, SeriesDataGenerator, SeriesDataset, TimeSeriesEstimator
from dlwp_amd.presets import unet_layers
grid, members, forwards = (88, 180), 256, 28
rng = np.random.default_rng(3)
n_t = members + 3
dates = (np.datetime64('2010-01-01T00') + np.arange(n_t) * np.timedelta64(6, 'h')).astype('datetime64[s]')
series = rng.standard_normal((n_t, 2, 1) + grid).astype(np.float32)
ds = SeriesDataset(series, {'sample': dates, 'variable': np.array(['z', 'tau']), 'level': np.array([500]),
                            'lat': np.linspace(88., -88., grid[0]), 'lon': np.arange(0., 360., 360. / grid[1])},
                   ('sample', 'variable', 'level', 'lat', 'lon'))
d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
d.build_model(unet_layers((6,) + grid, cout=4), loss='mse', optimizer='adam', metrics=['mae'])
gen = SeriesDataGenerator(d, ds, input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=64)
est = TimeSeriesEstimator(d, gen)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); est.predict(2 * forwards); ts.append(time.perf_counter() - t0)
print('api call: best %.1f ms, all %s' % (1e3 * min(ts), [round(1e3 * t, 1) for t in ts]))
