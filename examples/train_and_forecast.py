#!/usr/bin/env python3
"""
The reference's examples/train.py + examples/plot_forecasts.py flow (non-recurrent U-Net), written with the REFERENCE's
imports, on synthetic data (the reanalysis files are not part of either repository).  Only two things differ from a
reference script: the first import line, and ArrayDataset standing in for xarray.open_dataset.

    python examples/train_and_forecast.py [--grid 36x72] [--samples 256] [--epochs 3] [--latitude-dependent]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dlwp_amd.compat  # noqa: E402,F401  (registers DLWP.* and keras.* on the HIP back end)

from DLWP.model import DLWPNeuralNet, DataGenerator, ArrayDataset  # noqa: E402
from DLWP.custom import EarlyStoppingMin, RNNResetStates  # noqa: E402
from DLWP.util import save_model, load_model, train_test_split_ind  # noqa: E402
from keras.callbacks import History  # noqa: E402
from keras.losses import mean_squared_error  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', default='36x72')
    ap.add_argument('--samples', type=int, default=256)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--model-file', default='/tmp/dlwp_amd_example')
    ap.add_argument('--latitude-dependent', action='store_true',
                    help="the output layer is DLWP.custom.RowConnected2D (examples/train_functional.py:53, 191-196)")
    a = ap.parse_args()
    lat, lon = (int(v) for v in a.grid.split('x'))

    #%% "Open data": (sample, time_step, varlev, lat, lon), two variables, two input / output time steps
    rng = np.random.default_rng(0)
    base = rng.standard_normal((a.samples + 2, 2, lat, lon)).astype(np.float32)
    base = 0.5 * base + 0.25 * np.roll(base, 1, axis=-1) + 0.25 * np.roll(base, -1, axis=-1)
    predictors = np.stack([base[:-2], base[1:-1]], axis=1)[:a.samples]
    targets = np.stack([base[1:-1], base[2:]], axis=1)[:a.samples]
    n_sample = predictors.shape[0]
    train_set, val_set = train_test_split_ind(n_sample, n_sample // 4, method='last')
    train_data = ArrayDataset(predictors[train_set], targets[train_set])
    validation_data = ArrayDataset(predictors[val_set], targets[val_set])

    #%% Build a model and the data generators (examples/train.py:92-133)
    dlwp = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    generator = DataGenerator(dlwp, train_data, batch_size=a.batch_size, shuffle=True)
    val_generator = DataGenerator(dlwp, validation_data, batch_size=a.batch_size)

    #%% Compile the model structure (examples/train.py:159-219, non-recurrent part)
    cs = generator.convolution_shape
    cf = {'data_format': 'channels_first'}

    def block(k, filters, size, dilation, activation):
        return (('PeriodicPadding2D', ((0, k),), dict(cf)), ('ZeroPadding2D', ((k, 0),), dict(cf)),
                ('Conv2D', (filters, size), dict(cf, dilation_rate=dilation, padding='valid', activation=activation)))
    layers = list(block(2, 32, 3, 2, 'tanh'))
    layers[0][2]['input_shape'] = cs
    layers += [('MaxPooling2D', (2,), dict(cf))] + list(block(1, 64, 3, 1, 'tanh'))
    layers += [('MaxPooling2D', (2,), dict(cf))] + list(block(1, 128, 3, 1, 'tanh'))
    layers += [('UpSampling2D', (2,), dict(cf))] + list(block(1, 64, 3, 1, 'tanh'))
    layers += [('UpSampling2D', (2,), dict(cf))] + list(block(2, 32, 3, 2, 'tanh'))
    if a.latitude_dependent:        # per-latitude filters in the output layer (examples/train_functional.py:191-196)
        layers += [('PeriodicPadding2D', ((0, 2),), dict(cf)), ('ZeroPadding2D', ((2, 0),), dict(cf)),
                   ('RowConnected2D', (cs[0], 5), dict(cf, padding='valid', activation='linear'))]
    else:
        layers += list(block(2, cs[0], 5, 1, 'linear'))
    try:
        dlwp.build_model(tuple(layers), loss=mean_squared_error, optimizer='adam', metrics=['mae'], gpus=1)
    except ValueError:
        for layer in dlwp.base_model.layers:
            print(layer.name, layer.output_shape)
        raise
    dlwp.base_model.summary()

    #%% Train, evaluate, and save the model (examples/train.py:249-277)
    start_time = time.time()
    history = History()
    early = EarlyStoppingMin(min_epochs=1, monitor='val_loss', min_delta=0., patience=50, restore_best_weights=True, verbose=1)
    dlwp.fit_generator(generator, epochs=a.epochs, verbose=1, validation_data=val_generator, use_multiprocessing=True,
                       callbacks=[history, RNNResetStates(), early])
    print("\\nTrain time -- %s seconds --" % (time.time() - start_time))
    save_model(dlwp, a.model_file, history=history)
    score = dlwp.evaluate(*val_generator.generate([], scale_and_impute=False), verbose=0)
    print('Validation loss:', score[0])
    print('Validation mean absolute error:', score[1])

    #%% Forecast (examples/plot_forecasts.py:225-239)
    dlwp2, hist = load_model(a.model_file, history=True)
    p_val, t_val = DataGenerator(dlwp2, validation_data, batch_size=216).generate([], scale_and_impute=False)
    time_series = dlwp2.predict_timeseries(p_val, 8)
    print('forecast series', time_series.shape, 'finite:', bool(np.isfinite(time_series).all()))
    assert np.array_equal(time_series, dlwp.predict_timeseries(p_val, 8)), 'reloaded model must forecast identically'
    return score, time_series


if __name__ == '__main__':
    main()
