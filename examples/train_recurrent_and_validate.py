#!/usr/bin/env python3
"""
The reference's default flow -- examples/train.py with `model_is_recurrent = True` (ConvLSTM2D front end, l2 kernel
regulariser, anomaly-correlation loss) on a SeriesDataGenerator with an insolation input, followed by the
TimeSeriesEstimator rollout of examples/validate.py:191-205 -- written with the REFERENCE's imports on synthetic data.
Differences from a reference script: the `dlwp_amd.compat` import, and SeriesDataset standing in for xarray.open_dataset.

    python examples/train_recurrent_and_validate.py [--grid 32x64] [--times 160] [--epochs 2]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dlwp_amd.compat  # noqa: E402,F401  (registers DLWP.* and keras.* on the HIP back end)

from DLWP.model import DLWPNeuralNet, SeriesDataGenerator, TimeSeriesEstimator  # noqa: E402
from DLWP.custom import EarlyStoppingMin, RNNResetStates, anomaly_correlation_loss, latitude_weighted_loss  # noqa: E402
from keras.callbacks import History  # noqa: E402
from keras.regularizers import l2  # noqa: E402
from dlwp_amd.model import SeriesDataset  # noqa: E402  (stands in for xr.open_dataset(predictor_file))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', default='32x64')
    ap.add_argument('--times', type=int, default=160)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--batch-size', type=int, default=16)
    ap.add_argument('--forecast-steps', type=int, default=6)
    a = ap.parse_args()
    n_lat, n_lon = (int(v) for v in a.grid.split('x'))
    lambda_ = 1.e-4
    io_time_steps = 2

    #%% "Open data": one continuous 6-hourly series of two variables on one level
    rng = np.random.default_rng(0)
    raw = rng.standard_normal((a.times, 2, 1, n_lat, n_lon)).astype(np.float32)
    for t in range(1, a.times):                                     # a smooth, slowly advected signal
        raw[t] = 0.8 * np.roll(raw[t - 1], 1, axis=-1) + 0.2 * raw[t]
    dates = (np.datetime64('2007-01-01T00') + np.arange(a.times) * np.timedelta64(6, 'h')).astype('datetime64[s]')
    lats = np.linspace(80., -80., n_lat)
    coords = {'sample': dates, 'variable': np.array(['HGT', 'THICK']), 'level': np.array([500]), 'lat': lats,
              'lon': np.arange(0., 360., 360. / n_lon)}
    dims = ('sample', 'variable', 'level', 'lat', 'lon')
    n_val = a.times // 4
    train_data = SeriesDataset(raw[:-n_val], dict(coords, sample=dates[:-n_val]), dims)
    validation_data = SeriesDataset(raw[-n_val:], dict(coords, sample=dates[-n_val:]), dims)

    #%% Model object and generators (examples/train.py:92-133)
    dlwp = DLWPNeuralNet(is_convolutional=True, is_recurrent=True, time_dim=io_time_steps, scaler_type=None,
                         scale_targets=False)
    generator = SeriesDataGenerator(dlwp, train_data, input_time_steps=io_time_steps, output_time_steps=io_time_steps,
                                    add_insolation=True, batch_size=a.batch_size, shuffle=True, load='minimal')
    val_generator = SeriesDataGenerator(dlwp, validation_data, input_time_steps=io_time_steps,
                                        output_time_steps=io_time_steps, add_insolation=True, batch_size=a.batch_size)

    #%% Layers (examples/train.py:139-221): ConvLSTM2D front end + convolutional encoder / decoder
    cs = generator.convolution_shape            # (time, variables + insolation, lat, lon)
    cso = generator.output_convolution_shape    # (time, variables, lat, lon)
    cf = {'data_format': 'channels_first'}

    def block(k, filters, size, dilation, activation):
        return (('PeriodicPadding2D', ((0, k),), dict(cf)), ('ZeroPadding2D', ((k, 0),), dict(cf)),
                ('Conv2D', (filters, size), dict(cf, dilation_rate=dilation, padding='valid', activation=activation)))
    layers = (
        ('PeriodicPadding3D', ((0, 0, 2),), dict(cf, input_shape=cs)),
        ('ZeroPadding3D', ((0, 2, 0),), dict(cf)),
        ('ConvLSTM2D', (4 * cs[1], 3), dict(cf, dilation_rate=2, padding='valid', activation='tanh',
                                            return_sequences=True, kernel_regularizer=l2(lambda_))),
        ('Reshape', ((4 * cs[0] * cs[1], cs[2], cs[3]),), None),
    ) + block(2, 32, 3, 2, 'tanh') + (('MaxPooling2D', (2,), dict(cf)),) + block(1, 64, 3, 1, 'tanh') + \
        (('UpSampling2D', (2,), dict(cf)),) + block(1, 32, 3, 1, 'tanh') + \
        block(2, cso[0] * cso[1], 5, 1, 'linear') + (('Reshape', (cso,), None),)

    #%% Loss: latitude-weighted anomaly correlation with an MSE regulariser (examples/train.py:224-231)
    mean = generator.generate([], scale_and_impute=False)[1].mean(axis=0, keepdims=True)
    loss = latitude_weighted_loss(anomaly_correlation_loss(mean=mean.reshape((1, -1) + cso[-2:]), regularize_mean='mse'),
                                  lats=lats, output_shape=(cso[0] * cso[1],) + cso[-2:], axis=-2, weighting='midlatitude')
    dlwp.build_model(layers, loss=loss, optimizer='adam', metrics=['mae'], gpus=1)
    print(dlwp.base_model.summary())

    #%% Train (examples/train.py:253-263)
    history = History()
    early = EarlyStoppingMin(min_epochs=1, monitor='val_loss', min_delta=0., patience=5, restore_best_weights=True)
    dlwp.fit_generator(generator, epochs=a.epochs, verbose=1, validation_data=val_generator,
                       use_multiprocessing=True, callbacks=[history, RNNResetStates(), early])

    #%% Forecast every validation sample (examples/validate.py:191-205)
    estimator = TimeSeriesEstimator(dlwp, val_generator)
    time_series = estimator.predict(a.forecast_steps, verbose=0)
    print('forecast', time_series.dims, time_series.shape)
    return history.history, time_series


if __name__ == '__main__':
    main()
