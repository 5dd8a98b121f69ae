"""
Forecast error measures of the validation scripts (reference DLWP/model/verify.py:17-102, called from
examples/validate.py / plot_forecasts.py on the output of predict_timeseries): plain numpy on host arrays -- the rollout
that produces the forecast is the device path, the error reduction over a few hundred MB is not hot.
Pinned by tests/golden/verify.npz (the reference's own functions run by oracle/make_golden.py).
"""
import numpy as np

_METHODS = ('mse', 'mae', 'rmse')


def _error(diff, method, axis):
    if method == 'mae':
        return np.nanmean(np.abs(diff), axis=axis)
    mse = np.nanmean(diff ** 2., axis=axis)
    return np.sqrt(mse) if method == 'rmse' else mse


def _check(method):
    if method not in _METHODS:
        raise ValueError("'method' must be 'mse', 'rmse', or 'mae'")


def forecast_error(forecast, valid, method='mse', axis=None):
    """Error of a time-series forecast (forecast step first).  `valid` either carries the same leading forecast-step axis
    -- then the mean runs over `axis` (default: everything but the step) -- or is the plain verification series, in which
    case forecast step f made from sample i is compared with valid[i + f]."""
    _check(method)
    if forecast.ndim == valid.ndim:
        ax = tuple(range(1, valid.ndim)) if axis is None else axis
        return _error(valid - forecast, method, ax)
    n_val = valid.shape[0]
    return np.array([_error(valid[f:] - forecast[f, :n_val - f], method, axis) for f in range(forecast.shape[0])])


def persistence_error(predictors, valid, n_fhour, method='mse', axis=None):
    """Error of forecasting "no change": step f compares valid[i + f] with predictors[i]."""
    _check(method)
    n = valid.shape[0]
    return np.array([_error(valid[f:] - predictors[:n - f], method, axis) for f in range(n_fhour)])


def climo_error(valid, n_fhour, method='mse', axis=None):
    """Error of forecasting the mean of the verification data itself, per forecast step (the reference shortens the
    sample window with the step, so the values differ slightly between steps)."""
    _check(method)
    n = valid.shape[0]
    mean = np.nanmean(valid, axis=0)
    return np.array([_error(valid[:n - f] - mean, method, axis) for f in range(n_fhour)])
