"""SeriesDataGenerator + util.insolation against golden vectors produced by the reference's own code (oracle/make_golden.py
executes DLWP/model/generators.py:323-629 and DLWP/util.py:300-352 under the numpy stub; tests/golden/series.npz)."""
import os

import numpy as np
import pandas as pd
import pytest

from dlwp_amd import util
from dlwp_amd.model import DLWPNeuralNet, LabeledArray, SeriesDataGenerator, SeriesDataset

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'series.npz'))
DATES = G['dates'].astype('datetime64[s]')

CASES = {
    'a': dict(rec=False, kw=dict(input_time_steps=2, output_time_steps=2, batch_size=4)),
    'b': dict(rec=True, kw=dict(input_time_steps=2, output_time_steps=2, batch_size=4)),
    'c': dict(rec=False, kw=dict(input_sel={'variable': ['z']}, output_sel={'variable': ['t'], 'level': [850]},
                                 input_time_steps=3, output_time_steps=1, interval=2, batch_size=5)),
    'd': dict(rec=False, kw=dict(input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4)),
    'e': dict(rec=True, kw=dict(input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4)),
    'f': dict(rec=False, kw=dict(input_time_steps=2, output_time_steps=2, sequence=3, batch_size=3)),
    'g': dict(rec=False, kw=dict(input_time_steps=1, output_time_steps=1, batch_size=6, shuffle=True)),
}


def _ds(series=None):
    s = G['S'] if series is None else series
    return SeriesDataset(s, {'sample': DATES[:s.shape[0]], 'variable': np.array(['z', 't']), 'level': np.array([500, 850]),
                             'lat': G['lat'], 'lon': G['lon']}, ('sample', 'variable', 'level', 'lat', 'lon'))


def _gen(tag, ds=None):
    case = CASES[tag]
    m = DLWPNeuralNet(is_convolutional=True, is_recurrent=case['rec'], time_dim=case['kw']['input_time_steps'],
                      scaler_type=None, scale_targets=False)
    np.random.seed(7)
    return SeriesDataGenerator(m, ds or _ds(), **case['kw'])


def test_insolation_matches_the_reference():
    lat, lon = G['lat'].copy(), G['lon'].copy()
    sol = util.insolation(DATES, lat, lon)
    assert sol.dtype == np.float32 and sol.shape == (14, 6, 8)
    assert np.array_equal(lat, G['lat'])                                   # inputs untouched
    assert np.abs(sol - G['insolation']).max() < 1e-6
    assert np.abs(util.insolation(DATES[:3], lat, lon, S=2.) - G['insolation_S2']).max() < 2e-6
    assert (sol >= 0).all() and (sol == 0).any() and sol.max() < 1.1      # night side clipped
    # 2-D lat / lon of one shape are accepted, mixed ranks are not
    lon2, lat2 = np.meshgrid(lon, lat)
    assert np.array_equal(util.insolation(DATES[:2], lat2, lon2), sol[:2])
    with pytest.raises(ValueError):
        util.insolation(DATES[:2], lat2, lon)
    assert util.day_of_year(pd.Timestamp('2003-01-02 12:00')) == 1.5


@pytest.mark.parametrize('tag', sorted(CASES))
def test_series_generator_matches_the_reference(tag):
    g = _gen(tag)
    assert len(g) == int(G['%s_len' % tag])
    for prop in ('shape', 'dense_shape', 'convolution_shape', 'shape_2d', 'output_shape', 'output_dense_shape',
                 'output_convolution_shape', 'output_shape_2d'):
        assert tuple(getattr(g, prop)) == tuple(G['%s_%s' % (tag, prop)]), prop
    assert g.n_features == int(G['%s_n_features' % tag]) and g.output_n_features == int(G['%s_output_n_features' % tag])
    assert np.array_equal(g._indices, G['%s_indices' % tag])               # shuffle order of the legacy RandomState
    for b in (0, len(g) - 1):
        X, y = g[b]
        assert X.dtype == np.float32 and np.array_equal(X, G['%s_X%d' % (tag, b)])
        if isinstance(y, list):
            assert len(y) == CASES[tag]['kw']['sequence']
            for k, yy in enumerate(y):
                assert np.array_equal(yy, G['%s_y%d_%d' % (tag, b, k)])
        else:
            assert np.array_equal(y, G['%s_y%d' % (tag, b)])
    Xa, ya = g.generate([], scale_and_impute=False)
    assert np.array_equal(Xa, G['%s_Xall' % tag])
    if not isinstance(ya, list):
        assert np.array_equal(ya, G['%s_yall' % tag])
    assert np.array_equal(g[-1][0], g[len(g) - 1][0])
    with pytest.raises(IndexError):
        g[len(g)]


def test_series_generator_windows_by_hand_and_nan_removal():
    g = _gen('c')                                       # 3 input steps of z (2 levels), target t850, 2 steps ahead
    X, y = g.generate([1, 4], scale_and_impute=False)
    S = G['S']
    assert X.shape == (2, 6, 6, 8) and y.shape == (2, 1, 6, 8)
    assert np.array_equal(X[1], S[4:7, 0].reshape(6, 6, 8))               # samples 4, 5, 6 of z, time-step major
    assert np.array_equal(y[1, 0], S[4 + 3 + 2 - 1, 1, 1])                # first target = last input + interval
    # NaN in a target block of a sequence drops that sample from the predictors and from EVERY block
    s2 = S.copy()
    s2[7, 0, 0, 2, 3] = np.nan
    gf = _gen('f', _ds(s2))                              # sample i covers series steps i .. i+7 (2 inputs + 3 x 2 targets)
    X, ys = gf.generate([0, 1, 2, 3], scale_and_impute=False)
    assert X.shape[0] == 0 and all(t.shape[0] == 0 for t in ys) and len(ys) == 3     # every window contains step 7
    g2 = SeriesDataGenerator(gf.model, _ds(s2), input_time_steps=2, output_time_steps=2, batch_size=3)
    X2, y2 = g2.generate([3, 8], scale_and_impute=False)                  # 3 -> steps 3..6 clean, 8 -> steps 8..11 clean
    assert X2.shape[0] == 2
    X3, y3 = g2.generate([5, 6], scale_and_impute=False)                  # both windows contain step 7
    assert X3.shape[0] == 0 and y3.shape[0] == 0


def test_labeled_array_selection_errors():
    da = _ds().predictors
    assert da.sel(variable=['t']).shape == (14, 1, 2, 6, 8) and da.sel(level=850).shape == (14, 2, 6, 8)
    assert isinstance(da, LabeledArray) and da.lat.values.shape == (6,)
    with pytest.raises(KeyError):
        da.sel(variable=['q'])
    with pytest.raises(KeyError):
        da.sel(height=[2])
    with pytest.raises(ValueError, match="'predictors'"):
        SeriesDataGenerator(DLWPNeuralNet(scaler_type=None), object())


@pytest.mark.parametrize('tag', sorted(CASES))
def test_generate_inputs_is_generate_without_the_targets(tag):
    """TimeSeriesEstimator.predict reads the predictors of generate([], scale_and_impute=False) and only the SHAPE of the
    targets (DLWP/model/extensions.py:171-172, 199-203): generate_inputs builds just that, bit for bit."""
    g = _gen(tag)
    made = g.generate_inputs()
    p, t = g.generate([], scale_and_impute=False)
    if CASES[tag]['kw'].get('sequence'):
        assert made is None                                # a list of target blocks: the estimator goes through generate()
        return
    assert made is not None and np.array_equal(made[0], p) and made[0].dtype == p.dtype and tuple(made[1]) == tuple(t.shape)
    # a NaN anywhere in the series: samples may be dropped, so the full generate() decides
    s = G['S'].copy()
    s[5, 0, 0, 1, 1] = np.nan
    assert _gen(tag, _ds(s)).generate_inputs() is None
