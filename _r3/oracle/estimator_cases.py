"""TEST INFRASTRUCTURE ONLY.  The TimeSeriesEstimator cases shared by oracle/make_golden.py (which runs the REFERENCE's
DLWP/model/extensions.py:136-303 on them under the numpy-backed xarray stub and stores the results in
tests/golden/estimator.npz) and tests/test_estimator.py (which runs dlwp_amd's restatement on the same inputs)."""
import numpy as np

#: SeriesDataGenerator keyword arguments / predict() arguments per case.  The dataset is the one of the 'series' goldens:
#: 14 six-hourly samples, variables (z, t) x levels (500, 850) on a 6 x 8 grid ('varlev': the same data flattened to a
#: 'varlev' dimension with labels 'z/500' ...).
CASES = {
    # inputs == outputs: the plain autoregressive rollout (every channel and time step is overwritten by the forecast)
    'same': dict(gen=dict(input_time_steps=2, output_time_steps=2, batch_size=4), predict=dict(steps=5)),
    # fewer output steps and a subset of the variables: the rest keeps coming from the data, re-indexed
    'fewer': dict(gen=dict(input_sel={'level': [500]}, output_sel={'variable': ['z'], 'level': [500]},
                           input_time_steps=2, output_time_steps=1, batch_size=4), predict=dict(steps=3)),
    'fewer_impute': dict(gen=dict(input_sel={'level': [500]}, output_sel={'variable': ['z'], 'level': [500]},
                                  input_time_steps=2, output_time_steps=1, batch_size=4),
                         predict=dict(steps=3, impute=True)),
    # analytically known insolation channel, refreshed for the rows past the data
    'sol': dict(gen=dict(input_time_steps=1, output_time_steps=1, add_insolation=True, batch_size=4),
                predict=dict(steps=3)),
    'sol2': dict(gen=dict(input_time_steps=2, output_time_steps=2, add_insolation=True, batch_size=4),
                 predict=dict(steps=4)),
    # more output than input steps: first or last predicted times seed the next call
    'more_first': dict(gen=dict(input_time_steps=1, output_time_steps=2, batch_size=4), predict=dict(steps=4)),
    'more_last': dict(gen=dict(input_time_steps=1, output_time_steps=2, batch_size=4),
                      predict=dict(steps=4, prefer_first_times=False)),
    # a gap between the last input and the first output step
    'interval': dict(gen=dict(input_time_steps=2, output_time_steps=2, interval=2, batch_size=4), predict=dict(steps=4)),
    # variable selection with different input / output orderings
    'swap': dict(gen=dict(input_sel={'variable': ['t', 'z'], 'level': [850]}, output_sel={'variable': ['z'], 'level': [850]},
                          input_time_steps=2, output_time_steps=2, batch_size=4), predict=dict(steps=3)),
}
#: the same estimator on a dataset whose predictors carry one flattened 'varlev' dimension.  keep_time_dim=True only exists
#: here: on a (variable, level) dataset the reference's final transpose (extensions.py:302) names six dimensions for a
#: seven-dimensional array and raises.
VARLEV_CASES = {
    'varlev_same_keep': dict(gen=dict(input_time_steps=2, output_time_steps=2, batch_size=4),
                             predict=dict(steps=4, keep_time_dim=True)),
    'varlev_more_keep': dict(gen=dict(input_time_steps=1, output_time_steps=2, batch_size=4),
                             predict=dict(steps=4, keep_time_dim=True)),
    'varlev_same': dict(gen=dict(input_time_steps=2, output_time_steps=2, batch_size=4), predict=dict(steps=3)),
    'varlev_fewer': dict(gen=dict(input_sel={'varlev': ['z/500', 't/500']}, output_sel={'varlev': ['z/500']},
                                  input_time_steps=2, output_time_steps=1, batch_size=4), predict=dict(steps=3)),
}


def mixing_model(c_in, c_out, seed=3):
    """A stand-in network with a forecast one can tell channel routing errors from: out[:, j] = tanh(sum_i A[j, i] p[:, i]
    + b[j]) over the channel axis (axis 1) of an (n, c_in, h, w) input, float32, NaN rows stay NaN."""
    rng = np.random.RandomState(seed)
    A = rng.uniform(-0.6, 0.6, size=(c_out, c_in)).astype(np.float32)
    b = rng.uniform(-0.2, 0.2, size=(c_out,)).astype(np.float32)

    def predict(p, **kwargs):
        p = np.asarray(p, dtype=np.float32)
        assert p.ndim == 4 and p.shape[1] == c_in, (p.shape, c_in)
        out = np.einsum('ji,nihw->njhw', A, p) + b[None, :, None, None]
        return np.tanh(out).astype(np.float32)
    return predict
