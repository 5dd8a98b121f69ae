"""DeviceLoader (dlwp_amd/model/generators.py): the pinned-host -> HBM feed behind fit_generator -- the reference's
fit_generator(generator, use_multiprocessing=True) over a keras.utils.Sequence (DLWP/model/models.py:216-228, generators.py:137-159).
Here on the CPU device: the worker thread's host gathers (the library's dlwp_host_gather_rows for plain row gathers, generate()
otherwise) must hand out exactly the generator's batches, in order, for shuffled epochs, ragged last batches and rank shards."""
import ctypes

import numpy as np
import torch

from dlwp_amd import _lib
from dlwp_amd.model import ArrayDataset, DataGenerator, DLWPNeuralNet
from dlwp_amd.model.generators import DeviceLoader


def _gen(dtype, n=23, batch=5, shuffle=True, remove_nan=True, nan_at=None):
    rng = np.random.default_rng(5)
    P = rng.standard_normal((n, 2, 3, 6, 8)).astype(dtype)
    T = rng.standard_normal((n, 2, 3, 6, 8)).astype(dtype)
    if nan_at is not None:
        P[nan_at, 0, 0, 0, 0] = np.nan
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=2, scaler_type=None, scale_targets=False)
    np.random.seed(7)
    return DataGenerator(d, ArrayDataset(P, T), batch_size=batch, shuffle=shuffle, remove_nan=remove_nan)


def test_host_gather_rows_copies_the_named_rows():
    src = np.arange(7 * 300000, dtype=np.float32).reshape(7, 300000)          # 1.2 MB rows: the threaded path
    rows = np.array([6, 0, 3, 3, 5], dtype=np.int64)
    for threads in (1, 4):
        dst = np.full((5, 300000), -1, dtype=np.float32)
        rc = _lib.lib.dlwp_host_gather_rows(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src.ctypes.data),
                                            rows.ctypes.data_as(ctypes.c_void_p), len(rows), src.shape[1] * 4, src.shape[0], threads)
        assert rc == 0 and np.array_equal(dst, src[rows])
    bad = np.array([7], dtype=np.int64)
    assert _lib.lib.dlwp_host_gather_rows(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src.ctypes.data),
                                          bad.ctypes.data_as(ctypes.c_void_p), 1, 4, src.shape[0], 1) == _lib.EINVAL


def test_plain_row_gathers_take_the_native_path_and_equal_generate():
    gen = _gen(np.float32)
    assert gen.batch_sources() is not None and gen.convolution_shape == (6, 6, 8)
    dev = torch.device('cpu')
    for epoch in range(2):
        want = [gen[i] for i in range(len(gen))]
        got = [(X.clone(), y.clone()) for X, y in DeviceLoader(gen, dev)]
        assert len(got) == len(want) == 5 and got[-1][0].shape[0] == 3              # ragged last batch
        for (X, y), (Xw, yw) in zip(got, want):
            assert X.dtype == torch.float32 and np.array_equal(X.numpy(), Xw) and np.array_equal(y.numpy(), yw)
        gen.on_epoch_end()                                                           # another shuffle


def test_generators_that_transform_their_batches_keep_the_generic_path():
    assert _gen(np.float64).batch_sources() is None                                  # a conversion, not a copy
    g = _gen(np.float32, nan_at=4)
    assert g.batch_sources() is None                                                 # a NaN sample to drop
    want = [g[i] for i in range(len(g))]
    got = [(X.clone(), y.clone()) for X, y in DeviceLoader(g, torch.device('cpu'))]
    assert sum(x.shape[0] for x, _ in got) == 22
    for (X, y), (Xw, yw) in zip(got, want):
        assert np.array_equal(X.numpy(), Xw.astype(np.float32)) and np.array_equal(y.numpy(), yw.astype(np.float32))
    assert _gen(np.float32, nan_at=4, remove_nan=False).batch_sources() is not None  # NaNs stay: still a plain gather


def test_rank_shards_of_the_native_path_cover_every_row_once():
    gen = _gen(np.float32, n=22, batch=8, shuffle=True)
    whole = [gen[i] for i in range(len(gen))]
    parts = [[(X.clone(), y.clone(), ng) for X, y, ng in DeviceLoader(gen, torch.device('cpu'), shard=(r, 3)).iter_batches()]
             for r in range(3)]
    for i, (Xw, yw) in enumerate(whole):
        X = np.concatenate([parts[r][i][0].numpy() for r in range(3)])
        y = np.concatenate([parts[r][i][1].numpy() for r in range(3)])
        assert np.array_equal(X, Xw) and np.array_equal(y, yw) and parts[0][i][2] == Xw.shape[0]
