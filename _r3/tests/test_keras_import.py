"""Keras HDF5 checkpoints (what the reference's save_model writes as '<name>.keras', DLWP/util.py:126-153; load_model,
util.py:156-192): the pure-numpy container reader (dlwp_amd/hdf5_lite.py) against files written by a real libhdf5
(tests/golden/keras_*.h5 from oracle/make_keras_h5.py, h5py 3.3 / HDF5 1.10 in the build container's other interpreter), and
the importer that rebuilds the model from `model_config` + `model_weights`."""
import os
import pickle

import numpy as np
import pytest

from dlwp_amd import hdf5_lite, serialization, util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _expected(tag):
    e = np.load(os.path.join(GOLDEN, 'keras_h5_expected.npz'))
    return {k.split('|', 1)[1]: e[k] for k in e.files if k.startswith(tag + '|')}


def test_reader_matches_libhdf5_on_the_keras_layout():
    for tag, fn in (('seq', 'keras_sequential.h5'), ('fun', 'keras_functional.h5')):
        f = hdf5_lite.File(os.path.join(GOLDEN, fn))
        assert hdf5_lite.is_hdf5(os.path.join(GOLDEN, fn))
        assert f.attrs['keras_version'] in ('2.2.4', b'2.2.4') and 'model_config' in f.attrs
        mw = f['model_weights']
        exp = _expected(tag)
        seen = 0
        for lname in [n.decode() if isinstance(n, bytes) else n for n in mw.attrs['layer_names']]:
            g = mw[lname]
            for wn in [n.decode() if isinstance(n, bytes) else n for n in np.asarray(g.attrs['weight_names']).reshape(-1)]:
                d = g[wn]
                want = exp['%s/%s' % (lname, wn.split('/')[-1].split(':')[0])]
                assert d.shape == want.shape and d.dtype == want.dtype and np.array_equal(d[...], want)
                seen += 1
        assert seen == len(exp) and seen >= 4
        with pytest.raises(KeyError):
            f['model_weights/nothing_here']


def test_reader_container_features():
    f = hdf5_lite.File(os.path.join(GOLDEN, 'keras_container.h5'))
    exp = _expected('con')
    assert f.attrs['vlen'] == 'a python str attribute' and np.array_equal(f.attrs['ints'], np.arange(7))
    assert f.attrs['scalar_f32'] == np.float32(1.5)
    assert sorted(f['many'].keys()) == ['d%02d' % i for i in range(60)]          # several symbol-table nodes
    for i in (0, 31, 59):
        assert np.array_equal(f['many/d%02d' % i][...], exp['many/d%02d' % i])
    assert np.array_equal(f['chunked'][...], exp['chunked'])                     # chunk B-tree, ragged edge chunks
    assert np.array_equal(f['gzip_shuffle'][...], exp['gzip_shuffle'])           # deflate + shuffle filters
    assert f['scalar'][...] == 7 and f['scalar'].shape == ()
    g = f['chunks']                                                              # Keras splits attributes > 64 KB
    names = []
    k = 0
    while 'layer_names%d' % k in g.attrs:
        names += list(g.attrs['layer_names%d' % k])
        k += 1
    assert k > 1 and np.array_equal(np.asarray(names), exp['chunk_names'])
    with open(os.path.join(GOLDEN, 'padding.npz'), 'rb') as fh:
        assert fh.read(8) != hdf5_lite.SIGNATURE
    assert not hdf5_lite.is_hdf5(os.path.join(GOLDEN, 'padding.npz'))


def test_reader_on_the_newest_file_format_covers_compact_groups_and_names_what_it_refuses():
    f = hdf5_lite.File(os.path.join(GOLDEN, 'keras_container_latest.h5'))      # superblock 3, 'OHDR' headers, link messages
    exp = _expected('new')
    assert sorted(f.keys()) == ['chunked', 'g', 'many'] and np.array_equal(f.attrs['a'], np.arange(3))
    assert np.array_equal(f['g/x'][...], exp['g/x']) and np.array_equal(f['g/y'][...], exp['g/y'])
    assert list(f['g'].attrs['names']) == [b'ab', b'cde']
    with pytest.raises(NotImplementedError, match='dense link storage'):
        f['many']
    with pytest.raises(NotImplementedError, match='chunk'):
        f['chunked'][...]


def test_import_sequential_checkpoint():
    m = serialization.load_model_file(os.path.join(GOLDEN, 'keras_sequential.h5'))      # routed by the file signature
    exp = _expected('seq')
    assert [type(lay).__name__ for lay in m.layers if lay._weights] == ['Conv2D', 'Conv2D']
    ws = m.get_weights()
    assert np.array_equal(ws[0], exp['conv2d_1/kernel']) and np.array_equal(ws[1], exp['conv2d_1/bias'])
    assert np.array_equal(ws[2], exp['conv2d_2/kernel']) and np.array_equal(ws[3], exp['conv2d_2/bias'])
    assert m.input_shape == (None, 2, 10, 12) and m.output_shape == (None, 2, 10, 12)
    assert [lay.name for lay in m.layers][1:4] == ['periodic_padding2d_1', 'zero_padding2d_1', 'conv2d_1']
    c1 = [lay for lay in m.layers if lay.name == 'conv2d_1'][0]
    assert c1.kernel_regularizer.l2 == pytest.approx(1e-4) and c1.activation == 'tanh' and c1.kernel_size == (5, 5)
    # compiled from training_config: Adam(lr=5e-4), mse, mae
    assert type(m.optimizer).__name__ == 'Adam' and m.optimizer.lr == pytest.approx(5e-4)
    assert m.metrics_names == ['loss', 'mean_absolute_error']
    # pooling in the producer's epilogue, halos fused: the usual plan
    assert [op.kind for op in m.infer_plan.ops] == ['conv', 'conv']


def test_import_functional_checkpoint_with_a_shared_layer_and_row_connected_output():
    m = serialization.import_keras_hdf5(os.path.join(GOLDEN, 'keras_functional.h5'))
    exp = _expected('fun')
    shared = [lay for lay in m.layers if lay.name == 'shared'][0]
    row = [lay for lay in m.layers if lay.name == 'row'][0]
    assert shared._calls == 2 and type(row).__name__ == 'RowConnected2D'
    assert np.array_equal(shared.get_weights()[0], exp['shared/kernel'])
    assert np.array_equal(row.get_weights()[0], exp['row/kernel']) and np.array_equal(row.get_weights()[1], exp['row/bias'])
    kinds = [op.kind for op in m.infer_plan.ops]
    assert kinds.count('conv') == 2 and kinds[-1] == 'rowconv' and m.output_shape == (None, 3, 8, 12)


def test_reference_style_load_model_reads_a_keras_checkpoint(tmp_path):
    """DLWP.util.load_model: '<name>.pkl' (the wrapper without its Keras model) + '<name>.keras' (HDF5)."""
    from dlwp_amd.model import DLWPNeuralNet
    import shutil
    base = os.path.join(str(tmp_path), 'ref_model')
    d = DLWPNeuralNet(is_convolutional=True, is_recurrent=False, time_dim=1, scaler_type=None, scale_targets=False)
    with open(base + '.pkl', 'wb') as fh:
        pickle.dump(d, fh)
    shutil.copy(os.path.join(GOLDEN, 'keras_sequential.h5'), base + '.keras')
    d2 = util.load_model(base)
    assert isinstance(d2, DLWPNeuralNet) and d2.model is d2.base_model
    assert np.array_equal(d2.model.get_weights()[0], _expected('seq')['conv2d_1/kernel'])


def test_lambda_layers_are_refused_with_a_pointer():
    import json
    cfg = {'class_name': 'Lambda', 'config': {'name': 'lambda_1', 'function': ['4wEAAAA...', None, None]}}
    with pytest.raises(NotImplementedError, match='slice_layer'):
        # the importer's layer factory is reached through a tiny fake config: patch a copy of the sequential file's JSON
        f = hdf5_lite.File(os.path.join(GOLDEN, 'keras_sequential.h5'))
        mc = json.loads(f.attrs['model_config'] if isinstance(f.attrs['model_config'], str) else f.attrs['model_config'].decode())
        mc['config']['layers'].insert(3, cfg)
        f.attrs['model_config'] = json.dumps(mc)
        import dlwp_amd.hdf5_lite as H
        orig = H.File
        H.File = lambda path: f
        try:
            serialization.import_keras_hdf5('unused')
        finally:
            H.File = orig
