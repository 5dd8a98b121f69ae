#!/usr/bin/env python
"""
Fixture generator for the Keras-HDF5 import path.  TEST INFRASTRUCTURE ONLY -- never imported by the product.

Writes small HDF5 checkpoints in the layout `keras.models.save_model` uses (Keras 2.2.x `engine/saving.py`; the reference's
save_model calls it, DLWP/util.py:141-144): root attributes keras_version / backend / model_config (JSON) / training_config,
group model_weights with attribute layer_names and one group per layer holding attribute weight_names and the datasets
`<layer>/<weight>:0` -- through a REAL libhdf5 (h5py), so that dlwp_amd/hdf5_lite.py is validated against the genuine
container format.  Keras itself is not installed anywhere in the image: the JSON configs below restate what Keras 2.2.4's
`get_config()` returns for these layers ("parity unpinned" for the Keras side; the HDF5 side is pinned).

Run with the build container's OTHER interpreter, the only one that has h5py:
    /opt/conda/bin/python3.9 oracle/make_keras_h5.py        (writes tests/golden/keras_*.h5 + keras_h5_expected.npz)
"""
import json
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
CF = 'channels_first'
GLOROT = {'class_name': 'VarianceScaling', 'config': {'scale': 1.0, 'mode': 'fan_avg', 'distribution': 'uniform', 'seed': None}}
ZEROS = {'class_name': 'Zeros', 'config': {}}


def pad_cfg(name, pad, first_shape=None):
    c = {'name': name, 'trainable': True, 'padding': [list(p) for p in pad], 'data_format': CF}
    if first_shape is not None:
        c.update(batch_input_shape=[None] + list(first_shape), dtype='float32')
    return c


def conv_cfg(name, filters, ks, dil, act, l2=None):
    return {'name': name, 'trainable': True, 'filters': filters, 'kernel_size': [ks, ks], 'strides': [1, 1], 'padding': 'valid',
            'data_format': CF, 'dilation_rate': [dil, dil], 'activation': act, 'use_bias': True, 'kernel_initializer': GLOROT,
            'bias_initializer': ZEROS,
            'kernel_regularizer': None if l2 is None else {'class_name': 'L1L2', 'config': {'l1': 0.0, 'l2': l2}},
            'bias_regularizer': None, 'activity_regularizer': None, 'kernel_constraint': None, 'bias_constraint': None}


def save_attr_list(group, name, data):
    """keras.engine.saving.save_attributes_to_hdf5_group: split into chunks when the attribute would exceed 64 KB."""
    limit = 64512
    arr = np.asarray(data)
    n = 1
    chunks = np.array_split(arr, n)
    while any(c.nbytes > limit for c in chunks):
        n += 1
        chunks = np.array_split(arr, n)
    if n > 1:
        for k, c in enumerate(chunks):
            group.attrs['%s%d' % (name, k)] = c
    else:
        group.attrs[name] = arr


def write(path, class_name, config, weights, training=None, expected=None, tag=''):
    with h5py.File(path, 'w') as f:
        f.attrs['keras_version'] = '2.2.4'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = json.dumps({'class_name': class_name, 'config': config}).encode('utf8')
        if training is not None:
            f.attrs['training_config'] = json.dumps(training).encode('utf8')
        g = f.create_group('model_weights')
        names = [spec['config']['name'] for spec in config['layers']]
        save_attr_list(g, 'layer_names', [n.encode('utf8') for n in names])
        g.attrs['backend'] = 'tensorflow'.encode('utf8')
        g.attrs['keras_version'] = '2.2.4'.encode('utf8')
        for n in names:
            lg = g.create_group(n)
            ws = weights.get(n, [])
            save_attr_list(lg, 'weight_names', [('%s/%s:0' % (n, wn)).encode('utf8') for wn, _ in ws])
            for wn, val in ws:
                d = lg.create_dataset('%s/%s:0' % (n, wn), val.shape, dtype=val.dtype)
                d[...] = val
                if expected is not None:
                    expected['%s|%s/%s' % (tag, n, wn)] = val


def main():
    rng = np.random.default_rng(20240601)
    expected = {}
    # ---- Sequential: the 2-conv network of BASELINE config 1 (examples/train.py:159-169, 211-219) + pooling / up-sampling
    cs = (2, 10, 12)
    layers = [
        {'class_name': 'PeriodicPadding2D', 'config': pad_cfg('periodic_padding2d_1', ((0, 0), (2, 2)), cs)},
        {'class_name': 'ZeroPadding2D', 'config': pad_cfg('zero_padding2d_1', ((2, 2), (0, 0)))},
        {'class_name': 'Conv2D', 'config': conv_cfg('conv2d_1', 8, 5, 1, 'tanh', l2=1e-4)},
        {'class_name': 'MaxPooling2D', 'config': {'name': 'max_pooling2d_1', 'trainable': True, 'pool_size': [2, 2], 'padding': 'valid',
                                                  'strides': [2, 2], 'data_format': CF}},
        {'class_name': 'UpSampling2D', 'config': {'name': 'up_sampling2d_1', 'trainable': True, 'size': [2, 2], 'data_format': CF,
                                                  'interpolation': 'nearest'}},
        {'class_name': 'PeriodicPadding2D', 'config': pad_cfg('periodic_padding2d_2', ((0, 0), (1, 1)))},
        {'class_name': 'ZeroPadding2D', 'config': pad_cfg('zero_padding2d_2', ((1, 1), (0, 0)))},
        {'class_name': 'Conv2D', 'config': conv_cfg('conv2d_2', 2, 3, 1, 'linear')},
    ]
    w = {'conv2d_1': [('kernel', rng.standard_normal((5, 5, 2, 8)).astype(np.float32) * 0.1),
                      ('bias', rng.standard_normal(8).astype(np.float32) * 0.1)],
         'conv2d_2': [('kernel', rng.standard_normal((3, 3, 8, 2)).astype(np.float32) * 0.1),
                      ('bias', rng.standard_normal(2).astype(np.float32) * 0.1)]}
    training = {'optimizer_config': {'class_name': 'Adam', 'config': {'lr': 0.0005, 'beta_1': 0.9, 'beta_2': 0.999, 'decay': 0.0,
                                                                     'epsilon': 1e-07, 'amsgrad': False}},
                'loss': 'mse', 'metrics': ['mae'], 'sample_weight_mode': None, 'loss_weights': None}
    write(os.path.join(OUT, 'keras_sequential.h5'), 'Sequential', {'name': 'sequential_1', 'layers': layers}, w, training,
          expected, 'seq')
    # ---- functional Model: a shared convolution applied twice + concatenate + RowConnected2D output
    cs = (3, 8, 12)
    def node(*srcs):
        return [[[s, k, 0, {}] for s, k in srcs]]
    fl = [
        {'name': 'input_0', 'class_name': 'InputLayer', 'inbound_nodes': [],
         'config': {'batch_input_shape': [None] + list(cs), 'dtype': 'float32', 'sparse': False, 'name': 'input_0'}},
        {'name': 'pp', 'class_name': 'PeriodicPadding2D', 'config': pad_cfg('pp', ((0, 0), (1, 1))),
         'inbound_nodes': node(('input_0', 0)) + node(('shared', 0))},
        {'name': 'zp', 'class_name': 'ZeroPadding2D', 'config': pad_cfg('zp', ((1, 1), (0, 0))),
         'inbound_nodes': node(('pp', 0)) + node(('pp', 1))},
        {'name': 'shared', 'class_name': 'Conv2D', 'config': conv_cfg('shared', 3, 3, 1, 'tanh'),
         'inbound_nodes': node(('zp', 0)) + node(('zp', 1))},
        {'name': 'cat', 'class_name': 'Concatenate', 'config': {'name': 'cat', 'trainable': True, 'axis': 1},
         'inbound_nodes': node(('shared', 0), ('shared', 1))},
        {'name': 'pp2', 'class_name': 'PeriodicPadding2D', 'config': pad_cfg('pp2', ((0, 0), (2, 2))), 'inbound_nodes': node(('cat', 0))},
        {'name': 'zp2', 'class_name': 'ZeroPadding2D', 'config': pad_cfg('zp2', ((2, 2), (0, 0))), 'inbound_nodes': node(('pp2', 0))},
        {'name': 'row', 'class_name': 'RowConnected2D', 'inbound_nodes': node(('zp2', 0)),
         'config': {'name': 'row', 'trainable': True, 'filters': 3, 'kernel_size': [5, 5], 'strides': [1, 1], 'padding': 'valid',
                    'data_format': CF, 'activation': 'linear', 'use_bias': True, 'kernel_initializer': GLOROT,
                    'bias_initializer': ZEROS, 'kernel_regularizer': None, 'bias_regularizer': None, 'activity_regularizer': None,
                    'kernel_constraint': None, 'bias_constraint': None}},
    ]
    fw = {'shared': [('kernel', rng.standard_normal((3, 3, 3, 3)).astype(np.float32) * 0.2),
                     ('bias', rng.standard_normal(3).astype(np.float32) * 0.1)],
          'row': [('kernel', rng.standard_normal((8, 5, 5, 6, 3)).astype(np.float32) * 0.1),
                  ('bias', rng.standard_normal((8, 1, 3)).astype(np.float32) * 0.1)]}
    write(os.path.join(OUT, 'keras_functional.h5'), 'Model',
          {'name': 'model_1', 'layers': fl, 'input_layers': [['input_0', 0, 0]], 'output_layers': [['row', 0, 0]]}, fw, None,
          expected, 'fun')
    # ---- container features beyond what Keras writes: many links (several symbol nodes), chunked + gzip + shuffle, scalars,
    #      float64 / integers, a python-str (variable-length) attribute, a layer_names list long enough to be split in chunks
    with h5py.File(os.path.join(OUT, 'keras_container.h5'), 'w') as f:
        f.attrs['vlen'] = 'a python str attribute'
        f.attrs['ints'] = np.arange(7, dtype=np.int64)
        f.attrs['scalar_f32'] = np.float32(1.5)
        many = f.create_group('many')
        for i in range(60):
            a = np.full((3,), i, dtype=np.int32)
            many.create_dataset('d%02d' % i, data=a)
            expected['con|many/d%02d' % i] = a
        a = rng.standard_normal((37, 21))
        f.create_dataset('chunked', data=a, chunks=(8, 5))
        expected['con|chunked'] = a
        a = np.arange(240, dtype=np.float32).reshape(12, 20)
        f.create_dataset('gzip_shuffle', data=a, chunks=(5, 20), compression='gzip', shuffle=True)
        expected['con|gzip_shuffle'] = a
        d = f.create_dataset('scalar', (), dtype=np.int64)
        d[()] = 7
        expected['con|scalar'] = np.int64(7)
        names = [('layer_with_a_long_name_%05d' % i).encode('utf8') for i in range(3000)]
        save_attr_list(f.create_group('chunks'), 'layer_names', names)
        expected['con|chunk_names'] = np.asarray(names)
    # ---- the newest file format (libver='latest': superblock 3, version-2 object headers, link messages): what the reader
    #      covers of it (compact groups, contiguous data) and what it must refuse by name (dense groups, new chunk indexes)
    with h5py.File(os.path.join(OUT, 'keras_container_latest.h5'), 'w', libver='latest') as f:
        f.attrs['a'] = np.arange(3)
        g = f.create_group('g')
        a = rng.standard_normal((4, 5)).astype(np.float32)
        g.create_dataset('x', data=a)
        expected['new|g/x'] = a
        g.create_dataset('y', data=np.arange(6, dtype=np.int16))
        expected['new|g/y'] = np.arange(6, dtype=np.int16)
        g.attrs['names'] = np.array([b'ab', b'cde'])
        f.create_dataset('chunked', data=np.arange(20.).reshape(4, 5), chunks=(2, 5))
        many = f.create_group('many')
        for i in range(30):
            many.create_dataset('d%02d' % i, data=np.full(2, i))
    np.savez_compressed(os.path.join(OUT, 'keras_h5_expected.npz'), **expected)
    for fn in sorted(os.listdir(OUT)):
        if fn.startswith('keras_'):
            print('%-28s %8d bytes' % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == '__main__':
    main()
