"""
Model files.  The reference writes `<name>.keras` as Keras HDF5 (DLWP/util.py:126-153).  So does this package (r6:
export_keras_hdf5, the layout of keras.engine.saving.save_model through the pure-numpy container writer of dlwp_amd.hdf5_lite --
h5py is absent here) and it READS such files (import_keras_hdf5): checkpoints move both ways between the reference and this
package.  The older format of this package -- a numpy .npz archive under the same name: a JSON description of the layer graph +
compile arguments, every weight array in KERAS LAYOUT (conv kernels (kh, kw, cin, cout), biases (cout,)) under the key
`<layer name>/<weight name>` -- is still read (load_model_file routes by the file's signature) and written on request
(format='npz' / DLWP_SAVE_FORMAT=npz).
"""
import io
import json

import numpy as np

FORMAT = 'dlwp_amd-model-v1'


def _layer_config(lay):
    from . import layers as L
    cfg = {'name': lay.name}
    if isinstance(lay, L.InputLayer):
        cfg['input_shape'] = list(lay.batch_input_shape[1:])
    elif isinstance(lay, (L._Pad2DBase, L._Pad3DBase)):
        cfg.update(padding=[list(p) for p in lay.padding], data_format=lay.data_format)
        if hasattr(lay, 'tf_mode'):          # TFPadding2D
            cfg.update(mode=lay.tf_mode, constant_values=lay.constant_values)
    elif isinstance(lay, L.ConvLSTM2D):
        from .regularizers import L1L2
        cfg.update(filters=lay.filters, kernel_size=list(lay.kernel_size), padding=lay.padding,
                   data_format=lay.data_format, dilation_rate=list(lay.dilation_rate), activation=lay.activation,
                   recurrent_activation=lay.recurrent_activation, use_bias=lay.use_bias,
                   unit_forget_bias=lay.unit_forget_bias, return_sequences=lay.return_sequences)
        if isinstance(lay.kernel_regularizer, L1L2):
            cfg['kernel_regularizer'] = {'l2': lay.kernel_regularizer.l2}
    elif isinstance(lay, L.Conv2D):
        from .regularizers import L1L2
        cfg.update(filters=lay.filters, kernel_size=list(lay.kernel_size), padding=lay.padding,
                   data_format=lay.data_format, dilation_rate=list(lay.dilation_rate), activation=lay.activation,
                   use_bias=lay.use_bias)
        if isinstance(lay.kernel_regularizer, L1L2):
            cfg['kernel_regularizer'] = {'l2': lay.kernel_regularizer.l2}
    elif isinstance(lay, L.RowConnected2D):
        from .regularizers import L1L2
        cfg.update(filters=lay.filters, kernel_size=list(lay.kernel_size), strides=list(lay.strides), padding=lay.padding,
                   data_format=lay.data_format, activation=lay.activation, use_bias=lay.use_bias)
        if isinstance(lay.kernel_regularizer, L1L2):
            cfg['kernel_regularizer'] = {'l2': lay.kernel_regularizer.l2}
    elif isinstance(lay, (L.MaxPooling2D, L.UpSampling2D)):
        cfg.update(data_format=lay.data_format)
    elif isinstance(lay, L.Reshape):
        cfg.update(target_shape=list(lay.target_shape))
    elif isinstance(lay, L.ChannelSlice):
        cfg.update(start=lay.start, end=lay.end, axis=lay.axis)
    elif isinstance(lay, L.Concatenate):
        cfg.update(axis=lay.axis)
    else:
        raise NotImplementedError('cannot serialise layer %s (%s)' % (lay.name, type(lay).__name__))
    if lay.batch_input_shape is not None and not isinstance(lay, L.InputLayer):
        cfg['input_shape'] = list(lay.batch_input_shape[1:])
    return cfg


def _loss_config(loss):
    from .custom import LossSpec
    if isinstance(loss, LossSpec):
        return {'spec': {'kind': loss.kind, 'regularize': loss.regularize, 'scale': loss.scale, 'name': loss.__name__,
                         'has_mean': loss.mean is not None, 'has_row_weights': loss.row_weights is not None}}
    return loss if isinstance(loss, str) else getattr(loss, '__name__', None)


def describe(model):
    from . import plan as P
    order = P.toposort(model.outputs)
    layers, lidx, nodes, nidx = [], {}, [], {}
    for t in order:
        if id(t.layer) not in lidx:
            lidx[id(t.layer)] = len(layers)
            layers.append({'class': type(t.layer).__name__, 'config': _layer_config(t.layer)})
        nidx[t.uid] = len(nodes)
        nodes.append({'layer': lidx[id(t.layer)], 'inputs': [nidx[i.uid] for i in t.inputs]})
    arch = {'format': FORMAT, 'class': type(model).__name__, 'name': model.name, 'layers': layers, 'nodes': nodes,
            'inputs': [nidx[t.uid] for t in model.inputs], 'outputs': [nidx[t.uid] for t in model.outputs]}
    if model.optimizer is not None:
        opt = model.optimizer
        arch['compile'] = {
            'optimizer': {'class': type(opt).__name__,
                          'config': {k: v for k, v in vars(opt).items() if isinstance(v, (int, float))}},
            'loss': _loss_config(model.loss),
            'metrics': [m if isinstance(m, str) else getattr(m, '__name__', None) for m in model.metrics],
            'loss_weights': list(model.loss_weights) if model.loss_weights is not None else None}
    return arch, layers, order


def save_model_file(model, path, format=None):
    """`format`: 'keras' = Keras HDF5 (what the reference writes, readable by keras.models.load_model), 'npz' = this package's own
    archive, None = DLWP_SAVE_FORMAT or 'keras' -- falling back to 'npz' (with a warning) for a graph Keras' format cannot name."""
    import os
    fmt = format or os.environ.get('DLWP_SAVE_FORMAT', 'keras')
    if fmt == 'keras':
        try:
            return export_keras_hdf5(model, path)
        except NotImplementedError as e:
            if format == 'keras':
                raise
            import warnings
            warnings.warn('%s: %s -- written in the npz format of this package instead' % (path, e))
    elif fmt != 'npz':
        raise ValueError("format must be 'keras' or 'npz', got %r" % (fmt,))
    arch, _, order = describe(model)
    arrays = {}
    seen = set()
    for t in order:
        lay = t.layer
        if id(lay) in seen:
            continue
        seen.add(id(lay))
        for (nm, _), a in zip(lay._weights, lay.get_weights()):
            arrays['%s/%s' % (lay.name, nm)] = a
    from .custom import LossSpec
    if isinstance(model.loss, LossSpec):
        if model.loss.mean is not None:
            arrays['__loss_mean__'] = model.loss.mean
        if model.loss.row_weights is not None:
            arrays['__loss_row_weights__'] = model.loss.row_weights
    # optimizer slots (Adam m / v, SGD velocity), as keras save_model keeps the optimizer weights: a resumed run continues
    # with the moments that belong to `iterations` (flat buffers in the order of the weights above)
    tr = getattr(model, '_trainer', None)
    if tr is not None and tr.opt_state is not None:
        for k, t in enumerate(tr.opt_state):
            arrays['__opt_state_%d__' % k] = t.detach().cpu().numpy()
    buf = io.BytesIO()
    np.savez(buf, __arch__=np.frombuffer(json.dumps(arch).encode('utf-8'), dtype=np.uint8), **arrays)
    with open(path, 'wb') as f:
        f.write(buf.getvalue())


# ------------------------------------------------------------------------------------------------------------------ #
# Keras HDF5 checkpoints: what the reference's save_model writes as '<name>.keras' (DLWP/util.py:126-153: model.save)
# ------------------------------------------------------------------------------------------------------------------ #

# ------------------------------------------------------------------------------------------------------------------ #
# WRITING Keras HDF5 checkpoints (r6): what the reference's save_model produces -- `model.save('<name>.keras')`,
# DLWP/util.py:141-144 -- in the layout of keras.engine.saving.save_model (Keras 2.2.x): root attributes keras_version / backend /
# model_config (JSON) / training_config (JSON), group model_weights (attributes layer_names, backend, keras_version; one group
# per layer with attribute weight_names and the datasets <layer>/<weight>:0), group optimizer_weights.  The container is written
# by dlwp_amd.hdf5_lite (pure numpy).  The layer configs restate Keras 2.2.4's get_config() for the layers this package has
# (the inverse of _keras_layer_kwargs; initialisers are the Keras defaults -- the weights are in the file).  Two things Keras'
# format cannot carry ride in a private group `dlwp_amd` that Keras ignores: the arrays of a custom loss (the reference re-creates
# its loss closures in the loading script and passes them as custom_objects, DLWP/util.py:171-174) and nothing else.
# A channel slice is a keras Lambda in the reference (marshalled bytecode): written as class ChannelSlice, which this package's
# importer knows and a real Keras needs as a custom object.
# ------------------------------------------------------------------------------------------------------------------ #

_GLOROT = {'class_name': 'VarianceScaling', 'config': {'scale': 1.0, 'mode': 'fan_avg', 'distribution': 'uniform', 'seed': None}}
_ZEROS = {'class_name': 'Zeros', 'config': {}}
_ORTHO = {'class_name': 'Orthogonal', 'config': {'gain': 1.0, 'seed': None}}


def _keras_layer_spec(lay):
    """{'class_name', 'config'} of a layer as Keras 2.2.4 would describe it"""
    from . import layers as L
    from .regularizers import L1L2
    c = {'name': lay.name, 'trainable': True}
    cname = type(lay).__name__

    def reg(r):
        return {'class_name': 'L1L2', 'config': {'l1': 0.0, 'l2': float(r.l2)}} if isinstance(r, L1L2) else None
    tail = {'bias_regularizer': None, 'activity_regularizer': None, 'kernel_constraint': None, 'bias_constraint': None}
    if isinstance(lay, L.InputLayer):
        c = {'batch_input_shape': [None] + list(lay.batch_input_shape[1:]), 'dtype': 'float32', 'sparse': False, 'name': lay.name}
    elif isinstance(lay, (L._Pad2DBase, L._Pad3DBase)):
        c.update(padding=[list(p) for p in lay.padding], data_format=lay.data_format)
        if hasattr(lay, 'tf_mode'):
            c.update(mode=lay.tf_mode, constant_values=lay.constant_values)
    elif isinstance(lay, L.ConvLSTM2D):
        c.update(return_sequences=lay.return_sequences, return_state=False, go_backwards=False, stateful=False, unroll=False,
                 filters=lay.filters, kernel_size=list(lay.kernel_size), strides=[1, 1], padding=lay.padding,
                 data_format=lay.data_format, dilation_rate=list(lay.dilation_rate), activation=lay.activation,
                 recurrent_activation=lay.recurrent_activation, use_bias=lay.use_bias, kernel_initializer=_GLOROT,
                 recurrent_initializer=_ORTHO, bias_initializer=_ZEROS, unit_forget_bias=lay.unit_forget_bias,
                 kernel_regularizer=reg(lay.kernel_regularizer), recurrent_regularizer=None, recurrent_constraint=None,
                 dropout=0.0, recurrent_dropout=0.0, **tail)
    elif isinstance(lay, L.Conv2D):
        c.update(filters=lay.filters, kernel_size=list(lay.kernel_size), strides=[1, 1], padding=lay.padding,
                 data_format=lay.data_format, dilation_rate=list(lay.dilation_rate), activation=lay.activation or 'linear',
                 use_bias=lay.use_bias, kernel_initializer=_GLOROT, bias_initializer=_ZEROS,
                 kernel_regularizer=reg(lay.kernel_regularizer), **tail)
    elif isinstance(lay, L.RowConnected2D):
        c.update(filters=lay.filters, kernel_size=list(lay.kernel_size), strides=list(lay.strides), padding=lay.padding,
                 data_format=lay.data_format, activation=lay.activation or 'linear', use_bias=lay.use_bias,
                 kernel_initializer=_GLOROT, bias_initializer=_ZEROS, kernel_regularizer=reg(lay.kernel_regularizer), **tail)
    elif isinstance(lay, L.MaxPooling2D):
        c.update(pool_size=[2, 2], padding='valid', strides=[2, 2], data_format=lay.data_format)
    elif isinstance(lay, L.UpSampling2D):
        c.update(size=[2, 2], data_format=lay.data_format, interpolation='nearest')
    elif isinstance(lay, L.Reshape):
        c.update(target_shape=list(lay.target_shape))
    elif isinstance(lay, L.ChannelSlice):
        c.update(start=lay.start, end=lay.end, axis=lay.axis)
    elif isinstance(lay, L.Concatenate):
        c.update(axis=lay.axis)
    else:
        raise NotImplementedError('no Keras description of layer %s (%s)' % (lay.name, cname))
    if lay.batch_input_shape is not None and not isinstance(lay, L.InputLayer):
        c.update(batch_input_shape=[None] + list(lay.batch_input_shape[1:]), dtype='float32')
    return {'class_name': cname, 'config': c}


def keras_model_config(model):
    """(model_config dict, the layers in Keras order) -- a Sequential when the graph is one chain from one input, else a Model"""
    from . import layers as L
    from . import plan as P
    order = P.toposort(model.outputs)
    seen, layers = set(), []
    for t in order:
        if id(t.layer) not in seen:
            seen.add(id(t.layer))
            layers.append(t.layer)
    names = [lay.name for lay in layers]
    if len(set(names)) != len(names):
        raise ValueError('layer names must be unique in a Keras checkpoint: %r' % names)
    chain = (len(model.inputs) == 1 and len(model.outputs) == 1 and len(order) == len(layers) and
             all(len(t.inputs) == (0 if isinstance(t.layer, L.InputLayer) else 1) for t in order) and
             all(order[k].inputs[0] is order[k - 1] for k in range(1, len(order))))
    if chain and (type(model).__name__ == 'Sequential' or getattr(model, '_keras_class', None) == 'Sequential'):
        specs = [_keras_layer_spec(lay) for lay in layers if not isinstance(lay, L.InputLayer)]
        first = specs[0]['config']
        if 'batch_input_shape' not in first:
            first.update(batch_input_shape=[None] + list(model.inputs[0].shape), dtype='float32')
        return {'class_name': 'Sequential', 'config': {'name': model.name, 'layers': specs}}, \
            [lay for lay in layers if not isinstance(lay, L.InputLayer)]
    # functional: node k of a layer = its k-th application, in graph order
    node_of, count = {}, {}
    for t in order:
        k = count.get(id(t.layer), 0)
        count[id(t.layer)] = k + 1
        node_of[t.uid] = (t.layer.name, k)
    specs = []
    for lay in layers:
        sp = _keras_layer_spec(lay)
        sp['name'] = lay.name
        sp['inbound_nodes'] = [[[node_of[i.uid][0], node_of[i.uid][1], 0, {}] for i in t.inputs]
                               for t in order if t.layer is lay and t.inputs]
        specs.append(sp)
    return {'class_name': 'Model',
            'config': {'name': model.name, 'layers': specs,
                       'input_layers': [[node_of[t.uid][0], node_of[t.uid][1], 0] for t in model.inputs],
                       'output_layers': [[node_of[t.uid][0], node_of[t.uid][1], 0] for t in model.outputs]}}, layers


def _attr_chunks(group, name, items):
    """keras.engine.saving.save_attributes_to_hdf5_group: a list of names as ONE fixed-length-string attribute, or as name0, name1,
    ... when it would exceed the 64 KB an object header message holds"""
    arr = np.array([i.encode('utf-8') for i in items]) if items else np.zeros((0,), dtype='S1')
    n = 1
    chunks = np.array_split(arr, n)
    while any(c.nbytes > 64512 for c in chunks):
        n += 1
        chunks = np.array_split(arr, n)
    if n > 1:
        for k, c in enumerate(chunks):
            group.attrs['%s%d' % (name, k)] = c
    else:
        group.attrs[name] = arr


def export_keras_hdf5(model, path):
    """Write `model` (a compiled or uncompiled dlwp_amd.engine.Model) as a Keras 2.2-layout HDF5 checkpoint."""
    from . import hdf5_lite
    from .custom import LossSpec
    config, layers = keras_model_config(model)
    root = hdf5_lite.GroupSpec()
    root.attrs['keras_version'] = b'2.2.4'
    root.attrs['backend'] = b'tensorflow'
    root.attrs['model_config'] = json.dumps(config).encode('utf-8')
    mw = root.create_group('model_weights')
    _attr_chunks(mw, 'layer_names', [lay.name for lay in layers])
    mw.attrs['backend'] = b'tensorflow'
    mw.attrs['keras_version'] = b'2.2.4'
    for lay in layers:
        g = mw.create_group(lay.name)
        pairs = list(zip([nm for nm, _ in lay._weights], lay.get_weights()))
        _attr_chunks(g, 'weight_names', ['%s/%s:0' % (lay.name, nm) for nm, _ in pairs])
        for nm, a in pairs:
            g.create_dataset('%s/%s:0' % (lay.name, nm), np.ascontiguousarray(a, dtype=np.float32))
    if model.optimizer is not None:
        opt = model.optimizer
        ocfg = {k: v for k, v in vars(opt).items() if isinstance(v, (int, float)) and k != 'iterations'}
        if type(opt).__name__ == 'Adam':
            ocfg['amsgrad'] = False
        loss = model.loss
        tc = {'optimizer_config': {'class_name': type(opt).__name__, 'config': ocfg},
              'loss': loss if isinstance(loss, (str, dict, list)) else getattr(loss, '__name__', 'loss'),
              'metrics': [m if isinstance(m, str) else getattr(m, '__name__', None) for m in model.metrics],
              'sample_weight_mode': None, 'loss_weights': list(model.loss_weights) if model.loss_weights is not None else None}
        root.attrs['training_config'] = json.dumps(tc).encode('utf-8')
        if isinstance(loss, LossSpec):
            priv = root.create_group('dlwp_amd')
            priv.attrs['loss_spec'] = json.dumps(_loss_config(loss)['spec']).encode('utf-8')
            if loss.mean is not None:
                priv.create_dataset('loss_mean', np.asarray(loss.mean, dtype=np.float32))
            if loss.row_weights is not None:
                priv.create_dataset('loss_row_weights', np.asarray(loss.row_weights, dtype=np.float32))
        # optimizer state in Keras' order -- Adam: iterations, the first moments of every weight, the second moments, one
        # (1,)-shaped placeholder per weight (amsgrad off: keras.optimizers.Adam keeps K.zeros(1) there); SGD: iterations, moments
        tr = getattr(model, '_trainer', None)
        if tr is not None and tr.opt_state is not None:
            ow = root.create_group('optimizer_weights')
            oname = type(opt).__name__
            names, arrays = ['%s/iterations:0' % oname], [np.asarray(int(opt.iterations), dtype=np.int64)]
            slots = [t.detach().cpu().numpy() for t in tr.opt_state]
            k = 0
            for slot in slots:
                for lay, nm, off, numel, shape in tr.entries:
                    names.append('training/%s/Variable%s:0' % (oname, '' if k == 0 else '_%d' % k))
                    arrays.append(np.ascontiguousarray(slot[off:off + numel].reshape(shape), dtype=np.float32))
                    k += 1
            if oname == 'Adam':
                for _ in tr.entries:
                    names.append('training/%s/Variable_%d:0' % (oname, k))
                    arrays.append(np.zeros((1,), dtype=np.float32))
                    k += 1
            _attr_chunks(ow, 'weight_names', names)
            for nm, a in zip(names, arrays):
                ow.create_dataset(nm, a)
    hdf5_lite.write_file(path, root)



def _keras_layer_kwargs(cls, cfg):
    """A Keras layer config (keras `Layer.get_config()` as stored in `model_config`) -> constructor arguments of this
    package's layer of the same name: initialisers / constraints / activity regularisers are dropped (the weights come from the
    file), `batch_input_shape` becomes `input_shape`, an l2 kernel regulariser is kept, unknown keys the constructor does not
    take are ignored."""
    import inspect
    from .regularizers import L1L2
    kw = {}
    params = inspect.signature(cls.__init__).parameters
    for k, v in cfg.items():
        if k == 'batch_input_shape':
            if v is not None:
                kw['input_shape'] = tuple(v[1:])
            continue
        if k in ('dtype', 'sparse') or k.endswith('_initializer') or k.endswith('_constraint') or k == 'activity_regularizer':
            continue
        if k.endswith('_regularizer'):
            if isinstance(v, dict) and k == 'kernel_regularizer':
                c = v.get('config', {})
                if c.get('l1', 0.0):
                    raise NotImplementedError('l1 kernel regulariser in the checkpoint')
                kw[k] = L1L2(l2=float(c.get('l2', 0.0)))
            continue
        if k not in params and k not in ('name', 'trainable'):      # (the base Layer takes name / trainable / input_shape)
            # a Keras option this layer does not implement: silent only at the value that changes nothing
            if k != 'implementation' and _KERAS_NEUTRAL.get(k, _MISSING) != v:
                import warnings
                warnings.warn('checkpoint layer %r: Keras option %s=%r has no counterpart in %s and is ignored'
                              % (cfg.get('name'), k, v, cls.__name__))
            continue
        if isinstance(v, list):
            v = tuple(tuple(e) if isinstance(e, list) else e for e in v)
        kw[k] = v
    return kw


_MISSING = object()
#: Keras 2.2 defaults of constructor arguments the layers here do not take: dropping them at these values changes nothing
_KERAS_NEUTRAL = {'return_state': False, 'go_backwards': False, 'stateful': False, 'unroll': False, 'dropout': 0.0,
                  'recurrent_dropout': 0.0, 'interpolation': 'nearest', 'strides': (1, 1), 'unit_forget_bias': True,
                  'use_bias': True, 'data_format': 'channels_first', 'padding': 'valid', 'dilation_rate': (1, 1)}


def import_keras_hdf5(path, custom_objects=None, device=None, compile=True):
    """Build a Model from a Keras 2.x HDF5 checkpoint (`keras.models.save_model` layout: root attribute `model_config` = JSON
    of the Sequential / functional graph, group `model_weights/<layer>/<weight names>`, optional `training_config`).  The
    container is read by dlwp_amd.hdf5_lite (no h5py for this interpreter).  Layers are looked up by class name in
    keras.layers / DLWP.custom as this package provides them; a `Lambda` (the reference's slice_layer) carries marshalled
    Python bytecode and cannot be imported -- NotImplementedError names it."""
    from . import custom as C
    from . import engine, hdf5_lite
    from . import layers as L
    f = hdf5_lite.File(path)

    def text(v):
        return v.decode('utf-8') if isinstance(v, (bytes, np.bytes_)) else str(v)

    def attr_list(group, name):      # keras.engine.saving.load_attributes_from_hdf5_group: large lists are split into chunks
        if name in group.attrs:
            return [text(n) for n in np.asarray(group.attrs[name]).reshape(-1)]
        out, k = [], 0
        while '%s%d' % (name, k) in group.attrs:
            out += [text(n) for n in np.asarray(group.attrs['%s%d' % (name, k)]).reshape(-1)]
            k += 1
        return out
    if 'model_config' not in f.attrs:
        raise ValueError('%s holds no model_config (a weights-only file? build the model and use load_weights semantics)' % path)
    config = json.loads(text(f.attrs['model_config']))
    registry = {}
    for mod in (L, C):
        registry.update({k: v for k, v in vars(mod).items() if isinstance(v, type)})
    registry.update(custom_objects or {})

    def make(spec):
        cname = spec['class_name']
        if cname == 'Lambda':
            raise NotImplementedError('layer %r is a keras Lambda (marshalled Python code, e.g. DLWP.custom.slice_layer): rebuild '
                                      'the graph with dlwp_amd.custom.slice_layer and load the weights by name'
                                      % spec['config'].get('name'))
        if cname not in registry:
            raise NotImplementedError('layer class %r of the checkpoint has no counterpart here' % cname)
        cls = registry[cname]
        return cls(**_keras_layer_kwargs(cls, spec['config']))
    cls_name, mc = config['class_name'], config['config']
    layer_specs = mc if isinstance(mc, list) else mc['layers']          # Keras < 2.2.3 stored a Sequential as a bare list
    by_name = {}
    if cls_name == 'Sequential':
        objs = []
        for spec in layer_specs:
            if spec['class_name'] == 'InputLayer':
                continue
            objs.append(make(spec))
            by_name[spec['config']['name']] = objs[-1]
        first = layer_specs[0]['config']
        shp = first.get('batch_input_shape')
        if shp is None:
            raise ValueError('the first layer of the Sequential checkpoint has no batch_input_shape')
        t = L.Input(shape=tuple(shp[1:]))
        x0 = t
        for lay in objs:
            t = lay(t)
        model = engine.Model(inputs=x0, outputs=t, name=mc.get('name') if isinstance(mc, dict) else None, device=device)
        model._keras_class = 'Sequential'          # (a later save describes it as what it was)
    elif cls_name == 'Model':
        tensors = {}                                   # (layer name, node index) -> output tensor
        pending = []
        for spec in layer_specs:
            nm = spec['name']
            if spec['class_name'] == 'InputLayer':
                tensors[(nm, 0)] = L.Input(shape=tuple(spec['config']['batch_input_shape'][1:]), name=nm)
                continue
            by_name[nm] = make(spec)
            for k, node in enumerate(spec['inbound_nodes']):
                pending.append((nm, k, [(i[0], i[1]) for i in node]))
        while pending:                                 # nodes in dependency order, whatever order the file lists them in
            progressed = False
            for item in list(pending):
                nm, k, srcs = item
                if all(s in tensors for s in srcs):
                    ins = [tensors[s] for s in srcs]
                    tensors[(nm, k)] = by_name[nm](ins if isinstance(by_name[nm], L.Concatenate) or len(ins) > 1 else ins[0])
                    pending.remove(item)
                    progressed = True
            if not progressed:
                raise ValueError('the checkpoint graph has a cycle or a missing layer')
        model = engine.Model(inputs=[tensors[(i[0], i[1])] for i in mc['input_layers']],
                             outputs=[tensors[(o[0], o[1])] for o in mc['output_layers']], name=mc.get('name'), device=device)
    else:
        raise NotImplementedError('checkpoint of a %r' % cls_name)
    # ---- weights, by layer name: model_weights/<layer>/<weight name>, in the order of the layer's weight_names attribute
    mw = f['model_weights'] if 'model_weights' in f else f
    for lname in attr_list(mw, 'layer_names'):
        g = mw[lname]
        names = attr_list(g, 'weight_names')
        if not names:
            continue
        if lname not in by_name:
            raise ValueError('checkpoint layer %r holds weights but is not in the graph' % lname)
        by_name[lname].set_weights([np.asarray(g[n][...]) for n in names])
    tc = f.attrs.get('training_config') if compile else None
    if tc is not None:
        tc = json.loads(text(tc))
        from . import training
        oc = tc.get('optimizer_config', {})
        ocls = getattr(training, oc.get('class_name', 'Adam'), None)
        loss = tc.get('loss')
        import warnings
        if isinstance(loss, str) and loss in (custom_objects or {}):
            loss = custom_objects[loss]         # the reference's closures (`lat_loss`, `acc_loss`) arrive by name, custom.py:956-1088
        elif 'dlwp_amd' in f and 'loss_spec' in f['dlwp_amd'].attrs:
            # a checkpoint this package wrote (export_keras_hdf5): the custom loss with its arrays rides in a private group
            priv = f['dlwp_amd']
            sp = json.loads(text(priv.attrs['loss_spec']))
            loss = C.LossSpec(sp['kind'], sp['regularize'], np.asarray(priv['loss_mean'][...]) if sp['has_mean'] else None,
                              np.asarray(priv['loss_row_weights'][...]) if sp['has_row_weights'] else None, sp['scale'], sp['name'])
        known = isinstance(loss, C.LossSpec) or (isinstance(loss, str) and
                                                  loss in ('mse', 'mean_squared_error', 'mae', 'mean_absolute_error'))
        if ocls is not None and known:
            ocfg = {k: v for k, v in oc.get('config', {}).items() if isinstance(v, (int, float))}
            metrics = [m for m in (tc.get('metrics') or []) if m in ('mae', 'mse', 'mean_absolute_error', 'mean_squared_error')]
            ocfg.pop('amsgrad', None)
            model.compile(optimizer=ocls(**ocfg), loss=loss, metrics=metrics, loss_weights=tc.get('loss_weights'))
            if 'optimizer_weights' in f:
                # keras.engine.saving.load_model: optimizer.set_weights(values in the order of weight_names) -- Adam: iterations,
                # first moments, second moments (+ one placeholder per weight); SGD: iterations, moments
                import torch
                ow = f['optimizer_weights']
                vals = [np.asarray(ow[n][...]) for n in attr_list(ow, 'weight_names')]
                tr = model._trainer
                n_par = len(tr.entries)
                n_slots = 2 if oc.get('class_name') == 'Adam' else 1
                shapes = [tuple(e[4]) for e in tr.entries]
                ok = len(vals) >= 1 + n_slots * n_par and all(
                    tuple(vals[1 + sl * n_par + i].shape) == shapes[i] for sl in range(n_slots) for i in range(n_par))
                if ok:
                    model.optimizer.iterations = int(np.asarray(vals[0]).reshape(-1)[0])
                    tr.opt_state = tuple(
                        torch.from_numpy(np.concatenate([np.asarray(vals[1 + sl * n_par + i], dtype=np.float32).reshape(-1)
                                                         for i in range(n_par)])).to(model.device) for sl in range(n_slots))
                else:
                    warnings.warn('%s: the optimizer state of the checkpoint does not match the weights of the graph; training '
                                  'resumes with fresh moments' % path)
        else:
            warnings.warn('%s: training_config with loss %r / optimizer %r is not recognised (pass the loss through '
                          'custom_objects); the model is returned UNCOMPILED' % (path, loss, oc.get('class_name')))
    return model


def load_model_file(path, custom_objects=None, device=None):
    from . import custom as C
    from . import engine
    from . import layers as L
    from . import hdf5_lite
    if hdf5_lite.is_hdf5(path):          # a real Keras checkpoint (reference DLWP/util.py:141-144)
        return import_keras_hdf5(path, custom_objects=custom_objects, device=device)
    with open(path, 'rb') as f:
        data = np.load(io.BytesIO(f.read()), allow_pickle=False)
    arch = json.loads(bytes(data['__arch__']).decode('utf-8'))
    if arch.get('format') != FORMAT:
        raise ValueError('%s is not a %s file' % (path, FORMAT))
    registry = {}
    for mod in (L, C):
        registry.update({k: v for k, v in vars(mod).items() if isinstance(v, type)})
    registry.update(custom_objects or {})
    objs = []
    for spec in arch['layers']:
        cfg = dict(spec['config'])
        cls = registry[spec['class']]
        if 'padding' in cfg and isinstance(cfg['padding'], list):
            cfg['padding'] = tuple(tuple(p) for p in cfg['padding'])
        for k in ('kernel_size', 'dilation_rate', 'strides', 'target_shape', 'input_shape'):
            if k in cfg and isinstance(cfg[k], list):
                cfg[k] = tuple(cfg[k])
        if isinstance(cfg.get('kernel_regularizer'), dict):
            from .regularizers import L1L2
            cfg['kernel_regularizer'] = L1L2(l2=cfg['kernel_regularizer']['l2'])
        objs.append(cls(**cfg))
    tensors = []
    for node in arch['nodes']:
        lay = objs[node['layer']]
        if isinstance(lay, L.InputLayer):
            tensors.append(L.KTensor(lay.batch_input_shape[1:], lay, ()))
        else:
            ins = [tensors[i] for i in node['inputs']]
            tensors.append(lay(ins if isinstance(lay, L.Concatenate) else ins[0]))
    model = engine.Model(inputs=[tensors[i] for i in arch['inputs']], outputs=[tensors[i] for i in arch['outputs']],
                         name=arch.get('name'), device=device)
    for lay in objs:
        if lay._weights:
            lay.set_weights([data['%s/%s' % (lay.name, nm)] for nm, _ in lay._weights])
    comp = arch.get('compile')
    if comp:
        from . import training
        ocls = getattr(training, comp['optimizer']['class'])
        ocfg = dict(comp['optimizer']['config'])
        iters = int(ocfg.pop('iterations', 0))
        opt = ocls(**ocfg)
        opt.iterations = iters
        loss = comp['loss']
        if isinstance(loss, dict):
            sp = loss['spec']
            loss = C.LossSpec(sp['kind'], sp['regularize'], data['__loss_mean__'] if sp['has_mean'] else None,
                              data['__loss_row_weights__'] if sp['has_row_weights'] else None, sp['scale'], sp['name'])
        model.compile(optimizer=opt, loss=loss, metrics=comp['metrics'], loss_weights=comp['loss_weights'])
        slots = []
        while '__opt_state_%d__' % len(slots) in data.files:
            slots.append(data['__opt_state_%d__' % len(slots)])
        tr = model._trainer
        if slots and all(a.size == tr.flat_params.numel() for a in slots):
            import torch
            tr.opt_state = tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(model.device)
                                 for a in slots)
        elif iters:
            opt.iterations = 0      # a file without the moments: restart the bias correction with them (m = v = 0)
    return model
