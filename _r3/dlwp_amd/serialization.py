"""
Model files.  The reference writes `<name>.keras` as Keras HDF5 (DLWP/util.py:126-153); h5py is absent here, so files written
by THIS package hold a numpy .npz archive under the same name (Keras HDF5 checkpoints are READ: import_keras_hdf5, through the
pure-numpy container reader dlwp_amd.hdf5_lite): a JSON description of the layer graph + compile arguments, and every
weight array in KERAS LAYOUT (conv kernels (kh, kw, cin, cout), biases (cout,)) under the Keras-style key
`<layer name>/<weight name>` -- an offline converter can move real Keras checkpoints in either direction.
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


def save_model_file(model, path):
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
        known = isinstance(loss, C.LossSpec) or (isinstance(loss, str) and
                                                  loss in ('mse', 'mean_squared_error', 'mae', 'mean_absolute_error'))
        if ocls is not None and known:
            ocfg = {k: v for k, v in oc.get('config', {}).items() if isinstance(v, (int, float))}
            metrics = [m for m in (tc.get('metrics') or []) if m in ('mae', 'mse', 'mean_absolute_error', 'mean_squared_error')]
            model.compile(optimizer=ocls(**ocfg), loss=loss, metrics=metrics, loss_weights=tc.get('loss_weights'))
            if 'optimizer_weights' in f:
                warnings.warn('%s: the optimizer state of the checkpoint (iteration count and slot variables) is not imported; '
                              'training resumes with fresh moments' % path)
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
