"""
Model files.  The reference writes `<name>.keras` as Keras HDF5 (DLWP/util.py:126-153); h5py is absent here, so the same
file name holds a numpy .npz archive instead: a JSON description of the layer graph + compile arguments, and every
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


def load_model_file(path, custom_objects=None, device=None):
    from . import custom as C
    from . import engine
    from . import layers as L
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
