"""
Layer objects with the Keras names and constructor signatures the reference resolves through its registry
(`util.get_from_class('keras.layers', name)`, DLWP/model/models.py:97-103).  They are *descriptions*: a layer records
its hyper-parameters and output shape and owns its weights (device tensors in Keras layout); it never computes.  The
graph of layers is lowered by dlwp_amd.plan into fused libdlwp_hip.so launches.

Shapes are per-sample tuples without the batch axis, as in Keras `input_shape=`.
"""
import itertools
import math

import numpy as np

_uid = itertools.count(1)
_name_counts = {}


def _auto_name(cls_name):
    # keras-style snake_case + running index: Conv2D -> conv2d_1, PeriodicPadding2D -> periodic_padding2d_1
    out = []
    for i, ch in enumerate(cls_name):
        if ch.isupper() and i and (cls_name[i - 1].islower() or (i + 1 < len(cls_name) and cls_name[i + 1].islower())):
            out.append('_')
        out.append(ch.lower())
    base = ''.join(out).replace('2_d', '2d').replace('3_d', '3d')
    _name_counts[base] = _name_counts.get(base, 0) + 1
    return '%s_%d' % (base, _name_counts[base])


def normalize_data_format(value):
    """Keras: None -> the image_data_format default, which is 'channels_last'."""
    if value is None:
        return 'channels_last'
    if value not in ('channels_first', 'channels_last'):
        raise ValueError("data_format must be 'channels_first' or 'channels_last', got %r" % (value,))
    return value


def normalize_padding(padding, rank):
    """int | tuple of ints | tuple of pairs  ->  ((lo, hi),) * rank   (keras ZeroPadding2D/3D argument forms)."""
    if isinstance(padding, (int, np.integer)):
        return tuple((int(padding), int(padding)) for _ in range(rank))
    if not hasattr(padding, '__len__') or len(padding) != rank:
        raise ValueError('`padding` should have %d elements. Found: %r' % (rank, padding))
    out = []
    for p in padding:
        if isinstance(p, (int, np.integer)):
            out.append((int(p), int(p)))
        else:
            p = tuple(int(v) for v in p)
            if len(p) != 2:
                raise ValueError('each `padding` entry must be an int or a tuple of 2 ints. Found: %r' % (p,))
            out.append(p)
    for lo, hi in out:
        if lo < 0 or hi < 0:
            raise ValueError('negative padding is not supported: %r' % (padding,))
    return tuple(out)


class KTensor(object):
    """Symbolic tensor: per-sample shape + the node (layer application) that produced it."""

    def __init__(self, shape, layer=None, inputs=(), index=0):
        self.shape = tuple(int(s) for s in shape)
        self.layer = layer
        self.inputs = tuple(inputs)
        self.index = index
        self.uid = next(_uid)

    @property
    def _keras_shape(self):
        return (None,) + self.shape

    def __repr__(self):
        return '<KTensor %s from %s>' % ((None,) + self.shape, self.layer.name if self.layer else 'input')


class Layer(object):
    """Base class; also the extension point DLWP.custom-style layers subclass."""

    def __init__(self, input_shape=None, name=None, trainable=True, **kwargs):
        kwargs.pop('batch_input_shape', None)
        kwargs.pop('dtype', None)
        if kwargs:
            raise TypeError('unexpected keyword arguments for %s: %r' % (type(self).__name__, sorted(kwargs)))
        self.name = name or _auto_name(type(self).__name__)
        self.trainable = trainable
        self.batch_input_shape = (None,) + tuple(input_shape) if input_shape is not None else None
        self.input_shape = None
        self.output_shape = None
        self.built = False
        self._weights = []          # list of (name, device tensor) -- filled by build()
        self._calls = 0

    # -- protocol ------------------------------------------------------------------------------------------------ #
    def compute_output_shape(self, input_shape):
        """per-sample input shape(s) -> per-sample output shape"""
        return tuple(input_shape)

    def build(self, input_shape, device, rng):
        self.built = True

    def __call__(self, inputs):
        multi = isinstance(inputs, (list, tuple))
        ins = list(inputs) if multi else [inputs]
        for t in ins:
            if not isinstance(t, KTensor):
                raise TypeError('%s must be called on symbolic tensors (Input(...) or the output of another layer); '
                                'got %r' % (self.name, type(t)))
        in_shape = [t.shape for t in ins] if multi else ins[0].shape
        out_shape = self.compute_output_shape(in_shape)
        if self._calls == 0:
            self.input_shape = [(None,) + tuple(s) for s in in_shape] if multi else (None,) + tuple(in_shape)
            self.output_shape = (None,) + tuple(out_shape)
        self._calls += 1
        return KTensor(out_shape, self, ins, self._calls - 1)

    # -- weights --------------------------------------------------------------------------------------------------- #
    @property
    def weights(self):
        return [w for _, w in self._weights]

    def get_weights(self):
        return [w.detach().cpu().numpy() for _, w in self._weights]

    def set_weights(self, arrays):
        import torch
        if len(arrays) != len(self._weights):
            raise ValueError('layer %s expects %d weight arrays, got %d' % (self.name, len(self._weights), len(arrays)))
        for (nm, w), a in zip(self._weights, arrays):
            a = np.asarray(a, dtype=np.float32)
            if tuple(a.shape) != tuple(w.shape):
                raise ValueError('layer %s weight %s: shape %s != %s' % (self.name, nm, a.shape, tuple(w.shape)))
            w.copy_(torch.from_numpy(np.ascontiguousarray(a)))

    def count_params(self):
        return int(sum(int(np.prod(w.shape)) for _, w in self._weights))

    def get_config(self):
        return {'name': self.name, 'trainable': self.trainable}


class InputLayer(Layer):
    def __init__(self, input_shape=None, name=None, **kwargs):
        super(InputLayer, self).__init__(input_shape=input_shape, name=name, **kwargs)
        self.output_shape = self.batch_input_shape
        self.input_shape = self.batch_input_shape


def Input(shape=None, name=None, **kwargs):
    """keras.layers.Input: a symbolic placeholder with per-sample `shape` (examples/train_functional.py:154)."""
    if shape is None:
        raise ValueError('Input() needs shape=')
    layer = InputLayer(input_shape=tuple(shape), name=name or _auto_name('Input'))
    return KTensor(tuple(shape), layer, ())


class _Pad2DBase(Layer):
    """Shared argument handling of ZeroPadding2D / PeriodicPadding2D / FillPadding2D (they differ only in mode)."""
    mode = 0  # _lib.PAD_ZERO

    def __init__(self, padding=(1, 1), data_format=None, **kwargs):
        super(_Pad2DBase, self).__init__(**kwargs)
        self.padding = normalize_padding(padding, 2)
        self.data_format = normalize_data_format(data_format)

    def compute_output_shape(self, s):
        if len(s) != 3:
            raise ValueError('%s expects 4D input (batch + 3), got per-sample shape %r' % (self.name, s))
        (t, b), (l, r) = self.padding
        if self.data_format == 'channels_first':
            return (s[0], s[1] + t + b, s[2] + l + r)
        return (s[0] + t + b, s[1] + l + r, s[2])

    def get_config(self):
        cfg = super(_Pad2DBase, self).get_config()
        cfg.update({'padding': self.padding, 'data_format': self.data_format})
        return cfg


class ZeroPadding2D(_Pad2DBase):
    """keras.layers.ZeroPadding2D -- the pole-row halo of every reference stack (examples/train.py:163 ...)."""
    mode = 0


class Conv2D(Layer):
    """keras.layers.Conv2D: stride 1, 'valid' | 'same', dilation, bias, activation in {None,'linear','tanh','relu'},
    data_format='channels_first' (what every reference call site passes: examples/train.py:164-219).
    Weights: kernel (kh, kw, cin, cout) glorot_uniform, bias (cout,) zeros -- Keras layout and initialisers."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', data_format=None, dilation_rate=(1, 1),
                 activation=None, use_bias=True, kernel_initializer='glorot_uniform', bias_initializer='zeros',
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, **kwargs):
        super(Conv2D, self).__init__(**kwargs)
        self.filters = int(filters)
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, (int, np.integer)) else tuple(kernel_size)
        st = (strides, strides) if isinstance(strides, (int, np.integer)) else tuple(strides)
        dl = (dilation_rate, dilation_rate) if isinstance(dilation_rate, (int, np.integer)) else tuple(dilation_rate)
        if len(ks) != 2 or len(st) != 2 or len(dl) != 2:
            raise ValueError('kernel_size / strides / dilation_rate must be an int or a pair')
        if tuple(st) != (1, 1):
            raise NotImplementedError('Conv2D: only strides=1 is implemented (the reference never strides)')
        if padding not in ('valid', 'same'):
            raise ValueError("Conv2D padding must be 'valid' or 'same'")
        if callable(activation):
            activation = getattr(activation, '__name__', None)
        if activation not in (None, 'linear', 'tanh', 'relu'):
            raise NotImplementedError('Conv2D activation %r is not implemented (linear, tanh, relu are)' % (activation,))
        if kernel_initializer not in ('glorot_uniform', 'zeros') or bias_initializer not in ('zeros',):
            raise NotImplementedError('initialisers: kernel glorot_uniform|zeros, bias zeros')
        self.kernel_size = tuple(int(k) for k in ks)
        self.strides = (1, 1)
        self.padding = padding
        self.data_format = normalize_data_format(data_format)
        if self.data_format != 'channels_first':
            raise NotImplementedError("Conv2D: data_format='channels_first' is required (pass it explicitly, as the "
                                      "reference scripts do; Keras' default is channels_last)")
        self.dilation_rate = tuple(int(d) for d in dl)
        self.activation = activation or 'linear'
        self.use_bias = bool(use_bias)
        self.kernel_initializer = kernel_initializer
        self.kernel_regularizer = kernel_regularizer   # l2 handled by the trainer when it is a dlwp_amd.regularizers.L2
        self.kernel = None
        self.bias = None

    def same_halo(self):
        """Keras 'same' for stride 1: total pad = dil*(k-1), smaller half first."""
        tot_h = self.dilation_rate[0] * (self.kernel_size[0] - 1)
        tot_w = self.dilation_rate[1] * (self.kernel_size[1] - 1)
        return (tot_h // 2, tot_h - tot_h // 2, tot_w // 2, tot_w - tot_w // 2)

    def compute_output_shape(self, s):
        if len(s) != 3:
            raise ValueError('%s expects 4D input, got per-sample shape %r' % (self.name, s))
        c, h, w = s
        if self.padding == 'same':
            return (self.filters, h, w)
        ho = h - self.dilation_rate[0] * (self.kernel_size[0] - 1)
        wo = w - self.dilation_rate[1] * (self.kernel_size[1] - 1)
        if ho <= 0 or wo <= 0:
            raise ValueError('%s: kernel %r with dilation %r does not fit the input %r' %
                             (self.name, self.kernel_size, self.dilation_rate, s))
        return (self.filters, ho, wo)

    def build(self, input_shape, device, rng):
        import torch
        if self.built:
            if self.kernel.shape[2] != input_shape[0]:
                raise ValueError('%s was built for %d input channels, now called with %d' %
                                 (self.name, self.kernel.shape[2], input_shape[0]))
            return
        kh, kw = self.kernel_size
        cin = int(input_shape[0])
        if self.kernel_initializer == 'glorot_uniform':
            limit = math.sqrt(6.0 / (kh * kw * cin + kh * kw * self.filters))
            k = rng.uniform(-limit, limit, size=(kh, kw, cin, self.filters)).astype(np.float32)
        else:
            k = np.zeros((kh, kw, cin, self.filters), dtype=np.float32)
        self.kernel = torch.from_numpy(k).to(device)
        self._weights = [('kernel', self.kernel)]
        if self.use_bias:
            self.bias = torch.zeros(self.filters, dtype=torch.float32, device=device)
            self._weights.append(('bias', self.bias))
        self.built = True

    def get_config(self):
        cfg = super(Conv2D, self).get_config()
        cfg.update({'filters': self.filters, 'kernel_size': self.kernel_size, 'padding': self.padding,
                    'data_format': self.data_format, 'dilation_rate': self.dilation_rate,
                    'activation': self.activation, 'use_bias': self.use_bias})
        return cfg


class RowConnected2D(Layer):
    """DLWP.custom.RowConnected2D (reference DLWP/custom.py:695-837; a keras LocallyConnected2D subclass): a convolution whose
    filters are shared along a row only.  kernel (output_rows, kh, kw, cin, filters), bias (output_rows, 1, filters)
    (custom.py:800-816), glorot_uniform / zeros as Keras initialises them for those shapes (every axis in front of the last
    two counts as receptive field).  As in the reference only padding='valid' exists; on the HIP path the layer needs
    strides 1 and data_format='channels_first' (the call sites: examples/train_functional.py:191-196)."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', data_format=None, activation=None,
                 use_bias=True, kernel_initializer='glorot_uniform', bias_initializer='zeros', kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None, **kwargs):
        super(RowConnected2D, self).__init__(**kwargs)
        self.filters = int(filters)
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, (int, np.integer)) else tuple(kernel_size)
        st = (strides, strides) if isinstance(strides, (int, np.integer)) else tuple(strides)
        if len(ks) != 2 or len(st) != 2:
            raise ValueError('kernel_size / strides must be an int or a pair')
        if str(padding).lower() != 'valid':        # keras LocallyConnected2D.__init__
            raise ValueError('Invalid border mode for LocallyConnected2D (only "valid" is supported): ' + str(padding))
        if tuple(st) != (1, 1):
            raise NotImplementedError('RowConnected2D: only strides=1 is implemented (the reference never strides it)')
        if callable(activation):
            activation = getattr(activation, '__name__', None)
        if activation not in (None, 'linear', 'tanh', 'relu'):
            raise NotImplementedError('RowConnected2D activation %r is not implemented (linear, tanh, relu are)' % (activation,))
        if kernel_initializer not in ('glorot_uniform', 'zeros') or bias_initializer not in ('zeros',):
            raise NotImplementedError('initialisers: kernel glorot_uniform|zeros, bias zeros')
        self.kernel_size = tuple(int(k) for k in ks)
        self.strides = (1, 1)
        self.padding = 'valid'
        self.dilation_rate = (1, 1)
        self.data_format = normalize_data_format(data_format)
        if self.data_format != 'channels_first':
            raise NotImplementedError("RowConnected2D: data_format='channels_first' is required (pass it explicitly, as the "
                                      "reference scripts do; Keras' default is channels_last)")
        self.activation = activation or 'linear'
        self.use_bias = bool(use_bias)
        self.kernel_initializer = kernel_initializer
        self.kernel_regularizer = kernel_regularizer
        self.kernel = None
        self.bias = None
        self.output_row = self.output_col = None
        self.kernel_shape = None

    def compute_output_shape(self, s):
        if len(s) != 3:
            raise ValueError('%s expects 4D input, got per-sample shape %r' % (self.name, s))
        c, h, w = s
        ho, wo = h - self.kernel_size[0] + 1, w - self.kernel_size[1] + 1
        if ho <= 0 or wo <= 0:
            raise ValueError('%s: kernel %r does not fit the input %r' % (self.name, self.kernel_size, s))
        return (self.filters, ho, wo)

    def build(self, input_shape, device, rng):
        """input_shape: per-sample (cin, h, w) AFTER the halo in front -- the row count of the weights is the output height
        (custom.py:794-805)."""
        import torch
        cin, h, w = (int(v) for v in input_shape)
        kh, kw = self.kernel_size
        rows, cols = h - kh + 1, w - kw + 1
        if self.built:
            if tuple(self.kernel.shape) != (rows, kh, kw, cin, self.filters):
                raise ValueError('%s was built for a kernel of shape %r, now called on an input that needs %r' %
                                 (self.name, tuple(self.kernel.shape), (rows, kh, kw, cin, self.filters)))
            return
        self.output_row, self.output_col = rows, cols
        self.kernel_shape = (rows, kh, kw, cin, self.filters)
        if self.kernel_initializer == 'glorot_uniform':
            rf = rows * kh * kw                       # keras.initializers._compute_fans on a rank-5 shape
            limit = math.sqrt(6.0 / (rf * cin + rf * self.filters))
            k = rng.uniform(-limit, limit, size=self.kernel_shape).astype(np.float32)
        else:
            k = np.zeros(self.kernel_shape, dtype=np.float32)
        self.kernel = torch.from_numpy(k).to(device)
        self._weights = [('kernel', self.kernel)]
        if self.use_bias:
            self.bias = torch.zeros((rows, 1, self.filters), dtype=torch.float32, device=device)
            self._weights.append(('bias', self.bias))
        self.built = True

    def get_config(self):
        cfg = super(RowConnected2D, self).get_config()
        cfg.update({'filters': self.filters, 'kernel_size': self.kernel_size, 'strides': self.strides,
                    'padding': self.padding, 'data_format': self.data_format, 'activation': self.activation,
                    'use_bias': self.use_bias})
        return cfg


class _Pad3DBase(Layer):
    """ZeroPadding3D / PeriodicPadding3D argument handling (keras ZeroPadding3D forms).  On the HIP path these layers
    pad the recurrent (T, C, H, W) tensor in front of ConvLSTM2D (examples/train.py:144-147): channels_first makes T the
    'channel' axis, so `padding` addresses (C, H, W)."""
    mode = 0

    def __init__(self, padding=(1, 1, 1), data_format=None, **kwargs):
        super(_Pad3DBase, self).__init__(**kwargs)
        self.padding = normalize_padding(padding, 3)
        self.data_format = normalize_data_format(data_format)

    def compute_output_shape(self, s):
        if len(s) != 4:
            raise ValueError('%s expects 5D input (batch + 4), got per-sample shape %r' % (self.name, s))
        p = self.padding
        if self.data_format == 'channels_first':
            return (s[0],) + tuple(s[1 + k] + p[k][0] + p[k][1] for k in range(3))
        return tuple(s[k] + p[k][0] + p[k][1] for k in range(3)) + (s[3],)

    def get_config(self):
        cfg = super(_Pad3DBase, self).get_config()
        cfg.update({'padding': self.padding, 'data_format': self.data_format})
        return cfg


class ZeroPadding3D(_Pad3DBase):
    """keras.layers.ZeroPadding3D -- the pole-row halo in front of ConvLSTM2D (examples/train.py:147)."""
    mode = 0


class _ConvPart(object):
    """One of the two convolutions of a ConvLSTM2D step, presented to the planner / executor like a Conv2D layer: it
    reads the parent's weights through properties, so re-homed (flattened) parameters stay visible."""

    def __init__(self, parent, which):
        self.parent, self.which = parent, which
        self.name = '%s/%s' % (parent.name, 'input_conv' if which == 'kernel' else 'recurrent_conv')
        self.kernel_size = parent.kernel_size
        self.dilation_rate = parent.dilation_rate if which == 'kernel' else (1, 1)
        self.filters = 4 * parent.filters
        self.activation = 'linear'

    @property
    def kernel(self):
        return getattr(self.parent, self.which)

    @property
    def bias(self):
        return self.parent.bias if self.which == 'kernel' else None


class ConvLSTM2D(Layer):
    """keras.layers.ConvLSTM2D as the reference uses it (examples/train.py:148-155, train_functional.py:207-219):
    data_format='channels_first' input (T, C, H, W), stride 1, input convolution 'valid' | 'same' with dilation,
    recurrent convolution 'same' / zero halo / no dilation, gates i, f, c, o, activation tanh,
    recurrent_activation hard_sigmoid | sigmoid, unit_forget_bias, return_sequences.  Weights in Keras order and layout:
    kernel (kh, kw, C, 4F) glorot_uniform, recurrent_kernel (kh, kw, F, 4F) orthogonal, bias (4F,) zeros with the
    forget block at 1.  States start at zero on every call (stateful=False, as everywhere in the reference)."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', data_format=None, dilation_rate=(1, 1),
                 activation='tanh', recurrent_activation='hard_sigmoid', use_bias=True,
                 kernel_initializer='glorot_uniform', recurrent_initializer='orthogonal', bias_initializer='zeros',
                 unit_forget_bias=True, kernel_regularizer=None, recurrent_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, recurrent_constraint=None, bias_constraint=None,
                 return_sequences=False, go_backwards=False, stateful=False, dropout=0., recurrent_dropout=0., **kwargs):
        super(ConvLSTM2D, self).__init__(**kwargs)
        self.filters = int(filters)
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, (int, np.integer)) else tuple(kernel_size)
        st = (strides, strides) if isinstance(strides, (int, np.integer)) else tuple(strides)
        dl = (dilation_rate, dilation_rate) if isinstance(dilation_rate, (int, np.integer)) else tuple(dilation_rate)
        if tuple(st) != (1, 1):
            raise NotImplementedError('ConvLSTM2D: only strides=1 is implemented')
        if padding not in ('valid', 'same'):
            raise ValueError("ConvLSTM2D padding must be 'valid' or 'same'")
        if callable(activation):
            activation = getattr(activation, '__name__', None)
        if activation not in (None, 'linear', 'tanh', 'relu'):
            raise NotImplementedError('ConvLSTM2D activation %r is not implemented' % (activation,))
        if recurrent_activation not in ('hard_sigmoid', 'sigmoid'):
            raise NotImplementedError('ConvLSTM2D recurrent_activation %r is not implemented' % (recurrent_activation,))
        if go_backwards or stateful or dropout or recurrent_dropout:
            raise NotImplementedError('ConvLSTM2D: go_backwards / stateful / dropout are not implemented')
        if any(k % 2 == 0 for k in ks):
            raise NotImplementedError("ConvLSTM2D: even kernel sizes (asymmetric 'same' recurrent halo) are not implemented")
        self.kernel_size = tuple(int(k) for k in ks)
        self.strides = (1, 1)
        self.padding = padding
        self.data_format = normalize_data_format(data_format)
        if self.data_format != 'channels_first':
            raise NotImplementedError("ConvLSTM2D: data_format='channels_first' is required")
        self.dilation_rate = tuple(int(d) for d in dl)
        self.activation = activation or 'linear'
        self.recurrent_activation = recurrent_activation
        self.use_bias = bool(use_bias)
        self.unit_forget_bias = bool(unit_forget_bias)
        self.kernel_regularizer = kernel_regularizer
        self.return_sequences = bool(return_sequences)
        self.kernel = self.recurrent_kernel = self.bias = None
        self.input_part = _ConvPart(self, 'kernel')
        self.recurrent_part = _ConvPart(self, 'recurrent_kernel')

    def same_halo(self):
        tot_h = self.dilation_rate[0] * (self.kernel_size[0] - 1)
        tot_w = self.dilation_rate[1] * (self.kernel_size[1] - 1)
        return (tot_h // 2, tot_h - tot_h // 2, tot_w // 2, tot_w - tot_w // 2)

    def compute_output_shape(self, s):
        if len(s) != 4:
            raise ValueError('%s expects 5D input (batch, time, channels, rows, cols), got per-sample shape %r' %
                             (self.name, s))
        t, c, h, w = s
        if self.padding == 'same':
            ho, wo = h, w
        else:
            ho = h - self.dilation_rate[0] * (self.kernel_size[0] - 1)
            wo = w - self.dilation_rate[1] * (self.kernel_size[1] - 1)
        if ho <= 0 or wo <= 0:
            raise ValueError('%s: kernel %r with dilation %r does not fit the input %r' %
                             (self.name, self.kernel_size, self.dilation_rate, s))
        return (t, self.filters, ho, wo) if self.return_sequences else (self.filters, ho, wo)

    def build(self, input_shape, device, rng):
        import torch
        cin = int(input_shape[1])
        if self.built:
            if self.kernel.shape[2] != cin:
                raise ValueError('%s was built for %d input channels, now called with %d' %
                                 (self.name, self.kernel.shape[2], cin))
            return
        kh, kw = self.kernel_size
        f = self.filters
        limit = math.sqrt(6.0 / (kh * kw * cin + kh * kw * 4 * f))
        k = rng.uniform(-limit, limit, size=(kh, kw, cin, 4 * f)).astype(np.float32)
        # keras.initializers.Orthogonal: SVD of a normal matrix flattened to (prod(shape[:-1]), shape[-1])
        a = rng.normal(0.0, 1.0, (kh * kw * f, 4 * f))
        u, _, vt = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == a.shape else vt
        r = q.reshape(kh, kw, f, 4 * f).astype(np.float32)
        self.kernel = torch.from_numpy(k).to(device)
        self.recurrent_kernel = torch.from_numpy(np.ascontiguousarray(r)).to(device)
        self._weights = [('kernel', self.kernel), ('recurrent_kernel', self.recurrent_kernel)]
        if self.use_bias:
            b = np.zeros(4 * f, dtype=np.float32)
            if self.unit_forget_bias:
                b[f:2 * f] = 1.0
            self.bias = torch.from_numpy(b).to(device)
            self._weights.append(('bias', self.bias))
        self.built = True

    def get_config(self):
        cfg = super(ConvLSTM2D, self).get_config()
        cfg.update({'filters': self.filters, 'kernel_size': self.kernel_size, 'padding': self.padding,
                    'data_format': self.data_format, 'dilation_rate': self.dilation_rate,
                    'activation': self.activation, 'recurrent_activation': self.recurrent_activation,
                    'use_bias': self.use_bias, 'unit_forget_bias': self.unit_forget_bias,
                    'return_sequences': self.return_sequences})
        return cfg


class MaxPooling2D(Layer):
    """keras.layers.MaxPooling2D(2): 2x2 / stride 2 / 'valid' (examples/train.py:171,181)."""

    def __init__(self, pool_size=(2, 2), strides=None, padding='valid', data_format=None, **kwargs):
        super(MaxPooling2D, self).__init__(**kwargs)
        ps = (pool_size, pool_size) if isinstance(pool_size, (int, np.integer)) else tuple(pool_size)
        st = ps if strides is None else ((strides, strides) if isinstance(strides, (int, np.integer)) else tuple(strides))
        if tuple(ps) != (2, 2) or tuple(st) != (2, 2) or padding != 'valid':
            raise NotImplementedError("MaxPooling2D: only pool_size=2, strides=2, padding='valid' is implemented")
        self.pool_size, self.strides, self.padding = (2, 2), (2, 2), 'valid'
        self.data_format = normalize_data_format(data_format)
        if self.data_format != 'channels_first':
            raise NotImplementedError("MaxPooling2D: data_format='channels_first' is required")

    def compute_output_shape(self, s):
        return (s[0], s[1] // 2, s[2] // 2)


class UpSampling2D(Layer):
    """keras.layers.UpSampling2D(2), nearest (examples/train.py:191,201)."""

    def __init__(self, size=(2, 2), data_format=None, interpolation='nearest', **kwargs):
        super(UpSampling2D, self).__init__(**kwargs)
        sz = (size, size) if isinstance(size, (int, np.integer)) else tuple(size)
        if tuple(sz) != (2, 2) or interpolation != 'nearest':
            raise NotImplementedError("UpSampling2D: only size=2, interpolation='nearest' is implemented")
        self.size = (2, 2)
        self.data_format = normalize_data_format(data_format)
        if self.data_format != 'channels_first':
            raise NotImplementedError("UpSampling2D: data_format='channels_first' is required")

    def compute_output_shape(self, s):
        return (s[0], s[1] * 2, s[2] * 2)


class Reshape(Layer):
    """keras.layers.Reshape(target_shape): a relabelling of the contiguous per-sample block (examples/train.py:220)."""

    def __init__(self, target_shape, **kwargs):
        super(Reshape, self).__init__(**kwargs)
        self.target_shape = tuple(int(v) for v in target_shape)

    def compute_output_shape(self, s):
        n_in = int(np.prod(s))
        tgt = list(self.target_shape)
        if tgt.count(-1) > 1:
            raise ValueError('Reshape: at most one -1')
        if -1 in tgt:
            known = int(np.prod([v for v in tgt if v != -1]))
            tgt[tgt.index(-1)] = n_in // max(known, 1)
        if int(np.prod(tgt)) != n_in:
            raise ValueError('total size of new array must be unchanged: %r -> %r' % (s, self.target_shape))
        return tuple(tgt)


class ChannelSlice(Layer):
    """What `slice_layer(start, end, axis=1)` builds (DLWP/custom.py:675-692): a window of the channel axis."""

    def __init__(self, start, end, step=None, axis=1, **kwargs):
        super(ChannelSlice, self).__init__(**kwargs)
        if step not in (None, 1):
            raise NotImplementedError('slice_layer: only step=None/1 is implemented')
        if axis != 1:
            raise NotImplementedError('slice_layer: only axis=1 (channels_first channel axis) is implemented')
        self.start, self.end, self.axis = start, end, axis

    def window(self, c):
        lo, hi, _ = slice(self.start, self.end).indices(c)
        if hi <= lo:
            raise ValueError('slice_layer(%r, %r) selects no channels out of %d' % (self.start, self.end, c))
        return lo, hi

    def compute_output_shape(self, s):
        lo, hi = self.window(s[0])
        return (hi - lo,) + tuple(s[1:])


class Concatenate(Layer):
    """keras.layers.Concatenate(axis=1) for channels_first maps (examples/train_functional.py:255,259,266,270)."""

    def __init__(self, axis=-1, **kwargs):
        super(Concatenate, self).__init__(**kwargs)
        self.axis = axis

    def compute_output_shape(self, shapes):
        if not isinstance(shapes, list) or len(shapes) < 2:
            raise ValueError('Concatenate needs a list of at least 2 inputs')
        rank = len(shapes[0]) + 1
        ax = self.axis if self.axis >= 0 else rank + self.axis
        if ax != 1:
            raise NotImplementedError('Concatenate: only the channel axis (axis=1, channels_first) is implemented')
        for s in shapes[1:]:
            if tuple(s[1:]) != tuple(shapes[0][1:]):
                raise ValueError('Concatenate: inputs must match except on the channel axis: %r' % (shapes,))
        return (sum(s[0] for s in shapes),) + tuple(shapes[0][1:])


def concatenate(inputs, axis=-1, **kwargs):
    return Concatenate(axis=axis, **kwargs)(list(inputs))
