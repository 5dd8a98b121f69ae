"""
numpy restatement of the reference hot path (float64 arithmetic unless stated).  TEST INFRASTRUCTURE ONLY -- see
oracle/__init__.py for who may import this and for the pinning status of each part.

All `file:line` citations are relative to /root/reference.
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------------------------ #
# padding  (PINNED by tests/golden/padding.npz)
# ------------------------------------------------------------------------------------------------------------------ #

def normalize_padding(padding, rank=2):
    """Keras ZeroPadding{2,3}D argument forms -> ((lo, hi),)*rank.  int | tuple of ints | tuple of pairs.
    (The reference inherits this from keras.layers.ZeroPadding2D, DLWP/custom.py:139,187-189.)"""
    if isinstance(padding, (int, np.integer)):
        return tuple((int(padding), int(padding)) for _ in range(rank))
    padding = tuple(padding)
    if len(padding) != rank:
        raise ValueError('`padding` should have %d elements, got %r' % (rank, padding))
    out = []
    for p in padding:
        if isinstance(p, (int, np.integer)):
            out.append((int(p), int(p)))
        else:
            p = tuple(int(v) for v in p)
            if len(p) != 2:
                raise ValueError('each padding entry must be an int or a pair, got %r' % (p,))
            out.append(p)
    return tuple(out)


def _spatial_axes(ndim, data_format):
    # channels_first: (N, C, d1..dk); channels_last: (N, d1..dk, C)
    return tuple(range(2, ndim)) if data_format == 'channels_first' else tuple(range(1, ndim - 1))


def _wrap_axis(a, axis, lo, hi):
    """One axis of the reference's periodic pad: cat(a[n-lo:n], a, a[0:hi]) -- DLWP/custom.py:197-204.
    Like the reference's slices this does NOT tile; lo/hi > n is invalid there (negative slice start) and here."""
    n = a.shape[axis]
    if lo > n or hi > n:
        raise ValueError('periodic padding (%d, %d) exceeds the axis length %d' % (lo, hi, n))
    head = np.take(a, range(n - lo, n), axis=axis)
    tail = np.take(a, range(0, hi), axis=axis)
    return np.concatenate([head, a, tail], axis=axis)


def periodic_padding2d(x, padding, data_format='channels_first'):
    """PeriodicPadding2D.call, DLWP/custom.py:191-214: pad the horizontal (W) first, then the vertical (H) using the
    already W-padded tensor, so corners are wrap-of-wrap."""
    (t, b), (l, r) = normalize_padding(padding, 2)
    ah, aw = _spatial_axes(x.ndim, data_format)
    y = _wrap_axis(x, aw, l, r)
    return _wrap_axis(y, ah, t, b)


def periodic_padding3d(x, padding, data_format='channels_first'):
    """PeriodicPadding3D.call, DLWP/custom.py:275-306: last spatial axis first, then the middle, then the first."""
    pads = normalize_padding(padding, 3)
    axes = _spatial_axes(x.ndim, data_format)
    y = x
    for ax, (lo, hi) in reversed(list(zip(axes, pads))):
        y = _wrap_axis(y, ax, lo, hi)
    return y


def _edge_axis(a, axis, lo, hi):
    first = np.take(a, [0], axis=axis)
    last = np.take(a, [a.shape[axis] - 1], axis=axis)
    return np.concatenate([first] * lo + [a] + [last] * hi, axis=axis)


def fill_padding2d(x, padding, data_format='channels_first'):
    """FillPadding2D.call, DLWP/custom.py:359-402: replicate edge rows first, then edge columns of the row-padded
    tensor."""
    (t, b), (l, r) = normalize_padding(padding, 2)
    ah, aw = _spatial_axes(x.ndim, data_format)
    y = _edge_axis(x, ah, t, b)
    return _edge_axis(y, aw, l, r)


def zero_padding2d(x, padding, data_format='channels_first'):
    """Keras ZeroPadding2D (third-party; call sites examples/train.py:163,173,183,193,203,212).  UNPINNED."""
    (t, b), (l, r) = normalize_padding(padding, 2)
    ah, aw = _spatial_axes(x.ndim, data_format)
    widths = [(0, 0)] * x.ndim
    widths[ah], widths[aw] = (t, b), (l, r)
    return np.pad(x, widths)


PAD_ZERO, PAD_WRAP, PAD_EDGE = 0, 1, 2


def pad2d_modes(x, pads, mode_h, mode_w):
    """Per-axis-mode pad of an NCHW tensor -- the form the fused HIP halo uses (include/dlwp_hip.h dlwp_pad2d).
    pads = (top, bottom, left, right).  Equivalent to composing the layer functions above one axis at a time."""
    t, b, l, r = pads
    def np_mode(name):
        return lambda a, ax, lo, hi: np.pad(a, [(lo, hi) if i == ax else (0, 0) for i in range(a.ndim)], mode=name)
    # 3 / 4: tf.pad 'REFLECT' / 'SYMMETRIC' (TFPadding2D, reference custom.py:527-600) == numpy's modes of the same names
    fn = {PAD_ZERO: np_mode('constant'), PAD_WRAP: _wrap_axis, PAD_EDGE: _edge_axis, 3: np_mode('reflect'),
          4: np_mode('symmetric')}
    y = fn[mode_w](x, x.ndim - 1, l, r)
    return fn[mode_h](y, x.ndim - 2, t, b)


def pad2d_modes_grad(dy, x_shape, pads, mode_h, mode_w):
    """Adjoint of pad2d_modes: fold the halo of dy back onto the interior (wrap: add to the periodic image;
    edge: add to the border row/column; zero: drop)."""
    t, b, l, r = pads
    H, W = x_shape[-2:]

    def fold(a, axis, lo, hi, n, mode):
        a = np.moveaxis(a, axis, -1)
        core = a[..., lo:lo + n].copy()
        if mode == PAD_WRAP:
            if lo:
                core[..., n - lo:] += a[..., :lo]
            if hi:
                core[..., :hi] += a[..., lo + n:]
        elif mode == PAD_EDGE:
            if lo:
                core[..., 0] += a[..., :lo].sum(-1)
            if hi:
                core[..., n - 1] += a[..., lo + n:].sum(-1)
        elif mode in (3, 4):       # mirror halos: padded position p is the image of one interior position
            off = 0 if mode == 3 else 1
            for p in range(lo):
                core[..., lo - p - off] += a[..., p]
            for p in range(lo + n, lo + n + hi):
                core[..., 2 * n - 2 + off - (p - lo)] += a[..., p]
        return np.moveaxis(core, -1, axis)
    g = fold(np.asarray(dy, dtype=np.float64), dy.ndim - 2, t, b, H, mode_h)
    return fold(g, dy.ndim - 1, l, r, W, mode_w)


# ------------------------------------------------------------------------------------------------------------------ #
# convolution / pooling / activations  (UNPINNED: Keras semantics)
# ------------------------------------------------------------------------------------------------------------------ #

def conv2d(x, w_hwio, bias=None, dilation=1, activation='linear'):
    """Keras Conv2D(padding='valid', strides=1, data_format='channels_first'): cross-correlation (no kernel flip),
    y[n,co,i,j] = b[co] + sum_{ci,u,v} x[n,ci,i+u*d,j+v*d] * w[u,v,ci,co]; call sites examples/train.py:164-219.
    float64 direct sum."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w_hwio, dtype=np.float64)
    kh, kw, cin, cout = w.shape
    n, c, h, wd = x.shape
    assert c == cin, (c, cin)
    d = int(dilation)
    ho, wo = h - d * (kh - 1), wd - d * (kw - 1)
    y = np.zeros((n, cout, ho, wo), dtype=np.float64)
    for u in range(kh):
        for v in range(kw):
            patch = x[:, :, u * d:u * d + ho, v * d:v * d + wo]            # [n, ci, ho, wo]
            y += np.einsum('nchw,co->nohw', patch, w[u, v], optimize=True)
    if bias is not None:
        y += np.asarray(bias, dtype=np.float64)[None, :, None, None]
    return activate(y, activation)


def conv2d_grads(x, w_hwio, dz, dilation=1):
    """Gradients of the valid cross-correlation above w.r.t. x, w and the bias, given dz = dL/d(pre-activation)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w_hwio, dtype=np.float64)
    dz = np.asarray(dz, dtype=np.float64)
    kh, kw, cin, cout = w.shape
    d = int(dilation)
    ho, wo = dz.shape[-2:]
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for u in range(kh):
        for v in range(kw):
            sl = (slice(None), slice(None), slice(u * d, u * d + ho), slice(v * d, v * d + wo))
            dx[sl] += np.einsum('nohw,co->nchw', dz, w[u, v], optimize=True)
            dw[u, v] = np.einsum('nchw,nohw->co', x[sl], dz, optimize=True)
    return dx, dw, dz.sum(axis=(0, 2, 3))


def row_conv2d(x, kernel, strides=(1, 1)):
    """DLWP.custom.row_conv2d, channels_first (reference DLWP/custom.py:840-896): output row i is the 'valid' convolution
    of the input rows slice(i * stride_row, i * stride_col + kh) (:881 -- the END uses the COLUMN stride; identical for
    equal strides, which is all the call sites pass) with kernel[i] (kh, kw, cin, cout), strides applied inside that
    convolution (:887); the rows are concatenated along the row axis (:891).  x: (n, cin, h, w); kernel: (rows, kh, kw, cin,
    cout).  float64 direct sum; K.conv2d itself is third-party (Keras cross-correlation)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    rows, kh, kw, cin, cout = k.shape
    sr, sc = strides
    out = []
    for i in range(rows):
        xi = x[:, :, i * sr:i * sc + kh, :]
        hi, wi = xi.shape[2], xi.shape[3]
        ho, wo = (hi - kh) // sr + 1, (wi - kw) // sc + 1
        y = np.zeros((x.shape[0], cout, ho, wo), dtype=np.float64)
        for u in range(kh):
            for v in range(kw):
                patch = xi[:, :, u:u + (ho - 1) * sr + 1:sr, v:v + (wo - 1) * sc + 1:sc]
                y += np.einsum('nchw,co->nohw', patch, k[i, u, v], optimize=True)
        out.append(y)
    return np.concatenate(out, axis=2)


def row_bias_channels_first(bias, rows, filters):
    """How the (rows, 1, filters) bias of RowConnected2D (reference DLWP/custom.py:812) lands on a channels_first output:
    K.bias_add (:834) is third-party -- Keras 2.2's tensorflow backend RESHAPES (does not transpose) a bias of rank
    ndim(x) - 1 to (1, filters, rows, 1) for 'channels_first', so output channel f, row r receives flat element
    f * rows + r of the stored array.  Returns that (filters, rows) view.  UNPINNED (Keras semantics)."""
    return np.asarray(bias).reshape(-1)[:rows * filters].reshape(filters, rows)


def row_connected2d(x, kernel, bias=None, activation='linear', strides=(1, 1)):
    """RowConnected2D.call, channels_first (reference DLWP/custom.py:825-837): row_conv2d, bias add, activation."""
    y = row_conv2d(x, kernel, strides)
    if bias is not None:
        rows, filters = np.asarray(kernel).shape[0], np.asarray(kernel).shape[-1]
        y = y + row_bias_channels_first(np.asarray(bias, dtype=np.float64), rows, filters)[None, :, :, None]
    return activate(y, activation)


def row_connected2d_grads(x, kernel, dz):
    """Gradients of row_conv2d (stride 1) w.r.t. x, the kernel and the stored (rows, 1, filters) bias, given dz."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    dz = np.asarray(dz, dtype=np.float64)
    rows, kh, kw, cin, cout = k.shape
    wo = dz.shape[3]
    dx, dk = np.zeros_like(x), np.zeros_like(k)
    for i in range(rows):
        for u in range(kh):
            for v in range(kw):
                dx[:, :, i + u, v:v + wo] += np.einsum('now,co->ncw', dz[:, :, i, :], k[i, u, v], optimize=True)
                dk[i, u, v] = np.einsum('ncw,now->co', x[:, :, i + u, v:v + wo], dz[:, :, i, :], optimize=True)
    db = dz.sum(axis=(0, 3)).reshape(-1).reshape(rows, 1, cout)     # (filters, rows) flat -> the stored shape (see above)
    return dx, dk, db


def activate(z, activation):
    if activation in (None, 'linear'):
        return z
    if activation == 'tanh':
        return np.tanh(z)
    if activation == 'relu':
        return np.maximum(z, 0.)
    raise ValueError('activation %r not restated' % (activation,))


def activation_grad(y, dy, activation):
    """dL/dz from dL/dy and the layer OUTPUT y."""
    if activation in (None, 'linear'):
        return dy
    if activation == 'tanh':
        return dy * (1. - y * y)
    if activation == 'relu':
        return dy * (y > 0)
    raise ValueError(activation)


def maxpool2(x):
    """Keras MaxPooling2D(2): 2x2 window, stride 2, 'valid' (floor) -- examples/train.py:171,181."""
    n, c, h, w = x.shape
    h2, w2 = h // 2, w // 2
    v = x[:, :, :h2 * 2, :w2 * 2].reshape(n, c, h2, 2, w2, 2)
    return v.max(axis=(3, 5))


def maxpool2_grad(x, dy):
    """Route dy to the FIRST maximal element of each window in row-major window order (TF MaxPoolGrad convention)."""
    n, c, h, w = x.shape
    h2, w2 = h // 2, w // 2
    v = x[:, :, :h2 * 2, :w2 * 2].reshape(n, c, h2, 2, w2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h2, w2, 4)
    arg = v.argmax(axis=-1)
    g = np.zeros(v.shape, dtype=np.float64)
    np.put_along_axis(g, arg[..., None], np.asarray(dy, dtype=np.float64)[..., None], axis=-1)
    dx = np.zeros(x.shape, dtype=np.float64)
    dx[:, :, :h2 * 2, :w2 * 2] = g.reshape(n, c, h2, w2, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h2 * 2, w2 * 2)
    return dx


def upsample2(x):
    """Keras UpSampling2D(2), nearest: y[i,j] = x[i//2, j//2] -- examples/train.py:191,201."""
    return x.repeat(2, axis=-2).repeat(2, axis=-1)


def upsample2_grad(dy):
    n, c, h, w = dy.shape
    return np.asarray(dy, dtype=np.float64).reshape(n, c, h // 2, 2, w // 2, 2).sum(axis=(3, 5))


def mse(y_true, y_pred):
    """Keras 'mse' reduced to the scalar the train loop reports (mean over every element)."""
    return float(np.mean(np.square(np.asarray(y_pred, np.float64) - np.asarray(y_true, np.float64))))


def mae(y_true, y_pred):
    return float(np.mean(np.abs(np.asarray(y_pred, np.float64) - np.asarray(y_true, np.float64))))


def glorot_uniform(shape_hwio, rng):
    """Keras glorot_uniform for a conv kernel: limit = sqrt(6 / (fan_in + fan_out)), fan_in = kh*kw*cin,
    fan_out = kh*kw*cout."""
    kh, kw, cin, cout = shape_hwio
    limit = math.sqrt(6. / (kh * kw * cin + kh * kw * cout))
    return rng.uniform(-limit, limit, size=shape_hwio).astype(np.float32)


def adam_keras_step(p, m, v, g, iteration, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, decay=0.):
    """One Keras-2.2-form Adam update, as the reference's own tracker restates it (DLWP/custom.py:34-40):
    t = it+1; lr' = lr/(1+decay*it); lr_t = lr'*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps)."""
    t = iteration + 1
    lr_ = lr / (1. + decay * iteration)
    lr_t = lr_ * math.sqrt(1. - beta_2 ** t) / (1. - beta_1 ** t)
    m = beta_1 * m + (1. - beta_1) * g
    v = beta_2 * v + (1. - beta_2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + epsilon)
    return p, m, v


# ------------------------------------------------------------------------------------------------------------------ #
# ConvLSTM2D  (PARITY UNPINNED: the arithmetic lives in Keras 2.2.x, keras/layers/convolutional_recurrent.py
# ConvLSTM2DCell.call, absent here; restated from its published algorithm.  Call sites: examples/train.py:148-155,
# examples/train_functional.py:207-219.  Cross-checked against oracle/torch_ref.conv_lstm2d, an independent restatement
# on torch-CPU F.conv2d.)
# ------------------------------------------------------------------------------------------------------------------ #

def hard_sigmoid(z):
    """Keras' default recurrent_activation: clip(0.2 z + 0.5, 0, 1)."""
    return np.clip(0.2 * z + 0.5, 0.0, 1.0)


def zero_padding3d(x, padding, data_format='channels_first'):
    """keras ZeroPadding3D on (N, A, d1, d2, d3): pads the last three axes for channels_first."""
    pads = normalize_padding(padding, 3)
    if data_format != 'channels_first':
        return np.pad(x, ((0, 0),) + pads + ((0, 0),))
    return np.pad(x, ((0, 0), (0, 0)) + pads)


def conv_lstm2d(x, kernel, recurrent_kernel, bias, dilation=1, padding='valid', activation='tanh',
                recurrent_activation='hard_sigmoid', return_sequences=True, bf16_storage=False, bf16_kernels=(),
                fused_cell_update=False):
    """x: (N, T, C, H, W) channels_first, already padded for the 'valid' input convolution.  kernel (kh,kw,C,4F),
    recurrent_kernel (kh,kw,F,4F), bias (4F,); gate order i, f, c, o.  Per step (ConvLSTM2DCell.call):
        z  = conv(x_t, kernel, dilation, padding) + bias + conv(h_{t-1}, recurrent_kernel, 'same', no dilation)
        i, f, o = rec_act(z_i), rec_act(z_f), rec_act(z_o);  c_t = f*c_{t-1} + i*act(z_c);  h_t = o*act(c_t)
    with h_{-1} = c_{-1} = 0.  Returns (N, T, F, Ho, Wo) or the last h (N, F, Ho, Wo).
    The product's config-4 mode: bf16_storage -- the gate pre-activations conv(x_t) + bias and conv(h_{t-1}) and every h_t
    are rounded to bfloat16 when they are stored (c_t and the gate arithmetic stay float32); bf16_kernels -- which of the
    two convolutions ('kernel', 'recurrent') run on the bf16 matrix cores: their kernel and input are rounded to bf16;
    fused_cell_update -- the convolution that completes a step's pre-activations (the input convolution on the first step,
    the recurrent one afterwards) applies the cell update in its epilogue (dlwp_convlstm_conv_fwd): ITS pre-activations are
    never stored, hence not rounded; the other convolution's (the input convolution's from the second step on) are;
    fused_cell_update='step': a later step is one launch computing both convolutions -- nothing stored, nothing rounded."""
    x = np.asarray(x, dtype=np.float64)
    if 'kernel' in bf16_kernels:
        kernel, x = round_bf16(kernel), round_bf16(x)
    if 'recurrent' in bf16_kernels:
        recurrent_kernel = round_bf16(recurrent_kernel)
    rnd = round_bf16 if bf16_storage else (lambda v: v)
    n, t_len = x.shape[:2]
    kh, kw, _, f4 = kernel.shape
    f = f4 // 4
    rec = {'hard_sigmoid': hard_sigmoid, 'sigmoid': lambda z: 1.0 / (1.0 + np.exp(-z))}[recurrent_activation]
    rkh, rkw = recurrent_kernel.shape[:2]
    h = c = None
    outs = []
    for t in range(t_len):
        xt = x[:, t]
        if padding == 'same':
            ph, pw = dilation * (kh - 1), dilation * (kw - 1)
            xt = np.pad(xt, ((0, 0), (0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)))
        z = conv2d(xt, kernel, bias, dilation, 'linear')
        # 'step': every step of t >= 1 is ONE launch (dlwp_convlstm_step_fwd) -- no pre-activation is ever stored
        if not ((fused_cell_update and h is None) or fused_cell_update == 'step'):
            z = rnd(z)
        if h is not None:
            hp = np.pad(h, ((0, 0), (0, 0), ((rkh - 1) // 2, rkh - 1 - (rkh - 1) // 2),
                            ((rkw - 1) // 2, rkw - 1 - (rkw - 1) // 2)))
            zh = conv2d(hp, recurrent_kernel, None, 1, 'linear')
            z = z + (zh if fused_cell_update else rnd(zh))
        zi, zf, zc, zo = z[:, :f], z[:, f:2 * f], z[:, 2 * f:3 * f], z[:, 3 * f:]
        c_new = rec(zi) * activate(zc, activation)
        if c is not None:
            c_new = c_new + rec(zf) * c
        c = c_new
        h = rnd(rec(zo) * activate(c, activation))
        outs.append(h)
    return np.stack(outs, axis=1) if return_sequences else h


def init_conv_lstm_weights(cin, filters, ks, rng):
    """Keras initialisers: kernel glorot_uniform over the whole (kh,kw,cin,4F) tensor, recurrent kernel here also
    glorot_uniform (Keras uses an orthogonal matrix; any fixed numbers serve a parity test), bias zeros with the forget
    block = 1 (unit_forget_bias)."""
    k = glorot_uniform(tuple(ks) + (cin, 4 * filters), rng)
    r = glorot_uniform(tuple(ks) + (filters, 4 * filters), rng)
    b = np.zeros(4 * filters, dtype=np.float32)
    b[filters:2 * filters] = 1.0
    return k, r, b


# ------------------------------------------------------------------------------------------------------------------ #
# Conv2D on a 2x nearest-neighbour up-sampled tensor restated on the tensor itself (the product's inference plan does
# this for the decoder layers of examples/train.py:191-219; csrc/phase.hip).  Restated here independently, from the
# definition: tap u of output row 2i + a reads up-sampled row 2i + a + u - pad, i.e. source row i + floor((a+u-pad)/2).
# ------------------------------------------------------------------------------------------------------------------ #

def phase_weights(w_hwio, bias, pad_top, pad_left):
    """w (kh,kw,cin,cout) -> (w2 (kh2,kw2,cin,4*cout), b2 (4*cout) | None, (lo_h, hi_h, lo_w, hi_w)): column
    (2a + b)*cout + co of w2 is the kernel of output phase (a, b) on the low-resolution tensor."""
    w = np.asarray(w_hwio, dtype=np.float64)
    kh, kw, cin, cout = w.shape
    off_h = [[(a + u - pad_top) // 2 for u in range(kh)] for a in (0, 1)]
    off_w = [[(b + v - pad_left) // 2 for v in range(kw)] for b in (0, 1)]
    lo_h, hi_h = min(min(r) for r in off_h), max(max(r) for r in off_h)
    lo_w, hi_w = min(min(r) for r in off_w), max(max(r) for r in off_w)
    w2 = np.zeros((hi_h - lo_h + 1, hi_w - lo_w + 1, cin, 4 * cout))
    for a in (0, 1):
        for b in (0, 1):
            col = (2 * a + b) * cout
            for u in range(kh):
                for v in range(kw):
                    w2[off_h[a][u] - lo_h, off_w[b][v] - lo_w, :, col:col + cout] += w[u, v]
    b2 = None if bias is None else np.tile(np.asarray(bias, dtype=np.float64), 4)
    return w2, b2, (lo_h, hi_h, lo_w, hi_w)


def depth_to_space2(y, cout):
    """(N, 4*cout, H, W) phase-major -> (N, cout, 2H, 2W)."""
    n, c4, h, w = y.shape
    out = np.zeros((n, cout, 2 * h, 2 * w), dtype=y.dtype)
    for a in (0, 1):
        for b in (0, 1):
            out[:, :, a::2, b::2] = y[:, (2 * a + b) * cout:(2 * a + b + 1) * cout]
    return out


# ------------------------------------------------------------------------------------------------------------------ #
# a tiny interpreter for the reference's (layer_name, args, kwargs) stacks  (examples/train.py:142-221)
# ------------------------------------------------------------------------------------------------------------------ #

def _conv_args(args, kwargs):
    filters = args[0] if len(args) > 0 else kwargs['filters']
    ks = args[1] if len(args) > 1 else kwargs['kernel_size']
    ks = (ks, ks) if isinstance(ks, int) else tuple(ks)
    dil = kwargs.get('dilation_rate', 1)
    dil = dil if isinstance(dil, int) else dil[0]
    return int(filters), ks, int(dil), kwargs.get('activation', None)


def init_weights(layers, in_channels, rng):
    """glorot_uniform kernels / zero biases for every Conv2D in a layer stack; returns [(w_hwio, b), ...]."""
    out, c = [], in_channels
    for name, args, kwargs in layers:
        if name == 'Conv2D':
            filters, ks, _, _ = _conv_args(args or (), kwargs or {})
            out.append((glorot_uniform(ks + (c, filters), rng), np.zeros(filters, dtype=np.float32)))
            c = filters
        elif name == 'RowConnected2D':       # needs the output row count: init_row_connected_weights()
            raise ValueError('init_weights: give RowConnected2D weights through init_row_connected_weights (row count)')
        elif name == 'ConvLSTM2D':
            # in_channels is then the per-time-step channel count C of the (T, C, H, W) input
            filters, ks, _, _ = _conv_args(args or (), kwargs or {})
            out.append(init_conv_lstm_weights(c, filters, ks, rng))
            c = filters
        elif name == 'Reshape':
            c = (args or kwargs['target_shape'])[0][0] if args else kwargs['target_shape'][0]
    return out


def init_row_connected_weights(rows, ks, cin, filters, rng, bias_scale=0.):
    """Keras glorot_uniform on the (rows, kh, kw, cin, filters) kernel of RowConnected2D (reference DLWP/custom.py:800-810):
    keras.initializers._compute_fans treats every axis in front of the last two as receptive field, so
    fan_in = rows kh kw cin, fan_out = rows kh kw filters.  Bias (rows, 1, filters): zeros (Keras default) or, for tests
    that must see the bias layout, uniform(-bias_scale, bias_scale)."""
    kh, kw = ks
    limit = np.sqrt(6.0 / (rows * kh * kw * (cin + filters)))
    k = rng.uniform(-limit, limit, size=(rows, kh, kw, cin, filters)).astype(np.float32)
    b = rng.uniform(-bias_scale, bias_scale, size=(rows, 1, filters)).astype(np.float32) if bias_scale else \
        np.zeros((rows, 1, filters), dtype=np.float32)
    return k, b


def round_bf16(a):
    """Round float values to the nearest bfloat16 (ties to even) and return them as float64: what a bf16 store + load
    of a float32 value does (v_cvt_pk_bf16_f32)."""
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    out = r.view(np.float32).astype(np.float64)
    return np.where(np.isnan(f), np.nan, out)


def run_layers(layers, x, weights, record=None, bf16_activations=False, bf16_weights=(), bf16_lstm=None, lstm_fused=False):
    """Execute a sequential stack exactly as the reference graph is laid out (one op per layer, unfused).
    bf16_activations: every Conv2D output except the model output is rounded to bfloat16 (the product's config-4 storage);
    bf16_weights: indices (among the weighted layers) of the Conv2D layers whose kernel is rounded to bfloat16 as well
    and whose input is rounded to bfloat16 (a no-op unless it is the float32 model input): the layers the product runs on
    the bf16 matrix cores; bf16_lstm: None, or the subset of ('kernel', 'recurrent') of the ConvLSTM2D convolutions that
    run on the bf16 matrix cores -- the ConvLSTM2D then also stores zx, zh and h as bfloat16 (conv_lstm2d)."""
    x = np.asarray(x, dtype=np.float64)
    wi = 0
    n_weighted = sum(1 for nm, _, _ in layers if nm in ('Conv2D', 'ConvLSTM2D', 'RowConnected2D'))
    for name, args, kwargs in layers:
        args, kwargs = args or (), kwargs or {}
        fmt = kwargs.get('data_format', 'channels_first')
        if name == 'PeriodicPadding2D':
            x = periodic_padding2d(x, args[0] if args else kwargs.get('padding', (1, 1)), fmt)
        elif name == 'FillPadding2D':
            x = fill_padding2d(x, args[0] if args else kwargs.get('padding', (1, 1)), fmt)
        elif name == 'ZeroPadding2D':
            x = zero_padding2d(x, args[0] if args else kwargs.get('padding', (1, 1)), fmt)
        elif name == 'TFPadding2D':      # tf.pad(mode) on the two spatial axes (custom.py:585-590), channels_first
            (t, b), (l, r) = normalize_padding(args[0] if args else kwargs.get('padding', (1, 1)), 2)
            m = {'CONSTANT': PAD_ZERO, 'REFLECT': 3, 'SYMMETRIC': 4}[kwargs.get('mode', 'CONSTANT').upper()]
            x = pad2d_modes(x, (t, b, l, r), m, m)
        elif name == 'Conv2D':
            _, _, dil, act = _conv_args(args, kwargs)
            w, b = weights[wi]
            if wi in bf16_weights:
                w, x = round_bf16(w), round_bf16(x)
            wi += 1
            x = conv2d(x, w, b, dil, act or 'linear')
            if bf16_activations and wi < n_weighted:
                x = round_bf16(x)       # config 4: every Conv2D output except the model output is stored as bfloat16
        elif name == 'PeriodicPadding3D':
            x = periodic_padding3d(x, args[0] if args else kwargs.get('padding', (1, 1, 1)), fmt)
        elif name == 'ZeroPadding3D':
            x = zero_padding3d(x, args[0] if args else kwargs.get('padding', (1, 1, 1)), fmt)
        elif name == 'ConvLSTM2D':
            _, _, dil, act = _conv_args(args, kwargs)
            k, r, b = weights[wi]
            wi += 1
            x = conv_lstm2d(x, k, r, b, dil, kwargs.get('padding', 'valid'), act or 'tanh',
                            kwargs.get('recurrent_activation', 'hard_sigmoid'), kwargs.get('return_sequences', False),
                            bf16_storage=bf16_lstm is not None, bf16_kernels=bf16_lstm or (), fused_cell_update=lstm_fused)
        elif name == 'RowConnected2D':
            _, _, _, act = _conv_args(args, kwargs)
            w, b = weights[wi]
            wi += 1
            x = row_connected2d(x, w, b, act or 'linear')
        elif name == 'MaxPooling2D':
            x = maxpool2(x)
        elif name == 'UpSampling2D':
            x = upsample2(x)
        elif name == 'Reshape':
            x = x.reshape((x.shape[0],) + tuple(args[0]))
        else:
            raise ValueError('layer %r not restated' % name)
        if record is not None:
            record.append((name, x))
    return x


# ------------------------------------------------------------------------------------------------------------------ #
# rollout bookkeeping  (PINNED by tests/golden/rollout.npz)
# ------------------------------------------------------------------------------------------------------------------ #

def _merge_time(series, n_slots, n_sample, time_dim, feature_shape, keep_time_dim):
    # DLWP/model/models.py:294-300 / 448-451: (T,N,time_dim,V,...) -> (T*time_dim, N, V, ...)
    series = series.reshape((n_slots, n_sample, time_dim, -1) + tuple(feature_shape[1:]))
    if keep_time_dim:
        return series
    order = (0, 2, 1) + tuple(range(3, series.ndim))
    return series.transpose(order).reshape((n_slots * time_dim, n_sample, -1) + tuple(feature_shape[1:]))


def predict_timeseries_nn(predict, predictors, time_steps, time_dim, is_recurrent=False, step_sequence=False,
                          keep_time_dim=False):
    """DLWPNeuralNet.predict_timeseries, DLWP/model/models.py:247-301.  `predict` maps a state array to the next one.
    Output is float32 whatever `predict` returns (:270) and is NOT truncated to time_steps."""
    time_steps = int(time_steps)
    if time_steps < 1:
        raise ValueError('time_steps must be an int > 0')
    n_fwd = time_steps if step_sequence else int(math.ceil(float(time_steps) / time_dim))
    state = np.array(predictors, copy=True)
    n_sample = state.shape[0]
    feature_shape = state.shape[2:] if is_recurrent else state.shape[1:]
    series = np.full((n_fwd,) + predictors.shape, np.nan, dtype=np.float32)
    for t in range(n_fwd):
        if not step_sequence:
            state = predict(state)                                          # :292
            series[t] = state
            continue
        out = predict(state)                                                # :281-290
        if is_recurrent:
            state = np.concatenate([state[:, 1:], out[:, :1]], axis=1)
        else:
            split = (n_sample, time_dim, -1) + tuple(feature_shape[1:])
            o5, s5 = out.reshape(split), state.reshape(split)
            state = np.concatenate([s5[:, 1:], o5[:, :1]], axis=1).reshape(predictors.shape)
        series[t] = out
    merged = _merge_time(series, n_fwd, n_sample, time_dim, feature_shape, keep_time_dim or step_sequence)
    if step_sequence and not keep_time_dim:
        merged = merged[:, :, 0]                                            # :296-297
    return merged


def predict_timeseries_functional(predict, predictors, time_steps, time_dim, n_outputs=1, is_recurrent=False,
                                  keep_time_dim=False):
    """DLWPFunctional.predict_timeseries, DLWP/model/models.py:414-452.  `predict` returns one array
    (n_outputs == 1) or a list of n_outputs arrays; the last one seeds the next call (:443-446)."""
    time_steps = int(time_steps)
    if time_steps < 1:
        raise ValueError('time_steps must be an int > 0')
    n_calls = int(math.ceil(time_steps / n_outputs / time_dim))
    n_slots = n_calls * n_outputs
    state = np.array(predictors, copy=True)
    n_sample = state.shape[0]
    feature_shape = state.shape[2:] if is_recurrent else state.shape[1:]
    series = np.full((n_slots,) + predictors.shape, np.nan, dtype=np.float32)
    for t in range(n_calls):
        result = predict(state)
        if n_outputs == 1:
            # :447 np.stack(result, axis=0) of a single ARRAY stacks its samples: with n_outputs == 1 slot t receives
            # the (N, ...) array itself.
            state = np.array(result, copy=True)
            series[t] = result
        else:
            state = np.array(result[-1], copy=True)
            series[t * n_outputs:(t + 1) * n_outputs] = np.stack(result, axis=0)
    return _merge_time(series, n_slots, n_sample, time_dim, feature_shape, keep_time_dim)


def estimator_next_state(p, r, k, es, idx_in, idx_out, keep_inputs=True, prefer_first_times=True, mean=None, sol=None,
                         sol_idx=None):
    """The input of the NEXT model call of TimeSeriesEstimator.predict, DLWP/model/extensions.py:214-240, on plain index
    arithmetic (the reference writes it with xarray labels; PINNED end to end by tests/golden/estimator.npz through
    dlwp_amd.model.extensions' host loop, which is this function inlined).
      p (n, t_in, C_in, H, W): the inputs of this call;  r (n, t_out, C_out, H, W): its prediction
      :215  p.reindex(sample = sample + k dt): row i takes row i + k, rows past the data are NaN
      :219-221  impute: the last es rows = the mean input (mean: (t_in, C_in, H, W))
      :224-228  insolation: channel sol_idx of the last es rows = sol (min(es, n), t_in, H, W)
      :232-240  the predicted channels idx_in <- idx_out: the last es input steps (keep_inputs), else the first / last t_in
                predicted steps."""
    n, t_in = p.shape[:2]
    nxt = np.full_like(p, np.nan)
    if k < n:
        nxt[:n - k] = p[k:]
    if mean is not None:
        nxt[-es:] = mean[np.newaxis]
    if sol is not None:
        nxt[-es:, :, sol_idx] = sol
    if keep_inputs:
        nxt[:, t_in - es:, idx_in] = r[:, :, idx_out]
    elif prefer_first_times:
        nxt[:, :, idx_in] = r[:, :t_in][:, :, idx_out]
    else:
        nxt[:, :, idx_in] = r[:, -t_in:][:, :, idx_out]
    return nxt


# ------------------------------------------------------------------------------------------------------------------ #
# data feed  (PINNED by tests/golden/generator.npz)
# ------------------------------------------------------------------------------------------------------------------ #

def delete_nan_samples(predictors, targets, large_fill_value=False, threshold=None):
    """DLWP/util.py:238-268: drop every sample (row) with a NaN in either array (or with a NaN fraction >= threshold)."""
    if threshold is not None and not (0 <= threshold <= 1):
        raise ValueError("'threshold' must be between 0 and 1")
    if large_fill_value:
        predictors[np.abs(predictors) >= 1.e20] = np.nan
        targets[np.abs(targets) >= 1.e20] = np.nan
    p2 = predictors.reshape((predictors.shape[0], -1))
    t2 = targets.reshape((targets.shape[0], -1))
    if threshold is None:
        bad = np.isnan(p2).any(axis=1) | np.isnan(t2).any(axis=1)
    else:
        bad = (np.isnan(p2).mean(axis=1) >= threshold) | (np.isnan(t2).mean(axis=1) >= threshold)
    keep = ~bad
    return predictors[keep], targets[keep]


class DataGeneratorRef(object):
    """DataGenerator, DLWP/model/generators.py:19-159, over plain arrays laid out like the predictor file
    (sample, [time_step,] varlev..., lat, lon)."""

    def __init__(self, predictors, targets, has_time_step=True, is_convolutional=True, is_recurrent=False,
                 batch_size=32, shuffle=False, remove_nan=True):
        self.P, self.T = predictors, targets
        self.has_time_step = has_time_step
        self.is_convolutional, self.is_recurrent = is_convolutional, is_recurrent
        self.batch_size, self.shuffle, self.remove_nan = batch_size, shuffle, remove_nan
        self.n_sample = predictors.shape[0]
        self.on_epoch_end()

    @property
    def shape(self):                                                    # :51-59
        return self.P.shape[1:] if self.has_time_step else (1,) + self.P.shape[1:]

    @property
    def n_features(self):                                               # :61-66
        return int(np.prod(self.shape))

    @property
    def dense_shape(self):                                              # :68-77
        if self.is_recurrent:
            return (self.shape[0], self.n_features // self.shape[0])
        return (self.n_features,)

    def _conv_shape(self, keep_time):                                   # :79-101
        if keep_time:
            return (self.shape[0], int(np.prod(self.shape[1:-2]))) + tuple(self.shape[-2:])
        return (int(np.prod(self.shape[:-2])),) + tuple(self.P.shape[-2:])

    @property
    def convolution_shape(self):
        return self._conv_shape(self.is_recurrent)

    @property
    def shape_2d(self):
        return self._conv_shape(False)

    def on_epoch_end(self):                                             # :103-106 (legacy global RandomState)
        self.indices = np.arange(self.n_sample)
        if self.shuffle:
            np.random.shuffle(self.indices)

    def __len__(self):                                                  # :141
        return int(np.ceil(self.n_sample / self.batch_size))

    def generate(self, samples):                                        # :108-135
        sel = samples if len(samples) > 0 else slice(None)
        p, t = self.P[sel], self.T[sel]
        n = p.shape[0]
        p, t = p.reshape((n, -1)), t.reshape((n, -1))
        if self.remove_nan:
            p, t = delete_nan_samples(p, t)
        # NOTE: the reference keeps the PRE-deletion n (:113 vs :129), which raises whenever a sample was dropped
        # (SURVEY.md App. C); the goldens contain no dropped samples on this path.
        if self.is_convolutional:
            p = p.reshape((n,) + self.convolution_shape)
            t = t.reshape((n,) + self.convolution_shape)
        elif self.is_recurrent:
            p = p.reshape((n,) + self.dense_shape)
            t = t.reshape((n,) + self.dense_shape)
        return p, t

    def __getitem__(self, index):                                       # :143-159
        if int(index) < 0:
            index = len(self) + index
        if index > len(self):
            raise IndexError
        return self.generate(self.indices[index * self.batch_size:(index + 1) * self.batch_size])


# ------------------------------------------------------------------------------------------------------------------ #
# custom losses  (PINNED by tests/golden/losses.npz; "next" row of SURVEY.md section 8f)
# ------------------------------------------------------------------------------------------------------------------ #

def anomaly_correlation_loss(y_true, y_pred, mean=None, regularize_mean='mse'):
    """acc_loss with reverse=True, DLWP/custom.py:1036-1088, reduced to its batch-mean scalar."""
    yt = np.asarray(y_true, np.float64)
    yp = np.asarray(y_pred, np.float64)
    if mean is not None:
        yt_a, yp_a = yt - mean, yp - mean
    else:
        yt_a, yp_a = yt, yp
    a = np.mean(yp_a * yt_a) / np.sqrt(np.mean(yp_a ** 2) * np.mean(yt_a ** 2))
    if regularize_mean is None:
        return float(-a)
    if regularize_mean == 'mse':
        m = np.mean((yp - yt) ** 2)
    elif regularize_mean == 'mae':
        m = np.mean(np.abs(yp - yt))
    elif regularize_mean == 'global':
        m = np.abs((yt.mean() - yp.mean()) / yt.mean())
    elif regularize_mean == 'spatial':
        mt, mp = yt.mean(axis=(-2, -1)), yp.mean(axis=(-2, -1))
        m = np.mean(np.abs((mt - mp) / mt))
    else:
        raise ValueError(regularize_mean)
    return float(m - a)


def latitude_weights(lats, weighting='cosine'):
    """Weights of latitude_weighted_loss, DLWP/custom.py:975-978 (the FUNCTION form, whose 'midlatitude' formula
    differs from the class form at :926)."""
    lat = np.asarray(lats, np.float64) * np.pi / 180.
    w = np.cos(lat)
    if weighting == 'midlatitude':
        w = w + 0.5 * np.sin(2 * lat) ** 2
    return w
