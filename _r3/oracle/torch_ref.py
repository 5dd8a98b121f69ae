"""
torch-CPU float32 restatement of the reference graph, laid out UNFUSED exactly as the reference builds it
(wrap-pad copy -> zero-pad copy -> conv -> bias -> tanh -> pool / upsample; DLWP/custom.py:202-204,
examples/train.py:142-221) plus the reference-style host rollout loop with a full state copy per step
(DLWP/model/models.py:277-293).  TEST INFRASTRUCTURE ONLY (oracle/__init__.py): it is the second opinion for the
float64 numpy restatement, the autograd source for gradient checks, and the thing bench.py times as
`cpu_baseline` ("port": CPU restatement, torch-CPU/oneDNN -- never a Keras measurement).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import np_ref


def _pad_layer(x, name, padding):
    (t, b), (l, r) = np_ref.normalize_padding(padding, 2)
    if name == 'PeriodicPadding2D':
        W = x.shape[-1]
        x = torch.cat([x[..., W - l:], x, x[..., :r]], dim=-1)             # copy 1 (custom.py:202)
        H = x.shape[-2]
        return torch.cat([x[..., H - t:, :], x, x[..., :b, :]], dim=-2)    # copy 2 (custom.py:204)
    if name == 'ZeroPadding2D':
        return F.pad(x, (l, r, t, b))
    if name == 'FillPadding2D':
        if t or b:
            x = torch.cat([x[..., :1, :]] * t + [x] + [x[..., -1:, :]] * b, dim=-2)
        if l or r:
            x = torch.cat([x[..., :1]] * l + [x] + [x[..., -1:]] * r, dim=-1)
        return x
    raise ValueError(name)


def to_torch_weights(weights, dtype=torch.float32, requires_grad=False):
    """[(w_hwio, b)] numpy -> [(w_oihw, b)] torch."""
    out = []
    for item in weights:
        if len(item) == 3:      # ConvLSTM2D: (kernel, recurrent_kernel, bias)
            k, r, b = item
            out.append(tuple(torch.tensor(np.ascontiguousarray(np.transpose(a, (3, 2, 0, 1))), dtype=dtype,
                                          requires_grad=requires_grad) for a in (k, r)) +
                       (torch.tensor(np.asarray(b), dtype=dtype, requires_grad=requires_grad),))
            continue
        w, b = item
        if np.ndim(w) == 5:     # RowConnected2D: (rows, kh, kw, cin, cout) and the stored (rows, 1, cout) bias, kept as stored
            out.append((torch.tensor(np.asarray(w), dtype=dtype, requires_grad=requires_grad),
                        torch.tensor(np.asarray(b), dtype=dtype, requires_grad=requires_grad)))
            continue
        wt = torch.tensor(np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1))), dtype=dtype, requires_grad=requires_grad)
        bt = torch.tensor(np.asarray(b), dtype=dtype, requires_grad=requires_grad)
        out.append((wt, bt))
    return out


def run_layers(layers, x, tweights, record=None):
    wi = 0
    for name, args, kwargs in layers:
        args, kwargs = args or (), kwargs or {}
        if name in ('PeriodicPadding2D', 'ZeroPadding2D', 'FillPadding2D'):
            x = _pad_layer(x, name, args[0] if args else kwargs.get('padding', (1, 1)))
        elif name == 'TFPadding2D':
            (t, b), (l, r) = np_ref.normalize_padding(args[0] if args else kwargs.get('padding', (1, 1)), 2)
            mode = kwargs.get('mode', 'CONSTANT').upper()
            if mode == 'CONSTANT':
                x = F.pad(x, (l, r, t, b))
            elif mode == 'REFLECT':
                x = F.pad(x, (l, r, t, b), mode='reflect')
            else:       # SYMMETRIC: mirror with the border element = flipped border strips
                x = torch.cat([x[..., :l].flip(-1), x, x[..., x.shape[-1] - r:].flip(-1)], dim=-1)
                x = torch.cat([x[..., :t, :].flip(-2), x, x[..., x.shape[-2] - b:, :].flip(-2)], dim=-2)
        elif name == 'Conv2D':
            _, _, dil, act = np_ref._conv_args(args, kwargs)
            w, b = tweights[wi]
            wi += 1
            x = F.conv2d(x, w, None, stride=1, padding=0, dilation=dil)
            x = x + b.view(1, -1, 1, 1)
            if act == 'tanh':
                x = torch.tanh(x)
            elif act == 'relu':
                x = torch.relu(x)
        elif name == 'RowConnected2D':      # reference DLWP/custom.py:825-896: one convolution per output row, concatenated
            _, _, _, act = np_ref._conv_args(args, kwargs)
            w, b = tweights[wi]
            wi += 1
            rows, kh = w.shape[0], w.shape[1]
            x = torch.cat([F.conv2d(x[:, :, r:r + kh, :], w[r].permute(3, 2, 0, 1)) for r in range(rows)], dim=2)
            x = x + b.reshape(-1).reshape(w.shape[4], rows)[None, :, :, None]       # K.bias_add's reshape (np_ref)
            if act == 'tanh':
                x = torch.tanh(x)
            elif act == 'relu':
                x = torch.relu(x)
        elif name in ('PeriodicPadding3D', 'ZeroPadding3D'):
            # 3-D pads of the recurrent front end act on (N, T, C, H, W): fold T into the batch-side axes via numpy
            fn = np_ref.periodic_padding3d if name == 'PeriodicPadding3D' else np_ref.zero_padding3d
            x = torch.from_numpy(np.ascontiguousarray(fn(x.detach().numpy(), args[0] if args else (1, 1, 1),
                                                         kwargs.get('data_format', 'channels_first'))))
        elif name == 'ConvLSTM2D':
            _, _, dil, act = np_ref._conv_args(args, kwargs)
            k, r, b = tweights[wi]
            wi += 1
            x = conv_lstm2d(x, k, r, b, dil, act or 'tanh', kwargs.get('return_sequences', False))
        elif name == 'MaxPooling2D':
            x = F.max_pool2d(x, 2)
        elif name == 'UpSampling2D':
            x = F.interpolate(x, scale_factor=2, mode='nearest')
        elif name == 'Reshape':
            x = x.reshape((x.shape[0],) + tuple(args[0]))
        else:
            raise ValueError('layer %r not restated' % name)
        if record is not None:
            record.append((name, x))
    return x


def conv_lstm2d(x, k_oihw, r_oihw, b, dilation=1, activation='tanh', return_sequences=True):
    """Independent restatement of Keras' ConvLSTM2DCell on torch-CPU convolutions ('valid' input conv, 'same'
    recurrent conv, hard_sigmoid gates); weights OIHW as to_torch_weights produces them."""
    act = torch.tanh if activation == 'tanh' else (lambda v: v)
    hs = lambda v: torch.clamp(0.2 * v + 0.5, 0.0, 1.0)  # noqa: E731
    f = k_oihw.shape[0] // 4
    h = c = None
    outs = []
    for t in range(x.shape[1]):
        z = F.conv2d(x[:, t], k_oihw, b, dilation=dilation)
        if h is not None:
            z = z + F.conv2d(h, r_oihw, None, padding=(r_oihw.shape[2] // 2, r_oihw.shape[3] // 2))
        i, fg, g, o = hs(z[:, :f]), hs(z[:, f:2 * f]), act(z[:, 2 * f:3 * f]), hs(z[:, 3 * f:])
        c = i * g if c is None else fg * c + i * g
        h = o * act(c)
        outs.append(h)
    return torch.stack(outs, dim=1) if return_sequences else h


def rollout_host_loop(layers, tweights, state, forwards):
    """Reference-style rollout: numpy state in, model forward, numpy state out, two host copies per step."""
    p = np.array(state, copy=True)
    series = np.full((forwards,) + p.shape, np.nan, dtype=np.float32)
    with torch.no_grad():
        for t in range(forwards):
            out = run_layers(layers, torch.from_numpy(p), tweights).numpy()
            p = 1. * out
            series[t] = 1. * p
    return series


def adam_keras_step(p, m, v, g, iteration, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, decay=0.):
    """float32 torch form of np_ref.adam_keras_step (in place)."""
    t = iteration + 1
    lr_ = lr / (1. + decay * iteration)
    lr_t = lr_ * (1. - beta_2 ** t) ** 0.5 / (1. - beta_1 ** t)
    m.mul_(beta_1).add_(g, alpha=1. - beta_1)
    v.mul_(beta_2).addcmul_(g, g, value=1. - beta_2)
    p.sub_(lr_t * m / (v.sqrt() + epsilon))
