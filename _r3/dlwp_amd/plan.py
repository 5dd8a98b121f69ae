"""
Lowering of a layer graph to fused libdlwp_hip.so launches.

The reference executes every layer as its own TF op and materialises 3 padded copies in front of each convolution
(SURVEY.md 3.4).  Here padding, pooling, up-sampling and channel slicing are *lazy*: they only edit a `View` (which
stored buffer, which channel window, which loader transform, which halo).  A Conv2D consumes the view in one fused kernel
(dlwp_conv2d_fwd).  Only when a view reaches a consumer that cannot fuse it (a model output, a concatenate, a second
pad on the same axis with another mode, ...) is it materialised with the standalone kernels.

A plan is pure host data (no device access): it can be built and inspected on a machine without a GPU.
"""
import collections
import os

import numpy as np

from . import layers as L

PAD_ZERO, PAD_WRAP, PAD_EDGE, PAD_REFLECT, PAD_SYMMETRIC = 0, 1, 2, 3, 4
SRC_DIRECT, SRC_UPSAMPLE2, SRC_MAXPOOL2 = 0, 1, 2
ACT = {'linear': 0, None: 0, 'tanh': 1, 'relu': 2}
STATE_IN = -1


def OUT(o):
    return -2 - o


Halo = collections.namedtuple('Halo', 'top bottom left right mode_h mode_w')
NO_HALO = Halo(0, 0, 0, 0, PAD_ZERO, PAD_ZERO)


class View(object):
    """Lazy value of a symbolic NCHW tensor: pad(src_transform(buffer[:, c_off:c_off+c]))."""
    __slots__ = ('buf', 'c_off', 'c', 'c_total', 'h', 'w', 'src_mode', 'halo', 'shape')

    def __init__(self, buf, c_off, c, c_total, h, w, src_mode=SRC_DIRECT, halo=NO_HALO, shape=None):
        self.buf, self.c_off, self.c, self.c_total, self.h, self.w = buf, c_off, c, c_total, h, w
        self.src_mode, self.halo = src_mode, halo
        self.shape = shape      # logical per-sample shape when it is not (c, H, W) (after a Reshape)

    def copy(self, **kw):
        v = View(self.buf, self.c_off, self.c, self.c_total, self.h, self.w, self.src_mode, self.halo, self.shape)
        for k, val in kw.items():
            setattr(v, k, val)
        return v

    @property
    def src_hw(self):
        f = {SRC_DIRECT: lambda d: d, SRC_UPSAMPLE2: lambda d: 2 * d, SRC_MAXPOOL2: lambda d: d // 2}[self.src_mode]
        return f(self.h), f(self.w)

    @property
    def logical(self):
        h, w = self.src_hw
        return (self.c, h + self.halo.top + self.halo.bottom, w + self.halo.left + self.halo.right)

    @property
    def plain(self):
        return self.src_mode == SRC_DIRECT and self.halo == NO_HALO

    @property
    def full(self):
        return self.plain and self.c_off == 0 and self.c == self.c_total


class PlanOp(object):
    """One launch.  kind in {'conv','rowconv','pad','maxpool','upsample','copy','lstm','phasew','d2s'}."""

    def __init__(self, kind, src, dst, xs, **kw):
        self.kind, self.src, self.dst = kind, src, dst
        self.xs = tuple(xs)                 # per-sample (c, h, w) of the stored input this op reads
        self.layer = kw.pop('layer', None)  # Conv2D layer owning the weights
        self.halo = kw.pop('halo', NO_HALO)
        self.src_mode = kw.pop('src_mode', SRC_DIRECT)
        self.act = kw.pop('act', 0)
        self.in_c_off = kw.pop('in_c_off', 0)
        self.in_c_total = kw.pop('in_c_total', 0)
        self.out_c_off = kw.pop('out_c_off', 0)
        self.out_c_total = kw.pop('out_c_total', 0)
        self.inner = kw.pop('inner', 1)     # pad only: 1 = NCHW rows, C = NHWC
        self.out_shape = kw.pop('out_shape', None)   # per-sample (c, h, w) this op produces (window it writes)
        # lstm only: src = zx buffer, dst = h buffer; aux = (zh buffer | None, c_prev buffer | None, c_out buffer)
        self.aux = kw.pop('aux', None)
        self.out_pool = kw.pop('out_pool', False)   # conv only: MaxPooling2D(2) applied in the epilogue (inference plans)
        self.out_d2s = kw.pop('out_d2s', False)     # conv only: the 4 F phase channels stored interleaved (inference plans)
        self.rec_act = kw.pop('rec_act', 0)
        # conv with the ConvLSTM2D cell update in its epilogue (bfloat16 inference plans): lstm_f = F hidden channels, dst = the h
        # buffer, aux = (z_add buffer | None, c_prev buffer | None, c_out buffer), act / rec_act = the cell's activations
        self.lstm_f = kw.pop('lstm_f', 0)
        # ... and a WHOLE step in one launch (dlwp_convlstm_step_fwd; octet layout): this op is the recurrent convolution, src2
        # describes the input convolution of the same step: {'buf', 'xs' (c, h, w), 'layer', 'halo', 'in_c_off', 'in_c_total'}
        self.src2 = kw.pop('src2', None)
        # conv restated on a low-resolution source (inference plans; build_plan): geometry that overrides the layer's
        self.dil = kw.pop('dil', None)            # dilation (dh, dw)
        self.ksize = kw.pop('ksize', None)        # kernel size (kh, kw)
        self.filters = kw.pop('filters', None)    # output channels
        self.wparam = kw.pop('wparam', None)      # index into plan.phase_params: derived kernel / bias instead of the layer's
        self.alg_flops = kw.pop('alg_flops', None)   # algorithmic FLOPs per sample of the ORIGINAL layer
        assert not kw, kw

    @property
    def conv_geometry(self):
        """(filters, (kh, kw), (dh, dw)) this conv launch runs with."""
        lay = self.layer
        return (self.filters or lay.filters, tuple(self.ksize or lay.kernel_size), tuple(self.dil or lay.dilation_rate))

    def __repr__(self):
        extra = ''
        if self.kind == 'lstm':
            extra = ' aux%r h[%d:+%d/%d] act%d rec%d' % (self.aux, self.out_c_off, self.xs[0], self.out_c_total, self.act,
                                                        self.rec_act)
        if self.kind in ('conv', 'rowconv'):
            f, ks, dil = self.conv_geometry
            extra = ' %s k%s d%s src%d halo%s act%d cin[%d:+%d/%d] cout[%d:+%d/%d]%s%s' % (
                self.layer.name, ks, dil, self.src_mode, tuple(self.halo),
                self.act, self.in_c_off, self.xs[0], self.in_c_total, self.out_c_off, f,
                self.out_c_total, ' +pool' if self.out_pool else '', ' phase-kernels' if self.wparam is not None else '')
            if self.lstm_f:
                extra += ' +cell-update F=%d aux%r rec%d' % (self.lstm_f, self.aux, self.rec_act)
        return '<%s %s -> %s xs=%s%s>' % (self.kind, self.src, self.dst, self.xs, extra)


class Plan(object):
    def __init__(self):
        self.ops = []
        self.buffers = []        # per-sample (c_total, h, w) of each scratch buffer; index = buffer id
        self.input_shape = None  # per-sample
        self.output_shapes = []  # per-sample logical shapes of the model outputs
        self.output_store = []   # per-sample stored (c, h, w) of each output slot
        self.conv_layers = []    # unique Conv2D layers in first-use order
        self.phase_params = []   # derived (phase-summed) kernels: dicts layer, w2_shape, b2 (bool), pad_top, pad_left

    def new_buffer(self, c, h, w):
        self.buffers.append((int(c), int(h), int(w)))
        return len(self.buffers) - 1

    def buffer_shape(self, buf):
        if buf == STATE_IN:
            return self._in_store
        if buf <= -2:
            return self.output_store[-2 - buf]
        return self.buffers[buf]

    @property
    def n_launches(self):
        return len(self.ops)

    def conv_flops_per_sample(self):
        """sum over conv ops of 2*Ho*Wo*Cout*Cin*kh*kw."""
        tot = 0
        for op in self.ops:
            if op.kind in ('conv', 'rowconv'):
                if op.alg_flops is not None:
                    tot += op.alg_flops
                    continue
                kh, kw = op.layer.kernel_size
                co, ho, wo = getattr(op, 'conv_out_shape', None) or op.out_shape
                tot += 2 * ho * wo * co * op.xs[0] * kh * kw
                if op.src2 is not None:          # a whole ConvLSTM2D step: the input convolution's share
                    k2 = op.src2['layer'].kernel_size
                    tot += 2 * ho * wo * 4 * op.lstm_f * op.src2['xs'][0] * k2[0] * k2[1]
        return tot

    def algorithmic_bytes_per_sample(self, itemsize=4):
        """fused-forward algorithmic bytes: sum over launches of (input window + output window) + weights once."""
        tot = 0
        for op in self.ops:
            c, h, w = op.xs
            tot += c * h * w * itemsize
            co, ho, wo = op.out_shape
            tot += co * ho * wo * itemsize
        for lay in self.conv_layers:
            if lay.kernel is not None:
                tot += (int(np.prod(lay.kernel.shape)) + (int(np.prod(lay.bias.shape)) if lay.bias is not None else 0)) * itemsize
        return tot

    def bf16_buffers(self):
        """Scratch buffers that may be stored as bfloat16 (config 4: bf16 activations between the layers): written by a
        convolution (Conv2D output, ConvLSTM2D gate pre-activations), by the ConvLSTM2D cell update (the h sequence) or
        by a max-pooling of such a buffer, and read only by convolutions / max-pooling / the cell update.  Model inputs
        and outputs, the ConvLSTM2D cell state, and anything a copy / pad / up-sampling kernel touches stay float32."""
        ok = {}
        for op in self.ops:
            for b, role in ((op.src, 'r'), (op.dst, 'w')):
                if b >= 0:
                    ok[b] = ok.get(b, True) and op.kind in ('conv', 'maxpool', 'lstm') and not (role == 'w' and op.out_d2s)
            if op.kind == 'lstm' or (op.kind == 'conv' and op.lstm_f):
                zh, cp, co = op.aux
                for b in (cp, co):                                                # the cell state stays float32
                    if b is not None and b >= 0:
                        ok[b] = False
                if op.kind == 'conv' and zh is not None and zh >= 0:             # (read by this convolution's epilogue)
                    ok[zh] = ok.get(zh, True)
        for op in self.ops:                                                       # zx and zh of a step: the same type
            if op.kind == 'lstm' and op.aux[0] is not None and op.src >= 0 and op.aux[0] >= 0:
                both = ok.get(op.src, False) and ok.get(op.aux[0], False)
                ok[op.src] = ok[op.aux[0]] = both
        changed = True
        while changed:                  # a pooled copy is bf16 only if its source is (and vice versa)
            changed = False
            for op in self.ops:
                if op.kind == 'maxpool' and op.src >= 0 and op.dst >= 0 and ok.get(op.src, False) != ok.get(op.dst, False):
                    ok[op.src] = ok[op.dst] = False
                    changed = True
                if op.kind == 'maxpool' and (op.src < 0 or op.dst < 0):
                    for b in (op.src, op.dst):
                        if b >= 0 and ok.get(b, False):
                            ok[b] = False
                            changed = True
        return sorted(b for b, v in ok.items() if v)

    def describe(self):
        return '\n'.join(repr(op) for op in self.ops)


def _compose_halo(first, layer_pad, mode):
    """Halo after applying a padding layer (pads ((t,b),(l,r)), `mode`) on top of an existing lazy halo.
    Returns None when the two do not compose into one per-axis mode (caller materialises first)."""
    (t, b), (l, r) = layer_pad

    def axis(lo0, hi0, m0, lo1, hi1):
        if lo1 == 0 and hi1 == 0:
            return lo0, hi0, m0
        if lo0 == 0 and hi0 == 0:
            return lo1, hi1, mode
        if m0 == mode and mode in (PAD_ZERO, PAD_EDGE):
            return lo0 + lo1, hi0 + hi1, mode       # zero-of-zero / edge-of-edge extend; wrap-of-wrap does not
        return None
    ah = axis(first.top, first.bottom, first.mode_h, t, b)
    aw = axis(first.left, first.right, first.mode_w, l, r)
    if ah is None or aw is None:
        return None
    return Halo(ah[0], ah[1], aw[0], aw[1], ah[2], aw[2])


def _prefers_unfused_pool(cin, lay):
    """Ask the library (host logic only, works without a GPU) whether a pooled source should be materialised for this
    convolution; False when the library is not built."""
    try:
        from . import ops
        return ops.prefers_unfused_pool(cin, lay.filters, lay.kernel_size[0], lay.kernel_size[1],
                                        lay.dilation_rate[0], lay.dilation_rate[1])
    except (ImportError, OSError, AttributeError):
        return False


# Inference plans restate a Conv2D that reads a 2x up-sampled tensor on the low-resolution tensor (see build_plan).
# DLWP_RESTATE_UPSAMPLED=0 keeps the reference's own formulation (the fused up-sampling loader) for A/B comparisons.
RESTATE_UPSAMPLED = os.environ.get('DLWP_RESTATE_UPSAMPLED', '1') != '0'


def _phase_geometry(k, pad):
    """(k2, lo, hi): the distinct source offsets [lo, hi] the k taps of an axis reach on a 2x up-sampled tensor with a
    top / left halo of `pad` (csrc/phase.hip, ops.phase_geometry)."""
    offs = [(a + u - pad) // 2 for a in (0, 1) for u in range(k)]
    return max(offs) - min(offs) + 1, min(offs), max(offs)


def toposort(outputs):
    order, seen = [], set()

    def visit(t):
        if t.uid in seen:
            return
        seen.add(t.uid)
        for i in t.inputs:
            visit(i)
        order.append(t)
    for o in outputs:
        visit(o)
    return order


def _supports_out_pool(op):
    try:
        from . import ops
        f, ks, dil = op.conv_geometry
        cd = ops.make_conv(f, ks[0], ks[1], dil, ops.make_pad(*op.halo),
                           op.act, op.in_c_off, op.in_c_total, op.out_c_off, op.out_c_total, op.src_mode)
        return ops.supports_out_pool(op.xs, cd)
    except (ImportError, OSError, AttributeError):
        return False


def _supports_out_d2s(xs, f4, ks, halo, act, in_c_off, in_c_total):
    """can the restated layer's convolution store its 4 F phase channels interleaved (dlwp_conv2d.out_d2s)?"""
    try:
        from . import ops
        cd = ops.make_conv(f4, ks[0], ks[1], 1, ops.make_pad(*halo), act, in_c_off, in_c_total, 0, 0, SRC_DIRECT)
        return ops.supports_out_d2s(xs, cd)
    except (ImportError, OSError, AttributeError):
        return False


def _supports_lstm_conv(xs, part, halo, src_mode, act, rec_act, f, in_c_off, in_c_total, out_c_off, out_c_total, in_bf16):
    """can this convolution of a ConvLSTM2D step carry the cell update in its epilogue (dlwp_convlstm_conv_fwd)?"""
    try:
        from . import ops
        cd = ops.make_conv(4 * f, part.kernel_size[0], part.kernel_size[1], tuple(part.dilation_rate), ops.make_pad(*halo), act,
                           in_c_off, in_c_total, out_c_off, out_c_total, src_mode, lstm_f=f, lstm_rec_act=rec_act)
        return ops.convlstm_conv_supported(xs, cd, in_bf16, compute_bf16=not in_bf16)
    except (ImportError, OSError, AttributeError):
        return False


def _supports_lstm_step(lay, f, ho, wo, t_len, cin, v, halo, act, rec_act):
    """can ONE launch run a step t >= 1 of this ConvLSTM2D (dlwp_convlstm_step_fwd)?"""
    try:
        from . import ops
        ip, rp = lay.input_part, lay.recurrent_part
        rk = ((rp.kernel_size[0] - 1) // 2, (rp.kernel_size[1] - 1) // 2)
        cd_h = ops.make_conv(4 * f, rp.kernel_size[0], rp.kernel_size[1], 1, ops.make_pad(rk[0], rk[0], rk[1], rk[1], 0, 0), act,
                             0, t_len * f, f, t_len * f, SRC_DIRECT, lstm_f=f, lstm_rec_act=rec_act)
        cd_x = ops.make_conv(4 * f, ip.kernel_size[0], ip.kernel_size[1], tuple(ip.dilation_rate), ops.make_pad(*halo), 0,
                             v.c_off + cin, v.c_total, 0, 4 * f, v.src_mode)
        return v.h == ho and v.w == wo and ops.convlstm_step_supported((f, ho, wo), cd_h, (cin, v.h, v.w), cd_x)
    except (ImportError, OSError, AttributeError):
        return False


def build_plan(inputs, outputs, inference=False, fuse_d2s=True, fuse_lstm=False, fuse_lstm_step=True):
    """inputs: [KTensor] (exactly one), outputs: [KTensor].  Returns a Plan.  inference=True additionally moves a
    MaxPooling2D(2) that is the only consumer of a convolution into that convolution's epilogue (the pre-pooling tensor
    is never written; the training plan keeps it because the backward pass needs it)."""
    if len(inputs) != 1:
        raise NotImplementedError('exactly one model input is supported')
    plan = Plan()
    x_in = inputs[0]
    plan.input_shape = tuple(x_in.shape)
    order = toposort(outputs)
    consumers = collections.Counter()
    for t in order:
        for i in t.inputs:
            consumers[i.uid] += 1
    out_index = {}
    for o, t in enumerate(outputs):
        out_index.setdefault(t.uid, []).append(o)
    plan.output_shapes = [tuple(t.shape) for t in outputs]
    plan.output_store = [None] * len(outputs)

    def store_of(shape):
        """stored 3-D (c, h, w) block of a logical per-sample shape: everything in front of the last two axes is
        channels (the recurrent (T, C, H, W) layout is (T*C, H, W) in memory)."""
        if len(shape) < 3:
            raise NotImplementedError('dense (non-convolutional) tensors of per-sample shape %r are not on the HIP path'
                                      % (shape,))
        c = 1
        for d in shape[:-2]:
            c *= d
        return (c, shape[-2], shape[-1])

    plan._in_store = store_of(x_in.shape)
    views = {}

    producer = {}                      # scratch buffer -> the conv op that wrote all of it

    def emit(op):
        plan.ops.append(op)
        if op.kind == 'conv' and op.dst >= 0 and op.out_c_off == 0 and op.out_c_total == op.layer.filters:
            producer[op.dst] = op
        elif op.dst in producer:
            del producer[op.dst]
        if op.kind in ('conv', 'rowconv') and op.layer not in plan.conv_layers:
            plan.conv_layers.append(op.layer)
        return op

    def materialize(v, dst=None, dst_c_off=0, dst_c_total=None):
        """Turn a lazy view into real data.  With dst=None a new buffer is created (or the view is returned untouched if
        it is already a whole buffer).  With dst given, the data lands in channels [dst_c_off, +c) of that buffer."""
        cur = v
        steps = []
        if not (cur.c_off == 0 and cur.c == cur.c_total) and not cur.plain:
            steps.append('window')      # pool/pad kernels want a dense (n*c) plane run: extract the window first
        if cur.src_mode != SRC_DIRECT:
            steps.append('src')
        if cur.halo != NO_HALO:
            steps.append('halo')
        into_window = dst is not None and not (dst_c_off == 0 and (dst_c_total is None or dst_c_total == v.logical[0]))
        if dst is not None and (not steps or into_window or steps[-1] == 'window'):
            steps.append('final_copy')
        if not steps:
            return cur
        for k, step in enumerate(steps):
            last = k == len(steps) - 1
            tgt_direct = last and dst is not None and step != 'final_copy'
            if step == 'window':
                nb = plan.new_buffer(cur.c, cur.h, cur.w)
                emit(PlanOp('copy', cur.buf, nb, (cur.c, cur.h, cur.w), in_c_off=cur.c_off, in_c_total=cur.c_total,
                            out_c_off=0, out_c_total=cur.c, out_shape=(cur.c, cur.h, cur.w)))
                cur = cur.copy(buf=nb, c_off=0, c_total=cur.c)
            elif step == 'src':
                h2, w2 = cur.src_hw
                nb = dst if tgt_direct else plan.new_buffer(cur.c, h2, w2)
                emit(PlanOp('maxpool' if cur.src_mode == SRC_MAXPOOL2 else 'upsample', cur.buf, nb,
                            (cur.c, cur.h, cur.w), out_shape=(cur.c, h2, w2)))
                cur = cur.copy(buf=nb, h=h2, w=w2, src_mode=SRC_DIRECT)
            elif step == 'halo':
                c, hp, wp = cur.logical
                nb = dst if tgt_direct else plan.new_buffer(c, hp, wp)
                emit(PlanOp('pad', cur.buf, nb, (cur.c, cur.h, cur.w), halo=cur.halo, out_shape=(c, hp, wp)))
                cur = cur.copy(buf=nb, h=hp, w=wp, halo=NO_HALO)
            else:  # final_copy
                tot = dst_c_total if dst_c_total is not None else cur.c
                emit(PlanOp('copy', cur.buf, dst, (cur.c, cur.h, cur.w), in_c_off=cur.c_off, in_c_total=cur.c_total,
                            out_c_off=dst_c_off, out_c_total=tot, out_shape=(cur.c, cur.h, cur.w)))
                cur = View(dst, dst_c_off, cur.c, tot, cur.h, cur.w)
        return cur

    for t in order:
        lay = t.layer
        if isinstance(lay, L.InputLayer):
            if t.uid != x_in.uid:
                raise ValueError('graph reaches an Input that is not the model input')
            c, h, w = plan._in_store
            views[t.uid] = View(STATE_IN, 0, c, c, h, w, shape=tuple(t.shape) if len(t.shape) != 3 else None)
            continue
        ins = [views[i.uid] for i in t.inputs]
        outs = out_index.get(t.uid, [])

        if isinstance(lay, L._Pad2DBase):
            v = ins[0]
            if v.shape is not None:
                raise NotImplementedError('%s on a reshaped (non-3D) tensor' % lay.name)
            if lay.data_format == 'channels_last':
                # standalone only: buffer holds (H, W, C) per sample, rows of W*C floats
                v = materialize(v)
                hh, ww, cc = t.inputs[0].shape
                (tp, bt), (lf, rt) = lay.padding
                nb = plan.new_buffer(hh + tp + bt, ww + lf + rt, cc)
                emit(PlanOp('pad', v.buf, nb, (1, hh, ww), halo=Halo(tp, bt, lf, rt, lay.mode, lay.mode), inner=cc,
                            out_shape=(hh + tp + bt, ww + lf + rt, cc)))
                views[t.uid] = View(nb, 0, hh + tp + bt, hh + tp + bt, ww + lf + rt, cc)
                views[t.uid].shape = tuple(t.shape)
            else:
                halo = _compose_halo(v.halo, lay.padding, lay.mode)
                if halo is None:
                    v = materialize(v)
                    halo = _compose_halo(NO_HALO, lay.padding, lay.mode)
                hh, ww = v.src_hw
                if halo.mode_h == PAD_WRAP and max(halo.top, halo.bottom) > hh:
                    raise ValueError('%s: periodic row padding exceeds the input height %d' % (lay.name, hh))
                if halo.mode_w == PAD_WRAP and max(halo.left, halo.right) > ww:
                    raise ValueError('%s: periodic column padding exceeds the input width %d' % (lay.name, ww))
                views[t.uid] = v.copy(halo=halo)
        elif isinstance(lay, L._Pad3DBase):
            # (T, C, H, W) stored as (T*C, H, W): a 3-D pad that leaves the first (channel) axis alone is the 2-D halo
            v = ins[0]
            if lay.data_format != 'channels_first':
                raise NotImplementedError("%s: data_format='channels_first' is required" % lay.name)
            if lay.padding[0] != (0, 0):
                raise NotImplementedError('%s: padding of the first (channel) axis %r is not lowered to the HIP path'
                                          % (lay.name, lay.padding[0]))
            pad2 = (lay.padding[1], lay.padding[2])
            halo = _compose_halo(v.halo, pad2, lay.mode)
            if halo is None:
                shp = v.shape
                v = materialize(v)
                v.shape = shp
                halo = _compose_halo(NO_HALO, pad2, lay.mode)
            hh, ww = v.src_hw
            if halo.mode_h == PAD_WRAP and max(halo.top, halo.bottom) > hh:
                raise ValueError('%s: periodic row padding exceeds the input height %d' % (lay.name, hh))
            if halo.mode_w == PAD_WRAP and max(halo.left, halo.right) > ww:
                raise ValueError('%s: periodic column padding exceeds the input width %d' % (lay.name, ww))
            views[t.uid] = v.copy(halo=halo, shape=tuple(t.shape))
        elif isinstance(lay, L.ConvLSTM2D):
            v = ins[0]
            t_len, cin = t.inputs[0].shape[0], t.inputs[0].shape[1]
            if v.c != t_len * cin:
                raise ValueError('%s: input view has %d channels, expected T*C = %d' % (lay.name, v.c, t_len * cin))
            halo = v.halo
            if lay.padding == 'same':
                st, sb, sl, sr = lay.same_halo()
                halo2 = _compose_halo(halo, ((st, sb), (sl, sr)), PAD_ZERO)
                if halo2 is None:
                    v = materialize(v)
                    halo2 = Halo(st, sb, sl, sr, PAD_ZERO, PAD_ZERO)
                halo = halo2
            _, hl, wl = v.copy(halo=halo).logical
            ho = hl - lay.dilation_rate[0] * (lay.kernel_size[0] - 1)
            wo = wl - lay.dilation_rate[1] * (lay.kernel_size[1] - 1)
            f = lay.filters
            hbuf = plan.new_buffer(t_len * f, ho, wo)           # h_0 .. h_{T-1}, the return_sequences output
            # every step keeps its own pre-activations and cell state: they are the saved activations of the backward
            # pass (dlwp_convlstm_gates_bwd) -- T is the reference's time_dim (2), so this costs little
            cbufs = [plan.new_buffer(f, ho, wo) for _ in range(t_len)]
            rk = ((lay.kernel_size[0] - 1) // 2, (lay.kernel_size[1] - 1) // 2)
            rec_code = {'hard_sigmoid': 0, 'sigmoid': 1}[lay.recurrent_activation]
            rhalo = Halo(rk[0], rk[0], rk[1], rk[1], PAD_ZERO, PAD_ZERO)
            # bfloat16 inference: the convolution that completes a step's pre-activations applies the cell update in its
            # epilogue (the input convolution on the first step, the recurrent one afterwards): z_h (z_x on the first step)
            # is never stored and the gate kernel disappears.  The model input feeds the input convolution as float32.
            fused = (inference and fuse_lstm and v.buf == STATE_IN and
                     _supports_lstm_conv((cin, v.h, v.w), lay.input_part, halo, v.src_mode, ACT[lay.activation], rec_code, f,
                                         v.c_off, v.c_total, 0, t_len * f, False) and
                     (t_len == 1 or _supports_lstm_conv((f, ho, wo), lay.recurrent_part, rhalo, SRC_DIRECT, ACT[lay.activation],
                                                        rec_code, f, 0, t_len * f, f, t_len * f, True)))
            # ... and every later step as ONE launch where the library has the dual-source instance (octet layout; the executor
            # falls back to the two launches when it cannot keep the h sequence in octets)
            whole = (fused and fuse_lstm_step and t_len > 1 and os.environ.get('DLWP_LSTM_STEP', '1') != '0' and
                     _supports_lstm_step(lay, f, ho, wo, t_len, cin, v, halo, ACT[lay.activation], rec_code))
            zxs = [None if (fused and (step == 0 or whole)) else plan.new_buffer(4 * f, ho, wo) for step in range(t_len)]
            zhs = [None] + [None if fused else plan.new_buffer(4 * f, ho, wo) for _ in range(t_len - 1)]
            for step in range(t_len if fused else 0):
                if step == 0:
                    emit(PlanOp('conv', v.buf, hbuf, (cin, v.h, v.w), layer=lay.input_part, halo=halo, src_mode=v.src_mode,
                                act=ACT[lay.activation], rec_act=rec_code, lstm_f=f, aux=(None, None, cbufs[0]),
                                in_c_off=v.c_off, in_c_total=v.c_total, out_c_off=0, out_c_total=t_len * f,
                                out_shape=(f, ho, wo)))
                    continue
                if whole:
                    emit(PlanOp('conv', hbuf, hbuf, (f, ho, wo), layer=lay.recurrent_part, halo=rhalo, src_mode=SRC_DIRECT,
                                act=ACT[lay.activation], rec_act=rec_code, lstm_f=f, aux=(None, cbufs[step - 1], cbufs[step]),
                                in_c_off=(step - 1) * f, in_c_total=t_len * f, out_c_off=step * f, out_c_total=t_len * f,
                                out_shape=(f, ho, wo),
                                src2={'buf': v.buf, 'xs': (cin, v.h, v.w), 'layer': lay.input_part, 'halo': halo,
                                      'in_c_off': v.c_off + step * cin, 'in_c_total': v.c_total}))
                    continue
                emit(PlanOp('conv', v.buf, zxs[step], (cin, v.h, v.w), layer=lay.input_part, halo=halo,
                            src_mode=v.src_mode, act=0, in_c_off=v.c_off + step * cin, in_c_total=v.c_total, out_c_off=0,
                            out_c_total=4 * f, out_shape=(4 * f, ho, wo)))
                emit(PlanOp('conv', hbuf, hbuf, (f, ho, wo), layer=lay.recurrent_part, halo=rhalo, src_mode=SRC_DIRECT,
                            act=ACT[lay.activation], rec_act=rec_code, lstm_f=f, aux=(zxs[step], cbufs[step - 1], cbufs[step]),
                            in_c_off=(step - 1) * f, in_c_total=t_len * f, out_c_off=step * f, out_c_total=t_len * f,
                            out_shape=(f, ho, wo)))
            for step in range(0 if fused else t_len):
                emit(PlanOp('conv', v.buf, zxs[step], (cin, v.h, v.w), layer=lay.input_part, halo=halo,
                            src_mode=v.src_mode, act=0, in_c_off=v.c_off + step * cin, in_c_total=v.c_total, out_c_off=0,
                            out_c_total=4 * f, out_shape=(4 * f, ho, wo)))
                if step > 0:
                    emit(PlanOp('conv', hbuf, zhs[step], (f, ho, wo), layer=lay.recurrent_part,
                                halo=Halo(rk[0], rk[0], rk[1], rk[1], PAD_ZERO, PAD_ZERO), src_mode=SRC_DIRECT, act=0,
                                in_c_off=(step - 1) * f, in_c_total=t_len * f, out_c_off=0, out_c_total=4 * f,
                                out_shape=(4 * f, ho, wo)))
                emit(PlanOp('lstm', zxs[step], hbuf, (f, ho, wo), act=ACT[lay.activation],
                            rec_act={'hard_sigmoid': 0, 'sigmoid': 1}[lay.recurrent_activation],
                            aux=(zhs[step], cbufs[step - 1] if step > 0 else None, cbufs[step]),
                            out_c_off=step * f, out_c_total=t_len * f, out_shape=(f, ho, wo)))
            if lay.return_sequences:
                views[t.uid] = View(hbuf, 0, t_len * f, t_len * f, ho, wo, shape=(t_len, f, ho, wo))
            else:
                views[t.uid] = View(hbuf, (t_len - 1) * f, f, t_len * f, ho, wo)
        elif isinstance(lay, (L.MaxPooling2D, L.UpSampling2D)):
            v = ins[0]
            if v.shape is not None:
                raise NotImplementedError('%s on a reshaped tensor' % lay.name)
            if not v.plain:
                v = materialize(v)
            prod = producer.get(v.buf) if (inference and isinstance(lay, L.MaxPooling2D) and v.full and not outs) else None
            if (prod is not None and consumers[t.inputs[0].uid] == 1 and not isinstance(prod.layer, L._ConvPart)
                    and not prod.out_pool and _supports_out_pool(prod)):
                # the pooling runs in the producing convolution's epilogue: its buffer now holds the pooled tensor
                prod.out_pool = True
                c, h2, w2 = v.c, v.h // 2, v.w // 2
                prod.conv_out_shape = prod.out_shape        # what the convolution computes (FLOP accounting)
                prod.out_shape = (c, h2, w2)                # what it stores
                plan.buffers[v.buf] = (c, h2, w2)
                views[t.uid] = View(v.buf, 0, c, c, h2, w2)
            else:
                views[t.uid] = v.copy(src_mode=SRC_MAXPOOL2 if isinstance(lay, L.MaxPooling2D) else SRC_UPSAMPLE2)
        elif isinstance(lay, L.ChannelSlice):
            v = ins[0]
            if v.shape is not None:
                raise NotImplementedError('slice_layer on a reshaped tensor')
            lo, hi = lay.window(v.c)
            views[t.uid] = v.copy(c_off=v.c_off + lo, c=hi - lo)
        elif isinstance(lay, L.Reshape):
            v = materialize(ins[0])
            if not v.full:
                v = materialize(v, dst=plan.new_buffer(*v.logical))
            c, h, w = store_of(t.shape)
            if (c * h * w) != v.c * v.h * v.w:
                raise ValueError('Reshape size mismatch')
            # relabel the same contiguous block
            if v.buf >= 0:
                plan.buffers[v.buf] = (c, h, w)
                views[t.uid] = View(v.buf, 0, c, c, h, w, shape=tuple(t.shape) if len(t.shape) != 3 else None)
            else:
                nb = plan.new_buffer(v.c, v.h, v.w)
                emit(PlanOp('copy', v.buf, nb, (v.c, v.h, v.w), in_c_total=v.c, out_c_total=v.c,
                            out_shape=(v.c, v.h, v.w)))
                plan.buffers[nb] = (c, h, w)
                views[t.uid] = View(nb, 0, c, c, h, w, shape=tuple(t.shape) if len(t.shape) != 3 else None)
        elif isinstance(lay, L.Concatenate):
            c_tot = sum(v.logical[0] for v in ins)
            _, hh, ww = ins[0].logical
            if outs:
                o = outs[0]
                dst = OUT(o)
                plan.output_store[o] = (c_tot, hh, ww)
            else:
                dst = plan.new_buffer(c_tot, hh, ww)
            off = 0
            for v in ins:
                materialize(v, dst=dst, dst_c_off=off, dst_c_total=c_tot)
                off += v.logical[0]
            views[t.uid] = View(dst, 0, c_tot, c_tot, hh, ww)
        elif isinstance(lay, L.Conv2D):
            v = ins[0]
            if v.shape is not None and len(v.shape) != 3:
                raise NotImplementedError('%s on a non-3D tensor %r' % (lay.name, v.shape))
            if v.src_mode == SRC_MAXPOOL2 and _prefers_unfused_pool(v.c, lay):
                # the Winograd kernels read plain tensors: pool once with the standalone kernel, keep the halo lazy
                v = materialize(v.copy(halo=NO_HALO)).copy(halo=v.halo)
            halo = v.halo
            if lay.padding == 'same':
                st, sb, sl, sr = lay.same_halo()
                halo2 = _compose_halo(halo, ((st, sb), (sl, sr)), PAD_ZERO)
                if halo2 is None:
                    v = materialize(v)
                    halo2 = Halo(st, sb, sl, sr, PAD_ZERO, PAD_ZERO)
                halo = halo2
            _, hl, wl = v.copy(halo=halo).logical
            ho = hl - lay.dilation_rate[0] * (lay.kernel_size[0] - 1)
            wo = wl - lay.dilation_rate[1] * (lay.kernel_size[1] - 1)
            kh, kw = lay.kernel_size
            alg = 2 * ho * wo * lay.filters * v.c * kh * kw
            # ---- inference: a convolution on a 2x nearest-neighbour up-sampled tensor, restated on the tensor itself.
            # (a) dilation 2, even halo: tap u of output row 2i + a reads source row i + u - top/2 whatever a is, so the
            #     result is UpSampling2D(conv with dilation 1 and half the halo on the low-resolution tensor): a quarter
            #     of the multiplies, and the up-sampling stays lazy for the consumer.
            # both serve the training plan as well: (a) is a plain graph identity (forward, data and weight gradient of
            # the layer then run on the low-resolution tensors); (b) trains through the adjoints of its two linear maps
            # (dlwp_phase_weights_bwd, dlwp_space_to_depth2; training.py)
            # (a mirror halo WITHOUT the border element -- TFPadding2D 'REFLECT' -- does not commute with the up-sampling:
            #  up-sampled coordinate 2n reflects to source n - 1, low-resolution coordinate n to n - 2; the layer then keeps
            #  the reference's formulation.  'SYMMETRIC', periodic, zero and edge halos commute.)
            restate = (RESTATE_UPSAMPLED and v.src_mode == SRC_UPSAMPLE2 and
                       halo.mode_h != PAD_REFLECT and halo.mode_w != PAD_REFLECT)
            if (restate and tuple(lay.dilation_rate) == (2, 2) and
                    all(p % 2 == 0 for p in halo[:4]) and ho % 2 == 0 and wo % 2 == 0):
                h2 = Halo(halo.top // 2, halo.bottom // 2, halo.left // 2, halo.right // 2, halo.mode_h, halo.mode_w)
                dst = plan.new_buffer(lay.filters, ho // 2, wo // 2)
                emit(PlanOp('conv', v.buf, dst, (v.c, v.h, v.w), layer=lay, halo=h2, src_mode=SRC_DIRECT,
                            act=ACT[lay.activation], in_c_off=v.c_off, in_c_total=v.c_total, out_c_off=0,
                            out_c_total=lay.filters, out_shape=(lay.filters, ho // 2, wo // 2), dil=(1, 1), alg_flops=alg))
                views[t.uid] = View(dst, 0, lay.filters, lay.filters, ho // 2, wo // 2, src_mode=SRC_UPSAMPLE2)
            # (b) dilation 1: the k taps of an axis fall on k2 < k distinct source pixels; each of the 4 output phases is
            #     a k2 x k2 kernel of summed weights over the SAME window, so the layer runs as one convolution with
            #     4 x filters channels on the low-resolution tensor + a depth-to-space interleave (csrc/phase.hip).
            elif (restate and tuple(lay.dilation_rate) == (1, 1) and
                  ho == 2 * v.h and wo == 2 * v.w and
                  _phase_geometry(kh, halo.top)[0] * _phase_geometry(kw, halo.left)[0] < kh * kw):
                (kh2, lo_h, hi_h), (kw2, lo_w, hi_w) = _phase_geometry(kh, halo.top), _phase_geometry(kw, halo.left)
                h2 = Halo(-lo_h, hi_h, -lo_w, hi_w, halo.mode_h, halo.mode_w)
                pidx = len(plan.phase_params)
                plan.phase_params.append({'layer': lay, 'w2_shape': (kh2, kw2, v.c, 4 * lay.filters),
                                          'bias': lay.use_bias, 'pad_top': halo.top, 'pad_left': halo.left})
                emit(PlanOp('phasew', STATE_IN, STATE_IN, (v.c, 0, 0), layer=lay, wparam=pidx, halo=halo,
                            out_shape=(0, 0, 0)))
                fold = inference and fuse_d2s and _supports_out_d2s((v.c, v.h, v.w), 4 * lay.filters, (kh2, kw2), h2,
                                                           ACT[lay.activation], v.c_off, v.c_total)
                if outs:
                    dst = OUT(outs[0])
                    plan.output_store[outs[0]] = (lay.filters, ho, wo)
                else:
                    dst = plan.new_buffer(lay.filters, ho, wo)
                if fold:         # the convolution's epilogue stores the phases interleaved: no depth-to-space pass
                    op = PlanOp('conv', v.buf, dst, (v.c, v.h, v.w), layer=lay, halo=h2, src_mode=SRC_DIRECT,
                                act=ACT[lay.activation], in_c_off=v.c_off, in_c_total=v.c_total, out_c_off=0,
                                out_c_total=lay.filters, out_shape=(lay.filters, ho, wo), dil=(1, 1),
                                ksize=(kh2, kw2), filters=4 * lay.filters, wparam=pidx, alg_flops=alg, out_d2s=True)
                    op.conv_out_shape = (4 * lay.filters, v.h, v.w)
                    emit(op)
                else:
                    tmp = plan.new_buffer(4 * lay.filters, v.h, v.w)
                    emit(PlanOp('conv', v.buf, tmp, (v.c, v.h, v.w), layer=lay, halo=h2, src_mode=SRC_DIRECT,
                                act=ACT[lay.activation], in_c_off=v.c_off, in_c_total=v.c_total, out_c_off=0,
                                out_c_total=4 * lay.filters, out_shape=(4 * lay.filters, v.h, v.w), dil=(1, 1),
                                ksize=(kh2, kw2), filters=4 * lay.filters, wparam=pidx, alg_flops=alg))
                    emit(PlanOp('d2s', tmp, dst, (lay.filters, v.h, v.w), out_c_off=0, out_c_total=lay.filters,
                                out_shape=(lay.filters, ho, wo)))
                views[t.uid] = View(dst, 0, lay.filters, lay.filters, ho, wo)
            else:
              if outs:
                dst = OUT(outs[0])
                plan.output_store[outs[0]] = (lay.filters, ho, wo)
              else:
                dst = plan.new_buffer(lay.filters, ho, wo)
              emit(PlanOp('conv', v.buf, dst, (v.c, v.h, v.w), layer=lay, halo=halo, src_mode=v.src_mode,
                        act=ACT[lay.activation], in_c_off=v.c_off, in_c_total=v.c_total, out_c_off=0,
                        out_c_total=lay.filters, out_shape=(lay.filters, ho, wo)))
              views[t.uid] = View(dst, 0, lay.filters, lay.filters, ho, wo)
        elif isinstance(lay, L.RowConnected2D):
            # DLWP.custom.RowConnected2D (reference custom.py:695-837): per-row filters, dlwp_rowconv2d_fwd.  Its kernels read a
            # stored float32 tensor, directly or 2x up-sampled (a lazy pooling in front is materialised); the halo stays in the
            # loader.
            v = ins[0]
            if v.shape is not None and len(v.shape) != 3:
                raise NotImplementedError('%s on a non-3D tensor %r' % (lay.name, v.shape))
            if v.src_mode == SRC_MAXPOOL2:           # (a 2x up-sampling in front is resolved by the row kernels' loaders)
                v = materialize(v.copy(halo=NO_HALO)).copy(halo=v.halo)
            _, hl, wl = v.logical
            kh, kw = lay.kernel_size
            ho, wo = hl - kh + 1, wl - kw + 1
            if tuple(lay.kernel.shape) != (ho, kh, kw, v.c, lay.filters):
                raise ValueError('%s: kernel %r does not fit the input (%d rows of output, %d channels)' %
                                 (lay.name, tuple(lay.kernel.shape), ho, v.c))
            if outs:
                dst = OUT(outs[0])
                plan.output_store[outs[0]] = (lay.filters, ho, wo)
            else:
                dst = plan.new_buffer(lay.filters, ho, wo)
            emit(PlanOp('rowconv', v.buf, dst, (v.c, v.h, v.w), layer=lay, halo=v.halo, src_mode=v.src_mode,
                        act=ACT[lay.activation], in_c_off=v.c_off, in_c_total=v.c_total, out_c_off=0,
                        out_c_total=lay.filters, out_shape=(lay.filters, ho, wo)))
            views[t.uid] = View(dst, 0, lay.filters, lay.filters, ho, wo)
        else:
            raise NotImplementedError('layer %s (%s) has no HIP lowering' % (lay.name, type(lay).__name__))

        # a model output that was not produced in place by a conv / concatenate: materialise it into its slot
        for k, o in enumerate(outs):
            v = views[t.uid]
            if v.buf == OUT(o) and v.full:
                continue
            lc = v.logical
            plan.output_store[o] = lc
            # The value is the whole of a scratch buffer that exactly one launch writes and nothing reads (a convolution -- or its
            # depth-to-space pass -- behind a Reshape, which is a view: examples/train.py's recurrent stack ends that way): let
            # that launch write the output slot itself instead of copying the buffer (9 us per forward of config 4 at 8 members).
            if k == 0 and v.full and v.buf >= 0:
                def reads(op):
                    extra = [b for b in (op.aux or ()) if isinstance(b, int)] if op.kind in ('conv', 'lstm') else []
                    return op.src == v.buf or v.buf in extra
                writers = [op for op in plan.ops if op.dst == v.buf]
                w0 = writers[0] if len(writers) == 1 else None
                if (w0 is not None and w0.kind in ('conv', 'd2s', 'rowconv') and not w0.lstm_f and not w0.out_pool and
                        w0.out_c_off == 0 and w0.out_c_total in (0, lc[0]) and not any(reads(op) for op in plan.ops)):
                    w0.dst = OUT(o)
                    views[t.uid] = View(OUT(o), 0, lc[0], lc[0], lc[1], lc[2], shape=v.shape)
                    continue
            materialize(v, dst=OUT(o), dst_c_off=0, dst_c_total=lc[0])
            if k == 0:
                views[t.uid] = View(OUT(o), 0, lc[0], lc[0], lc[1], lc[2], shape=v.shape)
    for o, st in enumerate(plan.output_store):
        if st is None:
            raise RuntimeError('output %d was never produced' % o)
    return plan
