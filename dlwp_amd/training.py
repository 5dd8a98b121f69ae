"""
The train step behind model.fit / fit_generator / evaluate: forward (activations stay in the plan's buffers), 'mse'
loss + 'mae' metric, backward through the fused plan (dlwp_conv2d_bwd_data / _bwd_weight, activation / pooling /
up-sampling / halo adjoints), Keras-form Adam on ONE flat parameter buffer, and -- under torch.distributed -- one
all-reduce of ONE flat gradient buffer per step (RCCL over xGMI on the GPU box, gloo in the CPU tests).

Reference: keras Model.fit / fit_generator / evaluate as driven by DLWP/model/models.py:188-228, 303-316 and
examples/train.py:240,258-263,274; loss / metric / optimizer strings of examples/train.py:240.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import layers as L
from . import plan as P
from .custom import Callback, History


# ------------------------------------------------------------------------------------------------------------------ #
# optimizers (hyper-parameter holders; the update itself is a HIP kernel over the flat buffers)
# ------------------------------------------------------------------------------------------------------------------ #

class Optimizer(object):
    def __init__(self, lr, decay):
        self.lr, self.decay = float(lr), float(decay)
        self.iterations = 0


class Adam(Optimizer):
    """keras.optimizers.Adam (Keras 2.2 form: epsilon outside the sqrt, bias correction folded into lr_t)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False, **kwargs):
        super(Adam, self).__init__(lr, decay)
        if amsgrad:
            raise NotImplementedError('amsgrad is not implemented')
        self.beta_1, self.beta_2 = float(beta_1), float(beta_2)
        self.epsilon = 1e-7 if epsilon is None else float(epsilon)      # K.epsilon()


class SGD(Optimizer):
    def __init__(self, lr=0.01, momentum=0., decay=0., nesterov=False, **kwargs):
        super(SGD, self).__init__(lr, decay)
        if nesterov:
            raise NotImplementedError('nesterov momentum is not implemented')
        self.momentum = float(momentum)


def get_optimizer(spec):
    if isinstance(spec, Optimizer):
        return spec
    if isinstance(spec, str):
        table = {'adam': Adam, 'sgd': SGD}
        if spec.lower() not in table:
            raise NotImplementedError("optimizer %r is not implemented ('adam', 'sgd' are)" % spec)
        return table[spec.lower()]()
    raise TypeError('optimizer must be a name or a dlwp_amd.training.Optimizer instance')


def mean_squared_error(y_true, y_pred):  # marker objects accepted as loss= (the reference passes keras.losses.mean_squared_error)
    raise RuntimeError('marker only: pass it as loss=, the HIP loss kernel computes it')


def mean_absolute_error(y_true, y_pred):
    raise RuntimeError('marker only')


def _loss_name(loss):
    from .custom import LossSpec
    if isinstance(loss, LossSpec):
        return 'custom'
    name = loss if isinstance(loss, str) else getattr(loss, '__name__', None)
    if name in ('mse', 'MSE', 'mean_squared_error'):
        return 'mse'
    raise NotImplementedError("loss %r is not implemented on the HIP path ('mse' / mean_squared_error, "
                              "dlwp_amd.custom.anomaly_correlation_loss(...) and latitude_weighted_loss(...) are)"
                              % (loss,))


_METRIC_NAMES = {'mae': 'mean_absolute_error', 'mean_absolute_error': 'mean_absolute_error',
                 'mse': 'mean_squared_error', 'mean_squared_error': 'mean_squared_error'}


# ------------------------------------------------------------------------------------------------------------------ #
# flat parameter storage
# ------------------------------------------------------------------------------------------------------------------ #

def flatten_parameters(model):
    """Re-home every layer weight as a view into one contiguous fp32 buffer (values preserved).  Returns the buffer and
    [(layer, name, offset, numel, shape)]."""
    entries, total = [], 0
    seen = set()
    for lay in model.layers:
        if id(lay) in seen:
            continue
        seen.add(id(lay))
        for name, w in lay._weights:
            entries.append((lay, name, total, w.numel(), tuple(w.shape)))
            total += w.numel()
    flat = torch.empty(max(total, 1), dtype=torch.float32, device=model.device)
    for lay, name, off, numel, shape in entries:
        old = dict(lay._weights)[name]
        view = flat[off:off + numel].view(shape)
        view.copy_(old)
        setattr(lay, name, view)
        lay._weights = [(nm, view if nm == name else t) for nm, t in lay._weights]
    return flat, entries


# ------------------------------------------------------------------------------------------------------------------ #
# trainer
# ------------------------------------------------------------------------------------------------------------------ #

class _StepHandle(object):
    """Owner of a dlwp_train_step_t (csrc/tape.hip)."""

    def __init__(self, h):
        self.h = h

    def close(self):
        from . import _lib
        if self.h is not None:
            _lib.lib.dlwp_train_step_destroy(self.h)
            self.h = None

    def __del__(self):      # (never destroys here: _lib.bury)
        try:
            from . import _lib
            _lib.bury('step', self.h)
            self.h = None
        except Exception:  # noqa: BLE001
            pass


class _PhaseGrad(object):
    """The loss gradient of an output the plan produces as phase channels + depth-to-space, in phase layout: the gradient of
    plan buffer `buf` (written by convolution op `conv_k`); db: the bias gradient of the phase channels, or None."""

    def __init__(self, buf, conv_k, dz, db):
        self.buf, self.conv_k, self.dz, self.db = buf, conv_k, dz, db


class Trainer(object):
    def __init__(self, model):
        self.model = model
        self.plan = model.plan
        self.device = model.device
        self.loss_kind = _loss_name(model.loss)
        n_out = len(model.outputs)
        lw = model.loss_weights
        if lw is None:
            lw = [1.0] * n_out
        if isinstance(lw, dict):
            raise NotImplementedError('loss_weights as a dict')
        if len(lw) != n_out:
            raise ValueError('loss_weights has %d entries for %d outputs' % (len(lw), n_out))
        self.loss_weights = [float(v) for v in lw]
        self.metric_keys = []
        for m in model.metrics:
            key = m if isinstance(m, str) else getattr(m, '__name__', None)
            if key not in _METRIC_NAMES:
                raise NotImplementedError('metric %r is not implemented (mae, mse are)' % (m,))
            self.metric_keys.append(_METRIC_NAMES[key])
        out_names = [t.layer.name for t in model.outputs]
        if n_out == 1:
            self.metrics_names = ['loss'] + list(self.metric_keys)
        else:
            self.metrics_names = (['loss'] + ['%s_loss' % nm for nm in out_names] +
                                  ['%s_%s' % (nm, mk) for nm in out_names for mk in self.metric_keys])
        self.flat_params, self.entries = flatten_parameters(model)
        # gradients + (behind them) the [n_out, 7] loss table of a data-parallel step: ONE buffer, ONE all-reduce
        n_par = self.flat_params.numel()
        n_x = -(-(n_par + 7 * n_out) // 4) * 4        # (whole float4: the library's own exchange moves 16-byte units)
        self._flat_exchange = torch.zeros(n_x, dtype=torch.float32, device=self.device)
        self.flat_grads = self._flat_exchange[:n_par]
        self._loss_tail = self._flat_exchange[n_par:n_par + 7 * n_out].view(n_out, 7)
        self.opt_state = None
        self.dp = getattr(model, '_dp', None)
        self._grad_bufs = {}
        self._loss_out = None
        self._phase_out = None        # _phase_outputs(): outputs whose loss is taken on the phase channels
        self._dact_ops = None         # _dgrad_act_ops(): data gradients that carry the producer's activation backward
        self._loss_consts = None      # device copies of a custom loss's climatology / latitude weights
        self._params_dirty = False    # set by Model.set_weights / load: replicas re-align at the next collective step
        self._graphs = {}             # (n_local, n_global) -> captured training step (see _graph_step)
        self._graph_seen = {}
        self._iter_dev = None         # Adam's step number on the device (advanced inside the captured step)
        self._iter_shadow = None
        self._fold = None             # see _fold_ok
        self._prep_cache = {}         # batch size -> prepared weights of the step (see _prepare_step)
        self._side = None             # second stream: the weight gradients run beside the data-gradient chain
        self._loader_threads = 0      # > 0 while fit_generator feeds through a DeviceLoader (see _graph_ok)
        self._foreign = False         # set where a step launches through torch instead of the library (see _record_step)
        self._no_tape = set()         # step shapes whose recording was refused
        self.sync_parameters()

    # -- replicas ------------------------------------------------------------------------------------------------------ #
    def sync_parameters(self):
        """Data parallel: every replica takes rank 0's parameters (and optimizer state).  Layers draw their initial
        weights from per-process random streams, so without this the replicas would apply the same summed gradient to
        different weights (keras.utils.multi_gpu_model has ONE weight set by construction, models.py:104-109)."""
        dp = self.dp
        self._params_dirty = False
        if dp is None or dp.world <= 1:
            return
        dp.broadcast_(self.flat_params)
        if self.opt_state is not None:
            for t in self.opt_state:
                dp.broadcast_(t)

    # -- helpers ------------------------------------------------------------------------------------------------------ #
    def _grad_view(self, layer, name):
        if isinstance(layer, L._ConvPart):       # a ConvLSTM2D convolution: the weights belong to the parent layer
            layer, name = layer.parent, (layer.which if name == 'kernel' else 'bias')
        for lay, nm, off, numel, shape in self.entries:
            if lay is layer and nm == name:
                return self.flat_grads[off:off + numel].view(shape)
        raise KeyError(name)

    def _to_device(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def _targets(self, y, n):
        ys = list(y) if isinstance(y, (list, tuple)) else [y]
        if len(ys) != len(self.plan.output_store):
            raise ValueError('model has %d outputs but %d target arrays were given' % (len(self.plan.output_store), len(ys)))
        out = []
        for t, store in zip(ys, self.plan.output_store):
            t = self._to_device(t)
            if t.shape[0] != n or t.numel() != n * int(np.prod(store)):
                raise ValueError('target shape %r does not match the model output %r' % (tuple(t.shape), (n,) + tuple(store)))
            out.append(t.reshape((n,) + tuple(store)))
        return out

    # -- forward + loss ------------------------------------------------------------------------------------------------ #
    # -- a step's weight-side helpers in one launch each (csrc/batch.hip) ------------------------------------------------ #
    def _fold_ok(self):
        """The folded step (default; DLWP_TRAIN_FOLD=0 keeps one launch per helper): every prepared form of the weights --
        Winograd / packed-N forms for the forward and data-gradient convolutions, the flipped kernels -- is built by ONE
        launch in front of the forward, every final sum (weight-gradient slabs, bias-gradient and loss partials) by ONE launch
        behind the backward pass, and the weight gradients run on a second stream beside the data-gradient chain.  At 8
        samples per GPU that is 64 -> ~36 launches.  Plans it covers: Conv2D layers used once each (no ConvLSTM2D /
        RowConnected2D / shared layers)."""
        if self._fold is None:
            ok = self.device.type == 'cuda' and os.environ.get('DLWP_TRAIN_FOLD', '1') != '0'
            seen = set()
            for op in self.plan.ops:
                if op.kind in ('lstm', 'rowconv') or (op.kind == 'conv' and (op.lstm_f or isinstance(op.layer, L._ConvPart))):
                    ok = False
                if op.kind == 'conv':
                    ok = ok and id(op.layer) not in seen
                    seen.add(id(op.layer))
            # every deferred sum of a step in ONE flush (csrc/common.h: DLWP_MAX_BATCH_JOBS = 24; an early flush would run on
            # whichever stream filled the table)
            ok = ok and 2 * len(seen) + len(self.plan.output_store) <= 24 and 3 * len(seen) <= 24
            self._fold = bool(ok)
        return self._fold

    def _prepare_step(self, x):
        """Derived (phase-summed) kernels, then every prepared operand of this step's convolutions in one launch.  Returns
        {'fwd': {op index: tensor}, 'bwd': {op index: (tensor, stored)}}; the buffers are cached per batch size."""
        from . import _lib, ops
        ex = self.model.train_executor
        plan = self.plan
        n = int(x.shape[0])
        x = x.reshape((n,) + plan._in_store)
        for op in plan.ops:
            if op.kind == 'phasew':
                w2, b2 = ex.phase_buffers()[op.wparam]
                ops.phase_weights(op.layer.kernel, op.layer.bias, op.halo.top, op.halo.left, w2=w2, b2=b2)
        cache = self._prep_cache.get(n)
        if cache is None:
            if len(self._prep_cache) > 4:
                self._prep_cache.clear()
            cache = self._prep_cache[n] = {'fwd': {}, 'bwd': {}}
        bufs = ex.scratch(n)
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        ops.prepare_begin(self.device)
        try:
            for k, (op, d) in enumerate(zip(plan.ops, ex._descriptors())):
                if op.kind != 'conv':
                    continue
                kern = ex.conv_weights(op)[0]
                src = x if op.src == P.STATE_IN else bufs[op.src]
                dst_dtype = bufs[op.dst].dtype if op.dst >= 0 else torch.float32
                u = ops.conv2d_prepare(src, kern, d, out_dtype=dst_dtype, x_channels=op.xs[0], out=cache['fwd'].get(k))
                if u is not None:
                    cache['fwd'][k] = u
                if op.src == P.STATE_IN:
                    continue
                xs = _lib.Shape4(n, op.xs[0], op.xs[1], op.xs[2])
                old = cache['bwd'].get(k)
                stored = old[1] if old is not None else (op.src_mode == P.SRC_UPSAMPLE2 and
                                                         ops.conv2d_bwd_data_prepared_bytes(dev, xs, d, True) > 0)
                t = ops.conv2d_bwd_data_prepare(kern, d, xs, stored=stored, out=old[0] if old is not None else None)
                if t is not None:
                    cache['bwd'][k] = (t, stored)
        finally:
            ops.prepare_flush(self.device)
        return cache

    def _forward_loss(self, x, ys, want_grad, weight_scale=1.0, prep=None):
        """Returns (outs, loss table [n_out, 7] on device: col 0 custom-loss value, 1 mse, 2 mae, dys or None)."""
        from . import ops
        phase = self._phase_outputs()
        skip = tuple(k for k, _, _ in phase.values())
        if prep is not None:
            outs = self.model.train_executor.run(x, prepared=prep['fwd'], skip_phasew=True, skip_ops=skip)
        else:
            outs = self.model.train_executor.run(x, skip_ops=skip)
        n_out = len(outs)
        if self._loss_out is None or self._loss_out.shape[0] != n_out:
            self._loss_out = torch.zeros((n_out, 7), dtype=torch.float32, device=self.device)
        spec = self.model.loss if self.loss_kind == 'custom' else None
        if spec is not None and self._loss_consts is None:
            mean = None if spec.mean is None else torch.from_numpy(spec.mean).to(self.device).contiguous()
            roww = None if spec.row_weights is None else torch.from_numpy(spec.row_weights).to(self.device).contiguous()
            self._loss_consts = (mean, roww)
        dys = []
        for o, (yp, yt) in enumerate(zip(outs, ys)):
            lw = self.loss_weights[o] * weight_scale
            if o in phase:       # 'mse' on the phase channels of a restated output layer: no depth-to-space / space-to-depth pass
                _, d2s, conv_k = phase[o]
                lay = self.plan.ops[conv_k].layer
                yph = self.model.train_executor.scratch(x.shape[0])[d2s.src]
                dz = torch.empty_like(yph) if want_grad else None
                db = None
                if want_grad and lay.bias is not None and lay.activation == 'linear':     # db = the sums of dz: same pass
                    db = torch.empty(yph.shape[1], dtype=torch.float32, device=self.device)
                ops.mse_mae_phase(yph, yt.reshape((yph.shape[0], yph.shape[1] // 4, 2 * yph.shape[2], 2 * yph.shape[3])),
                                  self._loss_out[o, 1:3], dz, db, lw, ws_key=('mse', o) if prep is not None else None)
                dys.append(_PhaseGrad(d2s.src, conv_k, dz, db) if want_grad else None)
                continue
            dy = torch.empty_like(yp) if want_grad else None
            if spec is None:
                ops.mse_mae(yp, yt, self._loss_out[o, 1:3], dy, lw, ws_key=('mse', o) if prep is not None else None)
            else:
                mean, roww = self._loss_consts
                if yp.dim() != 4:
                    raise NotImplementedError('custom losses need (n, c, h, w) outputs')
                if mean is not None and mean.numel() != yp[0].numel():
                    raise ValueError('anomaly_correlation_loss mean has %d elements, the model output %d per sample'
                                     % (mean.numel(), yp[0].numel()))
                if roww is not None and roww.numel() != yp.shape[2]:
                    raise ValueError('latitude weights have %d rows, the model output %d' % (roww.numel(), yp.shape[2]))
                ops.loss_custom(yp, yt, self._loss_out[o], dy, lw * spec.scale, mean, roww, spec.kind, spec.regularize)
            dys.append(dy)
        return outs, self._loss_out, dys

    def _dgrad_act_ops(self):
        """{convolution op k: op index of the Conv2D whose activation output k alone reads} -- candidates for
        dlwp_conv2d_bwd_data_act: k's data gradient then multiplies by act'(that output) and sums that layer's bias gradient in its
        store phase, and the producer's dlwp_act_bwd_bias_grad launch disappears.  OFF unless DLWP_DGRAD_ACT=1: measured equal to
        the two launches on config 3 (batch 64: 1.385-1.391 vs 1.382-1.399 ms, batch 8: 0.410 vs 0.409 ms, same box) -- the store
        phase's extra read costs the data gradient what the separate pass cost."""
        if self._dact_ops is None:
            found = {}
            if os.environ.get('DLWP_DGRAD_ACT', '0') == '1':
                ops_ = self.plan.ops
                uses = {}
                for w in ops_:
                    if w.kind == 'conv' and w.layer is not None:
                        uses[id(w.layer)] = uses.get(id(w.layer), 0) + 1
                for k, op in enumerate(ops_):
                    if op.kind != 'conv' or op.src < 0 or op.src_mode != P.SRC_DIRECT or op.lstm_f or op.src2 is not None:
                        continue
                    chans = self.plan.buffers[op.src][0]
                    if op.in_c_off != 0 or op.xs[0] != chans:
                        continue
                    readers = [j for j, r in enumerate(ops_) if r.src == op.src or (r.src2 is not None and r.src2.get('buf') == op.src)
                               or (r.aux is not None and op.src in [a for a in r.aux if a is not None])]
                    writers = [j for j, w in enumerate(ops_) if w.dst == op.src]
                    if readers != [k] or len(writers) != 1:
                        continue
                    pw = ops_[writers[0]]
                    if (pw.kind == 'conv' and not pw.lstm_f and pw.wparam is None and not pw.out_pool and not pw.out_d2s and
                            pw.out_c_off == 0 and pw.conv_geometry[0] == chans and pw.layer is not None and
                            pw.layer.bias is not None and pw.layer.activation in ('tanh', 'relu') and
                            uses.get(id(pw.layer), 0) == 1 and writers[0] < k):
                        found[k] = writers[0]
            self._dact_ops = found
        return self._dact_ops

    def _phase_outputs(self):
        """{output index: (index of its 'd2s' op, that op, index of the convolution in front)} for outputs the plan restates as
        phase channels + depth-to-space (plan.py) and whose loss is the plain 'mse': the step then takes loss,
        gradient and bias gradient on the phase channels (ops.mse_mae_phase).  DLWP_PHASE_LOSS=0 keeps the separate passes."""
        if self._phase_out is None:
            found = {}
            if self.loss_kind != 'custom' and os.environ.get('DLWP_PHASE_LOSS', '1') != '0':
                ops_ = self.plan.ops
                for k, op in enumerate(ops_):
                    if op.kind != 'd2s' or op.dst >= 0 or op.dst == P.STATE_IN or op.out_c_off != 0:
                        continue
                    o = -2 - op.dst
                    writers = [j for j, w in enumerate(ops_) if w.dst == op.src]
                    readers = [j for j, r in enumerate(ops_) if r.src == op.src and j != k]
                    store = self.plan.output_store[o] if 0 <= o < len(self.plan.output_store) else None
                    if (len(writers) == 1 and not readers and ops_[writers[0]].kind == 'conv' and
                            ops_[writers[0]].wparam is not None and store is not None and store[0] == op.xs[0] and
                            sum(1 for w in ops_ if w.dst == op.dst) == 1):
                        found[o] = (k, op, writers[0])
            self._phase_out = found
        return self._phase_out

    # -- kernel regularisers (keras.regularizers.l2 on the ConvLSTM2D kernel, examples/train.py:154) --------------------- #
    def _regularized(self):
        from .regularizers import L1L2
        for lay, nm, off, numel, shape in self.entries:
            reg = getattr(lay, 'kernel_regularizer', None)
            if nm == 'kernel' and isinstance(reg, L1L2) and reg.l2 > 0:
                yield reg.l2, off, numel

    def _add_regularizer_gradients(self):
        from . import ops
        for l2, off, numel in self._regularized():
            ops.axpby(self.flat_params[off:off + numel], self.flat_grads[off:off + numel], 2.0 * l2, 1.0)

    def _regularizer_loss(self):
        """sum over regularised kernels of l2 * sum(w^2) -- Keras adds it to the reported total loss."""
        tot = 0.0
        for l2, off, numel in self._regularized():
            w = self.flat_params[off:off + numel]
            tot += l2 * float(torch.dot(w, w))
        return tot

    def _report(self, loss_vals, reg=None):
        """[loss, (per-output losses), metrics...] as python floats from the device loss table; reg: the kernel
        regularisers' penalty that belongs to it (None: at the current weights)."""
        v = loss_vals.detach().cpu().numpy().astype(np.float64)
        if self.dp is not None and getattr(self.dp, '_xchg', None) is not None:
            self.dp.oneshot_check()       # (a timed-out one-shot exchange: raise here, where the host looks at the step)
        return self._report_from(v, reg)

    def _report_from(self, v, reg=None):
        n_out = v.shape[0]
        if self.loss_kind == 'custom':
            per_out = [float(self.model.loss.scale * v[o, 0]) for o in range(n_out)]
        else:
            per_out = [float(v[o, 1]) for o in range(n_out)]
        total = float(sum(w * l for w, l in zip(self.loss_weights, per_out))) + (self._regularizer_loss() if reg is None else reg)
        col = {'mean_squared_error': 1, 'mean_absolute_error': 2}
        if n_out == 1:
            return [total] + [float(v[0, col[k]]) for k in self.metric_keys]
        return ([total] + per_out + [float(v[o, col[k]]) for o in range(n_out) for k in self.metric_keys])

    #: side streams of the weight gradients (DLWP_SIDE_STREAMS).  With the loader's copy stream and the main stream that is four -- the
    #: hardware queues a process has by default
    side_streams = 2

    def _ensure_side_streams(self):
        """The streams the weight gradients run on beside the data-gradient chain: probed to sit on hardware queues of their own
        (util.distinct_streams).  fit_generator calls this BEFORE it builds its DeviceLoader, whose copy stream then avoids them."""
        if self._side is None and self.device.type == 'cuda':
            from .util import distinct_streams
            k = max(1, int(os.environ.get('DLWP_SIDE_STREAMS', self.side_streams)))
            self._side = distinct_streams(self.device, k, [torch.cuda.current_stream(self.device)])
        return self._side or []

    # -- backward ------------------------------------------------------------------------------------------------------ #
    def _backward(self, x, outs, dys, prep=None):
        """prep (the folded step, _fold_ok): prepared data-gradient operands, a workspace per deferred final sum, the weight
        gradients on the side stream; the caller brackets forward + backward with ops.reductions_begin / _flush and runs
        the returned callables (work that needs the final sums) after the flush."""
        from . import _lib, ops
        plan = self.plan
        n = x.shape[0]
        post = []
        main = torch.cuda.current_stream(self.device) if prep is not None else None
        if prep is not None and self._side is None:
            self._ensure_side_streams()
        sides = self._side if prep is not None else None
        if sides is not None and torch.cuda.is_current_stream_capturing():
            # inside a captured step everything stays on ONE stream: replaying graphs with forked branches crashed inside
            # hipGraphLaunch now and then (r3: 2 of 3 full test runs), and the forks bought little there (0.49 vs ~0.47 ms at 8
            # samples: a fork / join costs ~10 us in a graph).  The eager step keeps the side streams (batch 64: 1.78 -> 1.68 ms).
            sides = None
        # Weight gradients leave the critical path (activation backward -> data gradient -> ...): they are queued and launched
        # on side streams in TWO batches with one fork each -- a fork / join costs ~10 us inside a captured graph
        # (profiles/r3_train_b8_timeline.txt): the first half, one after the other, beside the second half of the chain; the
        # rest in parallel behind it.
        wq, used = [], []
        n_wgrad = sum(1 for op in plan.ops if op.kind == 'conv')
        queued = [0]

        def launch_queued(parallel):
            started = set()
            for i, fn in enumerate(wq):
                st = sides[i % len(sides)] if parallel else sides[0]
                if id(st) not in started:
                    ops.stream_wait(st, main)    # behind everything issued on the main stream so far
                    started.add(id(st))
                    if st not in used:
                        used.append(st)
                with torch.cuda.stream(st):
                    fn()
            del wq[:]

        # (r3, eager steps: every weight gradient forks as soon as its operands exist -- the two-batch rule above was made for
        #  forks inside a captured graph, which is single-stream now; batch 64: the last layers' weight gradients no longer pile
        #  up behind the end of the chain.  DLWP_WGRAD_FORKS=batch keeps the two batches)
        each = os.environ.get('DLWP_WGRAD_FORKS', 'each') != 'batch'

        def on_side(fn):
            if sides is None:
                return fn()
            if each:
                st = sides[queued[0] % len(sides)]
                queued[0] += 1
                ops.stream_wait(st, main)
                if st not in used:
                    used.append(st)
                with torch.cuda.stream(st):
                    fn()
                return None
            wq.append(fn)
            queued[0] += 1
            if queued[0] == (n_wgrad + 1) // 2:
                launch_queued(False)
        # r4: where the step replays as ONE hipGraph on one stream (small batches: every launch under a round of workgroups), a
        # layer's weight gradient and its data gradient -- both read dz, neither the other's output -- leave in one launch
        # (ops.pair_begin / pair_end, csrc/conv_pair.hip).  The pair opens in front of the weight gradient and closes behind the data
        # gradient's convolution; what is issued in between (bias gradient) depends on neither.
        can_pair = (prep is not None and os.environ.get('DLWP_TRAIN_PAIR', '1') != '0' and self._graph_ok(int(n)) == 'graph')
        pair_open = [False]

        def pair_begin():
            if can_pair and not pair_open[0]:
                ops.pair_begin(self.device)
                pair_open[0] = True

        def pair_end():
            if pair_open[0]:
                pair_open[0] = False
                ops.pair_end(self.device)
        bufs = self.model.train_executor.scratch(n)
        x = x.reshape((n,) + plan._in_store)

        def tensor(i):
            if i >= 0:
                return bufs[i]
            return x if i == P.STATE_IN else outs[-2 - i]

        grads = {}          # buffer id -> gradient tensor
        written = {}        # buffer id -> list of (c_off, c) windows already holding a gradient

        def overlaps(buf, c_off, c):
            return any(not (c_off + c <= o or o + k <= c_off) for o, k in written.get(buf, []))

        # buffer id -> gradient of its MaxPooling2D(2) image, when that is the only gradient so far: the producing
        # convolution then takes pooling + activation backward (+ bias gradient) in one pass
        pending_pool = {}

        def materialise(buf):
            gp = pending_pool.pop(buf, None)
            if gp is not None:
                deposit(buf, 0, gp.shape[1], ops.maxpool2_bwd(tensor(buf), gp))

        def grad_of(buf):
            materialise(buf)
            g = grads.get(buf)
            if g is None:
                g = torch.empty_like(tensor(buf))
                grads[buf] = g
            return g

        preact = set()       # buffers whose gradient is already a PRE-activation gradient (dlwp_conv2d_bwd_data_act)
        phase_db = {}        # convolution op index -> bias gradient of its phase channels, already summed with the loss
        for o, dy in enumerate(dys):
            if isinstance(dy, _PhaseGrad):     # the gradient arrives on the phase channels: the 'd2s' op has no adjoint to run
                grads[dy.buf] = dy.dz
                written[dy.buf] = [(0, dy.dz.shape[1])]
                if dy.db is not None:
                    phase_db[dy.conv_k] = dy.db
                continue
            grads[P.OUT(o)] = dy
            written[P.OUT(o)] = [(0, dy.shape[1])]

        def deposit(buf, c_off, c, dense):
            """Put `dense` (n, c, h, w) into window [c_off, +c) of grad[buf]: overwrite on first touch, add after."""
            materialise(buf)
            if grads.get(buf) is None and c_off == 0 and tuple(dense.shape) == tuple(tensor(buf).shape) and \
                    dense.is_contiguous():
                grads[buf] = dense            # first gradient of the whole tensor: adopt it, no copy
                written[buf] = [(0, c)]
                return
            g = grad_of(buf)
            full = c_off == 0 and c == g.shape[1]
            if not overlaps(buf, c_off, c):
                if full:
                    if dense.data_ptr() != g.data_ptr():
                        self._foreign = True
                        g.view(-1).copy_(dense.reshape(-1))
                else:
                    ops.copy_channels(dense.reshape((n, c) + tuple(g.shape[2:])), g, c, 0, c_off)
                written.setdefault(buf, []).append((c_off, c))
            elif full and written[buf] == [(0, g.shape[1])]:
                ops.axpby(dense.reshape(-1), g.view(-1), 1.0, 1.0)
            elif all(any(o <= ch < o + k for o, k in written[buf]) for ch in range(c_off, c_off + c)):
                # the window already holds a gradient everywhere (e.g. h_{t-1}: read by the next layer as part of the
                # whole sequence AND by the recurrent convolution of step t): extract, add, put back
                cur = torch.empty((n, c) + tuple(g.shape[2:]), dtype=torch.float32, device=self.device)
                ops.copy_channels(g, cur, c, c_off, 0)
                ops.axpby(dense.reshape(-1), cur.view(-1), 1.0, 1.0)
                ops.copy_channels(cur, g, c, 0, c_off)
            else:
                raise NotImplementedError('partially overlapping channel windows in the backward pass')

        touched_layers = set()
        descs = self.model.train_executor._descriptors()
        for k, (op, d) in reversed(list(enumerate(zip(plan.ops, descs)))):
            key = (lambda what, k=k: (what, k)) if prep is not None else (lambda what: None)   # a scratch per deferred sum
            pooled_grad = None
            if op.dst in pending_pool:
                lay = op.layer if op.kind == 'conv' else None
                if (lay is not None and op.wparam is None and op.out_c_off == 0 and
                        op.conv_geometry[0] == tensor(op.dst).shape[1] and
                        (lay.bias is None or id(lay) not in touched_layers)):
                    pooled_grad = pending_pool.pop(op.dst)
                else:
                    materialise(op.dst)
            if op.dst not in grads and pooled_grad is None:
                continue                      # nothing downstream of this op contributes to the loss
            gD = grads.get(op.dst)
            src = tensor(op.src)
            if op.kind == 'd2s':               # adjoint of the depth-to-space interleave of a restated decoder layer
                grads[op.src] = ops.space_to_depth2(gD, op.xs[0], c_off=op.out_c_off)
                written[op.src] = [(0, 4 * op.xs[0])]
                continue
            if op.kind == 'conv':
                lay = op.layer
                y = tensor(op.dst)
                acc = id(lay) in touched_layers
                derived = op.wparam is not None        # the layer runs with phase-summed kernels (plan.phase_params)
                kern = self.model.train_executor.conv_weights(op)[0]
                n_out = op.conv_geometry[0]
                fused_bias = (pooled_grad is None and lay.activation != 'linear' and lay.bias is not None and not acc and
                              not derived and gD.shape[1] == lay.filters and tuple(y.shape) == tuple(gD.shape))
                if (pooled_grad is not None and op.src == P.STATE_IN and not acc and pooled_grad.is_contiguous() and
                        os.environ.get('DLWP_WGRAD_POOLED', '1') != '0' and
                        ops.conv2d_bwd_weight_pooled_supported(_lib.Shape4(n, op.xs[0], op.xs[1], op.xs[2]), d)):
                    # the first layer under MaxPooling2D(2): nobody needs its data gradient, so the pooling + activation
                    # backward is formed inside the weight gradient's loader and the gradient tensor is never stored
                    gb = self._grad_view(lay, 'bias') if lay.bias is not None else None
                    on_side(lambda src=src, y=y, pg=pooled_grad, lay=lay, gb=gb, d=d, act=op.act, key=key,
                            xs=_lib.Shape4(n, op.xs[0], op.xs[1], op.xs[2]):
                            ops.conv2d_bwd_weight_pooled(src, y, pg, self._grad_view(lay, 'kernel'), gb, d, xs, act,
                                                         ws_key=key('wgrad')))
                    touched_layers.add(id(lay))
                    if isinstance(lay, L._ConvPart):
                        touched_layers.add(id(lay.parent))
                    continue
                if op.dst in preact:           # the reader's data gradient left dz and this layer's bias gradient already
                    fused_bias = True
                elif pooled_grad is not None:  # the layer's only reader is MaxPooling2D(2): its backward rides along
                    fused_bias = lay.bias is not None
                    gD = ops.pool_act_bwd_bias_grad(y, pooled_grad, op.act,
                                                    self._grad_view(lay, 'bias') if fused_bias else None, ws_key=key('bias'))
                elif fused_bias:       # dz in place of dy and the bias gradient from the same pass
                    ops.act_bwd_bias_grad(y, gD, op.act, self._grad_view(lay, 'bias'), lay.filters, out=gD,
                                          ws_key=key('bias'))
                elif lay.activation != 'linear':
                    ops.act_bwd(y, gD, op.act, out=gD)          # dz in place of dy
                dz = gD
                xs = _lib.Shape4(n, op.xs[0], op.xs[1], op.xs[2])
                if op.src != P.STATE_IN:
                    pair_begin()           # (closed behind the data gradient's convolution, below)
                if derived:                # gradients of the derived kernels, folded back onto the layer's own
                    pp = plan.phase_params[op.wparam]
                    dw2 = torch.empty(tuple(kern.shape), dtype=torch.float32, device=self.device)
                    have_db = k in phase_db
                    db2 = phase_db[k] if have_db else (
                        torch.empty(n_out, dtype=torch.float32, device=self.device) if lay.bias is not None else None)

                    def derived_grads(src=src, dz=dz, dw2=dw2, db2=db2, d=d, xs=xs, n_out=n_out, key=key, have_db=have_db):
                        ops.conv2d_bwd_weight(src, dz, dw2, d, xs, ws_key=key('wgrad'))
                        if db2 is not None and not have_db:
                            ops.bias_grad(dz, db2, n_out, ws_key=key('bias'))
                    on_side(derived_grads)

                    def fold_back(dw2=dw2, db2=db2, lay=lay, pp=pp, acc=acc):
                        ops.phase_weights_bwd(dw2, db2, self._grad_view(lay, 'kernel'),
                                              self._grad_view(lay, 'bias') if lay.bias is not None else None,
                                              pp['pad_top'], pp['pad_left'], accumulate=acc)
                    if prep is not None:
                        post.append(fold_back)          # dw2 / db2 are final only behind the deferred sums
                    else:
                        fold_back()
                else:
                    on_side(lambda src=src, dz=dz, lay=lay, d=d, xs=xs, acc=acc, key=key: ops.conv2d_bwd_weight(
                        src, dz, self._grad_view(lay, 'kernel'), d, xs, accumulate=acc, ws_key=key('wgrad')))
                if lay.bias is not None and not fused_bias and not derived:
                    gb = self._grad_view(lay, 'bias')
                    if acc:
                        tmp = torch.empty_like(gb)
                        ops.bias_grad(dz, tmp, lay.filters)
                        ops.axpby(tmp, gb, 1.0, 1.0)
                    else:
                        on_side(lambda dz=dz, gb=gb, lay=lay, key=key: ops.bias_grad(dz, gb, lay.filters, ws_key=key('bias')))
                touched_layers.add(id(lay))
                if isinstance(lay, L._ConvPart):
                    touched_layers.add(id(lay.parent))
                if op.src == P.STATE_IN:
                    continue                  # no gradient w.r.t. the model input is needed
                cin = op.xs[0]
                c_total = src.shape[1]
                pb = prep['bwd'].get(k) if prep is not None else None     # (prepared operand, for the stored tensor?)
                if op.src_mode == P.SRC_DIRECT:
                    g = grad_of(op.src)
                    if not overlaps(op.src, op.in_c_off, cin):
                        # writes its channel window in place -- where this layer is the only reader of another Conv2D's
                        # activation output, already times act'(that output), with that layer's bias gradient from the same pass
                        done = False
                        kp = self._dgrad_act_ops().get(k)
                        if kp is not None:
                            pop = plan.ops[kp]
                            done = ops.conv2d_bwd_data_act(dz, kern, d, xs, g, src, pop.act, self._grad_view(pop.layer, 'bias'),
                                                           prepared=pb[0] if pb else None, ws_key=key('dact'))
                            if done:
                                preact.add(op.src)
                        if not done:
                            ops.conv2d_bwd_data(dz, kern, d, xs, g, prepared=pb[0] if pb else None)
                        pair_end()
                        written.setdefault(op.src, []).append((op.in_c_off, cin))
                    else:
                        dd = ops.make_conv(d.cout, d.kh, d.kw, (d.dil_h, d.dil_w), d.halo, d.act, 0, 0, d.out_c_off,
                                           d.out_c_total, d.src_mode)
                        tmp = torch.empty((n, cin) + tuple(src.shape[2:]), dtype=torch.float32, device=self.device)
                        ops.conv2d_bwd_data(dz, kern, dd, xs, tmp)
                        pair_end()
                        deposit(op.src, op.in_c_off, cin, tmp)
                else:
                    hin = 2 * op.xs[1] if op.src_mode == P.SRC_UPSAMPLE2 else op.xs[1] // 2
                    win = 2 * op.xs[2] if op.src_mode == P.SRC_UPSAMPLE2 else op.xs[2] // 2
                    if op.src_mode == P.SRC_UPSAMPLE2:    # 2x2 sum fused into the data-gradient kernel where it can be
                        dense = torch.empty((n, cin, op.xs[1], op.xs[2]), dtype=torch.float32, device=self.device)
                        if pb is not None and pb[1]:
                            ops.conv2d_bwd_data(dz, kern, d, xs, dense, prepared=pb[0], stored=True)
                            pair_end()
                        elif pb is not None or not ops.conv2d_bwd_data_stored(dz, kern, d, xs, dense):
                            tmp = torch.empty((n, cin, hin, win), dtype=torch.float32, device=self.device)
                            ops.conv2d_bwd_data(dz, kern, d, xs, tmp, prepared=pb[0] if pb else None)
                            pair_end()
                            dense = ops.upsample2_bwd(tmp)
                        pair_end()
                    else:
                        tmp = torch.empty((n, cin, hin, win), dtype=torch.float32, device=self.device)
                        ops.conv2d_bwd_data(dz, kern, d, xs, tmp, prepared=pb[0] if pb else None)
                        pair_end()
                        if op.in_c_off == 0 and cin == c_total:
                            xw = src
                        else:
                            xw = torch.empty((n, cin) + tuple(src.shape[2:]), dtype=torch.float32, device=self.device)
                            ops.copy_channels(src, xw, cin, op.in_c_off, 0)
                        dense = ops.maxpool2_bwd(xw, tmp)
                    deposit(op.src, op.in_c_off, cin, dense)
            elif op.kind == 'rowconv':         # RowConnected2D (reference custom.py:695-837): per-row filters
                lay = op.layer
                y = tensor(op.dst)
                acc = id(lay) in touched_layers
                if lay.activation != 'linear':
                    ops.act_bwd(y, gD, op.act, out=gD)          # dz in place of dy
                dz = gD
                xs = _lib.Shape4(n, op.xs[0], op.xs[1], op.xs[2])
                ops.rowconv2d_bwd_weight(src, dz, self._grad_view(lay, 'kernel'),
                                         self._grad_view(lay, 'bias') if lay.bias is not None else None, d, xs,
                                         accumulate=acc)
                touched_layers.add(id(lay))
                if op.src == P.STATE_IN:
                    continue
                cin = op.xs[0]
                if op.src_mode == P.SRC_UPSAMPLE2:     # gradient of the up-sampled tensor, then the 2x2 sums of its adjoint
                    dd = ops.make_conv(d.cout, d.kh, d.kw, 1, d.halo, d.act, 0, 0, d.out_c_off, d.out_c_total)
                    up = torch.empty((n, cin, 2 * op.xs[1], 2 * op.xs[2]), dtype=torch.float32, device=self.device)
                    ops.rowconv2d_bwd_data(dz, lay.kernel, dd, _lib.Shape4(n, cin, 2 * op.xs[1], 2 * op.xs[2]), up)
                    dense = ops.upsample2_bwd(up)
                else:
                    dd = ops.make_conv(d.cout, d.kh, d.kw, 1, d.halo, d.act, 0, 0, d.out_c_off, d.out_c_total)
                    dense = torch.empty((n, cin) + tuple(src.shape[2:]), dtype=torch.float32, device=self.device)
                    ops.rowconv2d_bwd_data(dz, lay.kernel, dd, xs, dense)
                deposit(op.src, op.in_c_off, cin, dense)
            elif op.kind == 'lstm':
                zh_i, cp_i, co_i = op.aux
                f = op.xs[0]
                win = (op.out_c_off, f)
                if not all(any(o <= ch < o + k for o, k in written.get(op.dst, [])) for ch in range(win[0], win[0] + f)):
                    if overlaps(op.dst, *win):
                        raise NotImplementedError('ConvLSTM2D output used through partially overlapping channel windows')
                    deposit(op.dst, win[0], f, torch.zeros((n, f) + tuple(gD.shape[2:]), dtype=torch.float32,
                                                           device=self.device))   # this h_t feeds nothing downstream
                dz, dcp = ops.convlstm_gates_bwd(src, tensor(zh_i) if zh_i is not None else None,
                                                 tensor(cp_i) if cp_i is not None else None, tensor(co_i), gD,
                                                 grads.get(co_i), f, h_c_off=op.out_c_off, act=op.act, rec_act=op.rec_act)
                grads[op.src] = dz
                written[op.src] = [(0, 4 * f)]
                if zh_i is not None:
                    grads[zh_i] = dz                 # z = zx + zh: the same gradient flows into both convolutions
                    written[zh_i] = [(0, 4 * f)]
                if cp_i is not None:
                    grads[cp_i] = dcp
                    written[cp_i] = [(0, f)]
            elif op.kind == 'copy':
                if op.src == P.STATE_IN:
                    continue
                c = op.xs[0]
                gw = torch.empty((n, c) + tuple(gD.shape[2:]), dtype=torch.float32, device=self.device)
                ops.copy_channels(gD, gw, c, op.out_c_off, 0)
                deposit(op.src, op.in_c_off, c, gw)
            elif op.kind == 'pad':
                if op.src == P.STATE_IN:
                    continue
                if op.inner > 1:
                    dense = ops.pad2d_bwd(gD.reshape((n, gD.shape[1], gD.shape[2], -1)) if gD.dim() == 4 else gD,
                                          (n, op.xs[1], op.xs[2], op.inner), d, channels_last=True)
                else:
                    dense = ops.pad2d_bwd(gD, (n,) + tuple(op.xs), d)
                deposit(op.src, 0, src.shape[1], dense)
            elif op.kind == 'maxpool':
                if op.src == P.STATE_IN:
                    continue
                if op.src >= 0 and op.src not in grads and op.src not in pending_pool:
                    pending_pool[op.src] = gD
                else:
                    deposit(op.src, 0, src.shape[1], ops.maxpool2_bwd(src, gD))
            elif op.kind == 'upsample':
                if op.src == P.STATE_IN:
                    continue
                deposit(op.src, 0, src.shape[1], ops.upsample2_bwd(gD))
            else:
                raise RuntimeError(op.kind)
        pair_end()          # (a layer whose data gradient took no convolution)
        # layers that received no gradient this step (unused by any output) must not keep a stale one
        for lay, nm, off, numel, shape in self.entries:
            if id(lay) not in touched_layers:
                self._foreign = True          # (torch's own launch: a recorded step would miss it)
                self.flat_grads[off:off + numel].zero_()
        if sides is not None:
            launch_queued(True)
            for st in used:
                ops.stream_wait(main, st)     # the weight gradients join before anything reads them
        return post

    def _forward_backward(self, x, ys, scale):
        """Forward + loss + backward of one step; gradients in flat_grads.  Returns (outs, loss table, dys)."""
        from . import ops
        if not self._fold_ok():
            outs, loss_vals, dys = self._forward_loss(x, ys, True, scale)
            self._backward(x, outs, dys)
            return outs, loss_vals, dys
        prep = self._prepare_step(x)
        ops.reductions_begin(self.device)
        post = []
        try:
            outs, loss_vals, dys = self._forward_loss(x, ys, True, scale, prep)
            post = self._backward(x, outs, dys, prep)
        finally:
            ops.reductions_flush(self.device)      # (always: the handle must not stay in recording mode)
        for fn in post:
            fn()
        return outs, loss_vals, dys

    # -- optimizer ----------------------------------------------------------------------------------------------------- #
    def _ensure_opt_state(self):
        opt = self.model.optimizer
        if self.opt_state is None:
            if isinstance(opt, Adam):
                self.opt_state = (torch.zeros_like(self.flat_params), torch.zeros_like(self.flat_params))
            else:
                self.opt_state = (torch.zeros_like(self.flat_params),)

    def _exchange_and_apply(self, dp):
        """The data-parallel half of a step: the flat gradient buffer (loss table in its tail) summed over the ranks, then the
        optimizer with grad_scale 1 / world.  RCCL (or torch.distributed) + the update as two launches; with DLWP_ALLREDUCE=oneshot
        and Adam ONE launch of the library's own exchange does both (csrc/xchg.hip)."""
        opt = self.model.optimizer
        if dp.wants_oneshot(self._flat_exchange):
            self._ensure_opt_state()
            if isinstance(opt, Adam):
                m, v = self.opt_state
                dp.oneshot_adam_(self._flat_exchange, self.flat_params.numel(), self.flat_params, m, v, opt.lr, opt.beta_1,
                                 opt.beta_2, opt.epsilon, opt.decay, opt.iterations, 1.0 / dp.world)
                opt.iterations += 1
                return
            dp.oneshot_all_reduce_(self._flat_exchange)
        else:
            dp.all_reduce_sum_(self._flat_exchange)        # gradients and loss table: one collective
        self._apply(1.0 / dp.world)

    def _apply(self, grad_scale=1.0):
        from . import ops
        opt = self.model.optimizer
        self._ensure_opt_state()
        if isinstance(opt, Adam):
            m, v = self.opt_state
            ops.adam_keras(self.flat_params, m, v, self.flat_grads, opt.iterations, opt.lr, opt.beta_1, opt.beta_2,
                           opt.epsilon, opt.decay, grad_scale)
        else:
            ops.sgd_keras(self.flat_params, self.opt_state[0], self.flat_grads, opt.iterations, opt.lr, opt.momentum,
                          opt.decay, grad_scale)
        opt.iterations += 1

    # -- public steps -------------------------------------------------------------------------------------------------- #
    def _host_rows(self, a, lo, hi):
        """rows [lo, hi) of a host array / device tensor WITHOUT touching the others (no upload of foreign rows)"""
        return a[lo:hi]

    def train_on_batch(self, x, y, return_device=False):
        """One optimisation step on a GLOBAL batch.  Under data parallelism every rank passes the same global batch (the
        Keras contract: one script, run by every rank) and trains on its own row shard -- only those rows are uploaded;
        the reported loss is the global-batch value on every rank.  Loaders that hold only the local rows call
        train_on_shard."""
        n_global = int(x.shape[0])
        dp = self.dp
        if dp is not None and dp.world > 1:
            lo, hi = dp.shard(n_global)
            ys = list(y) if isinstance(y, (list, tuple)) else [y]
            x = self._host_rows(x, lo, hi)
            ys = [self._host_rows(t, lo, hi) for t in ys]
            return self.train_on_shard(x, ys if isinstance(y, (list, tuple)) else ys[0], n_global, return_device)
        return self.train_on_shard(x, y, n_global, return_device)

    # -- the step as a hipGraph --------------------------------------------------------------------------------------------- #
    #: a batch shape seen this many times is captured (the first steps run eagerly: lazy allocations, scratch buffers)
    graph_after = 2

    #: DLWP_TRAIN_GRAPH unset: steps of at most this many samples x grid points replay as a captured graph
    # (r4, gpurun s22, ms per step graph / lanes -- the lanes on the streams the step was recorded on, DLWP_STEP_LANES_RECORDED --:
    #  8 samples 0.382 / 0.394, 12: 0.450 / 0.434, 16: 0.560 / 0.536, 24: 0.692 / 0.667, 64: 1.431 / 1.381; the graph form needs 0.04 ms
    #  of host time per step, the lanes 0.18)
    graph_below = 10 * 88 * 180

    def _graph_ok(self, n_local=None):
        """How a step of n_local samples runs: False -- launch by launch from Python (the eager step) -- or the form in which it is
        REPLAYED once its shape has been seen graph_after times:
          'graph' / 'lanes' / 'branches'  the library's step object (csrc/tape.hip, dlwp_train_step_*): the launch sequence is
                    recorded once, while an eager step runs, and replayed with ONE C call -- as one hipGraph on a single stream
                    ('graph'), launch by launch with the weight gradients on the step's side streams ('lanes'), or as one hipGraph
                    with those lanes as branches ('branches').  Plans the folded step covers (_fold_ok).  r4: this replaces both the
                    ~0.6 ms of Python per eager step and the torch.cuda.CUDAGraph capture, which raced a DeviceLoader's thread.
          'torch'   torch.cuda.CUDAGraph around the Python step (r2/r3): plans outside the folded step, DLWP_TRAIN_GRAPH=1 only.
        DLWP_TRAIN_GRAPH: '0' never replay; '1' always; unset: folded plans replay -- small steps (at most graph_below samples x
        grid points) as 'graph', larger ones as 'lanes' (their weight gradients gain from the side streams: batch 64 1.78 ->
        1.68 ms in r3).  DLWP_TRAIN_STEP=graph|lanes|branches|torch picks the form.  Never replayed: kernel regularisers (their
        penalty is read back to the host every step), SGD with decay (its rate is a launch argument), steps on the CPU device."""
        opt = self.model.optimizer
        mode = os.environ.get('DLWP_TRAIN_GRAPH', 'auto')
        if self.device.type != 'cuda' or mode == '0' or n_local is None:
            return False
        if any(True for _ in self._regularized()):
            return False
        if not (isinstance(opt, Adam) or (isinstance(opt, SGD) and opt.decay == 0.0)):
            return False
        form = os.environ.get('DLWP_TRAIN_STEP')
        if form not in (None, '', 'graph', 'lanes', 'branches', 'torch'):
            raise ValueError('DLWP_TRAIN_STEP=%r (graph, lanes, branches, torch)' % form)
        if not self._fold_ok():
            return 'torch' if (mode == '1' or form == 'torch') else False
        if form:
            return form
        store = self.plan._in_store
        return 'graph' if (mode == '1' or n_local * int(store[-1]) * int(store[-2]) <= self.graph_below) else 'lanes'

    # -- the step as a library object (dlwp_train_step_*) ------------------------------------------------------------------ #
    def _record_step(self, x, ys, n_global, scale, dp, form='graph'):
        """Runs ONE real step on (x, ys) while the library records its launches; returns the step entry, or {'refused': True} when
        the step contained device work the tape does not carry (it then stays eager).  Memory: every tensor the step allocates
        while recording comes from a private torch.cuda.MemPool that is kept with the entry, so the addresses the tape holds
        stay valid and nobody else is handed them.
        Two guards against a replay that silently misses a launch.  Structural: every stream-taking entry point of the
        library WITHOUT a tape record marks the tape foreign (DLWP_UNTAPED, csrc/common.h) and dlwp_train_step_create refuses it;
        torch-side launches set self._foreign.  By result (DLWP_TAPE_VALIDATE, default on): before the entry is cached the step
        runs twice more on a DIFFERENT batch from DIFFERENT weights (both derived from the recorded ones: nothing a missing
        launch leaves behind from the recorded step can pass for its result -- not even a weight preparation's) -- once launch by
        launch, once as the replay in the form it will run in (`form`) -- and parameters, optimizer slots, gradients and the loss
        table of the two must agree; the state the recorded step left is put back afterwards.  Two extra steps per recorded shape."""
        from . import _lib, ops
        opt = self.model.optimizer
        gx = x.clone()
        gys = [t.clone() for t in ys]
        if self._iter_dev is None:
            self._iter_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._lr_t = torch.zeros(1, dtype=torch.float32, device=self.device)
        if self.opt_state is None:
            raise RuntimeError('the optimizer slots must exist before the step is recorded')
        if dp is None and isinstance(opt, Adam):
            self._iter_dev.fill_(int(opt.iterations))
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = _lib.handle(dev)
        main = torch.cuda.current_stream(self.device)
        pool = torch.cuda.MemPool()
        self._foreign = False
        validate = os.environ.get('DLWP_TAPE_VALIDATE', '1') != '0'
        state0 = [t.clone() for t in [self.flat_params] + list(self.opt_state) + [self._iter_dev]] if validate else None
        def body(bx, bys):
            outs_, loss_, dys_ = self._forward_backward(bx, bys, scale)
            if dp is not None:       # the exchange and the update stay outside: a collective in between
                ops.axpby(loss_.view(-1), self._loss_tail.view(-1), scale, 0.0)
            elif isinstance(opt, Adam):
                m, v = self.opt_state
                ops.adam_keras_dev(self.flat_params, m, v, self.flat_grads, self._iter_dev, self._lr_t, opt.lr, opt.beta_1,
                                   opt.beta_2, opt.epsilon, opt.decay, 1.0)
            else:
                ops.sgd_keras(self.flat_params, self.opt_state[0], self.flat_grads, 0, opt.lr, opt.momentum, 0.0, 1.0)
            return outs_, loss_, dys_
        with torch.cuda.use_mem_pool(pool, device=self.device):
            _lib.check(_lib.lib.dlwp_train_step_record_begin(h, ctypes.c_void_p(main.cuda_stream)))
            try:
                outs, loss_vals, dys = body(gx, gys)
            except BaseException:
                _lib.lib.dlwp_train_step_record_abort(h)
                raise
            ins = [gx] + gys
            if self._foreign or len(ins) > 8:
                _lib.lib.dlwp_train_step_record_abort(h)
                return {'refused': True, 'loss': loss_vals}
            step = ctypes.c_void_p()
            dst = (ctypes.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
            floats = (ctypes.c_size_t * len(ins))(*[t.numel() for t in ins])
            rc = _lib.lib.dlwp_train_step_create(h, len(ins), dst, floats, ctypes.byref(step))
            if rc == _lib.EUNSUPPORTED:          # an entry point without a tape record ran inside the step: the shape stays eager
                return {'refused': True, 'loss': loss_vals, 'why': _lib.lib.dlwp_last_error().decode('utf-8', 'replace')}
            _lib.check(rc)
        keep = (outs, loss_vals, dys, dict(ops._workspaces), dict(ops._workspaces2),
                self.model.train_executor.scratch(int(x.shape[0])), self.model.train_executor.phase_buffers(),
                self._prep_cache.get(int(x.shape[0])), self._side, pool)
        ent = {'step': _StepHandle(step), 'x': gx, 'ys': gys, 'loss': loss_vals, 'keep': keep}
        if validate:
            why = self._validate_step(ent, state0, body, form, dp is not None)
            if why is not None:
                import warnings
                warnings.warn('dlwp_amd: the recorded training step did not reproduce the eager step (%s); batches of %d samples '
                              'keep running launch by launch' % (why, int(x.shape[0])), RuntimeWarning)
                ent['step'] = None                   # (its finaliser destroys the step object)
                return {'refused': True, 'loss': loss_vals, 'why': why}
        return ent

    def _validate_step(self, ent, state0, body, form, data_parallel=False):
        """The freshly recorded step against the launch-by-launch step on a batch and weights neither has seen (see _record_step);
        returns None when they agree, else what differed.  Leaves the state the recorded step left."""
        from . import _lib
        live = [self.flat_params] + list(self.opt_state) + [self._iter_dev]
        # (the loss table rides behind the gradients only in a data-parallel step: the recorded launches write it there)
        gbuf = self._flat_exchange if data_parallel else self.flat_grads
        after = [t.clone() for t in live] + [gbuf.clone(), ent['loss'].clone()]
        xv = ent['x'] * 0.75 + 0.125
        yvs = [t * 1.25 - 0.0625 for t in ent['ys']]

        def start():
            for t, t0 in zip(live, state0):
                t.copy_(t0)
            self.flat_params.mul_(1.0 + 2.0 ** -9)          # other weights: every prepared operand changes with them
        why = None
        try:
            start()
            _, loss_e, _ = body(xv, yvs)
            want = [t.clone() for t in live] + [gbuf.clone(), loss_e.clone()]
            start()
            ent['x'].copy_(xv)
            for d_, s_ in zip(ent['ys'], yvs):
                d_.copy_(s_)
            lanes_mode = _lib.STEP_LANES if os.environ.get('DLWP_TRAIN_LANES') == 'own' else _lib.STEP_LANES_RECORDED
            mode = {'lanes': lanes_mode, 'graph': _lib.STEP_GRAPH, 'branches': _lib.STEP_GRAPH_BRANCHES}.get(form, _lib.STEP_GRAPH)
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rc = _lib.lib.dlwp_train_step_launch(ent['step'].h, None, mode, stream)      # (the inputs are in the step's buffers)
            if rc != _lib.OK:
                why = 'replay failed: ' + _lib.lib.dlwp_last_error().decode('utf-8', 'replace')
            else:
                torch.cuda.synchronize(self.device)
                names = ['parameters'] + ['optimizer slot %d' % i for i in range(len(self.opt_state))] + \
                    ['step counter', 'gradients', 'loss table']
                for name, got, ref in zip(names, live + [gbuf, ent['loss']], want):
                    if got.dtype.is_floating_point:
                        if not bool(torch.isfinite(got).all()) and bool(torch.isfinite(ref).all()):
                            why = '%s are not finite' % name
                            break
                        if float((got - ref).abs().max()) > 1e-5 * float(ref.abs().max()) + 1e-30:
                            why = '%s differ by %.3g of %.3g' % (name, float((got - ref).abs().max()), float(ref.abs().max()))
                            break
                    elif not torch.equal(got, ref):
                        why = '%s differ' % name
                        break
        finally:
            for t, w in zip(live + [gbuf, ent['loss']], after):
                t.copy_(w)
        return why

    def _capture_step(self, x, ys, n_global, scale, dp):
        from . import ops
        opt = self.model.optimizer
        gx = x.clone()
        gys = [t.clone() for t in ys]
        if self._iter_dev is None:
            self._iter_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._lr_t = torch.zeros(1, dtype=torch.float32, device=self.device)
        if self.opt_state is None:
            raise RuntimeError('the optimizer slots must exist before the step is captured')
        import gc
        from ._lib import capture_lock
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        # No garbage collection while the stream is capturing: a collection cycle that finalises an old model's graphs, streams,
        # events or device memory (hipGraphExecDestroy, hipStreamDestroy, hipFree ...) in the middle of a capture aborts the
        # process or leaves a graph that crashes at launch (r3: seen as 'Fatal Python error: Aborted ... Garbage-collecting' inside
        # _capture_step and as segmentation faults in replay, 3 of 7 full test runs, once small steps were captured by default).
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._capture_locked(g, gx, gys, n_global, scale, dp, opt, capture_lock)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture_locked(self, g, gx, gys, n_global, scale, dp, opt, capture_lock):
        from . import ops
        x = gx
        with capture_lock, torch.cuda.graph(g, capture_error_mode='thread_local'):
            outs, loss_vals, dys = self._forward_backward(gx, gys, scale)
            if dp is not None:       # the exchange and the update stay outside: a collective in between
                ops.axpby(loss_vals.view(-1), self._loss_tail.view(-1), scale, 0.0)
            elif isinstance(opt, Adam):
                m, v = self.opt_state
                ops.adam_keras_dev(self.flat_params, m, v, self.flat_grads, self._iter_dev, self._lr_t, opt.lr, opt.beta_1,
                                   opt.beta_2, opt.epsilon, opt.decay, 1.0)
            else:
                ops.sgd_keras(self.flat_params, self.opt_state[0], self.flat_grads, 0, opt.lr, opt.momentum, 0.0, 1.0)
        # everything the captured launches point into must outlive the graph, whatever the caches do later
        keep = (outs, loss_vals, dys, dict(ops._workspaces), dict(ops._workspaces2),
                self.model.train_executor.scratch(int(x.shape[0])), self.model.train_executor.phase_buffers(),
                self._prep_cache.get(int(x.shape[0])), self._side)
        return {'graph': g, 'x': gx, 'ys': gys, 'loss': loss_vals, 'keep': keep}

    def _graph_step(self, x, ys, n_global, scale, dp):
        """Replays (capturing first, if needed) the step for this batch shape.  Returns the device loss table, or None when
        this shape has not been seen often enough yet (the caller then runs the step eagerly)."""
        from . import _lib
        _lib.drain_graveyard()          # a safe point: nothing is being recorded or replayed yet
        from . import ops
        opt = self.model.optimizer
        # the captured launches carry the optimizer's hyper-parameters as arguments: a changed rate (scheduler, callback,
        # load_model) must not replay the old ones
        hyper = tuple(float(getattr(opt, k)) for k in ('lr', 'decay', 'beta_1', 'beta_2', 'epsilon', 'momentum')
                      if hasattr(opt, k))
        form = self._graph_ok(int(x.shape[0]))
        key = (int(x.shape[0]), int(n_global), 0 if dp is None else dp.world, hyper, 'torch' if form == 'torch' else 'tape')
        if key in self._no_tape:
            return None
        ent = self._graphs.get(key)
        if ent is None:
            seen = self._graph_seen.get(key, 0) + 1
            self._graph_seen[key] = seen
            if seen <= self.graph_after:
                return None
            if len(self._graphs) >= 4:
                self._graphs.clear()
            if form == 'torch':
                ent = self._graphs[key] = self._capture_step(x, ys, n_global, scale, dp)
            else:
                ent = self._record_step(x, ys, n_global, scale, dp, form)     # (this IS a step: on x, ys)
                if ent.get('refused'):
                    self._no_tape.add(key)
                else:
                    self._graphs[key] = ent
                if dp is None:
                    opt.iterations += 1
                    self._iter_shadow = opt.iterations
                return ent['loss']
        if 'step' in ent:
            from . import _lib
            if dp is None and isinstance(opt, Adam) and self._iter_shadow != opt.iterations:
                self._iter_dev.fill_(int(opt.iterations))          # (after load_model / a manual change / eager steps)
            srcs = [x] + list(ys)
            if not all(s_.is_contiguous() and s_.dtype == torch.float32 and s_.numel() == d_.numel()
                       for s_, d_ in zip(srcs, [ent['x']] + ent['ys'])):
                srcs = [s_.to(torch.float32).contiguous() for s_ in srcs]
            ptrs = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
            # ('lanes': on the trainer's own side streams, the ones the step was recorded on -- they are kept in ent['keep'] --: streams
            #  the library creates later may share a hardware queue with the main stream; DLWP_TRAIN_LANES=own picks those)
            lanes_mode = _lib.STEP_LANES if os.environ.get('DLWP_TRAIN_LANES') == 'own' else _lib.STEP_LANES_RECORDED
            mode = {'lanes': lanes_mode, 'graph': _lib.STEP_GRAPH, 'branches': _lib.STEP_GRAPH_BRANCHES}[form]
            _lib.check(_lib.lib.dlwp_train_step_launch(ent['step'].h, ptrs, mode,
                                                       ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            if dp is None:
                opt.iterations += 1
                self._iter_shadow = opt.iterations
            return ent['loss']
        pairs = [(x, ent['x'])] + list(zip(ys, ent['ys']))
        if len(pairs) <= 8 and all(s.is_contiguous() and s.dtype == torch.float32 and s.numel() == d.numel() for s, d in pairs):
            ops.copy_many(pairs)            # the batch and its targets into the graph's buffers: one launch
        else:
            for s, d in pairs:
                d.copy_(s.reshape(d.shape))
        if dp is None and isinstance(opt, Adam) and self._iter_shadow != opt.iterations:
            self._iter_dev.fill_(int(opt.iterations))          # (after load_model / a manual change / eager steps)
        ent['graph'].replay()
        if dp is None:
            opt.iterations += 1
            self._iter_shadow = opt.iterations
        return ent['loss']

    def train_on_shard(self, x, y, n_global, return_device=False):
        """One optimisation step given THIS RANK's rows of a global batch of n_global samples (all of them when not data
        parallel).  Collective under data parallelism: every rank must call it, also with zero rows."""
        from . import ops
        x = self._to_device(x)
        n_local = int(x.shape[0])
        dp = self.dp if (self.dp is not None and self.dp.world > 1) else None
        if dp is None and n_local != int(n_global):
            raise ValueError('%d rows given for a batch of %d without data parallelism' % (n_local, n_global))
        if dp is not None and self._params_dirty:
            self.sync_parameters()
        if n_local > 0 and self._graph_ok(n_local):
            scale_g = 1.0 if dp is None else n_local * dp.world / float(n_global)
            x = x.reshape((n_local,) + tuple(x.shape[1:]))
            loss_vals = self._graph_step(x, self._targets(y, n_local), n_global, scale_g, dp)
            if loss_vals is not None:
                if dp is not None:
                    self._exchange_and_apply(dp)
                    loss_vals = self._loss_tail
                    ops.axpby(loss_vals.view(-1), loss_vals.view(-1), 0.0, 1.0 / dp.world)
                if return_device:
                    return loss_vals, 0.0
                return self._report(loss_vals, 0.0)
        # local means are averaged over ranks: weight each by its share so ragged shards stay exact
        scale = 1.0 if dp is None else n_local * dp.world / float(n_global)
        if n_local > 0:
            ys = self._targets(y, n_local)
            outs, loss_vals, dys = self._forward_backward(x, ys, scale)
            self._add_regularizer_gradients()
        else:                               # a rank without rows still takes part in the exchange: no data gradient, but
            self._flat_exchange.zero_()     # its share of the regularisers' (every rank adds it in full, the sum is / world)
            self._add_regularizer_gradients()
            loss_vals = None
        reg = self._regularizer_loss()      # Keras reports the penalty at the weights the step started from
        if dp is not None:
            if loss_vals is not None:
                ops.axpby(loss_vals.view(-1), self._loss_tail.view(-1), scale, 0.0)
            self._exchange_and_apply(dp)
            loss_vals = self._loss_tail
            ops.axpby(loss_vals.view(-1), loss_vals.view(-1), 0.0, 1.0 / dp.world)
        else:
            self._apply(1.0)
        if return_device:
            return loss_vals, reg
        return self._report(loss_vals, reg)

    def test_on_batch(self, x, y):
        x = self._to_device(x)
        ys = self._targets(y, x.shape[0])
        _, loss_vals, _ = self._forward_loss(x, ys, False)
        return self._report(loss_vals)

    # -- loops --------------------------------------------------------------------------------------------------------- #
    def _callbacks(self, callbacks, params):
        hist = History()
        cbs = [hist] + [c for c in (callbacks or [])]
        for c in cbs:
            if hasattr(c, 'set_model'):
                c.set_model(self.model)
            else:
                c.model = self.model
            if hasattr(c, 'set_params'):
                c.set_params(params)
        self.model.history = hist
        return hist, cbs

    @staticmethod
    def _call(cbs, name, *args):
        for c in cbs:
            fn = getattr(c, name, None)
            if fn is not None:
                fn(*args)

    def _run_epochs(self, epochs, initial_epoch, batches_fn, n_batches_fn, validate_fn, callbacks, verbose, on_epoch_end):
        params = {'epochs': epochs, 'metrics': self.metrics_names, 'verbose': verbose}
        hist, cbs = self._callbacks(callbacks, params)
        self.model.stop_training = False
        self._call(cbs, 'on_train_begin', {})
        for epoch in range(initial_epoch, epochs):
            t0 = time.time()
            self._call(cbs, 'on_epoch_begin', epoch, {})
            sums = None
            seen = 0
            pending = []
            # The epoch's logs are batch-size-weighted means of the per-batch values, and for the per-element losses those are
            # LINEAR in the step's loss table: the tables are summed on the device, weighted, by one small launch per step (r4: a
            # clone per step was a device-to-device memcpy between two graph launches -- 4 us and ~15 us of queue bubbles around it,
            # profiles/r4_loader_timeline.txt) and read once per epoch.  Whole-batch statistics (custom losses) keep their tables.
            acc, reg_sum = None, 0.0
            for bi, (X, y, n_glob) in enumerate(batches_fn(epoch)):      # X, y: this rank's rows of a batch of n_glob
                self._call(cbs, 'on_batch_begin', bi, {'batch': bi, 'size': n_glob})
                lv, reg = self.train_on_shard(X, y, n_glob, return_device=True)
                if self.loss_kind != 'custom' and lv.is_cuda and lv.dtype == torch.float32 and lv.is_contiguous():
                    from . import ops
                    if acc is None:
                        acc = torch.zeros_like(lv)
                    ops.axpby(lv.view(-1), acc.view(-1), float(n_glob), 1.0)
                    reg_sum += float(n_glob) * float(reg)
                    seen += n_glob
                else:
                    pending.append((lv.clone(), n_glob, reg))
                # convert lazily: one host sync per epoch unless a callback wants per-batch logs
                if any(getattr(type(c), 'on_batch_end', Callback.on_batch_end) is not Callback.on_batch_end
                       for c in cbs if isinstance(c, Callback)) or any(not isinstance(c, Callback) for c in cbs):
                    vals = self._report(lv, reg)
                    self._call(cbs, 'on_batch_end', bi, dict(zip(self.metrics_names, vals), batch=bi, size=n_glob))
                if self.model.stop_training:
                    break
            if pending:                  # one device-to-host copy for the whole epoch's loss tables
                tables = torch.stack([lv for lv, _, _ in pending]).cpu().numpy().astype(np.float64)
                for tab, (_, bs, reg) in zip(tables, pending):
                    vals = np.asarray(self._report_from(tab, reg), dtype=np.float64)
                    sums = vals * bs if sums is None else sums + vals * bs
                    seen += bs
            if acc is not None:          # sum over the batches of size x table -> size-weighted sums of the reported values
                vals = np.asarray(self._report_from(acc.cpu().numpy().astype(np.float64), 0.0), dtype=np.float64)
                vals[0] += reg_sum
                sums = vals if sums is None else sums + vals
            logs = dict(zip(self.metrics_names, (sums / max(seen, 1)).tolist())) if sums is not None else {}
            if self.dp is not None and getattr(self.dp, '_xchg', None) is not None:
                self.dp.oneshot_check()         # (the epoch's tables are on the host: the device has been synchronised anyway)
            if validate_fn is not None:
                vvals = validate_fn()
                logs.update({'val_' + k: v for k, v in zip(self.metrics_names, vvals)})
            if on_epoch_end is not None:
                on_epoch_end()
            if verbose:
                msg = ' - '.join('%s: %.4f' % (k, v) for k, v in logs.items())
                print('Epoch %d/%d - %.1fs - %s' % (epoch + 1, epochs, time.time() - t0, msg))
            self._call(cbs, 'on_epoch_end', epoch, logs)
            if self.dp is not None and self.dp.world > 1 and getattr(self.dp, 'mirror', False):
                # driver mode: callbacks (EarlyStopping ...) run on rank 0 only -- every rank takes its decision
                self.model.stop_training = bool(self.dp.broadcast_indices(np.array([int(bool(self.model.stop_training))]))[0])
            if self.model.stop_training:
                break
        self._call(cbs, 'on_train_end', {})
        return hist

    #: fit(x, y) keeps the whole training set in HBM when it takes at most this share of the free device memory (288 GB
    #: per MI355X: the reference's multi-year 2-degree sets are tens of GB); 0 disables (host gather + upload per batch)
    resident_fraction = 0.5

    def _make_resident(self, x, ys):
        """Upload numpy training arrays once (float32) if they fit; returns (x, ys, resident)."""
        if self.device.type != 'cuda' or self.resident_fraction <= 0:
            return x, ys, False
        arrays = [x] + list(ys)
        if all(isinstance(a, torch.Tensor) and a.is_cuda for a in arrays):
            return x, ys, True
        if any(isinstance(a, torch.Tensor) for a in arrays):
            return x, ys, False
        arrays = [np.asarray(a) for a in arrays]
        need = sum(int(a.size) * 4 for a in arrays)
        free, _ = torch.cuda.mem_get_info(self.device)
        if need > self.resident_fraction * free:
            return x, ys, False
        dev = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device) for a in arrays]
        return dev[0], dev[1:], True

    def fit(self, x, y, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None, shuffle=True,
            initial_epoch=0):
        batch_size = int(batch_size or 32)
        x = np.asarray(x) if not isinstance(x, torch.Tensor) else x
        n = x.shape[0]
        ys = list(y) if isinstance(y, (list, tuple)) else [y]
        x, ys, resident = self._make_resident(x, ys)

        dp = self.dp if (self.dp is not None and self.dp.world > 1) else None

        def batches(epoch):
            idx = np.arange(n)
            if shuffle:
                np.random.shuffle(idx)
            if dp is not None:             # one shuffle for all replicas: rank 0's
                idx = dp.broadcast_indices(idx)
            for lo in range(0, n, batch_size):
                sel = idx[lo:lo + batch_size]
                n_glob = len(sel)
                if dp is not None:         # this rank's rows of the global batch; nothing else is gathered or uploaded
                    a, b = dp.shard(n_glob)
                    sel = sel[a:b]
                if resident:           # the batch is gathered in HBM: no host copy, no PCIe transfer per step
                    sel = torch.from_numpy(np.ascontiguousarray(sel)).to(self.device)
                    yield x.index_select(0, sel), ([t.index_select(0, sel) for t in ys] if len(ys) > 1
                                                   else ys[0].index_select(0, sel)), n_glob
                else:
                    yield x[sel], ([t[sel] for t in ys] if len(ys) > 1 else ys[0][sel]), n_glob

        val = None
        if validation_data is not None:
            vx, vy = validation_data[0], validation_data[1]
            if not isinstance(vx, torch.Tensor):       # uploaded once, evaluated after every epoch
                vys = list(vy) if isinstance(vy, (list, tuple)) else [vy]
                vx, vys, _ = self._make_resident(np.asarray(vx), vys)
                vy = vys if isinstance(vy, (list, tuple)) else vys[0]
            val = lambda: self.evaluate(vx, vy, batch_size=batch_size, verbose=0, as_list=True)  # noqa: E731
        return self._run_epochs(epochs, initial_epoch, batches, None, val, callbacks, verbose, None)

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None, shuffle=True, initial_epoch=0):
        from .model.generators import DeviceLoader
        steps = int(steps_per_epoch) if steps_per_epoch is not None else len(generator)

        dp = self.dp if (self.dp is not None and self.dp.world > 1) else None
        shard = None if dp is None else (dp.rank, dp.world)

        loader = [None]          # one loader for the whole call: its pinned / device staging buffers serve every epoch

        def batches(epoch):
            order = list(range(steps))
            if dp is not None and hasattr(generator, '_indices'):
                # the generators shuffle with the process-global numpy RandomState (reference generators.py:103-106):
                # replicas must cut THE SAME batch i, so every rank takes rank 0's index list for this epoch
                generator._indices = dp.broadcast_indices(generator._indices)
            if self.device.type == 'cuda' or shard is not None:
                if loader[0] is None:
                    # (its copy stream on a hardware queue of its own: not the main stream's, not a weight-gradient stream's)
                    avoid = None
                    if self.device.type == 'cuda':
                        avoid = [torch.cuda.current_stream(self.device)] + list(self._ensure_side_streams() if self._fold_ok() else [])
                    loader[0] = DeviceLoader(generator, self.device, order=order, shard=shard, avoid_streams=avoid)
                for X, y, n_glob in loader[0].iter_batches(order):
                    yield X, y, n_glob
            else:
                for i in order:
                    X, y = generator[i]
                    yield X, y, int(X.shape[0])

        def end():
            if hasattr(generator, 'on_epoch_end'):
                generator.on_epoch_end()

        val = None
        if validation_data is not None:
            if isinstance(validation_data, (tuple, list)):
                vx, vy = validation_data[0], validation_data[1]
                val = lambda: self.evaluate(vx, vy, verbose=0, as_list=True)  # noqa: E731
            else:
                vgen = validation_data
                vsteps = int(validation_steps) if validation_steps is not None else len(vgen)
                val = lambda: self.evaluate_generator(vgen, vsteps)  # noqa: E731
        self._loader_threads += 1
        try:
            return self._run_epochs(epochs, initial_epoch, batches, None, val, callbacks, verbose, end)
        finally:
            self._loader_threads -= 1

    def evaluate_generator(self, generator, steps=None):
        steps = int(steps) if steps is not None else len(generator)
        sums, seen = None, 0
        for i in range(steps):
            X, y = generator[i]
            vals = np.asarray(self.test_on_batch(X, y), dtype=np.float64)
            bs = int(np.asarray(X).shape[0]) if not isinstance(X, torch.Tensor) else int(X.shape[0])
            sums = vals * bs if sums is None else sums + vals * bs
            seen += bs
        return (sums / max(seen, 1)).tolist()

    def evaluate(self, x, y, batch_size=None, verbose=1, as_list=False):
        batch_size = int(batch_size or 32)
        if self.loss_kind != 'custom':           # per-element means do not depend on it; larger chunks fill the GPU.
            batch_size = max(batch_size, 256)    # The anomaly-correlation loss is a whole-batch statistic: keep Keras' batches
        n = x.shape[0]
        ys = list(y) if isinstance(y, (list, tuple)) else [y]
        sums, seen = None, 0
        for lo in range(0, n, batch_size):
            yb = [t[lo:lo + batch_size] for t in ys]
            vals = np.asarray(self.test_on_batch(x[lo:lo + batch_size], yb if len(yb) > 1 else yb[0]), dtype=np.float64)
            bs = min(batch_size, n - lo)
            sums = vals * bs if sums is None else sums + vals * bs
            seen += bs
        out = (sums / max(seen, 1)).tolist()
        return out if (len(out) > 1 or as_list) else out[0]
