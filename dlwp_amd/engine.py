"""
Minimal Keras-shaped model containers (Sequential, functional Model) over the HIP back end: exactly the protocol the
reference's wrappers use: compile / fit / fit_generator / predict / evaluate / get_weights /
set_weights / reset_states / summary / layers / outputs / stop_training / save.

Forward execution = dlwp_amd.plan.Plan run by the Executor below: eager launches for predict, one captured hipGraph for
the autoregressive rollout.  torch is plumbing only (device buffers, streams, H2D/D2H copies).
"""
import ctypes
import os

import numpy as np
import torch

from . import layers as L
from . import plan as P
from .util import host_result_buffer


def default_device():
    if torch.cuda.is_available():
        idx = int(os.environ.get('LOCAL_RANK', torch.cuda.current_device()))
        if idx >= torch.cuda.device_count() and os.environ.get('DLWP_SHARE_GPUS') == '1':
            idx %= torch.cuda.device_count()          # testing only, see parallel.init
        return torch.device('cuda', idx)
    return torch.device('cpu')     # weights can be held for planning / inspection; every compute call will raise


class Executor(object):
    """Runs a Plan on one device for a given batch size.  Buffers are cached per batch size."""

    def __init__(self, plan, device, activation_dtype='float32'):
        self.plan, self.device = plan, device
        self._bufs = {}
        self._descs = None
        self._pool2 = None      # _pooled_too(): convolutions of a training forward that store their pooled image as well
        self._bf16 = set(plan.bf16_buffers()) if activation_dtype == 'bfloat16' else set()
        self._phase = None       # derived (phase-summed) kernels of plan.phase_params: [(w2, b2 | None)]
        self._oct = self._octet_buffers() if self._bf16 and os.environ.get('DLWP_BF16_O8', '1') != '0' else set()

    # -- buffers ----------------------------------------------------------------------------------------------------- #
    def scratch(self, n):
        b = self._bufs.get(n)
        if b is None:
            if len(self._bufs) > 4:
                self._bufs.clear()
            b = [torch.empty((n,) + s, dtype=torch.bfloat16 if i in self._bf16 else torch.float32, device=self.device)
                 for i, s in enumerate(self.plan.buffers)]
            self._bufs[n] = b
        return b

    def alloc_outputs(self, n):
        return [torch.empty((n,) + s, dtype=torch.float32, device=self.device) for s in self.plan.output_store]

    def phase_buffers(self):
        """Device tensors of the derived kernels (plan.phase_params), allocated once; filled by the 'phasew' ops."""
        if self._phase is None:
            self._phase = [(torch.empty(p['w2_shape'], dtype=torch.float32, device=self.device),
                            torch.empty(p['w2_shape'][3], dtype=torch.float32, device=self.device) if p['bias'] else None)
                           for p in self.plan.phase_params]
        return self._phase

    def conv_weights(self, op):
        """(kernel, bias) tensors a conv launch multiplies with: the layer's, or the derived ones of a restated layer."""
        if op.wparam is not None:
            return self.phase_buffers()[op.wparam]
        return op.layer.kernel, op.layer.bias

    def _descriptors(self):
        from . import ops
        if self._descs is None:
            descs = []
            for op in self.plan.ops:
                if op.kind in ('conv', 'rowconv'):
                    f, (kh, kw), dil = op.conv_geometry
                    descs.append(ops.make_conv(f, kh, kw, dil, ops.make_pad(*op.halo), op.act,
                                               op.in_c_off, op.in_c_total, op.out_c_off, op.out_c_total, op.src_mode,
                                               op.out_pool, op.out_d2s, op.lstm_f, op.rec_act))
                elif op.kind == 'pad':
                    descs.append(ops.make_pad(*op.halo))
                else:
                    descs.append(None)
            self._descs = descs
        return self._descs

    def _descriptor2(self, op):
        """descriptor of the INPUT convolution of a whole-step op (op.src2, plan.PlanOp)"""
        from . import ops
        s2 = op.src2
        lay = s2['layer']
        return ops.make_conv(4 * op.lstm_f, lay.kernel_size[0], lay.kernel_size[1], tuple(lay.dilation_rate),
                             ops.make_pad(*s2['halo']), 0, s2['in_c_off'], s2['in_c_total'], 0, 4 * op.lstm_f)

    def _conv_dtype(self, op, octets=None):
        """dtype code of a convolution launch: storage of its input / output buffers; in bfloat16 mode a float32-stored
        input (the model state) feeding a bf16-stored output may be rounded to bf16 by the kernel (DLWP_COMPUTE_BF16).
        Buffers of self._oct are stored in channel octets (DLWP_BF16_O8, include/dlwp_hip.h)."""
        from . import _lib
        octets = self._oct if octets is None else octets
        in16, out16 = op.src in self._bf16, op.dst in self._bf16
        code_in = _lib.BF16_O8 if op.src in octets else (_lib.BF16 if in16 else _lib.F32)
        code_out = _lib.BF16_O8 if op.dst in octets else (_lib.BF16 if out16 else _lib.F32)
        return _lib.dtype_io(code_in, code_out, compute_bf16=(out16 and not in16))

    def _octet_buffers(self):
        """The bfloat16 scratch buffers kept in channel OCTETS, (n, C/8, h, w, 8): a pixel's 8 consecutive channels are 16
        contiguous bytes -- the unit the bf16 matrix-core kernels stage and multiply, so their loaders and epilogues move
        16 / 8 bytes per lane in 256-byte runs instead of 2-byte elements of 8 channel planes (csrc/conv_fwd_bf16_kernel.h).
        A buffer qualifies when its channels (and every channel window on it) are whole octets and EVERY op that touches it is
        a convolution the library runs in that layout (dlwp_conv2d_supports_dtype); a ConvLSTM2D step with the cell update in
        its convolution's epilogue then keeps its float32 cell state in octets too, which only those convolutions touch.
        Model inputs / outputs stay NCHW.  DLWP_BF16_O8=0 switches the layout off."""
        from . import _lib
        plan = self.plan
        descs = self._descriptors()
        h = _lib.handle_or_none()
        cand = {b for b in self._bf16 if plan.buffers[b][0] % 8 == 0}
        fused = [op for op in plan.ops if op.kind == 'conv' and op.lstm_f]
        cells = {b for op in fused for b in op.aux[1:] if b is not None}

        def buffers_of(op):
            extra = [b for b in (op.aux or ()) if isinstance(b, int)] if op.kind in ('conv', 'lstm') else []
            return [op.src, op.dst] + extra
        cells_private = all((op.kind == 'conv' and op.lstm_f) or not (set(buffers_of(op)) & cells) for op in plan.ops)

        def settle():
            changed = True
            while changed and cand:
                changed = False
                for op, d in zip(plan.ops, descs):
                    mine = [b for b in buffers_of(op) if b in cand]
                    if not mine:
                        continue
                    ok = op.kind == 'conv'
                    if ok and op.src in cand:
                        tot = op.in_c_total if op.in_c_total else plan.buffers[op.src][0]
                        ok = op.xs[0] % 8 == 0 and op.in_c_off % 8 == 0 and tot % 8 == 0
                    if ok and op.dst in cand:
                        tot = op.out_c_total if op.out_c_total else plan.buffers[op.dst][0]
                        ok = not op.out_d2s and op.out_c_off % 8 == 0 and tot % 8 == 0
                    if ok and op.lstm_f:       # z_add and h share the launch's output layout; the cell state follows it
                        za = op.aux[0]
                        ok = cells_private and op.lstm_f % 8 == 0 and (za is None or (za in cand) == (op.dst in cand))
                    if ok and op.src2 is not None:     # a whole step per launch: h in octets on both sides, or not at all
                        ok = op.src in cand and op.dst in cand and bool(_lib.lib.dlwp_convlstm_step_supported(
                            h, _lib.Shape4(1, *[int(v) for v in op.xs]), ctypes.byref(d),
                            _lib.Shape4(1, *[int(v) for v in op.src2['xs']]), ctypes.byref(self._descriptor2(op)),
                            _lib.dtype_io(_lib.BF16_O8, _lib.BF16_O8)))
                    elif ok:
                        ok = bool(_lib.lib.dlwp_conv2d_supports_dtype(h, _lib.Shape4(1, *[int(v) for v in op.xs]),
                                                                      ctypes.byref(d), int(self._conv_dtype(op, cand))))
                    if not ok:
                        for b in mine:
                            cand.discard(b)
                        changed = True
        settle()
        # the fused steps of a ConvLSTM2D share one cell state: all of them write octets, or none does
        if fused and len({op.dst in cand for op in fused}) > 1:
            for op in fused:
                cand.discard(op.dst)
                if op.aux[0] is not None:
                    cand.discard(op.aux[0])
            settle()
        return cand

    def bf16_weight_layers(self, n=1):
        """The Conv2D layers this executor multiplies with bf16-rounded weights: the ones whose input buffer is stored as
        bfloat16 and whose geometry the bf16 matrix-core kernels cover (include/dlwp_hip.h:
        dlwp_conv2d_uses_bf16_weights).  Empty with float32 activation storage."""
        from . import _lib, ops
        out = []
        for op, d in zip(self.plan.ops, self._descriptors()):
            if op.kind == 'conv' and ops.uses_bf16_weights((n,) + tuple(op.xs), d, self._conv_dtype(op)):
                out.append(op.layer)
                if op.src2 is not None:          # a whole ConvLSTM2D step multiplies both kernels on the bf16 matrix cores
                    out.append(op.src2['layer'])
        return out

    # -- eager forward ----------------------------------------------------------------------------------------------- #
    def _pooled_too(self):
        """{conv op index: index of the 'maxpool' op that reads its whole float32 output} -- candidates for dlwp_conv2d_fwd_pool2 in
        the training forward (the executor of a training step passes `prepared` / `skip_ops`; plain run() calls keep the two
        launches).  DLWP_CONV_POOL2=0 switches it off."""
        if self._pool2 is None:
            found = {}
            if os.environ.get('DLWP_CONV_POOL2', '1') != '0' and not self._bf16:
                ops_ = self.plan.ops
                for k, op in enumerate(ops_):
                    if op.kind != 'conv' or op.lstm_f or op.out_pool or op.out_d2s or op.src2 is not None or op.dst < 0:
                        continue
                    readers = [j for j, r in enumerate(ops_) if r.src == op.dst]
                    writers = [j for j, w in enumerate(ops_) if w.dst == op.dst]
                    pools = [j for j in readers if ops_[j].kind == 'maxpool' and j > k]
                    if len(writers) == 1 and len(pools) == 1 and op.out_c_off == 0 and ops_[pools[0]].dst >= 0 and \
                            op.conv_geometry[0] == self.plan.buffers[op.dst][0]:
                        found[k] = pools[0]
            self._pool2 = found
        return self._pool2

    def run(self, x, outs=None, prepared=None, skip_phasew=False, skip_ops=()):
        """x: device tensor (n, ...) matching the model input; returns the list of output tensors (stored layout).
        prepared: {op index: tensor of ops.conv2d_prepare} -- those convolutions do not transform their weights again;
        skip_phasew: the derived kernels (plan.phase_params) are already up to date (the training step builds them, and every
        prepared form, in front of the forward); skip_ops: op indices left out (the training step takes its loss on the phase
        channels of a restated output layer: no depth-to-space pass, the output tensor stays unwritten)."""
        from . import ops
        n = x.shape[0]
        x = x.reshape((n,) + self.plan._in_store)
        bufs = self.scratch(n)
        outs = outs if outs is not None else self.alloc_outputs(n)

        def res(i):
            if i >= 0:
                return bufs[i]
            if i == P.STATE_IN:
                return x
            return outs[-2 - i]
        pooled_too = self._pooled_too() if prepared is not None or skip_ops else {}
        done = set()
        for k, (op, d) in enumerate(zip(self.plan.ops, self._descriptors())):
            if (op.kind == 'phasew' and skip_phasew) or k in skip_ops or k in done:
                continue
            src, dst = res(op.src), res(op.dst)
            if k in pooled_too:
                # training forward: the layer's output AND its MaxPooling2D(2) image from one launch (the pooling op is skipped)
                kp = pooled_too[k]
                kern, bias = self.conv_weights(op)
                if ops.conv2d(src, kern, bias, d, out=dst, x_channels=op.xs[0], prepared=prepared.get(k) if prepared else None,
                              out_pool2=res(self.plan.ops[kp].dst)) is not None:
                    done.add(kp)
                    continue
            if op.kind == 'conv' and op.src2 is not None:          # a whole ConvLSTM2D step (dlwp_convlstm_step_fwd)
                if op.dst not in self._oct:
                    raise RuntimeError('whole-step ConvLSTM2D launches need the h sequence in octets (Model.set_activation_dtype '
                                       're-plans without them otherwise)')
                kern, bias = self.conv_weights(op)
                za, cp, co = op.aux
                ops.convlstm_step(dst, res(op.src2['buf']), kern, op.src2['layer'].kernel, op.src2['layer'].bias, d,
                                  self._descriptor2(op), res(cp), res(co), op.src2['xs'][0])
            elif op.kind == 'conv' and op.lstm_f:
                kern, bias = self.conv_weights(op)
                za, cp, co = op.aux
                ops.convlstm_conv(src, kern, bias, d, dst, res(co), z_add=res(za) if za is not None else None,
                                  c_prev=res(cp) if cp is not None else None, x_channels=op.xs[0],
                                  compute_bf16=(op.dst in self._bf16 and op.src not in self._bf16),
                                  in_o8=op.src in self._oct, out_o8=op.dst in self._oct)
            elif op.kind == 'conv':
                kern, bias = self.conv_weights(op)
                ops.conv2d(src, kern, bias, d, out=dst, x_channels=op.xs[0],
                           compute_bf16=(op.dst in self._bf16 and op.src not in self._bf16),
                           prepared=prepared.get(k) if prepared else None,
                           in_o8=op.src in self._oct, out_o8=op.dst in self._oct)
            elif op.kind == 'rowconv':
                ops.rowconv2d(src, op.layer.kernel, op.layer.bias, d, out=dst, x_channels=op.xs[0])
            elif op.kind == 'phasew':
                w2, b2 = self.phase_buffers()[op.wparam]
                ops.phase_weights(op.layer.kernel, op.layer.bias, op.halo.top, op.halo.left, w2=w2, b2=b2)
                continue
            elif op.kind == 'd2s':
                ops.depth_to_space2(src, op.xs[0], out=dst, c_off=op.out_c_off)
            elif op.kind == 'pad':
                if op.inner > 1:
                    hh, ww = op.xs[1], op.xs[2]
                    ops.pad2d(src.reshape(n, hh, ww, op.inner), d, channels_last=True, out=dst)
                else:
                    ops.pad2d(src, d, out=dst)
            elif op.kind == 'maxpool':
                ops.maxpool2(src, out=dst)
            elif op.kind == 'upsample':
                ops.upsample2(src, out=dst)
            elif op.kind == 'copy':
                ops.copy_channels(src, dst, op.xs[0], op.in_c_off, op.out_c_off)
            elif op.kind == 'lstm':
                zh, cp, co = op.aux
                ops.convlstm_gates(src, res(zh) if zh is not None else None, res(cp) if cp is not None else None,
                                   res(co), dst, op.xs[0], h_c_off=op.out_c_off, act=op.act, rec_act=op.rec_act)
            else:
                raise RuntimeError(op.kind)
        return outs

    # -- hipGraph rollout -------------------------------------------------------------------------------------------- #
    @staticmethod
    def member_groups(n, pixels=15840):
        """How many parallel member chains a rollout of n members is captured as (dlwp_rollout_create_grouped).  Members are
        independent, so chains at different layers fill the gaps each other's launches leave (partly filled last rounds of
        workgroups, the drain at every kernel boundary).  Measured on one MI355X (profiles/r2i_rollout_member_groups.txt,
        r2q): two chains +7 % at 64 members (361.8 -> 387.4 k steps/s), +4 % at 32 members of config 5, +1.3 % at 256
        (400.6 -> 405.7 k); four or eight chains lose again (389 k at 256: the launches get too small); within noise or
        worse below 32 members of the 88 x 180 grid.  Default: TWO chains from members x grid points >= 0.5 M on (32 members of
        that grid, 8 of the 1-degree grid), one below; DLWP_ROLLOUT_GROUPS=g asks for g.  r4: between 0.2 M and 0.5 M points
        make_rollout measures one chain against two (whether two help there depends on how the member count fills rounds of workgroups)."""
        # what decides is the work per launch, members x grid points: 32 members of the 88 x 180 grid = 0.5 M points; the 1-degree
        # recurrent stack (config 4, 180 x 360) gains from 8 members on (r2y: 44.2 -> 45.8 k steps/s at 8, 48.2 -> 50.5 k at 16,
        # four chains 44.4 k), 4 members of config 5 (0.26 M points) do not
        env = os.environ.get('DLWP_ROLLOUT_GROUPS')
        g = int(env) if env and env != 'split' else (2 if n * int(pixels) >= 500000 else 1)
        g = max(1, min(g, max(n, 1)))
        while n % g:
            g -= 1
        return g

    #: work per launch (members x grid points) between which make_rollout MEASURES one chain against two (below: launches of a
    #: few tiles, two chains only add launches; above: the rule of member_groups, two chains)
    tune_groups_between = (200000, 500000)

    def make_rollout(self, state0, series, calls, groups=None):
        """Capture `calls` model applications.  state0: (n,)+input store; series: (calls*n_out, n)+store, contiguous.
        groups=None: member_groups' rule -- and, in the range where the better choice follows from how the member count happens
        to fill rounds of workgroups (r4, gpurun s23: config 5 at 4 members 49.7 k steps/s as one chain, 54.5 k as two; at 3 members
        49.9 / 49.9, at 6 members 61.9 / 64.5; 16 members of the 88 x 180 grid 235.8 / 238.8 k), by MEASUREMENT: one chain, two single-chain graphs on two
        probed streams (SplitRollout) and the forked graph are captured in turn, each is launched four times on the (zeroed) state; the
        first form that beats one chain by 2 % is kept.  Only where one chain and two run every
        convolution in the same split regime (ops.conv_split_count): the choice never changes a bit of the result."""
        if groups is None and os.environ.get('DLWP_ROLLOUT_GROUPS') == 'split' and int(state0.shape[0]) % 2 == 0:
            groups = 'split'
        if groups == 'split':
            return self._make_split_rollout(state0, series, calls)
        if groups is None and os.environ.get('DLWP_ROLLOUT_GROUPS') is None and os.environ.get('DLWP_ROLLOUT_TUNE', '1') != '0':
            n = int(state0.shape[0])
            work = n * int(self.plan._in_store[1]) * int(self.plan._in_store[2])
            if n >= 2 and n % 2 == 0 and self.tune_groups_between[0] <= work < self.tune_groups_between[1] and \
                    self.device.type == 'cuda' and self._same_split_regime(n, n // 2):
                best = None
                state0.zero_()
                # (two chains are built up to three times: a graph's branches run on streams the runtime creates when the graph is
                #  instantiated, and whether those land on different HARDWARE QUEUES depends on every stream the process has created
                #  before -- branches on one queue run one after the other.  A new instantiation draws new streams.)
                for g in (1, 'split', 2, 2):
                    if best is not None and best[2] != 1:
                        break
                    cand = self._make_split_rollout(state0, series, calls) if g == 'split' else \
                        self._make_rollout(state0, series, calls, g)
                    cand.launch()
                    torch.cuda.synchronize(self.device)
                    ts = []
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        cand.launch()
                        e1.record()
                        e1.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    t = min(ts)
                    if best is None or t < 0.98 * best[0]:      # (two chains must win by 2 %: they cost a second set of buffers)
                        if best is not None:
                            best[1].close()
                        best = (t, cand, g)
                    else:
                        cand.close()
                best[1].groups = best[2]
                return best[1]
        return self._make_rollout(state0, series, calls, groups)

    def _make_split_rollout(self, state0, series, calls):
        from .util import distinct_streams
        st = self.__dict__.get('_chain_streams')
        if st is None:
            st = self._chain_streams = distinct_streams(self.device, 2, [torch.cuda.current_stream(self.device)])
        return SplitRollout([self._make_rollout(state0, series, calls, 1, chain=(c, 2)) for c in range(2)], st, self.device)

    def _same_split_regime(self, n_a, n_b):
        """Do the plan's convolutions run in the same split-K regime (equal bits) at n_a and at n_b members per launch?"""
        from . import ops
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        for op, d in zip(self.plan.ops, self._descriptors()):
            if op.kind == 'conv' and not op.lstm_f and op.src2 is None:
                dt = self._conv_dtype(op)
                if ops.conv_split_count((n_a,) + tuple(op.xs), d, dt, dev) != ops.conv_split_count((n_b,) + tuple(op.xs), d, dt, dev):
                    return False
        return True

    def _make_rollout(self, state0, series, calls, groups=None, chain=None, span=None, ws=None, prepared=False, fed=None):
        """fed = (dlwp_feedback, state_b, sol | None, mean | None, calls of the whole rollout): the outputs are not the next inputs --
        a feedback launch between the calls builds the next state (dlwp_rollout_create_fed; one chain, one output); series:
        (calls, n) + output store; with span a time slice of it.
        chain = (index, count): a single-chain graph over members [index * n / count, (index + 1) * n / count) of state0 / series,
        with activation buffers of its own (the halves of a SplitRollout).
        span = (first call, number of calls): a TIME SLICE of the rollout over `series` -- the graph of calls [first, first + number),
        reading its state from the series slot the call before wrote (state0 for first = 0) -- one of several graphs launched one
        behind the other on a stream (StreamedRollout).  ws: the prepared-weights workspace to use (slices of one member count
        share it); prepared: the slice launched in front has filled it."""
        from . import _lib, ops
        _lib.drain_graveyard()          # a safe point: nothing is being captured yet
        n_total = int(state0.shape[0])
        n = n_total if chain is None else n_total // int(chain[1])
        first = 0 if chain is None else int(chain[0]) * n
        call0, calls = (0, int(calls)) if span is None else (int(span[0]), int(span[1]))
        n_out = len(self.plan.output_store)
        if fed is not None:
            if n_out != 1 or chain is not None:
                raise ValueError('a fed rollout is one member chain of a model with one output')
        else:
            for s in self.plan.output_store:
                if tuple(s) != tuple(self.plan._in_store):
                    raise ValueError('rollout needs every model output to have the input state shape %r, got %r' %
                                     (self.plan._in_store, s))
        bufs = self.scratch(n) if chain is None else \
            [torch.empty((n,) + s_, dtype=torch.bfloat16 if i in self._bf16 else torch.float32, device=self.device)
             for i, s_ in enumerate(self.plan.buffers)]
        table = [b for b in bufs]
        widx = {}
        for lay in self.plan.conv_layers:
            widx[id(lay)] = (len(table), len(table) + 1 if lay.bias is not None else -1)
            table.append(lay.kernel)
            if lay.bias is not None:
                table.append(lay.bias)
        pidx = []                                  # derived (phase-summed) kernels: (w2 index, b2 index | -1)
        for w2, b2 in self.phase_buffers():
            pidx.append((len(table), len(table) + 1 if b2 is not None else -1))
            table.append(w2)
            if b2 is not None:
                table.append(b2)
        kind = {'conv': _lib.OP_CONV2D, 'pad': _lib.OP_PAD2D, 'maxpool': _lib.OP_MAXPOOL2,
                'upsample': _lib.OP_UPSAMPLE2, 'copy': _lib.OP_COPYCH, 'lstm': _lib.OP_LSTM_GATES,
                'phasew': _lib.OP_PHASE_WEIGHTS, 'd2s': _lib.OP_DEPTH2SPACE, 'rowconv': _lib.OP_ROWCONV2D}
        arr = (_lib.Op * len(self.plan.ops))()
        for k, (op, d) in enumerate(zip(self.plan.ops, self._descriptors())):
            o = arr[k]
            o.kind, o.src, o.dst, o.w, o.b = kind[op.kind], op.src, op.dst, -1, -1
            o.src2, o.w2 = _lib.BUF_NONE, -1
            o.xs = _lib.Shape4(n, *op.xs)
            if op.kind == 'conv':
                o.w, o.b = pidx[op.wparam] if op.wparam is not None else widx[id(op.layer)]
                o.conv = d
                o.aux[0] = self._conv_dtype(op)
                if op.lstm_f:                          # cell update in the epilogue: z_add | NONE, c_prev | NONE, c_out
                    za, cp, co = op.aux
                    o.aux[1] = za if za is not None else _lib.BUF_NONE
                    o.aux[2] = cp if cp is not None else _lib.BUF_NONE
                    o.aux[3] = co
                if op.src2 is not None:                # a whole step: the input convolution rides along
                    il = op.src2['layer']
                    o.src2, o.w2, o.xs2_c = op.src2['buf'], widx[id(il)][0], op.src2['xs'][0]
                    o.conv2 = self._descriptor2(op)
                    o.b = widx[id(il)][1]                  # the layer's bias belongs to its input convolution
            elif op.kind == 'rowconv':                 # RowConnected2D: float32 buffers, per-row weights as stored
                o.w, o.b = widx[id(op.layer)]
                o.conv = d
                o.aux[0] = _lib.F32
            elif op.kind == 'phasew':                  # kernel -> derived kernel, once at the head of the graph
                o.src, o.b = widx[id(op.layer)]
                o.dst, b2i = pidx[op.wparam]
                o.aux[0] = b2i if b2i >= 0 else _lib.BUF_NONE
                kh, kw = op.layer.kernel_size
                o.conv.cout, o.conv.kh, o.conv.kw = op.layer.filters, kh, kw
                o.conv.halo.top, o.conv.halo.left = op.halo.top, op.halo.left
                o.xs = _lib.Shape4(n, op.xs[0], 1, 1)
            elif op.kind == 'd2s':
                o.conv.out_c_off, o.conv.out_c_total = op.out_c_off, op.out_c_total
            elif op.kind == 'maxpool':
                o.aux[0] = _lib.BF16 if op.src in self._bf16 else _lib.F32
            elif op.kind == 'pad':
                o.pad = d
                if op.inner > 1:
                    o.xs = _lib.Shape4(n, 1, op.xs[1], op.xs[2])
                    o.conv.in_c_total = op.inner
            elif op.kind == 'copy':
                o.conv.in_c_off, o.conv.in_c_total = op.in_c_off, op.in_c_total
                o.conv.out_c_off, o.conv.out_c_total = op.out_c_off, op.out_c_total
            elif op.kind == 'lstm':
                zh, cp, co = op.aux
                o.aux[0] = zh if zh is not None else _lib.BUF_NONE
                o.aux[1] = cp if cp is not None else _lib.BUF_NONE
                o.aux[2], o.aux[3] = co, op.rec_act + (256 if op.dst in self._bf16 else 0) + \
                    (512 if op.src in self._bf16 else 0)
                o.conv.act = op.act
                o.conv.out_c_off, o.conv.out_c_total = op.out_c_off, op.out_c_total
        ptrs = (ctypes.c_void_p * max(1, len(table)))(*[t.data_ptr() for t in table])
        member = int(np.prod(self.plan._in_store))
        slot = member * n_total                    # elements between two time slots of the series (all members of the rollout)
        out = ctypes.c_void_p()
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if fed is not None:
            groups = 1                             # rows exchange data between the calls: one chain
        groups = self.member_groups(n, self.plan._in_store[1] * self.plan._in_store[2]) if groups is None else int(groups)
        nbuf = len(bufs)
        sample_bytes = (ctypes.c_size_t * max(1, len(table)))(
            *[(t[0].numel() * t.element_size() if (i < nbuf and n > 0) else 0) for i, t in enumerate(table)])
        # prepared weights (Winograd / packed-N / bf16 layouts) live in memory WE own: the library allocates nothing
        ws_bytes = int(_lib.lib.dlwp_rollout_workspace_bytes(_lib.handle(dev), arr, len(self.plan.ops), groups))
        if ws is None:
            ws = torch.empty(max(1, (ws_bytes + 3) // 4), dtype=torch.float32, device=self.device)
        elif ws.numel() * 4 < ws_bytes:
            raise ValueError('rollout workspace of %d bytes, %d needed' % (ws.numel() * 4, ws_bytes))
        if fed is not None:
            fb, state_b, sol, mean, total_calls = fed
            slot_out = int(np.prod(self.plan.output_store[0])) * n
            _lib.check(_lib.lib.dlwp_rollout_create_fed(
                _lib.handle(dev), arr, len(self.plan.ops), ptrs, len(table), ctypes.c_void_p(state0.data_ptr()),
                ctypes.c_void_p(state_b.data_ptr()), ctypes.c_void_p(series.data_ptr() + 4 * call0 * slot_out), slot_out,
                int(calls), call0, int(call0 + calls < int(total_calls)), ctypes.byref(fb),
                ctypes.c_void_p(sol.data_ptr() if sol is not None else 0),
                ctypes.c_void_p(mean.data_ptr() if mean is not None else 0), _lib.F32 | (_lib.ROLLOUT_PREPARED if prepared else 0),
                ctypes.c_void_p(ws.data_ptr()), ws_bytes, ctypes.byref(out)))
            rg = RolloutGraph(out, keep=(table, state0, series, arr, ptrs, ws, fed), device=self.device)
            rg.groups, rg.ws = 1, ws
            return rg
        # (a time slice: the state comes from the slot in front of its first one, its series starts at its own first slot)
        state_ptr = state0.data_ptr() if call0 == 0 else series.data_ptr() + 4 * (call0 * n_out - 1) * slot
        series_ptr = series.data_ptr() + 4 * call0 * n_out * slot
        _lib.check(_lib.lib.dlwp_rollout_create_grouped(_lib.handle(dev), arr, len(self.plan.ops), ptrs, len(table),
                                                        sample_bytes, groups, ctypes.c_void_p(state_ptr + 4 * first * member),
                                                        ctypes.c_void_p(series_ptr + 4 * first * member), slot, int(calls), n_out,
                                                        _lib.F32 | (_lib.ROLLOUT_PREPARED if prepared else 0),
                                                        ctypes.c_void_p(ws.data_ptr()), ws_bytes, ctypes.byref(out)))
        rg = RolloutGraph(out, keep=(table, state0, series, arr, ptrs, ws), device=self.device)
        rg.groups = int(groups)
        rg.ws = ws
        return rg

    def make_streamed_rollout(self, n, calls, head_chunks=1):
        """The rollout as TIME SLICES (StreamedRollout): one graph per model call, launched one behind the other, so that the
        caller can start a finished forecast slot's copy to the host while the next call runs.  The first call is cut into
        `head_chunks` member chunks (graphs with activation buffers of their own) so that it can start on the first chunk of an
        input that is still being uploaded.  Same kernels on the same data as the one-graph rollout: the same bits."""
        n, calls, head_chunks = int(n), int(calls), max(1, int(head_chunks))
        while n % head_chunks:
            head_chunks -= 1
        n_out = len(self.plan.output_store)
        s0 = torch.empty((n,) + self.plan._in_store, dtype=torch.float32, device=self.device)
        series = torch.empty((calls * n_out, n) + self.plan._in_store, dtype=torch.float32, device=self.device)
        # one member chain per call (DLWP_STREAMED_GROUPS): a forked graph is launched on a stream of its own between two events
        # (csrc/rollout.hip) -- per CALL here, not per rollout
        g = int(os.environ.get('DLWP_STREAMED_GROUPS', '1'))
        while n % g:
            g -= 1
        if head_chunks == 1:
            head = [self._make_rollout(s0, series, calls, g, span=(0, 1))]
        else:
            # (the chunks have one member count, i.e. one layout of prepared weights: chunk 0 prepares, the others -- launched behind
            #  it on the same stream -- read its workspace)
            head = []
            for c in range(head_chunks):
                head.append(self._make_rollout(s0, series, calls, 1, chain=(c, head_chunks), span=(0, 1),
                                               ws=head[0].ws if head else None, prepared=bool(head)))
        tail = []
        for j in range(1, calls):
            tail.append(self._make_rollout(s0, series, calls, g, span=(j, 1), ws=tail[0].ws if tail else None, prepared=bool(tail)))
        return StreamedRollout(s0, series, head, tail, n_out, self.device)


def _mirrored(fn):
    """driver mode of dlwp_amd.parallel (a plain script that asked for gpus=n): the call goes to the worker ranks too"""
    import functools

    @functools.wraps(fn)
    def wrap(self, *args, **kwargs):
        drv = self.__dict__.get('_driver')
        if drv is not None and not drv.in_call and not drv.closed:
            return drv.call(self, fn.__name__, args, kwargs, fn)
        return fn(self, *args, **kwargs)
    return wrap


class RolloutGraph(object):
    """Owner of a captured rollout (dlwp_rollout_t).  launch() replays all forwards with one hipGraphLaunch."""

    def __init__(self, handle, keep, device):
        self._h, self._keep, self.device = handle, keep, device

    def launch(self):
        from . import _lib
        _lib.check(_lib.lib.dlwp_rollout_launch(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def close(self):
        if self._h is not None:
            from . import _lib
            _lib.lib.dlwp_rollout_destroy(self._h)
            self._h = None

    def __del__(self):      # (never destroys here: _lib.bury -- the next rollout capture or training step drains the graveyard)
        try:
            from . import _lib
            _lib.bury('rollout', self._h)
            self._h = None
        except Exception:  # noqa: BLE001
            pass


class StreamedRollout(object):
    """A rollout captured as one graph per model call (Executor.make_streamed_rollout): head[c] = call 0 on member chunk c,
    tail[j - 1] = call j on all members.  The owner launches them in order on ONE stream and records an event behind each: slot(s)
    [j * n_outputs, (j + 1) * n_outputs) of `series` are final when call j's event has passed -- that is when their copies to the
    host may start, under call j + 1.  Replaces the host loop of DLWP/model/models.py:277-293, whose every step ends with a
    device-to-host copy the next step waits for."""

    def __init__(self, s0, series, head, tail, n_out, device):
        self.s0, self.series, self.head, self.tail, self.n_out, self.device = s0, series, list(head), list(tail), n_out, device

    @property
    def calls(self):
        return 1 + len(self.tail)

    def chunk_bounds(self):
        n, k = int(self.s0.shape[0]), len(self.head)
        return [(c * (n // k), (c + 1) * (n // k)) for c in range(k)]

    def close(self):
        for g in self.head + self.tail:
            g.close()
        self.head, self.tail = [], []
    # (no finaliser: the graphs bury themselves, RolloutGraph.__del__)


class SplitRollout(object):
    """The members of a rollout as TWO single-chain graphs (Executor._make_rollout(chain=...)) launched side by side on two streams
    that were probed to sit on hardware queues of their own (util.distinct_streams) -- what a forked graph's branches do only when
    the streams the runtime draws for them happen to land on different queues.  Same kernels, same bits."""

    def __init__(self, graphs, streams, device):
        self._graphs, self._streams, self.device = list(graphs), list(streams), device
        self.groups = 'split'

    def launch(self):
        main = torch.cuda.current_stream(self.device)
        for g, s in zip(self._graphs, self._streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                g.launch()
        for s in self._streams:
            main.wait_stream(s)

    def close(self):
        for g in self._graphs:
            g.close()
        self._graphs = []
    # (no finaliser: the graphs bury themselves, RolloutGraph.__del__)


class Model(object):
    """Functional model: Model(inputs=Input(...), outputs=tensor | [tensors])."""

    def __init__(self, inputs=None, outputs=None, name=None, device=None, seed=None):
        self.name = name or 'model'
        self.device = device if device is not None else default_device()
        self.stop_training = False
        self.optimizer = None
        self.loss = None
        self.metrics = []
        self.metrics_names = ['loss']
        self.loss_weights = None
        self.history = None
        self._seed = seed
        self._trainer = None
        if inputs is not None:
            self._init_graph(inputs, outputs)

    # -- graph ------------------------------------------------------------------------------------------------------- #
    def _init_graph(self, inputs, outputs, rng=None):
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        order = P.toposort(self.outputs)
        self.layers = []
        for t in order:
            if t.layer not in self.layers:
                self.layers.append(t.layer)
        if rng is None:
            rng = np.random.RandomState(self._seed if self._seed is not None else np.random.randint(0, 2 ** 31 - 1))
        for t in order:
            if isinstance(t.layer, L.InputLayer):
                continue
            t.layer.build(t.inputs[0].shape, self.device, rng)
        self.plan = P.build_plan(self.inputs, self.outputs)                       # training: every activation kept
        self.infer_plan = P.build_plan(self.inputs, self.outputs, inference=True)   # predict / rollout
        self.activation_dtype = getattr(self, 'activation_dtype', 'float32')
        self.executor = Executor(self.infer_plan, self.device, self.activation_dtype)
        self._train_executor = None
        self.input_shape = (None,) + tuple(self.inputs[0].shape)
        shapes = [(None,) + tuple(o.shape) for o in self.outputs]
        self.output_shape = shapes[0] if len(shapes) == 1 else shapes

    # -- storage type of the activations between the layers (inference) ------------------------------------------------- #
    def set_activation_dtype(self, dtype):
        """'float32' (default) or 'bfloat16': how predict / predict_timeseries store the tensors BETWEEN convolutions
        (BASELINE.json config 4).  Model inputs, outputs, weights and all arithmetic stay float32; training always runs
        on float32 activations."""
        if dtype not in ('float32', 'bfloat16'):
            raise ValueError("activation dtype must be 'float32' or 'bfloat16'")
        if dtype != self.activation_dtype:
            self.activation_dtype = dtype
            # interleaved phase stores (dlwp_conv2d.out_d2s) belong to the float32 Winograd kernels; with bfloat16 storage
            # the restated layers run on the bf16 matrix cores and keep the separate depth-to-space pass
            # ... and the ConvLSTM2D cell update rides in a convolution's epilogue there (dlwp_convlstm_conv_fwd), provided
            # the h sequence really is stored as bfloat16
            bf = dtype == 'bfloat16'
            self.infer_plan = P.build_plan(self.inputs, self.outputs, inference=True, fuse_d2s=not bf,
                                           fuse_lstm=bf and os.environ.get('DLWP_LSTM_FUSE', '1') != '0')
            if bf and any(op.kind == 'conv' and op.lstm_f and op.dst not in self.infer_plan.bf16_buffers()
                          for op in self.infer_plan.ops):
                self.infer_plan = P.build_plan(self.inputs, self.outputs, inference=True, fuse_d2s=False)
            self.executor = Executor(self.infer_plan, self.device, dtype)
            if any(op.kind == 'conv' and op.src2 is not None and op.dst not in self.executor._oct for op in self.infer_plan.ops):
                # whole-step ConvLSTM2D launches exist in the octet layout only: two launches per step otherwise
                self.infer_plan = P.build_plan(self.inputs, self.outputs, inference=True, fuse_d2s=False, fuse_lstm=True,
                                               fuse_lstm_step=False)
                self.executor = Executor(self.infer_plan, self.device, dtype)
            self.__dict__.pop('_rollouts', None)
            self.__dict__.pop('_rollouts_streamed', None)
        return self

    @property
    def train_executor(self):
        """Executor of the training plan: float32 activations, every pre-pooling tensor materialised."""
        if self._train_executor is None:
            same = self.activation_dtype == 'float32' and len(self.infer_plan.ops) == len(self.plan.ops) and \
                not any(op.out_pool or op.out_d2s for op in self.infer_plan.ops)
            self._train_executor = self.executor if same else Executor(self.plan, self.device, 'float32')
        return self._train_executor

    # -- weights ----------------------------------------------------------------------------------------------------- #
    @property
    def weights(self):
        return [w for lay in self.layers for w in lay.weights]

    def get_weights(self):
        return [a for lay in self.layers for a in lay.get_weights()]

    @_mirrored
    def set_weights(self, arrays):
        arrays = list(arrays)
        k = 0
        for lay in self.layers:
            m = len(lay._weights)
            if m:
                lay.set_weights(arrays[k:k + m])
                k += m
        if k != len(arrays):
            raise ValueError('model has %d weight arrays, got %d' % (k, len(arrays)))
        if self._trainer is not None:          # data parallel: replicas re-align on rank 0's at the next training step
            self._trainer._params_dirty = True

    def count_params(self):
        return int(sum(lay.count_params() for lay in self.layers))

    def summary(self, print_fn=None):
        pr = print_fn or print
        pr('_' * 72)
        pr('%-34s %-26s %10s' % ('Layer (type)', 'Output Shape', 'Param #'))
        pr('=' * 72)
        for lay in self.layers:
            pr('%-34s %-26s %10d' % ('%s (%s)' % (lay.name, type(lay).__name__), str(lay.output_shape),
                                     lay.count_params()))
        pr('=' * 72)
        pr('Total params: %d' % self.count_params())
        pr('fused launches per forward: %d (%d conv)' % (self.infer_plan.n_launches,
                                                         sum(1 for o in self.infer_plan.ops if o.kind in ('conv', 'rowconv'))))
        pr('_' * 72)

    def reset_states(self):
        pass    # no stateful recurrent layers on this path

    # -- inference --------------------------------------------------------------------------------------------------- #
    def _logical_outputs(self, outs, n):
        res = [o.reshape((n,) + tuple(s)) for o, s in zip(outs, self.plan.output_shapes)]
        return res

    def predict_on_device(self, x):
        """x: float32 device tensor (n,)+input_shape  ->  device tensor (or list for multi-output models)."""
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        outs = self._logical_outputs(self.executor.run(x), x.shape[0])
        return outs[0] if len(outs) == 1 else outs

    def predict(self, x, batch_size=None, verbose=0, steps=None, **kwargs):
        """numpy in, numpy out (Keras contract).  batch_size only bounds the device working set: results do not depend
        on it (tests/test_gpu_model.py checks bit-identity)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if tuple(x.shape[1:]) != tuple(self.inputs[0].shape):
            raise ValueError('expected input of shape %r, got %r' % ((None,) + tuple(self.inputs[0].shape), x.shape))
        n = x.shape[0]
        chunk = int(batch_size) if batch_size else 2048
        chunk = max(1, min(max(chunk, 256), n)) if n else 1
        results = None
        if n > chunk and self.device.type == 'cuda':
            # several chunks: results go into pinned host arrays on a copy stream, under the next chunk's kernels
            from .util import io_streams
            copy_stream = io_streams(self.device)[0]          # (a hardware queue of its own: util.distinct_streams)
            copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            pinned = None
            for lo in range(0, n, chunk):
                xd = torch.from_numpy(x[lo:lo + chunk]).to(self.device, non_blocking=False)
                outs = self.predict_on_device(xd)
                outs = outs if isinstance(outs, list) else [outs]
                if pinned is None:
                    pinned = [host_result_buffer((n,) + tuple(o.shape[1:])) for o in outs]
                done = torch.cuda.Event()
                done.record()
                copy_stream.wait_event(done)
                with torch.cuda.stream(copy_stream):
                    for r, o in zip(pinned, outs):
                        r[lo:lo + chunk].copy_(o.contiguous(), non_blocking=True)
                for o in outs:
                    o.record_stream(copy_stream)
            copy_stream.synchronize()
            results = [r.numpy() for r in pinned]
            return results[0] if len(results) == 1 else results
        for lo in range(0, n, chunk):
            xd = torch.from_numpy(x[lo:lo + chunk]).to(self.device, non_blocking=False)
            outs = self.predict_on_device(xd)
            outs = outs if isinstance(outs, list) else [outs]
            host = [o.cpu().numpy() for o in outs]
            if results is None:
                results = [np.empty((n,) + h.shape[1:], dtype=np.float32) for h in host]
            for r, h in zip(results, host):
                r[lo:lo + chunk] = h
        if results is None:
            results = [np.empty((0,) + tuple(s), dtype=np.float32) for s in self.plan.output_shapes]
        return results[0] if len(results) == 1 else results

    def rollout_on_device(self, state0, calls, series=None, graph_cache=True):
        """Autoregressive rollout entirely in HBM: returns series (calls*n_outputs, n)+input_shape (device).
        The captured hipGraph is cached per (n, calls) and replayed on later calls."""
        n = state0.shape[0]
        n_out = len(self.outputs)
        key = (n, int(calls))
        cache = self.__dict__.setdefault('_rollouts', {})
        entry = cache.get(key) if graph_cache else None
        if entry is None:
            s0 = torch.empty((n,) + self.plan._in_store, dtype=torch.float32, device=self.device)
            ser = torch.empty((calls * n_out, n) + self.plan._in_store, dtype=torch.float32, device=self.device)
            g = self.executor.make_rollout(s0, ser, calls)
            entry = (g, s0, ser)
            if graph_cache:
                if len(cache) > 2:
                    cache.clear()
                cache[key] = entry
        g, s0, ser = entry
        s0.copy_(state0.reshape(s0.shape))
        g.launch()
        out = ser.reshape((calls * n_out, n) + tuple(self.inputs[0].shape))
        if series is not None:
            series.copy_(out)
            return series
        return out

    def _fed_entry(self, n, calls, src, shift, tail, sol, sol_map, mean, sliced):
        """the cached graphs + buffers of a fed rollout: one graph of all calls, or (sliced) one graph per call"""
        if len(self.plan.output_store) != 1:
            raise ValueError('a fed rollout needs a model with one output')
        c_in, h, w = self.plan._in_store
        c_out = int(self.plan.output_store[0][0])
        if tuple(self.plan.output_store[0][1:]) != (h, w):
            raise ValueError('a fed rollout needs outputs on the input grid: %r -> %r' % (self.plan._in_store, self.plan.output_store[0]))
        src = tuple(int(v) for v in src)
        sol_map = tuple(int(v) for v in sol_map) if sol is not None else None
        planes = int(sol.shape[2]) if sol is not None else 0
        key = (n, calls, src, int(shift), int(tail), sol_map, planes, mean is not None, bool(sliced))
        cache = self.__dict__.setdefault('_rollouts_fed', {})
        ent = cache.get(key)
        if ent is None:
            from . import ops
            for old in cache.values():
                for g in old[0]:
                    g.close()
            cache.clear()
            f32 = dict(dtype=torch.float32, device=self.device)
            sa, sb = torch.empty((n, c_in, h, w), **f32), torch.empty((n, c_in, h, w), **f32)
            ser = torch.empty((calls, n) + tuple(self.plan.output_store[0]), **f32)
            fb = ops.make_feedback(n, c_in, c_out, h * w, src, shift=shift, tail=tail, sol=sol_map, sol_planes=planes)
            sol_d = torch.empty((max(calls - 1, 1), fb.tail, planes, h, w), **f32) if sol is not None else None
            mean_d = torch.empty((c_in, h, w), **f32) if mean is not None else None
            fed = (fb, sb, sol_d, mean_d, calls)
            if not sliced:
                graphs = [self.executor._make_rollout(sa, ser, calls, fed=fed)]
            else:
                graphs = []
                for j in range(calls):
                    graphs.append(self.executor._make_rollout(sa, ser, calls, fed=fed, span=(j, 1),
                                                              ws=graphs[0].ws if graphs else None, prepared=bool(graphs)))
            ent = cache[key] = (graphs, sa, sb, ser, sol_d, mean_d)
        return ent

    @staticmethod
    def _fed_inputs(ent, calls, state0, sol, mean):
        _, sa, _, _, sol_d, mean_d = ent
        as_dev = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
        sa.copy_(as_dev(state0).reshape(sa.shape), non_blocking=True)
        if sol_d is not None and calls > 1:
            sol_d.copy_(as_dev(sol).reshape(sol_d.shape), non_blocking=True)
        if mean_d is not None:
            mean_d.copy_(as_dev(mean).reshape(mean_d.shape), non_blocking=True)

    def fed_rollout_on_device(self, state0, calls, src, shift=0, tail=0, sol=None, sol_map=None, mean=None):
        """A rollout whose outputs are NOT its next inputs, entirely in HBM (dlwp_rollout_create_fed): between two model calls ONE
        launch builds the next state from the old one (rows shifted by `shift`), the call's output (src[c] = -1 - j), the insolation
        block of the call (sol: (calls - 1, tail, planes) + grid, channel c takes plane sol_map[c]) and the mean state -- the
        bookkeeping TimeSeriesEstimator.predict does on the host with xarray between two model.predict round trips
        (DLWP/model/extensions.py:206-240) and the step_sequence shift of DLWP/model/models.py:280-290.
        state0: (n,) + input shape, device or host.  Returns the device series (calls, n) + output shape; cached per configuration."""
        n, calls = int(state0.shape[0]), int(calls)
        ent = self._fed_entry(n, calls, src, shift, tail, sol, sol_map, mean, sliced=False)
        self._fed_inputs(ent, calls, state0, sol, mean)
        ent[0][0].launch()
        return ent[3].reshape((calls, n) + tuple(self.plan.output_shapes[0]))

    def fed_rollout_to_host(self, state0, calls, src, shift=0, tail=0, sol=None, sol_map=None, mean=None, t_out=1, kept=None,
                            perm=None, time_major=True, blocks=None):
        """The same forecast with its series going home WHILE it runs, in the layout the reference returns
        (DLWP/model/extensions.py:243-302): one hipGraph per model call (call + feedback), launched back to back; behind call j, on
        a copy stream, its output (n, t_out, C) is arranged into its block of the result -- time first (kept, n, C) or, time_major
        False, (n, t_out, C); channels permuted by `perm` -- and leaves for ONE page-locked array under call j + 1.  blocks: how
        many (time-major: time steps; else calls) of the result to keep in all (the [:steps] cut).  Returns a numpy array
        (blocks | calls * kept, n, C, h, w) or (calls, n, t_out, C, h, w) lent from the pinned pool."""
        import ctypes as ct
        from . import _lib, util
        n, calls = int(state0.shape[0]), int(calls)
        ent = self._fed_entry(n, calls, src, shift, tail, sol, sol_map, mean, sliced=True)
        graphs, _, _, ser, _, _ = ent
        c_tot, h, w = self.plan.output_store[0]
        t_out = int(t_out)
        c = int(c_tot) // t_out
        kept = t_out if (kept is None or not time_major) else int(kept)
        total = calls * kept if time_major else calls
        blocks = total if blocks is None else min(int(blocks), total)
        host = util.pinned_results.take(((blocks, n, c, h, w) if time_major else (blocks, n, t_out, c, h, w)))
        per_block = n * c * h * w if time_major else n * t_out * c * h * w
        dev = self.device
        hnd = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
        main = torch.cuda.current_stream(dev)
        down = util.d2h_streams(dev)
        stage = self.__dict__.setdefault('_fed_stage', {})
        for k in range(len(down)):
            if k not in stage or stage[k].numel() < kept * n * c * h * w or stage[k].device != dev:
                stage[k] = torch.empty(kept * n * c * h * w, dtype=torch.float32, device=dev)
        perm_arr = (ct.c_int * c)(*[int(v) for v in (perm if perm is not None else range(c))])
        self._fed_inputs(ent, calls, state0, sol, mean)
        for s_ in down:
            s_.wait_stream(main)
        try:
            for j, g in enumerate(graphs):
                g.launch()
                first = j * kept if time_major else j
                take = min(kept if time_major else 1, blocks - first)
                if take <= 0:
                    continue
                ev = torch.cuda.Event()
                ev.record(main)
                k = j % len(down)
                down[k].wait_event(ev)
                with torch.cuda.stream(down[k]):
                    _lib.check(_lib.lib.dlwp_series_arrange(hnd, ct.c_void_p(ser[j].data_ptr()), ct.c_void_p(stage[k].data_ptr()), n,
                                                            t_out, c, h * w, kept, perm_arr, 1 if time_major else 0, _lib.F32,
                                                            ct.c_void_p(down[k].cuda_stream)))
                    host.view(-1)[first * per_block:(first + take) * per_block].copy_(stage[k][:take * per_block], non_blocking=True)
        except BaseException:
            for s_ in down:                    # (copies in flight may still write into the result array: drain, then recycle it)
                try:
                    s_.synchronize()
                except Exception:  # noqa: BLE001
                    pass
            util.pinned_results._give_back(host.view(-1))
            raise
        finally:
            for s_ in down:
                s_.synchronize()
            for s_ in down:                    # (the cached series / staging buffers are rewritten by the next rollout on the main stream)
                main.wait_stream(s_)
        return util.pinned_results.lend(host)

    def streamed_rollout(self, n, calls, head_chunks=4):
        """The time-sliced rollout of n members x `calls` model applications (Executor.make_streamed_rollout), cached like the
        one-graph rollouts of rollout_on_device."""
        key = (int(n), int(calls), int(head_chunks))
        cache = self.__dict__.setdefault('_rollouts_streamed', {})
        ent = cache.get(key)
        if ent is None:
            if cache:
                for old in cache.values():
                    old.close()
                cache.clear()
            ent = cache[key] = self.executor.make_streamed_rollout(n, calls, head_chunks)
        return ent

    # -- training (dlwp_amd.training) ---------------------------------------------------------------------------------- #
    @_mirrored
    def compile(self, optimizer='adam', loss=None, metrics=None, loss_weights=None, **kwargs):
        from . import training
        self.__dict__.pop('_rollouts', None)     # compile re-homes the weights into one flat buffer: drop captured graphs
        self.__dict__.pop('_rollouts_streamed', None)
        self.optimizer = training.get_optimizer(optimizer)
        self.loss = loss
        self.loss_weights = loss_weights
        self.metrics = list(metrics or [])
        self._trainer = training.Trainer(self)
        self.metrics_names = self._trainer.metrics_names

    def _need_trainer(self):
        if self._trainer is None:
            raise RuntimeError('You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.')
        return self._trainer

    @_mirrored
    def train_on_batch(self, x, y):
        return self._need_trainer().train_on_batch(x, y)

    def test_on_batch(self, x, y):
        return self._need_trainer().test_on_batch(x, y)

    @_mirrored
    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            shuffle=True, initial_epoch=0, **kwargs):
        return self._need_trainer().fit(x, y, batch_size=batch_size, epochs=epochs, verbose=verbose,
                                        callbacks=callbacks, validation_data=validation_data, shuffle=shuffle,
                                        initial_epoch=initial_epoch)

    @_mirrored
    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None,
                      validation_data=None, validation_steps=None, use_multiprocessing=False, workers=1,
                      max_queue_size=10, shuffle=True, initial_epoch=0, **kwargs):
        return self._need_trainer().fit_generator(generator, steps_per_epoch=steps_per_epoch, epochs=epochs,
                                                  verbose=verbose, callbacks=callbacks,
                                                  validation_data=validation_data, validation_steps=validation_steps,
                                                  shuffle=shuffle, initial_epoch=initial_epoch)

    def evaluate(self, x=None, y=None, batch_size=None, verbose=1, **kwargs):
        return self._need_trainer().evaluate(x, y, batch_size=batch_size, verbose=verbose)

    # -- persistence ---------------------------------------------------------------------------------------------------- #
    def save(self, path, format=None):
        """Keras HDF5 by default (what the reference's model.save writes, DLWP/util.py:141-144); format='npz': this package's archive"""
        from . import serialization
        serialization.save_model_file(self, path, format=format)


class Sequential(Model):
    """keras.models.Sequential: the first layer carries input_shape= (examples/train.py:159-162).  The graph, the
    weights of the new layer and the fused plan are (re)built at every add(), so shape errors surface where Keras
    raises them."""

    def __init__(self, layers=None, name=None, device=None, seed=None):
        super(Sequential, self).__init__(name=name or 'sequential', device=device, seed=seed)
        self._stack = []
        self._input_layer = None
        self.layers = []
        self._rng = np.random.RandomState(seed if seed is not None else np.random.randint(0, 2 ** 31 - 1))
        for lay in (layers or []):
            self.add(lay)

    def add(self, layer):
        if not isinstance(layer, L.Layer):
            raise TypeError('The added layer must be an instance of class Layer. Found: %r' % (layer,))
        if not self._stack and layer.batch_input_shape is None and not isinstance(layer, L.InputLayer):
            raise ValueError('The first layer in a Sequential model must get an `input_shape` argument.')
        self._stack.append(layer)
        try:
            self._rebuild()
        except Exception:
            self._stack.pop()
            raise

    def _rebuild(self):
        first = self._stack[0]
        if isinstance(first, L.InputLayer):
            self._input_layer = first
            rest = self._stack[1:]
        else:
            if self._input_layer is None:
                self._input_layer = L.InputLayer(input_shape=first.batch_input_shape[1:])
            rest = self._stack
        t = x_in = L.KTensor(self._input_layer.batch_input_shape[1:], self._input_layer, ())
        for lay in rest:
            lay._calls = 0
            t = lay(t)
        self._init_graph(x_in, t, rng=self._rng)
        self.layers = list(self._stack)
