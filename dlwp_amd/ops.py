"""
Tensor-level wrappers over the C ABI.  torch is used ONLY as plumbing here: device allocations (torch.empty), the
current HIP stream, and device indices.  Every arithmetic / data-movement operation is a libdlwp_hip.so kernel.
"""
import ctypes

import torch

from . import _lib
from ._lib import (ACT_LINEAR, ACT_RELU, ACT_TANH, PAD_EDGE, PAD_REFLECT, PAD_SYMMETRIC, PAD_WRAP, PAD_ZERO,  # noqa: F401
                   SRC_DIRECT, SRC_MAXPOOL2,
                   SRC_UPSAMPLE2, Conv2d, Pad2d, Shape4)

ACTIVATIONS = {None: ACT_LINEAR, 'linear': ACT_LINEAR, 'tanh': ACT_TANH, 'relu': ACT_RELU}


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(t):
    """the current HIP stream of t's device (torch.cuda.current_stream builds a Stream object per call: ~5 us, 50 times per
    training step; the raw handle is what the C ABI wants anyway)"""
    if _raw_stream is not None:
        idx = t.device.index
        return ctypes.c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _dev(t):
    if not t.is_cuda:
        raise RuntimeError('dlwp_amd ops need device (HIP) tensors; there is no CPU fallback')
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _check_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError('expected contiguous float32 tensors, got %s contiguous=%s' % (t.dtype, t.is_contiguous()))


def _check_act(*ts):
    """activation tensors: contiguous float32 or bfloat16 (storage types of the forward convolutions / max-pooling)"""
    for t in ts:
        if t is None:
            continue
        if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous():
            raise ValueError('expected contiguous float32 / bfloat16 tensors, got %s contiguous=%s' %
                             (t.dtype, t.is_contiguous()))


def storage_code(t):
    return _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32


def make_pad(top=0, bottom=0, left=0, right=0, mode_h=PAD_ZERO, mode_w=PAD_ZERO):
    return Pad2d(int(top), int(bottom), int(left), int(right), int(mode_h), int(mode_w))


def make_conv(cout, kh, kw, dil=1, halo=None, act=ACT_LINEAR, in_c_off=0, in_c_total=0, out_c_off=0, out_c_total=0,
              src_mode=SRC_DIRECT, out_pool=False, out_d2s=False, lstm_f=0, lstm_rec_act=0):
    dh, dw = (dil, dil) if isinstance(dil, int) else dil
    return Conv2d(int(cout), int(kh), int(kw), int(dh), int(dw), halo if halo is not None else make_pad(), int(act),
                  int(in_c_off), int(in_c_total), int(out_c_off), int(out_c_total), int(src_mode), int(bool(out_pool)),
                  int(bool(out_d2s)), int(lstm_f), int(lstm_rec_act))


def supports_out_pool(xs_chw, cd):
    """Planner hint: can a compiled kernel apply a following MaxPooling2D(2) in this convolution's epilogue?"""
    return bool(_lib.lib.dlwp_conv2d_supports_out_pool(_lib.handle_or_none(), Shape4(1, int(xs_chw[0]), int(xs_chw[1]), int(xs_chw[2])),
                                                       ctypes.byref(cd)))


def supports_out_d2s(xs_chw, cd):
    """Planner hint: can a compiled kernel store this convolution's 4 F phase channels interleaved (depth-to-space)?"""
    return bool(_lib.lib.dlwp_conv2d_supports_out_d2s(_lib.handle_or_none(), Shape4(1, int(xs_chw[0]), int(xs_chw[1]), int(xs_chw[2])),
                                                      ctypes.byref(cd)))


def conv_out_shape(xs, cd):
    ys = Shape4()
    _lib.check(_lib.lib.dlwp_conv2d_out_shape(xs, ctypes.byref(cd), ctypes.byref(ys)))
    return ys


def pad2d(x, pad, channels_last=False, out=None):
    """x: NCHW (or NHWC with channels_last=True) -> padded copy."""
    _check_f32(x)
    if channels_last:
        n, h, w, c = x.shape
        outer, inner = n, c
        oshape = (n, h + pad.top + pad.bottom, w + pad.left + pad.right, c)
    else:
        n, c, h, w = x.shape
        outer, inner = n * c, 1
        oshape = (n, c, h + pad.top + pad.bottom, w + pad.left + pad.right)
    y = out if out is not None else torch.empty(oshape, dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib.dlwp_pad2d_fwd(_lib.handle(_dev(x)), _ptr(x), _ptr(y), outer, h, w, inner, pad, _lib.F32,
                                       _stream(x)))
    return y


def pad2d_bwd(dy, x_shape, pad, channels_last=False):
    _check_f32(dy)
    if channels_last:
        n, h, w, c = x_shape
        outer, inner = n, c
    else:
        n, c, h, w = x_shape
        outer, inner = n * c, 1
    dx = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.lib.dlwp_pad2d_bwd(_lib.handle(_dev(dy)), _ptr(dy), _ptr(dx), outer, h, w, inner, pad, _lib.F32,
                                       _stream(dy)))
    return dx


def _code(t, o8=False):
    return _lib.BF16_O8 if o8 else storage_code(t)


def conv2d_prepare(x, w_hwio, cd, out_dtype=None, x_channels=None, compute_bf16=False, out=None, in_o8=False, out_o8=False):
    """Prepared weights for conv2d(x, ...) calls with exactly this input shape / storage (dlwp_conv2d_prepare), or None
    when the layer's kernel reads the HWIO weights directly.  out: a buffer an earlier call returned (refilled in place)."""
    _check_f32(w_hwio)
    n, c_total, h, w = x.shape
    cin = int(x_channels) if x_channels is not None else c_total
    if cd.in_c_total == 0 and cin != c_total:
        cd.in_c_total = c_total
    xs = Shape4(n, cin, h, w)
    out_code = storage_code(x) if out_dtype is None else {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}[out_dtype]
    dt = _lib.dtype_io(_code(x, in_o8), _lib.BF16_O8 if out_o8 else out_code, compute_bf16)
    nbytes = _lib.lib.dlwp_conv2d_prepared_bytes(_lib.handle(_dev(x)), xs, ctypes.byref(cd), dt)
    if nbytes == 0:
        return None
    u = out if out is not None else torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    if u.numel() * 4 < nbytes:
        raise ValueError('conv2d_prepare: the buffer holds %d bytes, %d needed' % (u.numel() * 4, nbytes))
    _lib.check(_lib.lib.dlwp_conv2d_prepare(_lib.handle(_dev(x)), _ptr(w_hwio), _ptr(u), xs, ctypes.byref(cd), dt,
                                            _stream(x)))
    return u


def conv2d(x, w_hwio, bias, cd, out=None, direct=False, x_channels=None, compute_bf16=False, prepared=None, in_o8=False,
           out_o8=False, out_pool2=None):
    """x: stored input (n, in_c_total, h, w); the conv reads `x_channels` (default: all) channels from cd.in_c_off.
    Returns (n, out_c_total, ho, wo); writes channels [out_c_off, out_c_off+cout).  compute_bf16: a float32 x may be
    rounded to bfloat16 so that the layer runs on the bf16 matrix cores (DLWP_COMPUTE_BF16).  prepared: the tensor
    conv2d_prepare returned for this call (the weights are then not transformed again).  in_o8 / out_o8: the bfloat16
    tensor x / out holds channel OCTETS, (n, C/8, h, w, 8) in the memory of an (n, C, h, w) tensor (DLWP_BF16_O8).
    out_pool2: a float32 (n, out_c_total, ho/2, wo/2) tensor that receives MaxPooling2D(2) of the output from the same launch
    (dlwp_conv2d_fwd_pool2); returns None -- nothing written -- where the layer's kernel cannot store both."""
    _check_act(x, out)
    _check_f32(w_hwio, bias)
    n, c_total, h, w = x.shape
    cin = int(x_channels) if x_channels is not None else c_total
    if tuple(w_hwio.shape) != (cd.kh, cd.kw, cin, cd.cout):
        raise ValueError('kernel shape %s does not match (kh,kw,cin,cout)=(%d,%d,%d,%d)' %
                         (tuple(w_hwio.shape), cd.kh, cd.kw, cin, cd.cout))
    if cd.in_c_total == 0 and cin != c_total:
        cd.in_c_total = c_total
    xs = Shape4(n, cin, h, w)
    ys = conv_out_shape(xs, cd)
    oc = cd.out_c_total if cd.out_c_total > 0 else ys.c
    if out is None:
        out = torch.empty((n, oc, ys.h, ys.w), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (n, oc, ys.h, ys.w):
        raise ValueError('output buffer shape %s != %s' % (tuple(out.shape), (n, oc, ys.h, ys.w)))
    fn = _lib.lib.dlwp_conv2d_fwd_direct if direct else _lib.lib.dlwp_conv2d_fwd
    if (in_o8 and x.dtype != torch.bfloat16) or (out_o8 and out.dtype != torch.bfloat16):
        raise ValueError('the octet layout is a bfloat16 storage')
    dt = _lib.dtype_io(_code(x, in_o8), _code(out, out_o8), compute_bf16)      # storage of x / y
    if out_pool2 is not None:
        _check_f32(x, out, out_pool2)
        if tuple(out_pool2.shape) != (n, oc, ys.h // 2, ys.w // 2) or not out_pool2.is_contiguous():
            raise ValueError('pooled output buffer shape %s != %s' % (tuple(out_pool2.shape), (n, oc, ys.h // 2, ys.w // 2)))
        rc = _lib.lib.dlwp_conv2d_fwd_pool2(_lib.handle(_dev(x)), _ptr(x), _ptr(w_hwio), _ptr(prepared), _ptr(bias), _ptr(out),
                                            _ptr(out_pool2), xs, ctypes.byref(cd), dt, _stream(x))
        if rc == _lib.EUNSUPPORTED:
            return None
        _lib.check(rc)
        return out
    if prepared is not None and not direct:
        _lib.check(_lib.lib.dlwp_conv2d_fwd_prepared(_lib.handle(_dev(x)), _ptr(x), _ptr(w_hwio), _ptr(prepared),
                                                     _ptr(bias), _ptr(out), xs, ctypes.byref(cd), dt, _stream(x)))
        return out
    _lib.check(fn(_lib.handle(_dev(x)), _ptr(x), _ptr(w_hwio), _ptr(bias), _ptr(out), xs, ctypes.byref(cd), dt,
                  _stream(x)))
    return out


def convlstm_conv_supported(xs_chw, cd, in_bf16, compute_bf16=False):
    """Planner hint: can this convolution (cd.lstm_f set) carry the ConvLSTM2D cell update in its epilogue, given the storage
    of its input (bfloat16, or float32 rounded by the loader)?"""
    dt = _lib.dtype_io(_lib.BF16 if in_bf16 else _lib.F32, _lib.BF16, compute_bf16)
    return bool(_lib.lib.dlwp_convlstm_conv_supported(_lib.handle_or_none(), Shape4(1, int(xs_chw[0]), int(xs_chw[1]), int(xs_chw[2])),
                                                      ctypes.byref(cd), dt))


def convlstm_conv(x, w_hwio, bias, cd, h_out, c_out, z_add=None, c_prev=None, x_channels=None, compute_bf16=False,
                  prepared=None, in_o8=False, out_o8=False):
    """One of the two convolutions of a ConvLSTM2D step with the cell update in its epilogue (dlwp_convlstm_conv_fwd):
    z = conv(x) + bias (+ z_add) is not stored; writes c_out (float32 (n, F, ho, wo)) and channels [cd.out_c_off, +F) of
    h_out.  z_add: bfloat16 (n, 4F, ho, wo) or None; c_prev: float32 or None."""
    _check_f32(w_hwio, bias, c_out, c_prev)
    n, c_total, h, w = x.shape
    cin = int(x_channels) if x_channels is not None else c_total
    f = int(cd.lstm_f)
    if tuple(w_hwio.shape) != (cd.kh, cd.kw, cin, 4 * f):
        raise ValueError('kernel shape %s does not match (kh,kw,cin,4F)=(%d,%d,%d,%d)' % (tuple(w_hwio.shape), cd.kh, cd.kw, cin, 4 * f))
    if cd.in_c_total == 0 and cin != c_total:
        cd.in_c_total = c_total
    xs = Shape4(n, cin, h, w)
    ys = conv_out_shape(xs, cd)
    if tuple(c_out.shape) != (n, f, ys.h, ys.w) or tuple(h_out.shape[2:]) != (ys.h, ys.w) or h_out.shape[0] != n:
        raise ValueError('convlstm_conv: c_out %s / h_out %s do not match (%d, %d, %d, %d)' %
                         (tuple(c_out.shape), tuple(h_out.shape), n, f, ys.h, ys.w))
    if z_add is not None and (z_add.dtype != torch.bfloat16 or tuple(z_add.shape) != (n, 4 * f, ys.h, ys.w)):
        raise ValueError('convlstm_conv: z_add must be bfloat16 (n, 4F, ho, wo)')
    if c_prev is not None and tuple(c_prev.shape) != tuple(c_out.shape):
        raise ValueError('convlstm_conv: c_prev shape')
    # out_o8: h_out AND z_add hold channel octets, c_prev / c_out float32 octets (n, F/8, h, w, 8)
    dt = _lib.dtype_io(_code(x, in_o8), _code(h_out, out_o8), compute_bf16)
    _lib.check(_lib.lib.dlwp_convlstm_conv_fwd(_lib.handle(_dev(x)), _ptr(x), _ptr(w_hwio), _ptr(prepared), _ptr(bias),
                                               _ptr(z_add), _ptr(c_prev), _ptr(c_out), _ptr(h_out), xs, ctypes.byref(cd), dt,
                                               _stream(x)))
    return h_out


def _step_args(h_seq, x, cd_h, cd_x, x_channels):
    n, _, hh, ww = h_seq.shape
    f = int(cd_h.lstm_f)
    if x.shape[0] != n or tuple(x.shape[2:]) != (hh, ww):
        raise ValueError('convlstm_step: state %r does not match the h sequence %r' % (tuple(x.shape), tuple(h_seq.shape)))
    if cd_h.in_c_total == 0 or cd_h.out_c_total == 0:
        cd_h.in_c_total = cd_h.out_c_total = h_seq.shape[1]
    if cd_x.in_c_total == 0:
        cd_x.in_c_total = x.shape[1]
    return Shape4(n, f, hh, ww), Shape4(n, int(x_channels), hh, ww), _lib.dtype_io(_lib.BF16_O8, _lib.BF16_O8)


def convlstm_step_supported(xs_h_chw, cd_h, xs_x_chw, cd_x):
    """Planner hint: can ONE launch run this ConvLSTM2D step (recurrent + input convolution + cell update,
    dlwp_convlstm_step_fwd)?  Octet layout only."""
    return bool(_lib.lib.dlwp_convlstm_step_supported(
        _lib.handle_or_none(), Shape4(1, *[int(v) for v in xs_h_chw]), ctypes.byref(cd_h), Shape4(1, *[int(v) for v in xs_x_chw]),
        ctypes.byref(cd_x), _lib.dtype_io(_lib.BF16_O8, _lib.BF16_O8)))


def convlstm_step_prepare(h_seq, x, w_h, w_x, cd_h, cd_x, x_channels):
    """Arranged weights of both kernels for convlstm_step calls of this geometry (dlwp_convlstm_step_prepare)."""
    _check_f32(w_h, w_x)
    xs_h, xs_x, dt = _step_args(h_seq, x, cd_h, cd_x, x_channels)
    hd = _lib.handle(_dev(h_seq))
    nbytes = _lib.lib.dlwp_convlstm_step_prepared_bytes(hd, xs_h, ctypes.byref(cd_h), xs_x, ctypes.byref(cd_x), dt)
    if nbytes == 0:
        raise _lib.DlwpError(_lib.EUNSUPPORTED, 'convlstm_step_prepare: this step has no dual-source instance')
    u = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=h_seq.device)
    _lib.check(_lib.lib.dlwp_convlstm_step_prepare(hd, _ptr(w_h), _ptr(w_x), _ptr(u), xs_h, ctypes.byref(cd_h), xs_x,
                                                   ctypes.byref(cd_x), dt, _stream(h_seq)))
    return u


def convlstm_step(h_seq, x, w_h, w_x, bias, cd_h, cd_x, c_prev, c_out, x_channels, prepared=None):
    """One ConvLSTM2D step t >= 1 in one launch (dlwp_convlstm_step_fwd): h_seq is the bfloat16 h sequence IN OCTETS -- cd_h's
    input window holds h_{t-1}, its output window receives h_t; x the float32 state (cd_x's channel window = x_t); c_prev / c_out
    the float32 cell state in octets."""
    _check_f32(w_h, w_x, bias, c_prev, c_out, x)
    if h_seq.dtype != torch.bfloat16 or not h_seq.is_contiguous():
        raise ValueError('convlstm_step: the h sequence must be a contiguous bfloat16 tensor (octet layout)')
    xs_h, xs_x, dt = _step_args(h_seq, x, cd_h, cd_x, x_channels)
    _lib.check(_lib.lib.dlwp_convlstm_step_fwd(_lib.handle(_dev(h_seq)), _ptr(h_seq), _ptr(x), _ptr(w_h), _ptr(w_x), _ptr(prepared),
                                               _ptr(bias), _ptr(c_prev), _ptr(c_out), _ptr(h_seq), xs_h, ctypes.byref(cd_h), xs_x,
                                               ctypes.byref(cd_x), dt, _stream(h_seq)))
    return h_seq


def maxpool2(x, out=None):
    _check_act(x, out)
    n, c, h, w = x.shape
    y = out if out is not None else torch.empty((n, c, h // 2, w // 2), dtype=x.dtype, device=x.device)
    if y.dtype != x.dtype:
        raise ValueError('maxpool2: input %s and output %s storage differ' % (x.dtype, y.dtype))
    _lib.check(_lib.lib.dlwp_maxpool2_fwd(_lib.handle(_dev(x)), _ptr(x), _ptr(y), Shape4(n, c, h, w), storage_code(x),
                                          _stream(x)))
    return y


def maxpool2_bwd(x, dy):
    _check_f32(x, dy)
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib.dlwp_maxpool2_bwd(_lib.handle(_dev(x)), _ptr(x), _ptr(dy), _ptr(dx), Shape4(n, c, h, w),
                                          _lib.F32, _stream(x)))
    return dx


def upsample2(x, out=None):
    _check_f32(x)
    n, c, h, w = x.shape
    y = out if out is not None else torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib.dlwp_upsample2_fwd(_lib.handle(_dev(x)), _ptr(x), _ptr(y), Shape4(n, c, h, w), _lib.F32,
                                           _stream(x)))
    return y


def upsample2_bwd(dy):
    _check_f32(dy)
    n, c, h2, w2 = dy.shape
    dx = torch.empty((n, c, h2 // 2, w2 // 2), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.lib.dlwp_upsample2_bwd(_lib.handle(_dev(dy)), _ptr(dy), _ptr(dx), Shape4(n, c, h2 // 2, w2 // 2),
                                           _lib.F32, _stream(dy)))
    return dx


def copy_channels(src, dst, c, src_c_off=0, dst_c_off=0):
    _check_f32(src, dst)
    n, sc, h, w = src.shape
    _lib.check(_lib.lib.dlwp_copy_channels(_lib.handle(_dev(src)), _ptr(src), _ptr(dst), n, int(c), h * w,
                                           int(src_c_off), sc, int(dst_c_off), dst.shape[1], _lib.F32, _stream(src)))
    return dst


REC_HARD_SIGMOID, REC_SIGMOID = 0, 1


def convlstm_gates(zx, zh, c_prev, c_out, h_out, f, h_c_off=0, act=1, rec_act=REC_HARD_SIGMOID):
    """ConvLSTM2D cell update: zx/zh (n, 4F, h, w) gate pre-activations (zh, c_prev may be None on the first step),
    c_out (n, F, h, w), h written to channels [h_c_off, +F) of h_out (n, h_c_total, h, w); h_out and zx / zh (both the
    same type) may be bfloat16 tensors (config 4 storage), the cell state is float32."""
    _check_f32(c_out)
    for t in (h_out, zx):
        if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous():
            raise ValueError('convlstm_gates: zx / h_out must be contiguous float32 or bfloat16 tensors')
    n, f4, h, w = zx.shape
    if f4 != 4 * f or tuple(c_out.shape) != (n, f, h, w) or tuple(h_out.shape[2:]) != (h, w) or h_out.shape[0] != n:
        raise ValueError('convlstm_gates: inconsistent shapes zx %r c_out %r h_out %r (F=%d)' %
                         (tuple(zx.shape), tuple(c_out.shape), tuple(h_out.shape), f))
    if c_prev is not None:
        _check_f32(c_prev)
    if zh is not None and (zh.dtype != zx.dtype or not zh.is_contiguous()):
        raise ValueError('convlstm_gates: zh must be stored like zx')
    nul = ctypes.c_void_p(0)
    _lib.check(_lib.lib.dlwp_convlstm_gates(_lib.handle(_dev(zx)), _ptr(zx), _ptr(zh) if zh is not None else nul,
                                            _ptr(c_prev) if c_prev is not None else nul, _ptr(c_out), _ptr(h_out), n,
                                            int(f), h * w, int(h_c_off), h_out.shape[1], int(act), int(rec_act),
                                            _lib.dtype_io(storage_code(zx), storage_code(h_out)), _stream(zx)))
    return h_out


def convlstm_gates_bwd(zx, zh, c_prev, c, dh, dc_in, f, h_c_off=0, act=1, rec_act=REC_HARD_SIGMOID, want_dc_prev=True):
    """Backward of convlstm_gates: returns (dz (n, 4F, h, w), dc_prev | None)."""
    _check_f32(zx, c, dh)
    n, f4, h, w = zx.shape
    dz = torch.empty_like(zx)
    dcp = torch.empty_like(c) if (c_prev is not None and want_dc_prev) else None
    _lib.check(_lib.lib.dlwp_convlstm_gates_bwd(_lib.handle(_dev(zx)), _ptr(zx), _ptr(zh), _ptr(c_prev), _ptr(c), _ptr(dh),
                                                _ptr(dc_in), _ptr(dz), _ptr(dcp), n, int(f), h * w, int(h_c_off),
                                                dh.shape[1], int(act), int(rec_act), _lib.F32, _stream(zx)))
    return dz, dcp


def series_merge_time(series, time_dim):
    """(T, N, time_dim*V, ...) -> (T*time_dim, N, V, ...)  -- DLWP/model/models.py:294-300."""
    _check_f32(series)
    t, n, c = series.shape[:3]
    v = c // time_dim
    rest = tuple(series.shape[3:])
    hw = 1
    for d in rest:
        hw *= d
    out = torch.empty((t * time_dim, n, v) + rest, dtype=series.dtype, device=series.device)
    _lib.check(_lib.lib.dlwp_series_merge_time(_lib.handle(_dev(series)), _ptr(series), _ptr(out), t, n, time_dim, v,
                                               hw, _lib.F32, _stream(series)))
    return out


def make_feedback(rows, state_c, out_c, hw, src, shift=0, tail=0, sol=None, sol_planes=0):
    """dlwp_feedback (include/dlwp_hip.h): src[c] >= 0 -> channel of the old state (row i + shift); -1 - j -> channel j of the
    model output of row i; sol[c] >= 0 -> plane of the tail rows' insolation block."""
    if state_c > _lib.FB_MAX_CHANNELS:
        raise ValueError('a fed rollout carries at most %d state channels, got %d' % (_lib.FB_MAX_CHANNELS, state_c))
    fb = _lib.Feedback()
    fb.rows, fb.state_c, fb.out_c, fb.hw = int(rows), int(state_c), int(out_c), int(hw)
    fb.shift, fb.tail, fb.sol_planes = int(shift), int(min(tail, rows)), int(sol_planes)
    for c in range(state_c):
        fb.src[c] = int(src[c])
        fb.sol[c] = int(sol[c]) if sol is not None else -1
    return fb


def state_feedback(old_state, out, fb, sol=None, mean=None, new_state=None):
    """The next input state of a forecast whose inputs and outputs differ (TimeSeriesEstimator.predict,
    DLWP/model/extensions.py:206-240; models.py:280-290): one launch, planes copied bit for bit.  old_state (rows, state_c, h, w),
    out (rows, out_c, h, w), sol (tail, sol_planes, h, w) | None, mean (state_c, h, w) | None."""
    _check_f32(old_state, out, sol, mean)
    if new_state is None:
        new_state = torch.empty_like(old_state)
    _check_f32(new_state)
    if old_state.numel() != fb.rows * fb.state_c * fb.hw or out.numel() != fb.rows * fb.out_c * fb.hw or \
            new_state.numel() != old_state.numel():
        raise ValueError('state_feedback: tensors of %d / %d elements for %d rows of %d -> %d channels x %d' %
                         (old_state.numel(), out.numel(), fb.rows, fb.state_c, fb.out_c, fb.hw))
    if sol is not None and sol.numel() != fb.tail * fb.sol_planes * fb.hw:
        raise ValueError('state_feedback: insolation block of %d elements, %d x %d x %d expected' %
                         (sol.numel(), fb.tail, fb.sol_planes, fb.hw))
    if mean is not None and mean.numel() != fb.state_c * fb.hw:
        raise ValueError('state_feedback: mean state of %d elements, %d x %d expected' % (mean.numel(), fb.state_c, fb.hw))
    _lib.check(_lib.lib.dlwp_state_feedback(_lib.handle(_dev(old_state)), _ptr(old_state), _ptr(out), _ptr(new_state), _ptr(sol),
                                            _ptr(mean), ctypes.byref(fb), _lib.F32, _stream(old_state)))
    return new_state


def conv_configs():
    """[(ks, dil, th, tw, waves, frags_per_wave, cout_frags, channel_chunk, pooled_loader, lds_bytes, flags)] of the
    compiled MFMA tiles (flags bit 0: position-split Winograd instance, dlwp_conv2d_config_flags)."""
    out = []
    info = (ctypes.c_int * 9)()
    lds = ctypes.c_int()
    for i in range(_lib.lib.dlwp_conv2d_num_configs()):
        _lib.check(_lib.lib.dlwp_conv2d_config_info(i, info, ctypes.byref(lds)))
        out.append(tuple(info) + (lds.value, int(_lib.lib.dlwp_conv2d_config_flags(i))))
    return out


def conv_launch_info(x_shape, cd, dtype=None, device_index=0):
    """What conv2d on a stored input of shape (n, cin, h, w) launches: [(config index, grid, block threads, executed
    matrix-core FLOPs, on the bf16 matrix cores?, input loader of a Winograd 8 x 32 launch: 0 elements / 1 column pairs / 2 source
    resolution)], one entry per kernel launch (dlwp_conv2d_launch_info)."""
    out = (_lib.LaunchInfo * 2)()
    n = ctypes.c_int(0)
    _lib.check(_lib.lib.dlwp_conv2d_launch_info(_lib.handle(device_index), Shape4(*[int(v) for v in x_shape]),
                                                ctypes.byref(cd), _lib.F32 if dtype is None else int(dtype), out,
                                                ctypes.byref(n)))
    return [(o.config, o.grid, o.block_threads, o.matrix_flops, bool(o.bf16_matrix), int(o.x_loader)) for o in out[:n.value]]


def force_conv_config(i):
    _lib.set_option(_lib.OPT_FORCE_CONV_CONFIG, int(i))


def set_few_stream(mode):
    """DLWP_OPT_FEW_STREAM: the streaming kernel for pooled 3x3 layers of at most four input channels (csrc/conv_fwd_few.hip):
    0 never, 1 from 8 tiles per workgroup on (default), 2 whenever the layer qualifies.  Returns the previous setting."""
    return int(_lib.set_option(_lib.OPT_FEW_STREAM, int(mode)))


def set_wino_xloader(mask):
    """DLWP_OPT_WINO_XLOADER: which input loaders the Winograd 8 x 32 instances may take -- bit 0 image-aligned column pairs, bit 1 an
    up-sampled source at source resolution, bit 2 edge pairs (the half-used last tile row's blocks take two column tiles each);
    7 = all (default), 0 = as before r5.  The same bits whatever the setting.  Returns
    the previous setting."""
    return int(_lib.set_option(_lib.OPT_WINO_XLOADER, int(mask)))


def set_splitk(mode):
    """DLWP_OPT_SPLITK: Winograd launches on small grids divide the input channels over several workgroups per output tile, the
    last arrival sums the partial tiles in index order (csrc/conv_fwd_k3d1s.hip): 0 never (default since r5), 1 by the library's rule,
    k >= 2 that many wherever the layer is eligible.  Returns the previous setting."""
    return int(_lib.set_option(_lib.OPT_SPLITK, int(mode)))


def conv_split_count(x_shape, cd, dtype=None, device_index=0):
    """The split regime of conv2d on a stored input of shape (n, cin, h, w): workgroups per output tile, 1 = unsplit
    (dlwp_conv2d_split_count).  Equal counts -> equal bits for a sample; across counts float32 round-off."""
    return int(_lib.lib.dlwp_conv2d_split_count(_lib.handle(device_index), Shape4(*[int(v) for v in x_shape]), ctypes.byref(cd),
                                                _lib.F32 if dtype is None else int(dtype)))


def prefers_unfused_pool(cin, cout, kh, kw, dil_h, dil_w):
    """Planner hint: materialise a MaxPooling2D in front of this convolution instead of fusing it into the loader?"""
    return bool(_lib.lib.dlwp_conv2d_prefers_unfused_pool(_lib.handle_or_none(), cin, cout, kh, kw, dil_h, dil_w))


def set_winograd(enable):
    """3x3 convolutions with >= 16 input and output channels run as Winograd F(2x2,3x3) by default."""
    _lib.set_option(_lib.OPT_WINOGRAD, 1 if enable else 0)


def set_bf16_mfma(enable):
    """Convolutions whose input is stored as bfloat16 multiply on the bf16 matrix cores (weights rounded to bf16) by
    default; False keeps them on the fp32 families.  Returns the previous setting."""
    return bool(_lib.set_option(_lib.OPT_BF16_MFMA, 1 if enable else 0))


def phase_geometry(k, pad):
    """(k2, lo, hi): the window of distinct source offsets [lo, hi] (size k2) that the k taps of a convolution on a 2x
    up-sampled axis reach, for the two output phases, with a top / left halo of `pad` on the up-sampled axis."""
    offs = [(a + u - pad) // 2 for a in (0, 1) for u in range(k)]
    return max(offs) - min(offs) + 1, min(offs), max(offs)


def phase_weights(w_hwio, bias, pad_top, pad_left, w2=None, b2=None):
    """Kernels of a Conv2D on a 2x up-sampled tensor restated on the tensor itself: (kh2, kw2, cin, 4*cout) with column
    (2a + b)*cout + co for output phase (a, b), and the bias repeated per phase (include/dlwp_hip.h: dlwp_phase_weights)."""
    _check_f32(w_hwio, bias)
    kh, kw, cin, cout = w_hwio.shape
    kh2, kw2 = phase_geometry(kh, pad_top)[0], phase_geometry(kw, pad_left)[0]
    if w2 is None:
        w2 = torch.empty((kh2, kw2, cin, 4 * cout), dtype=torch.float32, device=w_hwio.device)
    if b2 is None and bias is not None:
        b2 = torch.empty(4 * cout, dtype=torch.float32, device=w_hwio.device)
    _lib.check(_lib.lib.dlwp_phase_weights(_lib.handle(_dev(w_hwio)), _ptr(w_hwio), _ptr(bias), _ptr(w2), _ptr(b2), kh, kw,
                                           cin, cout, int(pad_top), int(pad_left), _lib.F32, _stream(w_hwio)))
    return w2, b2


def depth_to_space2(src, f, out=None, c_off=0):
    """(n, 4F, h, w) phase-major -> (n, F, 2h, 2w): out[:, c_off + co, 2i + a, 2j + b] = src[:, (2a + b)*F + co, i, j]."""
    _check_f32(src)
    n, c4, h, w = src.shape
    if c4 != 4 * f:
        raise ValueError('depth_to_space2: %d channels is not 4 x %d' % (c4, f))
    if out is None:
        out = torch.empty((n, f, 2 * h, 2 * w), dtype=torch.float32, device=src.device)
    _check_f32(out)
    if tuple(out.shape[2:]) != (2 * h, 2 * w) or out.shape[0] != n:
        raise ValueError('depth_to_space2: output shape %r' % (tuple(out.shape),))
    _lib.check(_lib.lib.dlwp_depth_to_space2(_lib.handle(_dev(src)), _ptr(src), _ptr(out), n, int(f), h, w, int(c_off),
                                             out.shape[1], _lib.F32, _stream(src)))
    return out


def space_to_depth2(src, f, c_off=0):
    """(n, c_total, 2h, 2w)[c_off:+f] -> (n, 4f, h, w) phase-major: the adjoint (= inverse) of depth_to_space2."""
    _check_f32(src)
    n, c_total, h2, w2 = src.shape
    out = torch.empty((n, 4 * f, h2 // 2, w2 // 2), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib.dlwp_space_to_depth2(_lib.handle(_dev(src)), _ptr(src), _ptr(out), n, int(f), h2 // 2, w2 // 2,
                                             int(c_off), c_total, _lib.F32, _stream(src)))
    return out


def phase_weights_bwd(dw2, db2, dw, db, pad_top, pad_left, accumulate=False):
    """Adjoint of phase_weights: dw (kh,kw,cin,cout) (+)= gather of dw2 over the 4 phases; db (+)= fold of db2."""
    _check_f32(dw2, dw, db2, db)
    kh, kw, cin, cout = dw.shape
    _lib.check(_lib.lib.dlwp_phase_weights_bwd(_lib.handle(_dev(dw2)), _ptr(dw2), _ptr(db2), _ptr(dw), _ptr(db), kh, kw, cin,
                                               cout, int(pad_top), int(pad_left), 1 if accumulate else 0, _lib.F32,
                                               _stream(dw2)))
    return dw


def uses_bf16_weights(xs, cd, dtype):
    """Does conv2d on an input of shape xs = (n, c, h, w) stored as `dtype` (a _lib.dtype_io code) multiply with weights
    rounded to bfloat16 (the bf16 matrix-core kernels)?  Host logic only."""
    return bool(_lib.lib.dlwp_conv2d_uses_bf16_weights(_lib.handle_or_none(), _lib.Shape4(*[int(v) for v in xs]),
                                                       ctypes.byref(cd), int(dtype)))


# ------------------------------------------------------------------------------------------------------------------ #
# training kernels
# ------------------------------------------------------------------------------------------------------------------ #

_workspaces = {}


def workspace(device, nbytes, key=None):
    """A growable per-device scratch allocation (bytes) for the *_bwd / loss kernels.  key: a scratch of its own for this
    caller (a deferred final sum reads its partials at dlwp_reductions_flush: nothing else may write there meanwhile)."""
    key = (device.type, device.index) if key is None else (device.type, device.index, key)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


_workspaces2 = {}


def workspace2(device, nbytes, key=None):
    """A second, small scratch allocation (reduction partials) that may be live next to `workspace`."""
    key = (device.type, device.index) if key is None else (device.type, device.index, key)
    ws = _workspaces2.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _workspaces2[key] = ws
    return ws


def conv_bwd_workspace_bytes(dev_index, xs, cd, which):
    out = ctypes.c_size_t()
    _lib.check(_lib.lib.dlwp_conv2d_bwd_workspace(_lib.handle(dev_index), xs, ctypes.byref(cd), int(which),
                                                  ctypes.byref(out)))
    return out.value


def conv2d_bwd_data(dz, w_hwio, cd, xs, dx, prepared=None, stored=False):
    """dz: (n, out_c_total, ho, wo); dx: preallocated gradient buffer (see include/dlwp_hip.h for its layout).
    prepared: the tensor conv2d_bwd_data_prepare built for (w_hwio, xs, cd, stored) -- the gradient's convolution is then the
    only launch; stored: gradient w.r.t. the STORED tensor of an up-sampled source (see conv2d_bwd_data_stored)."""
    _check_f32(dz, w_hwio, dx, prepared)
    d = _dev(dz)
    need = conv_bwd_workspace_bytes(d, xs, cd, 0)
    ws = workspace(dz.device, need)
    if prepared is not None:
        _lib.check(_lib.lib.dlwp_conv2d_bwd_data_prepared(_lib.handle(d), _ptr(dz), _ptr(prepared), _ptr(dx), xs,
                                                          ctypes.byref(cd), _lib.F32, _ptr(ws), ws.numel(),
                                                          1 if stored else 0, _stream(dz)))
        return dx
    fn = _lib.lib.dlwp_conv2d_bwd_data_stored if stored else _lib.lib.dlwp_conv2d_bwd_data
    _lib.check(fn(_lib.handle(d), _ptr(dz), _ptr(w_hwio), _ptr(dx), xs, ctypes.byref(cd), _lib.F32, _ptr(ws), ws.numel(),
                  _stream(dz)))
    return dx


def conv2d_bwd_data_act(dz, w_hwio, cd, xs, dx, x, act_in, db_in=None, prepared=None, ws_key=None):
    """conv2d_bwd_data whose result is multiplied by act'(x) in the store phase -- x: the layer's input = the activation output of
    the layer in front -- with that layer's bias gradient db_in (xs.c floats) from the same pass (dlwp_conv2d_bwd_data_act).
    Returns False, nothing written, where the gradient does not run on the instance with that store phase."""
    _check_f32(dz, w_hwio, dx, x, db_in, prepared)
    d = _dev(dz)
    need = conv_bwd_workspace_bytes(d, xs, cd, 3)
    ws = workspace(dz.device, need, ws_key)
    rc = _lib.lib.dlwp_conv2d_bwd_data_act(_lib.handle(d), _ptr(dz), _ptr(w_hwio), _ptr(prepared), _ptr(dx), xs, ctypes.byref(cd),
                                           _ptr(x), int(act_in), _ptr(db_in), _lib.F32, _ptr(ws), ws.numel(), _stream(dz))
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc)
    return True


def conv2d_bwd_data_prepared_bytes(dev_index, xs, cd, stored=False):
    """Bytes of the prepared operand of a data gradient (0: this gradient has no prepared form, e.g. `stored` on a layer
    without the summing epilogue)."""
    return int(_lib.lib.dlwp_conv2d_bwd_data_prepared_bytes(_lib.handle(dev_index), xs, ctypes.byref(cd), 1 if stored else 0))


def conv2d_bwd_data_prepare(w_hwio, cd, xs, stored=False, out=None):
    """The data gradient's operand from the layer's HWIO kernel (dlwp_conv2d_bwd_data_prepare); between prepare_begin /
    prepare_flush the work is only recorded."""
    _check_f32(w_hwio, out)
    d = _dev(w_hwio)
    nbytes = conv2d_bwd_data_prepared_bytes(d, xs, cd, stored)
    if nbytes == 0:
        return None
    u = out if out is not None else torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=w_hwio.device)
    if u.numel() * 4 < nbytes:
        raise ValueError('conv2d_bwd_data_prepare: the buffer holds %d bytes, %d needed' % (u.numel() * 4, nbytes))
    _lib.check(_lib.lib.dlwp_conv2d_bwd_data_prepare(_lib.handle(d), _ptr(w_hwio), _ptr(u), xs, ctypes.byref(cd),
                                                     1 if stored else 0, _stream(w_hwio)))
    return u


def _dev_index(device):
    return device.index if device.index is not None else torch.cuda.current_device()


def stream_wait(waiter, signaler):
    """`waiter` (a torch.cuda.Stream) waits for everything issued on `signaler` so far -- torch's Stream.wait_stream through the
    library (dlwp_stream_wait), so that a step being recorded (Trainer._record_step) keeps its fork / join edges."""
    dev = waiter.device.index if waiter.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.lib.dlwp_stream_wait(_lib.handle(dev), ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(signaler.cuda_stream)))


def prepare_begin(device):
    """Weight preparations from now on are recorded and built by ONE launch at prepare_flush (csrc/batch.hip)."""
    _lib.check(_lib.lib.dlwp_prepare_begin(_lib.handle(_dev_index(device))))


def _device_stream(device):
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(_dev_index(device)))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def prepare_flush(device):
    _lib.check(_lib.lib.dlwp_prepare_flush(_lib.handle(_dev_index(device)), _device_stream(device)))


def reductions_begin(device):
    """The final sums of conv2d_bwd_weight / the bias gradients / mse_mae from now on are recorded and done by ONE launch at
    reductions_flush; every such call needs a workspace of its own until then (ws_key)."""
    _lib.check(_lib.lib.dlwp_reductions_begin(_lib.handle(_dev_index(device))))


def reductions_flush(device):
    _lib.check(_lib.lib.dlwp_reductions_flush(_lib.handle(_dev_index(device)), _device_stream(device)))


def pair_begin(device):
    """From now until pair_end ONE conv2d_bwd_weight and ONE conv2d_bwd_data of INDEPENDENT tensors hand their launch over; pair_end
    issues both -- as one grid where a fused instance exists (dlwp_pair_begin / dlwp_pair_end, csrc/conv_pair.hip)."""
    _lib.check(_lib.lib.dlwp_pair_begin(_lib.handle(_dev_index(device))))


def pair_end(device):
    _lib.check(_lib.lib.dlwp_pair_end(_lib.handle(_dev_index(device)), _device_stream(device)))


def pair_fused_count(device):
    """pairs the device's handle has issued as ONE launch so far"""
    return int(_lib.lib.dlwp_pair_fused_count(_lib.handle(_dev_index(device))))


# ---- RowConnected2D (reference DLWP/custom.py:695-896) ------------------------------------------------------------------ #
def rowconv2d(x, kernel, bias, cd, out=None, direct=False, x_channels=None):
    """RowConnected2D.call / row_conv2d, channels_first: x stored (n, in_c_total, h, w); kernel (ho, kh, kw, cin, cout) --
    one filter set per OUTPUT ROW (custom.py:800-805); bias the stored (ho, 1, cout) array or None.  Returns
    (n, out_c_total, ho, wo), channels [out_c_off, +cout) written.  direct=True: the vector-ALU cross-check kernel."""
    _check_f32(x, kernel, bias, out)
    n, c_total, h, w = x.shape
    cin = int(x_channels) if x_channels is not None else c_total
    if cd.in_c_total == 0 and cin != c_total:
        cd.in_c_total = c_total
    xs = Shape4(n, cin, h, w)
    ys = conv_out_shape(xs, cd)
    if tuple(kernel.shape) != (ys.h, cd.kh, cd.kw, cin, cd.cout):
        raise ValueError('kernel shape %s does not match (rows,kh,kw,cin,cout)=%r' %
                         (tuple(kernel.shape), (ys.h, cd.kh, cd.kw, cin, cd.cout)))
    if bias is not None and bias.numel() != ys.h * cd.cout:
        raise ValueError('bias of %d elements, expected (rows, 1, cout) = %d' % (bias.numel(), ys.h * cd.cout))
    oc = cd.out_c_total if cd.out_c_total > 0 else ys.c
    if out is None:
        out = torch.empty((n, oc, ys.h, ys.w), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n, oc, ys.h, ys.w):
        raise ValueError('output buffer shape %s != %s' % (tuple(out.shape), (n, oc, ys.h, ys.w)))
    fn = _lib.lib.dlwp_rowconv2d_fwd_direct if direct else _lib.lib.dlwp_rowconv2d_fwd
    _lib.check(fn(_lib.handle(_dev(x)), _ptr(x), _ptr(kernel), _ptr(bias), _ptr(out), xs, ctypes.byref(cd), _lib.F32,
                  _stream(x)))
    return out


def rowconv2d_uses_matrix_cores(xs_nchw, cd, which=0):
    """Host logic: does pass `which` (0 forward, 1 data gradient, 2 weight gradient) of this geometry run on the MFMA kernels?"""
    return bool(_lib.lib.dlwp_rowconv2d_uses_matrix_cores(_lib.handle_or_none(), Shape4(*[int(v) for v in xs_nchw]),
                                                          ctypes.byref(cd), int(which)))


def rowconv2d_bwd_data(dz, kernel, cd, xs, dx):
    """dx: dense (n, cin, h, w) <- dL/dx of the row-connected layer (halo adjoint included)."""
    _check_f32(dz, kernel, dx)
    d = _dev(dz)
    need = ctypes.c_size_t(0)
    _lib.check(_lib.lib.dlwp_rowconv2d_bwd_workspace(_lib.handle(d), xs, ctypes.byref(cd), ctypes.byref(need)))
    ws = workspace(dz.device, need.value)
    _lib.check(_lib.lib.dlwp_rowconv2d_bwd_data(_lib.handle(d), _ptr(dz), _ptr(kernel), _ptr(dx), xs, ctypes.byref(cd),
                                                _lib.F32, _ptr(ws), ws.numel(), _stream(dz)))
    return dx


def rowconv2d_bwd_weight(x, dz, dw, db, cd, xs, accumulate=False):
    """dw (rows, kh, kw, cin, cout), db (rows, 1, cout) or None <- gradients of the kernel and the stored bias."""
    _check_f32(x, dz, dw, db)
    _lib.check(_lib.lib.dlwp_rowconv2d_bwd_weight(_lib.handle(_dev(x)), _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), xs,
                                                  ctypes.byref(cd), 1 if accumulate else 0, _lib.F32, _stream(x)))
    return dw


def conv2d_bwd_data_stored(dz, w_hwio, cd, xs, dx):
    """Up-sampled sources: gradient w.r.t. the stored tensor (n, cin, xs.h, xs.w) in one kernel.  Returns False when the
    layer has no kernel with the summing epilogue (nothing was written; use conv2d_bwd_data + upsample2_bwd)."""
    _check_f32(dz, w_hwio, dx)
    d = _dev(dz)
    need = conv_bwd_workspace_bytes(d, xs, cd, 0)
    ws = workspace(dz.device, need)
    rc = _lib.lib.dlwp_conv2d_bwd_data_stored(_lib.handle(d), _ptr(dz), _ptr(w_hwio), _ptr(dx), xs, ctypes.byref(cd),
                                              _lib.F32, _ptr(ws), ws.numel(), _stream(dz))
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc)
    return True


def conv2d_bwd_weight(x, dz, dw, cd, xs, accumulate=False, ws_key=None):
    _check_f32(x, dz, dw)
    d = _dev(x)
    need = conv_bwd_workspace_bytes(d, xs, cd, 1)
    ws = workspace(x.device, need, ws_key)
    _lib.check(_lib.lib.dlwp_conv2d_bwd_weight(_lib.handle(d), _ptr(x), _ptr(dz), _ptr(dw), xs, ctypes.byref(cd),
                                               int(bool(accumulate)), _lib.F32, _ptr(ws), ws.numel(), _stream(x)))
    return dw


def conv2d_bwd_weight_pooled_supported(xs, cd, dev_index=None):
    """Can dlwp_conv2d_bwd_weight_pooled take this layer (3x3, at most 4 input channels)?"""
    out = ctypes.c_size_t()
    d = torch.cuda.current_device() if dev_index is None else dev_index
    return _lib.lib.dlwp_conv2d_bwd_workspace(_lib.handle(d), xs, ctypes.byref(cd), 2, ctypes.byref(out)) == _lib.OK


def conv2d_bwd_weight_pooled(x, y, dpool, dw, db, cd, xs, act, accumulate=False, ws_key=None):
    """dw (and db, unless None) of a layer whose only reader is MaxPooling2D(2) and whose data gradient nobody needs, from its
    output y and the pooled tensor's gradient dpool (n, cout, ho/2, wo/2): the gradient tensor in between is never stored
    (include/dlwp_hip.h: dlwp_conv2d_bwd_weight_pooled)."""
    _check_f32(x, y, dpool, dw, db)
    n, cout, hp, wp = dpool.shape
    if cout != cd.cout or not dpool.is_contiguous() or (hp, wp) != (y.shape[2] // 2, y.shape[3] // 2):
        raise ValueError('conv2d_bwd_weight_pooled: pooled gradient %r does not match the layer output %r' %
                         (tuple(dpool.shape), tuple(y.shape)))
    d = _dev(x)
    need = conv_bwd_workspace_bytes(d, xs, cd, 2)
    ws = workspace(x.device, need, ws_key)
    _lib.check(_lib.lib.dlwp_conv2d_bwd_weight_pooled(_lib.handle(d), _ptr(x), _ptr(y), _ptr(dpool), _ptr(dw), _ptr(db), xs,
                                                     ctypes.byref(cd), int(act), int(bool(accumulate)), _lib.F32, _ptr(ws),
                                                     ws.numel(), _stream(x)))
    return dw


def act_bwd(y, dy, act, out=None):
    _check_f32(y, dy)
    dz = out if out is not None else torch.empty_like(dy)
    _lib.check(_lib.lib.dlwp_act_bwd(_lib.handle(_dev(dy)), _ptr(y), _ptr(dy), _ptr(dz), dy.numel(), int(act), _lib.F32,
                                     _stream(dy)))
    return dz


def bias_grad(dz, db, c, c_off=0, ws_key=None):
    _check_f32(dz, db)
    n, c_total, h, w = dz.shape
    ws = workspace2(dz.device, _lib.lib.dlwp_bias_grad_workspace(int(c)), ws_key)
    _lib.check(_lib.lib.dlwp_bias_grad(_lib.handle(_dev(dz)), _ptr(dz), _ptr(db), n, int(c), int(c_off), c_total, h * w,
                                       _ptr(ws), ws.numel(), _lib.F32, _stream(dz)))
    return db


def act_bwd_bias_grad(y, dy, act, db, c, c_off=0, out=None, ws_key=None):
    """dz = dy * act'(y) on channels [c_off, c_off + c) and db = sum of dz over (n, h, w), in one pass (dz may be dy)."""
    _check_f32(y, dy, db)
    dz = out if out is not None else torch.empty_like(dy)
    n, c_total, h, w = dy.shape
    ws = workspace2(dy.device, _lib.lib.dlwp_bias_grad_workspace(int(c)), ws_key)
    _lib.check(_lib.lib.dlwp_act_bwd_bias_grad(_lib.handle(_dev(dy)), _ptr(y), _ptr(dy), _ptr(dz), _ptr(db), n, int(c),
                                               int(c_off), c_total, h * w, int(act), _ptr(ws), ws.numel(), _lib.F32,
                                               _stream(dy)))
    return dz


def pool_act_bwd_bias_grad(y, dp, act, db=None, ws_key=None):
    """Backward of MaxPooling2D(2) + activation (+ bias gradient) of the convolution that produced y, in one pass:
    returns dz (shape of y) from the pooled tensor's gradient dp."""
    _check_f32(y, dp, db)
    n, c, h, w = y.shape
    assert tuple(dp.shape) == (n, c, h // 2, w // 2) and y.is_contiguous() and dp.is_contiguous()
    dz = torch.empty_like(y)
    ws = workspace2(y.device, _lib.lib.dlwp_bias_grad_workspace(int(c)), ws_key)
    _lib.check(_lib.lib.dlwp_pool_act_bwd_bias_grad(_lib.handle(_dev(y)), _ptr(y), _ptr(dp), _ptr(dz),
                                                    _ptr(db), _lib.Shape4(n, c, h, w), int(act),
                                                    _ptr(ws), ws.numel(), _lib.F32, _stream(y)))
    return dz


def mse_mae(y_pred, y_true, out2, dy=None, loss_weight=1.0, ws_key=None):
    """out2 (device, 2 floats) <- [mse, mae]; dy <- loss_weight * 2 (y_pred - y_true) / numel."""
    _check_f32(y_pred, y_true, out2, dy)
    d = _dev(y_pred)
    h = _lib.handle(d)
    need = _lib.lib.dlwp_mse_mae_workspace(h)
    ws = workspace(y_pred.device, need, ws_key)
    _lib.check(_lib.lib.dlwp_mse_mae(h, _ptr(y_pred), _ptr(y_true), y_pred.numel(), _ptr(out2), _ptr(dy),
                                     float(loss_weight), _ptr(ws), ws.numel(), _lib.F32, _stream(y_pred)))
    return out2


def mse_mae_phase(y_phase, y_true, out2, dz=None, db=None, loss_weight=1.0, ws_key=None):
    """dlwp_mse_mae on the phase channels (n, 4f, h, w) of a restated output layer against the target (n, f, 2h, 2w); dz
    (phase layout) and db (4f: the sums of dz) are optional outputs."""
    _check_f32(y_phase, y_true, out2, dz, db)
    n, f4, hh, ww = y_phase.shape
    f = f4 // 4
    if f4 != 4 * f or tuple(y_true.shape) != (n, f, 2 * hh, 2 * ww):
        raise ValueError('mse_mae_phase: phase tensor %r does not match the target %r' % (tuple(y_phase.shape), tuple(y_true.shape)))
    h = _lib.handle(_dev(y_phase))
    ws = workspace(y_phase.device, _lib.lib.dlwp_mse_mae_phase_workspace(f), ws_key)
    _lib.check(_lib.lib.dlwp_mse_mae_phase(h, _ptr(y_phase), _ptr(y_true), n, f, hh, ww, _ptr(out2), _ptr(dz), _ptr(db),
                                           float(loss_weight), _ptr(ws), ws.numel(), _lib.F32, _stream(y_phase)))
    return out2


def loss_custom(y_pred, y_true, stats7, dy=None, loss_weight=1.0, mean=None, row_weights=None, kind=0, regularize=0):
    """The reference's custom losses (include/dlwp_hip.h: dlwp_loss_custom).  y: (n, c, h, w) device tensors."""
    _check_f32(y_pred, y_true, stats7, dy, mean, row_weights)
    n, c, hh, ww = y_pred.shape
    d = _dev(y_pred)
    h = _lib.handle(d)
    ws = workspace(y_pred.device, _lib.lib.dlwp_loss_workspace(h, n, c))
    _lib.check(_lib.lib.dlwp_loss_custom(h, _ptr(y_pred), _ptr(y_true), n, c, hh, ww, _ptr(mean), _ptr(row_weights),
                                         int(kind), int(regularize), _ptr(stats7), _ptr(dy), float(loss_weight), _ptr(ws),
                                         ws.numel(), _lib.F32, _stream(y_pred)))
    return stats7


def adam_keras(p, m, v, g, iteration, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, decay=0.0, grad_scale=1.0):
    _check_f32(p, m, v, g)
    _lib.check(_lib.lib.dlwp_adam_keras(_lib.handle(_dev(p)), _ptr(p), _ptr(m), _ptr(v), _ptr(g), p.numel(), lr, beta_1,
                                        beta_2, epsilon, decay, int(iteration), grad_scale, _stream(p)))


def adam_keras_dev(p, m, v, g, iteration_dev, lr_t_scratch, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, decay=0.0,
                   grad_scale=1.0):
    """adam_keras with the step number in device memory (int64 tensor of one element, advanced by the call) -- the form a
    captured training step replays."""
    _check_f32(p, m, v, g, lr_t_scratch)
    if iteration_dev.dtype != torch.int64 or iteration_dev.numel() != 1:
        raise ValueError('iteration_dev must be one int64 on the device')
    _lib.check(_lib.lib.dlwp_adam_keras_dev(_lib.handle(_dev(p)), _ptr(p), _ptr(m), _ptr(v), _ptr(g), p.numel(), lr, beta_1,
                                            beta_2, epsilon, decay, _ptr(iteration_dev), _ptr(lr_t_scratch), grad_scale,
                                            _stream(p)))


def sgd_keras(p, vel, g, iteration, lr=0.01, momentum=0.0, decay=0.0, grad_scale=1.0):
    _check_f32(p, vel, g)
    _lib.check(_lib.lib.dlwp_sgd_keras(_lib.handle(_dev(p)), _ptr(p), _ptr(vel), _ptr(g), p.numel(), lr, momentum, decay,
                                       int(iteration), grad_scale, _stream(p)))


def copy_many(pairs):
    """[(src, dst), ...] (at most 8 contiguous float32 tensors of equal element counts per pair) copied by ONE launch."""
    pairs = list(pairs)
    if not pairs:
        return
    for s, d in pairs:
        _check_f32(s, d)
        if s.numel() != d.numel() or not s.is_contiguous() or not d.is_contiguous():
            raise ValueError('copy_many: contiguous tensors of equal size, please')
    k = len(pairs)
    srcs = (ctypes.c_void_p * k)(*[s.data_ptr() for s, _ in pairs])
    dsts = (ctypes.c_void_p * k)(*[d.data_ptr() for _, d in pairs])
    cnt = (ctypes.c_size_t * k)(*[s.numel() for s, _ in pairs])
    dst0 = pairs[0][1]
    _lib.check(_lib.lib.dlwp_copy_many(_lib.handle(_dev(dst0)), srcs, dsts, cnt, k, _stream(dst0)))


def axpby(x, y, a=1.0, b=1.0):
    """y <- a*x + b*y"""
    _check_f32(x, y)
    _lib.check(_lib.lib.dlwp_axpby(_lib.handle(_dev(x)), _ptr(x), _ptr(y), x.numel(), float(a), float(b), _stream(x)))
    return y


def wgrad_configs():
    """[(ks, dil, th, tw, cout_frags, waves, lds_bytes)] of the compiled weight-gradient tiles."""
    out = []
    info = (ctypes.c_int * 6)()
    lds = ctypes.c_int()
    for i in range(_lib.lib.dlwp_conv2d_wgrad_num_configs()):
        _lib.check(_lib.lib.dlwp_conv2d_wgrad_config_info(i, info, ctypes.byref(lds)))
        out.append(tuple(info) + (lds.value,))
    return out


def wgrad_config_forms():
    """[(input channels per workgroup, form)] per compiled weight-gradient instance; form 0 direct, 1 Winograd, 3 channel-block
    Winograd, 4 the streaming form for at most 4 input channels."""
    out = []
    cib, form = ctypes.c_int(), ctypes.c_int()
    for i in range(_lib.lib.dlwp_conv2d_wgrad_num_configs()):
        _lib.check(_lib.lib.dlwp_conv2d_wgrad_config_form(i, ctypes.byref(cib), ctypes.byref(form)))
        out.append((cib.value, form.value))
    return out


def force_wgrad_config(i):
    _lib.set_option(_lib.OPT_FORCE_WGRAD_CONFIG, int(i))
