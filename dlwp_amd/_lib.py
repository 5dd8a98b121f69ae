"""
ctypes binding of libdlwp_hip.so (include/dlwp_hip.h).  There is NO fallback: if the HIP library is missing or fails to
load, importing this module raises -- the product never computes on the CPU.
"""
import ctypes
import os
import re

# torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It MUST be in the
# process before libdlwp_hip.so is loaded so that our DT_NEEDED libamdhip64.so.7 binds to that same instance: two HIP
# runtimes in one process cannot both open the device ("no ROCm-capable device is detected").
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DLWP_LIB_PATH: profiling builds of the same library, tools/knockout_bf16.sh -- never a different implementation)
LIB_PATH = os.environ.get('DLWP_LIB_PATH') or os.path.join(_HERE, 'libdlwp_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'dlwp_hip.h')

OK, EINVAL, EUNSUPPORTED, EHIP, ERCCL = 0, -1, -2, -3, -4
OPT_WINOGRAD, OPT_BF16_MFMA, OPT_FORCE_CONV_CONFIG, OPT_FORCE_WGRAD_CONFIG, OPT_WINO_PAIRS, OPT_WGRAD_FILL = 0, 1, 2, 3, 4, 5
OPT_FEW_STREAM = 6
OPT_SPLITK = 7
OPT_WINO_XLOADER = 8
F32, BF16, BF16_O8 = 0, 1, 2
ROLLOUT_PREPARED = 0x100
PAD_ZERO, PAD_WRAP, PAD_EDGE, PAD_REFLECT, PAD_SYMMETRIC = 0, 1, 2, 3, 4
ACT_LINEAR, ACT_TANH, ACT_RELU = 0, 1, 2
SRC_DIRECT, SRC_UPSAMPLE2, SRC_MAXPOOL2 = 0, 1, 2
OP_CONV2D, OP_PAD2D, OP_MAXPOOL2, OP_UPSAMPLE2, OP_COPYCH, OP_LSTM_GATES, OP_PHASE_WEIGHTS, OP_DEPTH2SPACE = 0, 1, 2, 3, 4, 5, 6, 7
OP_ROWCONV2D = 8
BUF_NONE = -1000
STEP_LANES, STEP_GRAPH, STEP_GRAPH_BRANCHES, STEP_LANES_RECORDED = 0, 1, 2, 3


COMPUTE_BF16 = 0x20000


def dtype_io(dt_in, dt_out, compute_bf16=False):
    """DLWP_DTYPE_IO(in, out) of include/dlwp_hip.h: storage of a launch's input / output activations;
    compute_bf16: | DLWP_COMPUTE_BF16 (a float32-stored convolution input may be rounded to bfloat16)"""
    return 0x10000 | dt_in | (dt_out << 8) | (COMPUTE_BF16 if compute_bf16 else 0)
BUF_STATE_IN = -1


def BUF_OUT(o):
    return -2 - o


class Shape4(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int), ('c', ctypes.c_int), ('h', ctypes.c_int), ('w', ctypes.c_int)]


class Pad2d(ctypes.Structure):
    _fields_ = [('top', ctypes.c_int), ('bottom', ctypes.c_int), ('left', ctypes.c_int), ('right', ctypes.c_int),
                ('mode_h', ctypes.c_int), ('mode_w', ctypes.c_int)]


class Conv2d(ctypes.Structure):
    _fields_ = [('cout', ctypes.c_int), ('kh', ctypes.c_int), ('kw', ctypes.c_int), ('dil_h', ctypes.c_int),
                ('dil_w', ctypes.c_int), ('halo', Pad2d), ('act', ctypes.c_int), ('in_c_off', ctypes.c_int),
                ('in_c_total', ctypes.c_int), ('out_c_off', ctypes.c_int), ('out_c_total', ctypes.c_int),
                ('src_mode', ctypes.c_int), ('out_pool', ctypes.c_int), ('out_d2s', ctypes.c_int),
                ('lstm_f', ctypes.c_int), ('lstm_rec_act', ctypes.c_int)]


class Op(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int), ('src', ctypes.c_int), ('dst', ctypes.c_int), ('w', ctypes.c_int),
                ('b', ctypes.c_int), ('xs', Shape4), ('conv', Conv2d), ('pad', Pad2d), ('aux', ctypes.c_int * 4),
                ('src2', ctypes.c_int), ('w2', ctypes.c_int), ('xs2_c', ctypes.c_int), ('conv2', Conv2d)]


FB_MAX_CHANNELS = 128


class Feedback(ctypes.Structure):
    """dlwp_feedback (include/dlwp_hip.h): the state update between two model calls of a fed rollout"""
    _fields_ = [('rows', ctypes.c_int), ('state_c', ctypes.c_int), ('out_c', ctypes.c_int), ('hw', ctypes.c_int),
                ('shift', ctypes.c_int), ('tail', ctypes.c_int), ('sol_planes', ctypes.c_int),
                ('src', ctypes.c_int * FB_MAX_CHANNELS), ('sol', ctypes.c_int * FB_MAX_CHANNELS)]


class LaunchInfo(ctypes.Structure):
    _fields_ = [('config', ctypes.c_int), ('grid', ctypes.c_int), ('block_threads', ctypes.c_int),
                ('matrix_flops', ctypes.c_double), ('bf16_matrix', ctypes.c_int), ('x_loader', ctypes.c_int)]


class DlwpError(RuntimeError):
    """A non-zero status from libdlwp_hip.so (message from dlwp_last_error())."""

    def __init__(self, code, message):
        super(DlwpError, self).__init__('libdlwp_hip error %d: %s' % (code, message))
        self.code = code


def kernel_source_hash():
    """sha256[:16] over dlwp_amd/csrc/*.{h,hip}: profiles/*.json summaries carry it so that a counter measurement is
    only quoted for the kernel source it was taken on (bench.py, tools/parse_pmc.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.h')) + glob.glob(os.path.join(_HERE, 'csrc', '*.hip'))):
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def declared_symbols(header_path=HEADER_PATH):
    """Every function name include/dlwp_hip.h declares (used by the CPU test that checks the .so exports them all)."""
    text = open(header_path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dlwp_[a-z0-9_]+)\s*\(', text)))


if not os.path.exists(LIB_PATH):
    raise ImportError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` or '
                      '`make -C dlwp_amd/csrc`.  dlwp_amd has no CPU fallback.' % LIB_PATH)
lib = ctypes.CDLL(LIB_PATH)

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_P = ctypes.POINTER


def _sig(name, argtypes, restype=ctypes.c_int):
    fn = getattr(lib, name)
    fn.argtypes, fn.restype = argtypes, restype
    return fn


_sig('dlwp_version', [])
_sig('dlwp_last_error', [], ctypes.c_char_p)
_sig('dlwp_create', [_P(_vp), _i])
_sig('dlwp_destroy', [_vp])
_sig('dlwp_device_info', [_vp, _P(_i), _P(_i), ctypes.c_char_p, _sz])
_sig('dlwp_set_option', [_vp, _i, _i, _P(_i)])
_sig('dlwp_set_default_option', [_i, _i, _P(_i)])
_sig('dlwp_pad2d_fwd', [_vp, _vp, _vp, _i, _i, _i, _i, Pad2d, _i, _vp])
_sig('dlwp_pad2d_bwd', [_vp, _vp, _vp, _i, _i, _i, _i, Pad2d, _i, _vp])
_sig('dlwp_conv2d_out_shape', [Shape4, _P(Conv2d), _P(Shape4)])
_sig('dlwp_conv2d_fwd', [_vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_conv2d_fwd_pool2', [_vp, _vp, _vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_conv2d_fwd_direct', [_vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_conv2d_prepared_bytes', [_vp, Shape4, _P(Conv2d), _i], _sz)
_sig('dlwp_conv2d_prepare', [_vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_conv2d_fwd_prepared', [_vp, _vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_conv2d_num_configs', [])
_sig('dlwp_conv2d_config_info', [_i, _P(_i), _P(_i)])
_sig('dlwp_conv2d_config_flags', [_i])
_sig('dlwp_phase_geometry', [_i, _i, _P(_i), _P(_i), _P(_i)])
_sig('dlwp_phase_weights', [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_depth_to_space2', [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_space_to_depth2', [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_phase_weights_bwd', [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_conv2d_uses_bf16_weights', [_vp, Shape4, _P(Conv2d), _i])
_sig('dlwp_conv2d_prefers_unfused_pool', [_vp, _i, _i, _i, _i, _i, _i])
_sig('dlwp_conv2d_supports_out_pool', [_vp, Shape4, _P(Conv2d)])
_sig('dlwp_conv2d_supports_dtype', [_vp, Shape4, _P(Conv2d), _i])
_sig('dlwp_conv2d_supports_out_d2s', [_vp, Shape4, _P(Conv2d)])
_sig('dlwp_conv2d_pick_config', [_vp, Shape4, _P(Conv2d)])
_sig('dlwp_conv2d_launch_info', [_vp, Shape4, _P(Conv2d), _i, _P(LaunchInfo), _P(_i)])
_sig('dlwp_conv2d_split_count', [_vp, Shape4, _P(Conv2d), _i])
_sig('dlwp_conv2d_bwd_workspace', [_vp, Shape4, _P(Conv2d), _i, _P(_sz)])
_sig('dlwp_conv2d_bwd_data', [_vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp, _sz, _vp])
_sig('dlwp_conv2d_bwd_data_stored', [_vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp, _sz, _vp])
_sig('dlwp_conv2d_bwd_weight', [_vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _i, _vp, _sz, _vp])
_sig('dlwp_conv2d_bwd_data_act', [_vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _vp, _i, _vp, _i, _vp, _sz, _vp])
_sig('dlwp_conv2d_bwd_weight_pooled', [_vp, _vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _i, _i, _vp, _sz, _vp])
_sig('dlwp_conv2d_wgrad_num_configs', [])
_sig('dlwp_conv2d_wgrad_config_info', [_i, _P(_i), _P(_i)])
_sig('dlwp_conv2d_wgrad_config_form', [_i, _P(_i), _P(_i)])
_sig('dlwp_conv2d_wgrad_pick_config', [_vp, Shape4, _P(Conv2d)])
_sig('dlwp_rowconv2d_fwd', [_vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_rowconv2d_fwd_direct', [_vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_rowconv2d_uses_matrix_cores', [_vp, Shape4, _P(Conv2d), _i])
_sig('dlwp_rowconv2d_bwd_workspace', [_vp, Shape4, _P(Conv2d), _P(_sz)])
_sig('dlwp_rowconv2d_bwd_data', [_vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp, _sz, _vp])
_sig('dlwp_rowconv2d_bwd_weight', [_vp, _vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _i, _vp])
_sig('dlwp_act_bwd', [_vp, _vp, _vp, _vp, _sz, _i, _i, _vp])
_sig('dlwp_bias_grad_workspace', [_i], _sz)
_sig('dlwp_bias_grad', [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp])
_sig('dlwp_act_bwd_bias_grad', [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp])
_sig('dlwp_pool_act_bwd_bias_grad', [_vp, _vp, _vp, _vp, _vp, Shape4, _i, _vp, _sz, _i, _vp])
_sig('dlwp_mse_mae_workspace', [_vp], _sz)
_sig('dlwp_mse_mae', [_vp, _vp, _vp, _sz, _vp, _vp, ctypes.c_float, _vp, _sz, _i, _vp])
_sig('dlwp_mse_mae_phase_workspace', [_i], _sz)
_sig('dlwp_mse_mae_phase', [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _sz, _i, _vp])
_sig('dlwp_loss_workspace', [_vp, _i, _i], _sz)
_sig('dlwp_loss_custom', [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, ctypes.c_float, _vp, _sz, _i, _vp])
_sig('dlwp_adam_keras', [_vp, _vp, _vp, _vp, _vp, _sz] + [ctypes.c_float] * 5 + [ctypes.c_longlong, ctypes.c_float, _vp])
_sig('dlwp_copy_many', [_vp, _vp, _vp, _vp, _i, _vp])
_sig('dlwp_adam_keras_dev', [_vp, _vp, _vp, _vp, _vp, _sz] + [ctypes.c_float] * 5 + [_vp, _vp, ctypes.c_float, _vp])
_sig('dlwp_sgd_keras', [_vp, _vp, _vp, _vp, _sz] + [ctypes.c_float] * 3 + [ctypes.c_longlong, ctypes.c_float, _vp])
_sig('dlwp_axpby', [_vp, _vp, _vp, _sz, ctypes.c_float, ctypes.c_float, _vp])
_sig('dlwp_prepare_begin', [_vp])
_sig('dlwp_prepare_flush', [_vp, _vp])
_sig('dlwp_reductions_begin', [_vp])
_sig('dlwp_reductions_flush', [_vp, _vp])
_sig('dlwp_pair_begin', [_vp])
_sig('dlwp_pair_end', [_vp, _vp])
_sig('dlwp_pair_fused_count', [_vp], ctypes.c_longlong)
_sig('dlwp_conv2d_bwd_data_prepared_bytes', [_vp, Shape4, _P(Conv2d), _i], _sz)
_sig('dlwp_conv2d_bwd_data_prepare', [_vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_conv2d_bwd_data_prepared', [_vp, _vp, _vp, _vp, Shape4, _P(Conv2d), _i, _vp, _sz, _i, _vp])
_sig('dlwp_maxpool2_fwd', [_vp, _vp, _vp, Shape4, _i, _vp])
_sig('dlwp_maxpool2_bwd', [_vp, _vp, _vp, _vp, Shape4, _i, _vp])
_sig('dlwp_upsample2_fwd', [_vp, _vp, _vp, Shape4, _i, _vp])
_sig('dlwp_upsample2_bwd', [_vp, _vp, _vp, Shape4, _i, _vp])
_sig('dlwp_convlstm_gates', [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_convlstm_conv_fwd', [_vp] * 9 + [Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_convlstm_conv_supported', [_vp, Shape4, _P(Conv2d), _i])
_sig('dlwp_convlstm_step_supported', [_vp, Shape4, _P(Conv2d), Shape4, _P(Conv2d), _i])
_sig('dlwp_convlstm_step_prepared_bytes', [_vp, Shape4, _P(Conv2d), Shape4, _P(Conv2d), _i], _sz)
_sig('dlwp_convlstm_step_prepare', [_vp, _vp, _vp, _vp, Shape4, _P(Conv2d), Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_convlstm_step_fwd', [_vp] * 10 + [Shape4, _P(Conv2d), Shape4, _P(Conv2d), _i, _vp])
_sig('dlwp_convlstm_gates_bwd', [_vp] * 9 + [_i] * 8 + [_vp])
_sig('dlwp_copy_channels', [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_series_merge_time', [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp])
_sig('dlwp_rollout_workspace_bytes', [_vp, _P(Op), _i, _i], _sz)
_sig('dlwp_rollout_create', [_vp, _P(Op), _i, _P(_vp), _i, _vp, _vp, _sz, _i, _i, _i, _vp, _sz, _P(_vp)])
_sig('dlwp_rollout_create_grouped', [_vp, _P(Op), _i, _P(_vp), _i, _P(_sz), _i, _vp, _vp, _sz, _i, _i, _i, _vp, _sz, _P(_vp)])
_sig('dlwp_state_feedback', [_vp, _vp, _vp, _vp, _vp, _vp, _P(Feedback), _i, _vp])
_sig('dlwp_rollout_create_fed', [_vp, _P(Op), _i, _P(_vp), _i, _vp, _vp, _vp, _sz, _i, _i, _i, _P(Feedback), _vp, _vp, _i, _vp, _sz, _P(_vp)])
_sig('dlwp_series_arrange', [_vp, _vp, _vp, _i, _i, _i, _i, _i, _P(_i), _i, _i, _vp])
_sig('dlwp_rollout_launch', [_vp, _vp])
_sig('dlwp_rollout_destroy', [_vp])
_sig('dlwp_host_gather_rows', [_vp, _vp, _vp, ctypes.c_longlong, _sz, ctypes.c_longlong, _i])
_sig('dlwp_train_step_record_begin', [_vp, _vp])
_sig('dlwp_train_step_record_abort', [_vp])
_sig('dlwp_stream_wait', [_vp, _vp, _vp])
_sig('dlwp_train_step_create', [_vp, _i, _P(_vp), _P(_sz), _P(_vp)])
_sig('dlwp_train_step_info', [_vp, _P(_i), _P(_i), _P(_i)])
_sig('dlwp_train_step_launch', [_vp, _P(_vp), _i, _vp])
_sig('dlwp_train_step_destroy', [_vp])
_sig('dlwp_comm_unique_id', [_vp, _P(_sz)])
_sig('dlwp_comm_init_rank', [_P(_vp), _i, _i, _i, _vp, _sz])
_sig('dlwp_comm_info', [_vp, _P(_i), _P(_i), _P(_i)])
_sig('dlwp_allreduce_sum_f32', [_vp, _vp, _sz, _vp])
_sig('dlwp_broadcast_f32', [_vp, _vp, _sz, _i, _vp])
_sig('dlwp_comm_destroy', [_vp])
_sig('dlwp_set_crash_message', [ctypes.c_char_p])
_sig('dlwp_spin', [_vp, _i, _vp])
_sig('dlwp_xchg_create', [_vp, _i, _i, _sz, _vp, _P(_vp)])
_sig('dlwp_xchg_connect', [_vp, _vp])
_sig('dlwp_xchg_allreduce_sum_f32', [_vp, _vp, _sz, _vp])
_sig('dlwp_xchg_allreduce_adam', [_vp, _vp, _sz, _sz, _vp, _vp, _vp] + [ctypes.c_float] * 5 + [ctypes.c_longlong, ctypes.c_float, _vp])
_sig('dlwp_xchg_status', [_vp, _P(_i)])
_sig('dlwp_xchg_info', [_vp, _P(_i), _P(_i)])
_sig('dlwp_xchg_destroy', [_vp])


def check(rc):
    if rc != OK:
        raise DlwpError(rc, lib.dlwp_last_error().decode('utf-8', 'replace'))
    return rc


_handles = {}


#: held while a training step is being stream-captured (Trainer._capture_step) and by every other thread of this package
#: that issues device work (DeviceLoader's staging thread): HIP aborts a capture that another thread's stream operations
#: run into
import threading  # noqa: E402
capture_lock = threading.RLock()


def handle_or_none(device_index=None):
    """The handle of a device if there is a GPU, else None: the planner hints are pure host logic and run with the default
    options on a machine without one (CPU tests build plans)."""
    if not torch.cuda.is_available():
        return None
    return handle(torch.cuda.current_device() if device_index is None else device_index)


def set_option(option, value, device_index=None):
    """dlwp_set_option on the handle of `device_index` (default: the current device); returns the previous value."""
    prev = ctypes.c_int(0)
    if not torch.cuda.is_available():      # no device, no handle: the defaults the handle-less planner hints use
        check(lib.dlwp_set_default_option(int(option), int(value), ctypes.byref(prev)))
        return prev.value
    check(lib.dlwp_set_option(handle(torch.cuda.current_device() if device_index is None else device_index), int(option),
                              int(value), ctypes.byref(prev)))
    return prev.value


# Destruction of library objects that own hipGraphs / streams / events (dlwp_train_step_t, dlwp_rollout_t) is never done from a
# finaliser: the garbage collector runs finalisers at arbitrary points -- in the middle of a step that is being recorded, between two
# launches of another graph, under a DeviceLoader worker's copies -- and dlwp_*_destroy synchronises the device and tears down graphs,
# streams and events there (r4: a fault in hipGraphLaunch of a LATER graph; r5: one abort() inside a collection during _record_step in
# ~15 runs of the GPU suite).  A finaliser BURIES the handle; the owner of the next safe point -- the start of a training step, of a
# rollout capture, an explicit close() -- drains the graveyard.
_graveyard = []


def bury(kind, h):
    """from a finaliser: remember `h` ('step' | 'rollout') for destruction at the next safe point"""
    if h is not None:
        _graveyard.append((kind, h))


def drain_graveyard():
    """at a safe point (nothing of ours is being recorded or captured on this thread): destroy what finalisers buried"""
    while _graveyard:
        kind, h = _graveyard.pop()
        try:
            (lib.dlwp_train_step_destroy if kind == 'step' else lib.dlwp_rollout_destroy)(h)
        except Exception:  # noqa: BLE001
            pass


def handle(device_index=0):
    """One library handle per device, created on first use.  Raises DlwpError when there is no gfx950 GPU."""
    h = _handles.get(device_index)
    if h is None:
        out = _vp()
        check(lib.dlwp_create(ctypes.byref(out), int(device_index)))
        h = _handles[device_index] = out
    return h
