"""
The DLWP.custom names on the hot path, as descriptions for the HIP back end.

  PeriodicPadding3D   reference DLWP/custom.py:217-306   -> the same halo on the (T*C, H, W) store of the recurrent input
  PeriodicPadding2D   reference DLWP/custom.py:139-214   -> halo mode WRAP (fused into the next Conv2D's LDS loader,
                                                            or the standalone LDS-staged pad kernel)
  FillPadding2D       reference DLWP/custom.py:309-402   -> halo mode EDGE (pole-row replication)
  TFPadding2D / 3D    reference DLWP/custom.py:527-672   -> halo modes ZERO / REFLECT / SYMMETRIC
  slice_layer         reference DLWP/custom.py:675-692   -> a channel window, resolved as an input-channel offset
  RowConnected2D      reference DLWP/custom.py:695-837   -> per-row filters: dlwp_rowconv2d_fwd / _bwd_data / _bwd_weight
  row_conv2d          reference DLWP/custom.py:840-896   -> the same launch on device tensors
  EarlyStoppingMin    reference DLWP/custom.py:99-136
  RNNResetStates      reference DLWP/custom.py:94-96
  History             keras.callbacks.History (what examples/train.py:253 passes)
  RunHistory, Adam- / SGDLearningRateTracker, BatchHistory   reference DLWP/custom.py:32-91
"""
import numpy as np

from . import layers as _layers
from .layers import Layer  # noqa: F401  (re-export: custom layers subclass it)
from .layers import RowConnected2D  # noqa: F401  (DLWP.custom.RowConnected2D; the class lives with the other weighted layers)


class PeriodicPadding2D(_layers._Pad2DBase):
    """Periodic padding of rows and columns; corners are wrap-of-wrap (the reference pads W first, then H of the
    already W-padded tensor).  Like the reference's slices it does not tile: padding > axis length is an error
    (raised when the model is planned, where the shapes are known)."""
    mode = 1

    def compute_output_shape(self, s):
        out = super(PeriodicPadding2D, self).compute_output_shape(s)
        (t, b), (l, r) = self.padding
        h, w = (s[1], s[2]) if self.data_format == 'channels_first' else (s[0], s[1])
        if max(t, b) > h or max(l, r) > w:
            raise ValueError('%s: periodic padding %r exceeds the input size %dx%d' % (self.name, self.padding, h, w))
        return out


class FillPadding2D(_layers._Pad2DBase):
    """Edge-replicating padding (rows first, then columns of the row-padded tensor == np.pad(mode='edge'))."""
    mode = 2


class TFPadding2D(_layers._Pad2DBase):
    """tf.pad as a layer (reference DLWP/custom.py:527-600): mode 'CONSTANT' (zeros), 'REFLECT' (mirror without the border
    element) or 'SYMMETRIC' (mirror with it); same padding argument forms as ZeroPadding2D.  Like every halo here it is a
    lazy view fused into the consuming convolution's loader.  A non-zero `constant_values` is not lowered."""

    def __init__(self, padding=(1, 1), data_format=None, mode='CONSTANT', constant_values=0., **kwargs):
        super(TFPadding2D, self).__init__(padding=padding, data_format=data_format, **kwargs)
        modes = {'CONSTANT': 0, 'REFLECT': 3, 'SYMMETRIC': 4}
        if str(mode).upper() not in modes:
            raise ValueError("TFPadding2D mode must be one of 'CONSTANT', 'REFLECT', 'SYMMETRIC', got %r" % (mode,))
        self.tf_mode = str(mode).upper()
        self.mode = modes[self.tf_mode]
        self.constant_values = float(constant_values)
        if self.mode == 0 and self.constant_values != 0.:
            raise NotImplementedError('TFPadding2D: a non-zero constant_values is not implemented (zeros are)')

    def compute_output_shape(self, s):
        out = super(TFPadding2D, self).compute_output_shape(s)
        (t, b), (l, r) = self.padding
        h, w = (s[1], s[2]) if self.data_format == 'channels_first' else (s[0], s[1])
        lim_h, lim_w = (h - 1, w - 1) if self.mode == 3 else (h, w)
        if self.mode and (max(t, b) > lim_h or max(l, r) > lim_w):      # tf.pad's own limits
            raise ValueError('%s: %s padding %r exceeds the input size %dx%d' % (self.name, self.tf_mode, self.padding, h, w))
        return out

    def get_config(self):
        cfg = super(TFPadding2D, self).get_config()
        cfg.update({'mode': self.tf_mode, 'constant_values': self.constant_values})
        return cfg


class PeriodicPadding3D(_layers._Pad3DBase):
    """Periodic padding of the three trailing axes (reference DLWP/custom.py:217-306; last axis first, then the middle one
    of the already padded tensor, then the first).  On the HIP path it pads the (T, C, H, W) input of ConvLSTM2D
    (examples/train.py:144-146: padding (0, 0, 2)); a non-zero pad of the first (channel) axis is not lowered."""
    mode = 1

    def compute_output_shape(self, s):
        out = super(PeriodicPadding3D, self).compute_output_shape(s)
        dims = s[1:] if self.data_format == 'channels_first' else s[:3]
        for (lo, hi), d in zip(self.padding, dims):
            if max(lo, hi) > d:
                raise ValueError('%s: periodic padding %r exceeds the input size %r' % (self.name, self.padding, dims))
        return out


class FillPadding3D(_layers._Pad3DBase):
    """Edge-replicating 3-D padding (reference DLWP/custom.py:405-520)."""
    mode = 2


class TFPadding3D(_layers._Pad3DBase):
    """tf.pad on the three trailing axes as a layer (reference DLWP/custom.py:602-672): mode 'CONSTANT' (zeros), 'REFLECT' or
    'SYMMETRIC'; like the other 3-D pads it is lowered as a halo of the (T*C, H, W) store (the first axis is not padded)."""

    def __init__(self, padding=(1, 1, 1), data_format=None, mode='CONSTANT', constant_values=0., **kwargs):
        super(TFPadding3D, self).__init__(padding=padding, data_format=data_format, **kwargs)
        modes = {'CONSTANT': 0, 'REFLECT': 3, 'SYMMETRIC': 4}
        if str(mode).upper() not in modes:
            raise ValueError("TFPadding3D mode must be one of 'CONSTANT', 'REFLECT', 'SYMMETRIC', got %r" % (mode,))
        self.tf_mode = str(mode).upper()
        self.mode = modes[self.tf_mode]
        self.constant_values = float(constant_values)
        if self.mode == 0 and self.constant_values != 0.:
            raise NotImplementedError('TFPadding3D: a non-zero constant_values is not implemented (zeros are)')

    def get_config(self):
        cfg = super(TFPadding3D, self).get_config()
        cfg.update({'mode': self.tf_mode, 'constant_values': self.constant_values})
        return cfg


def slice_layer(start, end, step=None, axis=1):
    """Return a layer that slices `axis` -- the reference returns a keras Lambda; here it is a channel window that the
    consuming convolution reads in place (no copy)."""
    if axis < 0:
        raise ValueError("'slice_layer' can only work on a specified axis > 0")
    return _layers.ChannelSlice(start, end, step=step, axis=axis)


# ------------------------------------------------------------------------------------------------------------------ #
# custom losses (reference DLWP/custom.py:899-1093): descriptions consumed by the HIP loss kernels (dlwp_loss_custom)
# ------------------------------------------------------------------------------------------------------------------ #

def row_conv2d(inputs, kernel, kernel_size, strides, output_shape, data_format=None):
    """DLWP.custom.row_conv2d (reference DLWP/custom.py:840-896) on DEVICE tensors: the 2-D convolution whose weights are
    shared only along rows.  inputs: (batch, channels, rows, cols) for 'channels_first', (batch, rows, cols, channels) for
    'channels_last' (Keras' default when data_format is None); kernel: (output_rows, kh, kw, channels, filters);
    output_shape: (output_row, output_col), checked.  One launch (dlwp_rowconv2d_fwd) instead of output_row convolutions
    and a concatenate.  Strides other than 1 are not implemented (the reference never passes them)."""
    from . import ops
    from .layers import normalize_data_format
    fmt = normalize_data_format(data_format)
    if tuple(strides) != (1, 1):
        raise NotImplementedError('row_conv2d: only strides (1, 1) are implemented')
    x = inputs if fmt == 'channels_first' else inputs.permute(0, 3, 1, 2).contiguous()
    kh, kw = kernel_size
    if tuple(kernel.shape[1:3]) != (kh, kw):
        raise ValueError('row_conv2d: kernel %r does not have kernel_size %r' % (tuple(kernel.shape), (kh, kw)))
    y = ops.rowconv2d(x.contiguous(), kernel.contiguous(), None, ops.make_conv(kernel.shape[-1], kh, kw, 1))
    if tuple(y.shape[2:]) != tuple(output_shape):
        raise ValueError('row_conv2d: output is %r, output_shape says %r' % (tuple(y.shape[2:]), tuple(output_shape)))
    return y if fmt == 'channels_first' else y.permute(0, 2, 3, 1).contiguous()


class LossSpec(object):
    """What `anomaly_correlation_loss(...)` / `latitude_weighted_loss(...)` return: a description the trainer lowers to
    dlwp_loss_custom.  kind 0 = (latitude-weighted) mse, kind 1 = regularizer - anomaly correlation."""

    def __init__(self, kind, regularize=0, mean=None, row_weights=None, scale=1.0, name='loss'):
        self.kind, self.regularize, self.scale = kind, regularize, scale
        self.mean = None if mean is None else np.ascontiguousarray(mean, dtype=np.float32)
        self.row_weights = None if row_weights is None else np.ascontiguousarray(row_weights, dtype=np.float32)
        self.__name__ = name

    def __call__(self, y_true, y_pred):
        raise RuntimeError('%s is evaluated by the HIP loss kernel; pass it as loss= to build_model' % self.__name__)


def latitude_weights(lats, weighting='cosine'):
    """cos(lat) [+ 0.5 sin^2(2 lat) for 'midlatitude'] -- the function form of the reference (custom.py:975-978)."""
    if weighting not in ['cosine', 'midlatitude']:
        raise ValueError("'weighting' must be one of 'cosine' or 'midlatitude'")
    lat = np.asarray(lats, dtype=np.float32) * np.float32(np.pi / 180.)
    w = np.cos(lat)
    if weighting == 'midlatitude':
        w = w + np.float32(0.5) * np.sin(2 * lat) ** 2
    return w.astype(np.float32)


def anomaly_correlation_loss(mean=None, regularize_mean='mse', reverse=True):
    """Anomaly-correlation loss, `regularizer - ACC` (reference custom.py:1036-1088; the default loss of
    examples/train.py).  mean: climatology of shape (1,) + output shape, or None.  regularize_mean: None | 'mse' | 'mae'
    | 'global' | 'spatial'."""
    if mean is not None:
        assert len(mean.shape) > 1
        assert mean.shape[0] == 1
    if regularize_mean is not None:
        assert regularize_mean in ['global', 'spatial', 'mse', 'mae']
        reverse = True
    reg = {None: 0, 'mse': 1, 'mae': 2, 'global': 3, 'spatial': 4}[regularize_mean]
    return LossSpec(1, reg, None if mean is None else np.asarray(mean)[0], None, 1.0 if reverse else -1.0, 'acc_loss')


def latitude_weighted_loss(loss_function=None, lats=None, output_shape=(), axis=-2, weighting='cosine'):
    """Weight predictions and targets by a function of latitude before the loss (reference custom.py:956-991).
    loss_function: mean_squared_error (default) or the result of anomaly_correlation_loss(...)."""
    if weighting not in ['cosine', 'midlatitude']:
        raise ValueError("'weighting' must be one of 'cosine' or 'midlatitude'")
    w = None
    if lats is not None:
        if axis != -2 and axis != len(output_shape) - 2:
            raise NotImplementedError('latitude_weighted_loss: the latitude axis must be the second to last (axis=-2)')
        w = latitude_weights(lats, weighting)
    if isinstance(loss_function, LossSpec):
        return LossSpec(loss_function.kind, loss_function.regularize, loss_function.mean, w, loss_function.scale,
                        'lat_loss')
    name = loss_function if isinstance(loss_function, str) else getattr(loss_function, '__name__', 'mean_squared_error')
    if loss_function is not None and name not in ('mse', 'mean_squared_error'):
        raise NotImplementedError('latitude_weighted_loss over %r is not implemented' % (loss_function,))
    return LossSpec(0, 0, None, w, 1.0, 'lat_loss')


# compatibility names (reference custom.py:1091-1093): what load_model's custom_objects of older scripts look up
lat_loss = latitude_weighted_loss()
acc_loss = anomaly_correlation_loss()


# ------------------------------------------------------------------------------------------------------------------ #
# callbacks (host-side plumbing of examples/train.py:253-263)
# ------------------------------------------------------------------------------------------------------------------ #

class Callback(object):
    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class History(Callback):
    """Per-epoch record of the logs dict: history = {metric: [value per epoch]}, epoch = [indices]."""

    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class BatchHistory(Callback):
    """Per-batch record, one dict per epoch (reference DLWP/custom.py:54-68)."""

    def on_train_begin(self, logs=None):
        self.history = []
        self.epoch = 0

    def on_epoch_begin(self, epoch, logs=None):
        self.history.append({})

    def on_epoch_end(self, epoch, logs=None):
        self.epoch += 1

    def on_batch_end(self, batch, logs=None):
        for k, v in (logs or {}).items():
            self.history[self.epoch].setdefault(k, []).append(v)


class RunHistory(History):
    """History that also forwards every epoch's metrics to a run logger -- any object with `log(name, value)`, e.g. an AzureML
    `Run` (reference DLWP/custom.py:71-91; call sites Azure/train_tf.py:376, Azure/train_func.py:348)."""

    def __init__(self, run):
        super(RunHistory, self).__init__()
        self.epoch, self.history, self.run = [], {}, run

    def on_epoch_end(self, epoch, logs=None):
        super(RunHistory, self).on_epoch_end(epoch, logs)
        for k, v in (logs or {}).items():
            self.run.log(k, v)


def _effective_lr(optimizer):
    """lr / (1 + decay * iterations): the rate Keras' optimisers step with after `iterations` updates."""
    return float(optimizer.lr) / (1.0 + float(optimizer.decay) * float(optimizer.iterations))


class AdamLearningRateTracker(Callback):
    """Prints Adam's bias-corrected step size at the end of every epoch (reference DLWP/custom.py:32-41); `last_lr` keeps it."""

    def on_epoch_end(self, epoch, logs=None, beta_1=0.9, beta_2=0.999):
        opt = self.model.optimizer
        t = float(opt.iterations) + 1.0
        self.last_lr = _effective_lr(opt) * np.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
        print(' - LR: {:.6f}'.format(self.last_lr))


class SGDLearningRateTracker(Callback):
    """Prints SGD's decayed learning rate at the end of every epoch (reference DLWP/custom.py:44-51)."""

    def on_epoch_end(self, epoch, logs=None):
        self.last_lr = _effective_lr(self.model.optimizer)
        print(' - LR: {:.6f}'.format(self.last_lr))


class RNNResetStates(Callback):
    def on_epoch_begin(self, epoch, logs=None):
        self.model.reset_states()


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping semantics (monitor / min_delta / patience / mode / restore_best_weights)."""

    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto', baseline=None,
                 restore_best_weights=False):
        super(EarlyStopping, self).__init__()
        self.monitor, self.patience, self.verbose = monitor, patience, verbose
        self.baseline, self.restore_best_weights = baseline, restore_best_weights
        self.min_delta = abs(min_delta)
        if mode not in ('auto', 'min', 'max'):
            mode = 'auto'
        if mode == 'max' or (mode == 'auto' and 'acc' in monitor):
            self.monitor_op = np.greater
        else:
            self.monitor_op = np.less
            self.min_delta *= -1
        if self.monitor_op is np.greater:
            self.min_delta = abs(self.min_delta)
        self.wait = 0
        self.stopped_epoch = 0
        self.best = None
        self.best_weights = None

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        self.best = self.baseline if self.baseline is not None else (np.inf if self.monitor_op is np.less else -np.inf)

    def get_monitor_value(self, logs):
        value = (logs or {}).get(self.monitor)
        if value is None:
            import warnings
            warnings.warn('Early stopping conditioned on metric `%s` which is not available. Available metrics are: %s'
                          % (self.monitor, ','.join(sorted((logs or {}).keys()))), RuntimeWarning)
        return value

    def _improved(self, current):
        return self.monitor_op(current - self.min_delta, self.best)

    def on_epoch_end(self, epoch, logs=None):
        current = self.get_monitor_value(logs)
        if current is None:
            return
        if self._improved(current):
            self.best, self.wait = current, 0
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
            return
        self.wait += 1
        if self.wait >= self.patience:
            self.stopped_epoch = epoch
            self.model.stop_training = True
            if self.restore_best_weights and self.best_weights is not None:
                if self.verbose > 0:
                    print('Restoring model weights from the end of the best epoch')
                self.model.set_weights(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose > 0:
            print('Epoch %05d: early stopping' % (self.stopped_epoch + 1))


class EarlyStoppingMin(EarlyStopping):
    """EarlyStopping that does not even start counting before `min_epochs` epochs have run."""

    def __init__(self, min_epochs=0, **kwargs):
        super(EarlyStoppingMin, self).__init__(**kwargs)
        if not isinstance(min_epochs, int) or min_epochs < 0:
            raise ValueError('min_epochs must be an integer >= 0')
        self.min_epochs = min_epochs

    def on_epoch_end(self, epoch, logs=None):
        if epoch >= self.min_epochs:
            super(EarlyStoppingMin, self).on_epoch_end(epoch, logs)
