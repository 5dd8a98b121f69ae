"""
Import shim: `import dlwp_amd.compat` (or dlwp_amd.compat.install()) registers this package under the module names the
reference's scripts import, so code written against jweyn/DLWP + Keras runs unedited on the HIP back end:

    import dlwp_amd.compat                      # once, before the reference-style imports
    from DLWP.model import DLWPNeuralNet, DataGenerator
    from DLWP.custom import EarlyStoppingMin, RNNResetStates, PeriodicPadding2D, slice_layer
    from DLWP.util import save_model, load_model, train_test_split_ind
    from keras.layers import Input, ZeroPadding2D, Conv2D, MaxPooling2D, UpSampling2D, concatenate
    from keras.models import Model
    from keras.callbacks import History
    from keras.losses import mean_squared_error

Only the names on the hot path exist; anything else raises AttributeError / ImportError as a missing
Keras feature would.  Nothing is installed if a real `keras` or `DLWP` is already importable, unless force=True.
"""
import importlib.util
import sys
import types


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__dlwp_amd_shim__ = True
    sys.modules[name] = m
    return m


def install(force=False):
    from . import custom, engine, layers, regularizers, training, util
    from . import model as model_pkg
    from .model import extensions, generators, models
    have_keras = 'keras' in sys.modules or importlib.util.find_spec('keras') is not None
    have_dlwp = 'DLWP' in sys.modules or importlib.util.find_spec('DLWP') is not None
    if (have_keras or have_dlwp) and not force:
        real = [n for n, h in (('keras', have_keras), ('DLWP', have_dlwp)) if h and not getattr(sys.modules.get(n), '__dlwp_amd_shim__', False)]
        if real:
            raise ImportError('refusing to shadow the importable package(s) %r; call install(force=True)' % real)
    keras = _module('keras')
    keras.layers = _module('keras.layers', **{k: v for k, v in vars(layers).items() if not k.startswith('_')})
    keras.layers.convolutional = _module('keras.layers.convolutional', ZeroPadding2D=layers.ZeroPadding2D)
    keras.models = _module('keras.models', Model=engine.Model, Sequential=engine.Sequential)
    keras.callbacks = _module('keras.callbacks', Callback=custom.Callback, History=custom.History,
                              EarlyStopping=custom.EarlyStopping)
    keras.losses = _module('keras.losses', mean_squared_error=training.mean_squared_error,
                           mean_absolute_error=training.mean_absolute_error)
    keras.regularizers = _module('keras.regularizers', l2=regularizers.l2, l1_l2=regularizers.l1_l2,
                                 L1L2=regularizers.L1L2)
    keras.optimizers = _module('keras.optimizers', Adam=training.Adam, SGD=training.SGD)
    dlwp = _module('DLWP')
    dlwp.custom = _module('DLWP.custom', **{k: v for k, v in vars(custom).items() if not k.startswith('_')})
    dlwp.util = _module('DLWP.util', **{k: v for k, v in vars(util).items() if not k.startswith('_')})
    dlwp.model = _module('DLWP.model', DLWPNeuralNet=models.DLWPNeuralNet, DLWPFunctional=models.DLWPFunctional,
                         DataGenerator=generators.DataGenerator, ArrayDataset=generators.ArrayDataset,
                         SeriesDataGenerator=generators.SeriesDataGenerator,
                         TimeSeriesEstimator=extensions.TimeSeriesEstimator)
    dlwp.model.models = _module('DLWP.model.models', DLWPNeuralNet=models.DLWPNeuralNet,
                                DLWPFunctional=models.DLWPFunctional)
    dlwp.model.generators = _module('DLWP.model.generators', DataGenerator=generators.DataGenerator,
                                    ArrayDataset=generators.ArrayDataset,
                                    SeriesDataGenerator=generators.SeriesDataGenerator)
    return dlwp


install()
