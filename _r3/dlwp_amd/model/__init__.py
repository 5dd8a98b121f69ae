from .models import DLWPNeuralNet, DLWPFunctional  # noqa: F401
from .generators import (DataGenerator, ArrayDataset, SeriesDataGenerator, SeriesDataset,  # noqa: F401
                         LabeledArray)
from .extensions import TimeSeriesEstimator  # noqa: F401
from . import verify  # noqa: F401
