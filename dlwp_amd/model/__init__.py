from .models import DLWPNeuralNet, DLWPFunctional  # noqa: F401
from .generators import DataGenerator, ArrayDataset  # noqa: F401
