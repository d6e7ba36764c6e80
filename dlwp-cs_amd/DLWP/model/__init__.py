"""
DLWP.model of the MI355X engine: the cubed-sphere hot path of the reference package (reference DLWP/model/__init__.py
exports the same names for what is built here; DLWPNeuralNet, DataGenerator / SeriesDataGenerator (xarray), Preprocessor
and DLWPTorchNN are outside the hot-path scope, see DESIGN.md sections 1 and 7).
"""
from .models import DLWPFunctional                                  # noqa: F401
from .generators import ArrayDataGenerator, tf_data_generator      # noqa: F401
from .extensions import TimeSeriesEstimator                          # noqa: F401
