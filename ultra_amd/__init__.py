"""ultra_amd -- MI355X-native training-step hot path of ULTRA's offline unbiased-LTR pipeline.

Same plugin surface as the reference's ``ultra`` package for that path (``ultra_amd.utils``,
``.input_layer``, ``.learning_algorithm``, ``.ranking_model``; classes selected by dotted name in the
experiment JSON), backed by the hand-written HIP kernels of ``csrc/`` through a ctypes-bound C ABI
(include/ultr_hip.h).  There is no CPU fallback.
"""
from . import utils
from . import ranking_model
from . import learning_algorithm
from . import input_layer

__version__ = '0.1.0'
