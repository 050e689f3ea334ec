"""mentflow_amd — MI355X (gfx950) implementation of the MENT-Flow training-step hot path.

Same public names as the reference package ``mentflow`` for the hot path:
``MENTFlow``, ``generate``, ``simulate``, ``diagnostics``, ``entropy``, ``prior``, ``loss``, ``train``, ``utils``.
Compute = hand-written HIP kernels behind the C ABI in include/mentflow_hip.h; no CPU fallback.
"""
from .core import MENTFlow
from . import diagnostics
from . import dist
from . import entropy
from . import generate
from . import graph
from . import loss
from . import ops
from . import prior
from . import simulate
from . import train
from . import utils
from .utils import unravel
