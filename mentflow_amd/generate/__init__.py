"""Generative models (mentflow/generate/__init__.py)."""
from .base import GenerativeModel
from .flows import AutoregressiveFlow
from .build import build_generator
from .build import build_flow
from .nn import NNGenerator, NNTransform
