"""amira_amd — MI355X-native gene-mer de Bruijn graph engine with Amira's Python API.

The compute path is libamg.so (hand-written HIP for gfx950 behind the C ABI of
include/amg.h).  There is no CPU fallback: importing the package without the built
library, or building a graph without a HIP device, raises.
"""
from . import _ffi  # noqa: F401  (raises ImportError if libamg.so is missing)
from .engine import Engine  # noqa: F401
from .tokens import Vocabulary, tokenize  # noqa: F401
