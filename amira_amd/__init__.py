"""amira_amd — MI355X-native gene-mer de Bruijn graph engine with Amira's Python API.

The compute path is libamg.so (hand-written HIP for gfx950 behind the C ABI of
include/amg.h).  There is no CPU fallback: importing the package without the built
library, or building a graph without a HIP device, raises.

Module names mirror the reference package (amira.construct_graph -> amira_amd.construct_graph,
...), so `import amira_amd as amira`-style substitution works for the hot path.
"""
from . import _ffi  # noqa: F401  (raises ImportError if libamg.so is missing)
from .construct_edge import Edge  # noqa: F401
from .construct_gene import Gene, hashlib_hash  # noqa: F401
from .construct_gene_mer import GeneMer  # noqa: F401
from .construct_graph import GeneMerGraph  # noqa: F401
from .construct_node import Node  # noqa: F401
from .construct_read import Read  # noqa: F401
from .engine import Engine  # noqa: F401
from .graph_utils import (build_filtered_graph, build_graph, build_multiprocessed_graph,  # noqa: F401
                          cleaning_sweep, iterative_bubble_popping)
from .tokens import Vocabulary, tokenize  # noqa: F401

__version__ = "0.1.0"
