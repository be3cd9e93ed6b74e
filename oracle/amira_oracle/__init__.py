"""CPU oracle (test infrastructure only) — see oracle/README.md."""
from .values import Gene, GeneMer, Read, Node, Edge, hashlib_hash  # noqa: F401
from .graph import GeneMerGraph  # noqa: F401
