"""Stand-in for the absent third-party package `sourmash` (build container only): the reference is
imported with this on its path so that its bubble-popping code can run when goldens are generated.
MinHash is the restatement of sourmash's published algorithm in oracle/amira_oracle/minhash.py —
parity of that part is pinned by the reference-held test vectors only (see its docstring)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))                     # tests/golden/shims/sourmash
_ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))), "oracle")
if _ORACLE not in sys.path:
    sys.path.insert(0, _ORACLE)
from amira_oracle.minhash import MinHash  # noqa: E402,F401
