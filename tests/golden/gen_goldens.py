#!/usr/bin/env python3
"""Generate tests/golden/goldens.json from the REAL reference (build container only).

    PYTHONHASHSEED=0 python tests/golden/gen_goldens.py [--quick] [case ...]

Imports /root/reference/amira with the import-only shims in tests/golden/shims
(suffix_tree brute-force stand-in, sourmash / pysam stubs — SURVEY.md Appendix B), runs
the procedures of procedures.py against it and records digests + samples.  Nothing of
the reference travels: only the numbers written here.  /root/reference is not needed
(and not read) by any test.
"""
import json
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "shims"), "/root/reference", HERE]

import procedures as P  # noqa: E402

from amira.construct_gene import Gene  # noqa: E402  (the reference)
from amira.construct_gene_mer import GeneMer  # noqa: E402
from amira.construct_graph import GeneMerGraph  # noqa: E402
from amira.graph_utils import (choose_kmer_size, get_overall_mean_node_coverages,  # noqa: E402
                               iterative_bubble_popping)

from amira.pre_processing import process_pandora_json  # noqa: E402
from amira.result_utils import write_pandora_gene_calls  # noqa: E402

REFERENCE = types.SimpleNamespace(GeneMerGraph=GeneMerGraph, Gene=Gene, GeneMer=GeneMer,
                                  process_pandora_json=process_pandora_json,
                                  write_pandora_gene_calls=write_pandora_gene_calls,
                                  choose_kmer_size=choose_kmer_size,
                                  get_overall_mean_node_coverages=get_overall_mean_node_coverages,
                                  iterative_bubble_popping=iterative_bubble_popping)


def main():
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    quick = "--quick" in sys.argv
    path = os.path.join(HERE, "goldens.json")
    out = {}
    if args and os.path.exists(path):
        out = json.load(open(path))
    out["_meta"] = {"reference": "Danderson123/Amira v0.11.0", "hashseed": 0,
                    "python": sys.version.split()[0]}
    for name, (proc, pargs, slow) in P.CASES.items():
        if args and name not in args:
            continue
        if quick and slow:
            continue
        t = time.time()
        out[name] = proc(REFERENCE, *pargs)
        print(f"{name}: {time.time() - t:.1f}s", file=sys.stderr, flush=True)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))
    print("wrote", path, file=sys.stderr)


if __name__ == "__main__":
    main()
