"""Run one procedures.CASES entry in a child interpreter with PYTHONHASHSEED=0.

    python tests/golden/seed0.py <impl> <case>      impl: oracle | product
"""
import json
import os
import subprocess
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def run_case_seed0(impl, name, timeout=1800):
    env = dict(os.environ, PYTHONHASHSEED="0")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), impl, name], env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-4000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def _impl(kind):
    if kind == "oracle":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from amira_oracle import Gene, GeneMer, GeneMerGraph
        from amira_oracle.driver import choose_kmer_size, get_overall_mean_node_coverages
        from amira_oracle.front_end import process_pandora_json, write_pandora_gene_calls
    else:
        sys.path.insert(0, ROOT)
        from amira_amd import Gene, GeneMer, GeneMerGraph
        from amira_amd.graph_utils import choose_kmer_size, get_overall_mean_node_coverages
        from amira_amd.pre_processing import process_pandora_json
        from amira_amd.result_utils import write_pandora_gene_calls
    return types.SimpleNamespace(GeneMerGraph=GeneMerGraph, Gene=Gene, GeneMer=GeneMer,
                                 choose_kmer_size=choose_kmer_size,
                                 get_overall_mean_node_coverages=get_overall_mean_node_coverages,
                                 process_pandora_json=process_pandora_json,
                                 write_pandora_gene_calls=write_pandora_gene_calls)


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    import procedures as P

    proc, args, _ = P.CASES[sys.argv[2]]
    print(json.dumps(proc(_impl(sys.argv[1]), *args)))
