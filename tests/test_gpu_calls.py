"""f2 on the device path: gene calls loaded from JSON by the native loader (amg_calls_*) go into GeneMerGraph as
CSR arrays (no per-read Python work), through the cleaning sweep, and back out through the native writer — the
graphs and the corrected calls must equal what the dict path (the reference's input format) gives."""
import json

import numpy as np
import pytest

import procedures as P

pytestmark = pytest.mark.gpu


def _write(tmp_path, name, obj):
    p = tmp_path / name
    p.write_text(json.dumps(obj))
    return str(p)


def _graph_arrays(g):
    e = g._engine
    n, ed = e.nodes(), e.edges()
    tok_node, tok_dir = e.read_nodes()
    return {"tokens": n["tokens"], "coverage": n["coverage"], "first_dir": n["first_dir"], "component": n["component"],
            "src": ed["src"], "tgt": ed["tgt"], "sdir": ed["sdir"], "tdir": ed["tdir"], "ecov": ed["coverage"],
            "tok_node": tok_node, "tok_dir": tok_dir}


@pytest.mark.parametrize("case", [("fixture", "five", 3), ("fixture", "nine", 5), ("synth", 11, 600, 35, 200, 5, 0.03)])
def test_json_front_end_equals_dict_path(tmp_path, case):
    from amira_amd import GeneMerGraph
    from amira_amd.io import TokenizedPositions, load_gene_calls, write_gene_calls
    if case[0] == "fixture":
        calls, pos = P.fixture(case[1])
        k = case[2]
        fastq = P.FakeFastq({r: max([e for _, e in pos[r]] + [0]) + 50 for r in calls})
    else:
        _, seed, N, L, V, k, err = case
        calls, pos, fq = P.synth_inputs(seed, N, L, V, err)
        fastq = fq
    reads, gs, ge = load_gene_calls(_write(tmp_path, "calls.json", calls), _write(tmp_path, "pos.json", pos))
    tpos = TokenizedPositions(reads.read_ids, reads.read_offsets, gs, ge)
    with GeneMerGraph(reads, k, tpos) as a, GeneMerGraph(dict(calls), k, {r: list(p) for r, p in pos.items()}) as b:
        ga, gb = _graph_arrays(a), _graph_arrays(b)
        for key in ga:
            assert np.array_equal(ga[key], gb[key]), key
        for g in (a, b):
            g.filter_graph(3, 1)
        ca, pa = a.correct_reads(fastq)
        cb, pb = b.correct_reads(fastq)
        assert list(ca) == list(cb)
        assert all(ca[r] == cb[r] for r in cb)
        assert all([tuple(x) for x in pa[r]] == [tuple(x) for x in pb[r]] for r in cb)   # (JSON pairs are lists)
    # the corrected calls leave through the native writer as the reference's JSON
    from amira_amd.tokens import tokenize
    vocab, toks, offs, ids = tokenize(cb)
    out = str(tmp_path / "corrected.json")
    write_gene_calls(out, vocab, toks, offs, ids)
    assert json.load(open(out)) == {r: list(cb[r]) for r in cb}
