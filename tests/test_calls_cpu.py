"""Native gene-call loader / writer (amg_calls_*, host code) against the Python path: same
vocabulary order and hashes, same tokens, same positions, JSON round trip.  CPU only."""
import json
import os

import numpy as np
import pytest

import dump as D
import procedures as P


def _write(tmp_path, name, obj):
    p = tmp_path / name
    p.write_text(json.dumps(obj))
    return str(p)


@pytest.mark.parametrize("fixture", ["five", "nine"])
def test_native_loader_equals_python_tokenizer(tmp_path, fixture):
    from amira_amd.io import load_gene_calls, write_gene_calls
    from amira_amd.tokens import tokenize
    calls, pos = P.fixture(fixture)
    cj, pj = _write(tmp_path, "calls.json", calls), _write(tmp_path, "pos.json", pos)
    reads, gs, ge = load_gene_calls(cj, pj)
    vocab, toks, offs, read_ids = tokenize(calls)
    assert reads.read_ids == read_ids and reads.vocab.names == vocab.names
    assert reads.vocab.hashes == vocab.hashes                 # native sha256(pickle(name)) == hashlib
    assert np.array_equal(reads.tokens, toks) and np.array_equal(reads.read_offsets, offs)
    assert np.array_equal(gs, np.fromiter((p[0] for r in read_ids for p in pos[r]), np.int64))
    assert np.array_equal(ge, np.fromiter((p[1] for r in read_ids for p in pos[r]), np.int64))
    some = read_ids[:50] + read_ids[-5:]
    assert all(reads[r] == calls[r] for r in some) and len(reads) == len(calls)
    out = str(tmp_path / "back.json")
    write_gene_calls(out, reads.vocab, reads.tokens, reads.read_offsets, reads.read_ids)
    assert json.load(open(out)) == calls                       # write-back round trip


def test_gene_syntax_and_escapes(tmp_path):
    from amira_amd import _ffi
    from amira_amd.io import load_gene_calls
    from amira_amd.tokens import tokenize
    calls = {"réad \"1\"": ["+two words", "-café", "+a\\b", "-" + "x" * 300], "empty": [], "r2": ["-two_words"]}
    path = _write(tmp_path, "c.json", calls)
    reads = load_gene_calls(path)
    vocab, toks, offs, ids = tokenize(calls)
    assert reads.read_ids == ids and reads.vocab.names == vocab.names and reads.vocab.hashes == vocab.hashes
    assert np.array_equal(reads.tokens, toks) and np.array_equal(reads.read_offsets, offs)
    assert reads["r2"] == ["-two_words"] and reads["empty"] == []
    for bad in ({"r": ["gene_without_strand"]}, {"r": ["+"]}, {"r": [" "]}):
        with pytest.raises(_ffi.AmgError):
            load_gene_calls(_write(tmp_path, "bad.json", bad))
    with pytest.raises(_ffi.AmgError):
        load_gene_calls(str(tmp_path / "missing.json"))
