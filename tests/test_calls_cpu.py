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
    from amira_amd.io import write_gene_positions
    out_p = str(tmp_path / "back_pos.json")
    write_gene_positions(out_p, gs, ge, reads.read_offsets, reads.read_ids)
    # byte for byte what the reference's json.dumps(gene_position_dict) writes (result_utils.py:1260-1264)
    assert open(out_p).read() == json.dumps({r: [list(x) for x in pos[r]] for r in read_ids})
    # ... and from 32-bit position arrays (what a correction's read-back hands over when the positions fit)
    write_gene_positions(out_p, gs.astype(np.int32), ge.astype(np.int32), reads.read_offsets, reads.read_ids)
    assert open(out_p).read() == json.dumps({r: [list(x) for x in pos[r]] for r in read_ids})


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


def test_loader_and_writer_in_pieces(tmp_path, monkeypatch):
    """the file is cut at entry boundaries and parsed / written by several threads (AMG_CALLS_THREADS forces pieces on
    a small file): same result as one thread; a read id that ENDS with the bytes the cut looks for makes a piece fail
    on its own terms and the file goes down the single-threaded path — same result again; a malformed file is still
    reported with its offset"""
    from amira_amd import _ffi
    from amira_amd.io import load_gene_calls, write_gene_calls
    import random
    rng = random.Random(7)
    genes = [f"g{i}" for i in range(40)] + ["two words", "café", 'q"uote', "back\\slash"]
    calls, pos = {}, {}
    for i in range(700):
        n = rng.choice([0, 1, 3, 8, 30])
        calls[f"read_{i:04d}"] = [rng.choice("+-") + rng.choice(genes) for _ in range(n)]
        pos[f"read_{i:04d}"] = [[j * 100, j * 100 + rng.randint(1, 90)] for j in range(n)]
    tricky = dict(calls)
    tricky['odd], '] = ["+g1", "-g2"]                 # json.dumps writes ... "odd], ": ["+g1" ...: the cut pattern inside a key
    tricky_pos = dict(pos)
    tricky_pos['odd], '] = [[1, 2], [3, 4]]
    for data, positions in ((calls, pos), (tricky, tricky_pos)):
        cj, pj = _write(tmp_path, "c.json", data), _write(tmp_path, "p.json", positions)
        monkeypatch.setenv("AMG_CALLS_THREADS", "1")
        one, gs1, ge1 = load_gene_calls(cj, pj)
        for threads in ("2", "5", "16"):
            monkeypatch.setenv("AMG_CALLS_THREADS", threads)
            many, gs, ge = load_gene_calls(cj, pj)
            assert many.read_ids == one.read_ids == list(data) and many.vocab.names == one.vocab.names
            assert many.vocab.hashes == one.vocab.hashes
            assert np.array_equal(many.tokens, one.tokens) and np.array_equal(many.read_offsets, one.read_offsets)
            assert np.array_equal(gs, gs1) and np.array_equal(ge, ge1)
            out = str(tmp_path / f"back{threads}.json")
            write_gene_calls(out, many.vocab, many.tokens, many.read_offsets, many.read_ids)
            want = {r: [g[0] + g[1:].replace(" ", "_") for g in v] for r, v in data.items()}   # (construct_gene.py:57)
            assert json.load(open(out)) == want
    monkeypatch.setenv("AMG_CALLS_THREADS", "4")
    text = json.dumps(calls)
    cut = text.index('"read_0400"')
    bad = tmp_path / "bad.json"
    bad.write_text(text[:cut] + '"read_0400": ["gene_without_strand"], ' + text[cut:])
    with pytest.raises(_ffi.AmgError, match="Strand information missing"):
        load_gene_calls(str(bad))
    dup = tmp_path / "dup.json"
    dup.write_text(text[:-1] + ', "read_0003": ["+g1"]}')
    with pytest.raises(_ffi.AmgError, match="duplicate read id"):
        load_gene_calls(str(dup))


def test_writers_replace_a_longer_file(tmp_path, monkeypatch):
    """the writers do not truncate on open (a replaced file keeps its cached pages) and cut the file to its new length
    when they are done: a shorter result over a longer file leaves no tail, whatever the number of pieces; a path
    that cannot be written is an error, not a silent nothing"""
    from amira_amd import _ffi
    from amira_amd.io import load_gene_calls, write_gene_calls, write_gene_positions
    calls = {f"read_{i:03d}": [("+" if (i + j) % 3 else "-") + f"gene{(i * 7 + j) % 23}" for j in range(i % 9)]
             for i in range(300)}
    pos = {r: [[10 * j, 10 * j + 7] for j in range(len(v))] for r, v in calls.items()}
    cj, pj = _write(tmp_path, "c.json", calls), _write(tmp_path, "p.json", pos)
    reads, gs, ge = load_gene_calls(cj, pj)
    out_c, out_p = str(tmp_path / "out_c.json"), str(tmp_path / "out_p.json")
    for threads in ("1", "7"):
        monkeypatch.setenv("AMG_CALLS_THREADS", threads)
        for path in (out_c, out_p):
            with open(path, "w") as f:
                f.write("x" * 200_000)                       # longer than anything written below
        write_gene_calls(out_c, reads.vocab, reads.tokens, reads.read_offsets, reads.read_ids)
        write_gene_positions(out_p, gs, ge, reads.read_offsets, reads.read_ids)
        assert open(out_c).read() == json.dumps(calls) and open(out_p).read() == json.dumps(pos)
        # a handful of reads over the full files
        few = 5
        n_tok = int(reads.read_offsets[few])
        write_gene_calls(out_c, reads.vocab, reads.tokens[:n_tok], reads.read_offsets[:few + 1], reads.read_ids[:few])
        write_gene_positions(out_p, gs[:n_tok], ge[:n_tok], reads.read_offsets[:few + 1], reads.read_ids[:few])
        assert json.load(open(out_c)) == {r: calls[r] for r in list(calls)[:few]}
        assert json.load(open(out_p)) == {r: pos[r] for r in list(pos)[:few]}
    with pytest.raises(_ffi.AmgError, match="cannot write"):
        write_gene_calls(str(tmp_path / "no_such_dir" / "x.json"), reads.vocab, reads.tokens, reads.read_offsets,
                         reads.read_ids)


@pytest.mark.parametrize("case", ["front_end_five", "front_end_nine", "front_end_six_blanks"])
def test_reference_named_front_end_matches_the_reference(case):
    """process_pandora_json / write_pandora_gene_calls (pre_processing.py:44-63, result_utils.py:1260-1264) of the
    product against goldens from the real reference: every read kept, the genes of interest the reads contain in the
    order of the reference's list(set) (PYTHONHASHSEED=0, in a child interpreter), both files byte for byte"""
    from seed0 import run_case_seed0
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "goldens.json")))
    assert run_case_seed0("product", case) == gold[case]


def test_first_use_order_and_buffer_cache(tmp_path):
    """the native first-use pass against a plain loop, and a second load / write served from the kept buffers"""
    from amira_amd import pre_processing as pp
    from amira_amd.io import load_gene_calls
    rng = np.random.default_rng(3)
    genes = ["g%d" % i for i in range(50)]
    calls = {"r%d" % i: [("+" if rng.random() < 0.5 else "-") + genes[int(j)] for j in rng.integers(5, 50, int(rng.integers(0, 9)))]
             for i in range(400)}
    pos = {r: [[10 * i, 10 * i + 5] for i in range(len(v))] for r, v in calls.items()}
    cj, pj = _write(tmp_path, "c.json", calls), _write(tmp_path, "p.json", pos)
    wanted = ["g0", "g7", "g49", "g7", "nope", "g20", "g6"]
    seen = []
    for r in calls:
        for g in calls[r]:
            if g[1:] in wanted and g[1:] not in seen:
                seen.append(g[1:])
    reads = load_gene_calls(cj)
    assert pp._present_in_first_use_order(reads, wanted) == seen
    for _ in range(2):
        reads, got, positions = pp.process_pandora_json(cj, wanted, pj)
        assert set(got) == set(seen) and len(got) == len(seen)
        assert {r: reads[r] for r in reads} == calls
        assert {r: [list(x) for x in positions[r]] for r in positions} == pos
    assert pp.trim_buffers() >= 0
