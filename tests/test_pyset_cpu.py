"""The read-path clustering's block search runs in C++ (amira_amd/csrc/amg_cluster.hip) on device node ids, and the
order of its results is the order in which the reference's Python sets of node-hash tuples iterate.  That order is
reproduced by an emulation of CPython's set and tuple hash; here the emulation is checked against the running
interpreter's own sets on random operation scripts (adds, updates from other sets, comprehensions over a set), with
the element hashes Python itself reports."""
import ctypes as C
import random

import numpy as np
import pytest

from amira_amd import _ffi
from amira_amd._ffi import check, ptr


def _run_script(ops, key_hash, n_sets):
    ops = np.ascontiguousarray(np.asarray(ops, np.int32).reshape(-1, 3))
    kh = np.ascontiguousarray(key_hash, np.int64)
    out_keys = np.empty(max(len(kh) * n_sets, 1), np.int32)
    out_off = np.empty(n_sets + 1, np.int64)
    check(_ffi.lib.amg_pyset_script(ptr(ops), len(ops), ptr(kh), n_sets, ptr(out_keys), ptr(out_off)))
    return [out_keys[out_off[s]:out_off[s + 1]].tolist() for s in range(n_sets)]


def _random_key(rng):
    n = rng.choice([0, 1, 2, 3, 5, 8, 13, 30, 55])
    items = []
    for _ in range(n):
        r = rng.random()
        items.append(None if r < 0.05 else rng.getrandbits(256) - (1 << 255) if r < 0.9 else rng.randrange(-5, 5))
    return tuple(items)


@pytest.mark.parametrize("seed", range(12))
def test_set_emulation_follows_the_interpreter(seed):
    rng = random.Random(seed)
    n_keys = rng.choice([5, 40, 300, 2500])
    keys = []
    while len(keys) < n_keys:
        k = _random_key(rng)
        if k not in keys[-50:]:
            keys.append(k)
    keys = list(dict.fromkeys(keys))
    # suffix / prefix families, as the contexts are (many tuples sharing tails)
    base = tuple(rng.getrandbits(256) for _ in range(40))
    keys += [base[-i:] for i in range(1, 41)] + [base[:i] for i in range(1, 40)]
    keys = list(dict.fromkeys(keys))
    key_hash = [hash(k) for k in keys]
    n_sets = 6
    real = [set() for _ in range(n_sets)]
    ops = []
    for _ in range(rng.choice([20, 200, 3000])):
        r = rng.random()
        a = rng.randrange(n_sets)
        if r < 0.70:
            b = rng.randrange(len(keys))
            real[a].add(keys[b])
            ops.append((0, a, b))
        elif r < 0.90:
            b = rng.randrange(n_sets)
            real[a].update(real[b])
            ops.append((1, a, b))
        elif r < 0.93:
            real[a] = set()
            ops.append((2, a, 0))
        else:
            b = rng.randrange(n_sets)
            real[a] = {k for k in real[b]}
            ops.append((3, a, b))
    got = _run_script(ops, key_hash, n_sets)
    index = {k: i for i, k in enumerate(keys)}
    assert got == [[index[k] for k in s] for s in real]


def test_growth_through_every_resize():
    """one set grown key by key to 70 000 elements (past the 50 000 mark, where growth drops from 4x to 2x), and a
    copy filled by update() from it in chunks"""
    keys = [(i, i * i) for i in range(70000)]
    key_hash = [hash(k) for k in keys]
    real_a, real_b = set(), set()
    ops = []
    for i, k in enumerate(keys):
        real_a.add(k)
        ops.append((0, 0, i))
        if i % 9973 == 0:
            real_b.update(real_a)
            ops.append((1, 1, 0))
    got = _run_script(ops, key_hash, 2)
    index = {k: i for i, k in enumerate(keys)}
    assert got[0] == [index[k] for k in real_a] and got[1] == [index[k] for k in real_b]


def test_tuple_hash():
    rng = random.Random(3)
    for _ in range(300):
        t = _random_key(rng)
        lanes = np.asarray([hash(x) for x in t], np.int64)
        assert _ffi.lib.amg_py_tuple_hash(ptr(lanes) if len(lanes) else None, len(lanes)) == hash(t)
