"""Synthetic gene-call streams.

Two generators:

* :func:`loop_reads` — the per-read loop of Appendix C, draw for draw (start, error
  mask, error genes, flip).  Small N only; used for goldens and parity tests.
* :func:`block_reads` — the same model drawn block-wise (65 536 reads per block, one
  PCG64 stream per block keyed by (seed, block)), so any rank can generate any
  read range without the others and the union does not depend on the world size.
  Used by bench.py for the 100 k – 8 M read configurations.

Both return gene ids and strands as arrays; :func:`to_read_dict` renders the
reference's input format ({"r0000000": ["+g12", "-g7", ...]}), and
:func:`positions_for` / :func:`fake_fastq_lengths` the position layout of Appendix C.
"""
import numpy as np

BLOCK = 65536


def make_genome(rng, V, n_amr=0):
    """Circular genome: permutation of V genes with i.i.d. strands; optional planted
    multi-copy AMR genes (ids V..V+n_amr-1, copies 2 + (j % 2)) — Appendix C cfg 4."""
    genome = rng.permutation(V)
    strands = rng.integers(0, 2, V)
    if n_amr:
        genome = list(genome)
        strands = list(strands)
        for j in range(n_amr):
            for _ in range(2 + (j % 2)):
                p = int(rng.integers(0, len(genome)))
                genome.insert(p, V + j)
                strands.insert(p, int(rng.integers(0, 2)))
        genome = np.asarray(genome)
        strands = np.asarray(strands)
    return genome, strands


def gene_names(V, n_amr=0):
    return [f"g{i}" for i in range(V)] + [f"amr{j}" for j in range(n_amr)]


def loop_reads(seed, N, L, V, err=0.02, n_amr=0):
    """Appendix C generator, one read at a time.  Returns (gene_ids[N,L], strands[N,L])."""
    rng = np.random.default_rng(seed)
    genome, strands = make_genome(rng, V, n_amr)
    G = len(genome)
    ids = np.empty((N, L), dtype=np.int64)
    sts = np.empty((N, L), dtype=np.int64)
    for i in range(N):
        s = rng.integers(0, G)
        idx = (s + np.arange(L)) % G
        g = genome[idx]
        st = strands[idx]
        if err > 0:
            e = rng.random(L) < err
            g = np.where(e, rng.integers(0, V, L), g)
        if rng.random() < 0.5:
            g, st = g[::-1], 1 - st[::-1]
        ids[i], sts[i] = g, st
    return ids, sts


def block_reads(seed, lo, hi, L, V, err=0.02, n_amr=0):
    """Reads [lo, hi) of the block-wise stream.  Returns (gene_ids[n,L], strands[n,L])."""
    genome, strands = make_genome(np.random.default_rng([seed, 0xA11CE]), V, n_amr)
    G = len(genome)
    out_g, out_s = [], []
    b = lo // BLOCK
    while b * BLOCK < hi:
        rng = np.random.default_rng([seed, b])
        start = rng.integers(0, G, BLOCK)
        emask = rng.random((BLOCK, L)) < err
        egene = rng.integers(0, V, (BLOCK, L))
        flip = rng.random(BLOCK) < 0.5
        a, z = max(lo, b * BLOCK) - b * BLOCK, min(hi, (b + 1) * BLOCK) - b * BLOCK
        idx = (start[a:z, None] + np.arange(L)[None, :]) % G
        g = np.where(emask[a:z], egene[a:z], genome[idx])
        st = strands[idx]
        f = flip[a:z]
        g[f] = g[f, ::-1]
        st[f] = 1 - st[f, ::-1]
        out_g.append(g)
        out_s.append(st)
        b += 1
    return np.concatenate(out_g), np.concatenate(out_s)


def read_names(lo, hi):
    return [f"r{i:07d}" for i in range(lo, hi)]


def to_read_dict(ids, sts, names, first=0):
    reads = {}
    for i in range(ids.shape[0]):
        reads[f"r{first + i:07d}"] = [
            ("+" if s else "-") + names[g] for g, s in zip(ids[i].tolist(), sts[i].tolist())
        ]
    return reads


def positions_for(reads):
    """gene i of every read at [i*1000, i*1000+899] (lists, as json.load would give)."""
    return {r: [[i * 1000, i * 1000 + 899] for i in range(len(g))] for r, g in reads.items()}


def fake_fastq_lengths(reads):
    """Only len(fastq[read]['sequence']) is used on this path (construct_graph.py:1685)."""
    return {r: len(g) * 1000 + 100 for r, g in reads.items()}
