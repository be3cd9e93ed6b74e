#!/usr/bin/env python3
"""bench.py — read -> corrected gene-mer graph hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3-sweep|cfg3|cfg2|cfg4]

A "step" is one pass of the hot path over one batch of synthetic gene calls whose CSR
token arrays (and gene positions) are already resident in HBM:
  cfg3-sweep (default) 1 M reads x 60 genes, k=5, 20 k-gene vocabulary, 2 % substitutions:
             build -> filter_graph(3,1) -> correct_reads -> build ->
             remove_short_linear_paths(5) -> correct_reads -> build
             (the cleaning sweep of graph_utils.py:145-166; BASELINE.json configs[2])
  cfg3       the first build of that sweep only
  cfg2       100 k reads x 40 genes, k=5, 5 k-gene vocabulary: build + coverage (configs[1])
  cfg4       1 M error-free reads with 10 planted multi-copy AMR genes, k=5: build +
             assign_reads_to_genes through the reference-shaped Python API (configs[3])
Prints ONE JSON line (driver contract): `value` is the device-resident rate (inputs in HBM when
the clock starts, nothing read back); `e2e` (N = 1) is SURVEY 8(d)'s timed region — host CSR arrays
-> host graph arrays, H2D and D2H inside the clock, pinned buffers, the position upload overlapped
with the first build on a second stream; `roofline` prices the dominant kernel with HIP events of
the engine's own stream; `cpu_baseline` (sequential C restatement), `cpu_baseline_python` (the reference's cost
model) and `cpu_baseline_ncore` time the CPU oracles on this box.
Multi-GPU (torch.distributed.run, one rank per GPU, RCCL): every rank holds its own N-read
shard of the global stream (weak scaling); EVERY build of the step merges the per-shard node /
edge tables by key owner (all-to-all + all-gather, amira_amd/dist.py), so all ranks hold the
single-graph result; filtering and clipping run on that graph, correction on the local reads.
`--no-merge` builds the shards independently instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfg2": dict(N=100_000, L=40, V=5_000, k=5, err=0.02, seed=20250905 + 2, sweep=False,
                 desc="synthetic 100k reads x 40 genes, k=5, 5k-gene vocab: graph build + coverage"),
    "cfg3": dict(N=1_000_000, L=60, V=20_000, k=5, err=0.02, seed=20250905 + 3, sweep=False,
                 desc="synthetic 1M reads x 60 genes, k=5, 20k-gene vocab: graph build"),
    "cfg3-sweep": dict(N=1_000_000, L=60, V=20_000, k=5, err=0.02, seed=20250905 + 3, sweep=True,
                       desc="synthetic 1M reads x 60 genes, k=5, 20k-gene vocab: build + "
                            "error-correction sweep (build, filter(3,1), correct, build, clip(5), "
                            "correct, build)"),
    "cfg4": dict(N=1_000_000, L=60, V=20_000, k=5, err=0.0, seed=20250905 + 4, sweep=False, n_amr=10,
                 desc="synthetic 1M error-free reads x 60 genes with 10 planted multi-copy AMR genes, k=5, "
                      "20k-gene vocab: build + read-path clustering (assign_reads_to_genes)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def stage_bytes(stage, k, L, n_windows, n_reads, n_gapped):
    """ALGORITHMIC bytes one launch of the named stage's dominant kernel moves: SURVEY.md
    section 8(d)'s 56.9 B per gene-mer (k=5) split over the kernels that implement each term
    (DESIGN.md "Kernels"): tokens int32, node id int32, dir int8, counters uint32 RMW = 8 B,
    edge key 12 B, positions 2 x int64."""
    adj = (L - k) / (L - k + 1)
    per_window = {
        "node_upsert": 4.0 * L / (L - k + 1) + 5 + 4 * k,   # token read + (id, dir) write + key compare-read
        "node_count": 8.0,                                   # node counter RMW
        "edge_upsert": 12.0 * adj,                           # edge record key read per adjacency
        "graph_upsert": 4.0 * L / (L - k + 1) + 5 + 4 * k + 12.0 * adj,  # both of them: the fused table pass
        "edge_count": 8.0 * adj,                             # edge counter RMW per adjacency
    }
    if stage in per_window:
        return per_window[stage] * n_windows
    if stage == "correct_positions":   # per gapped read: x, y tokens + positions in and out
        return n_gapped * (L * 4 * 2 + L * 16 * 2)
    if stage == "correct_gapped":      # per gapped read: window ids+dirs, tokens in, genes out
        return n_gapped * ((L - k + 1) * 5 + L * 4 * 2)
    return None


def make_tokens(w, lo, hi):
    from amira_amd import synth
    from amira_amd.tokens import Vocabulary
    n_amr = w.get("n_amr", 0)
    ids, sts = synth.block_reads(w["seed"], lo, hi, w["L"], w["V"], w["err"], n_amr=n_amr)
    names = synth.gene_names(w["V"], n_amr)
    vocab = Vocabulary(names)
    rank = np.array([vocab.rank[n] for n in names], dtype=np.int64)
    r = rank[ids]
    toks = np.where(sts == 1, vocab.V + r, vocab.V - 1 - r).astype(np.int32)
    offs = (np.arange(hi - lo + 1, dtype=np.int64) * w["L"])
    return vocab, toks.reshape(-1), offs


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (pure-Python restatement of the reference, same sha256 + pickle work per
# gene-mer as construct_gene.py:5-10) on this box's host cores.  Test infrastructure used as the
# checker / yardstick only; nothing below is on the product path.
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from amira_amd import synth
    from amira_oracle import GeneMerGraph, driver, values
    return synth, GeneMerGraph, driver, values


def _cpu_sweep_or_build(w, first, n):
    """reads [first, first + n) of the workload's own stream through the oracle; returns (gene-mers, seconds).
    Runs in a thread with a 1 GB stack: the reference labels components by RECURSIVE depth-first search
    (construct_graph.py:911-918, recursion limit 50 000 at :27), which the oracle restates; ten thousand
    reads make one component of tens of thousands of nodes and overflow the default 8 MB stack."""
    import threading
    box = {}

    def run():
        try:
            box["out"] = _cpu_sweep_or_build_here(w, first, n)
        except BaseException as err:  # noqa: BLE001
            box["err"] = err

    old = threading.stack_size(1 << 30)
    try:
        t = threading.Thread(target=run)
        t.start()
        t.join()
    finally:
        threading.stack_size(old)
    if "err" in box:
        raise box["err"]
    return box["out"]


def _cpu_sweep_or_build_here(w, first, n):
    synth, GeneMerGraph, driver, values = _oracle()
    sys.setrecursionlimit(2_000_000)  # beyond the reference's 50 000 (:27): the sample's component may be larger
    values.CACHE_HASHES = False
    L, k = w["L"], w["k"]
    try:
        ids, sts = synth.block_reads(w["seed"], first, first + n, L, w["V"], w["err"], n_amr=w.get("n_amr", 0))
        reads = synth.to_read_dict(ids, sts, synth.gene_names(w["V"], w.get("n_amr", 0)), first=first)
        t = time.perf_counter()
        if w["sweep"]:
            pos = synth.positions_for(reads)
            fq = driver.FakeFastq(synth.fake_fastq_lengths(reads))
            t = time.perf_counter()
            driver.correction_sweep(reads, pos, k, fq, 3)
        else:
            GeneMerGraph(reads, k)
        return n * (L - k + 1), time.perf_counter() - t
    finally:
        values.CACHE_HASHES = True


def _cpu_worker(args):
    w, first, n = args
    return _cpu_sweep_or_build(w, first, n)


def cpu_baseline(w, frac=0.01, budget_s=12.0):
    """1 core — the reference pipeline always builds with cores=1 — on a 1 % read
    subsample of the SAME stream, the whole step (sweep workloads: the whole sweep)."""
    # sized to ~12 s of CPU work from a 250-read pilot, at most the 1 % subsample of SURVEY 8(d)
    pilot_done, pilot_s = _cpu_sweep_or_build(w, w["N"] - 250, 250)
    # (a deeper sample re-threads more reads per read: measured 1.8x from the 250-read pilot to a 0.1 % sample)
    per_read = pilot_s / 250 * (2.2 if w["sweep"] else 1.0)
    n = max(min(int(w["N"] * frac), int(budget_s / max(per_read, 1e-6))), 250)
    done, spent = _cpu_sweep_or_build(w, 0, n)
    what = "full sweep" if w["sweep"] else "build"
    return {"value": done / spent, "unit": "gene-mers/s", "cores": 1, "kind": "port",
            "sample": f"the first {n} reads ({100.0 * n / w['N']:.2g} %) of the same stream, {what}, pure-Python oracle "
                      f"with per-call sha256+pickle (the reference's cost model), {spent:.1f} s; sized to ~{budget_s:.0f} s of CPU "
                      f"work (the C port above carries the bounded-sample baseline the bench contract asks for) (SURVEY 8(d)'s 1 % sample would take ~{per_read * w['N'] * frac:.0f} s); "
                      f"extrapolates linearly in reads (depth is {n / w['N']:.2g}x the workload's, so fewer nodes survive "
                      f"the filter)"}


def cpu_baseline_c(w, frac=0.25):
    """1 core, the sequential C restatement of the reference's algorithm (oracle/token_sweep.c: integer tokens, hash
    tables and arrays instead of sha256 + pickle per call; pinned to the Python oracle in tests/test_token_oracle.py and
    the checker of the full-size GPU tests) on the first quarter of the same stream: what a careful single-threaded CPU
    implementation does, next to `cpu_baseline_python` (the reference's own cost model)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import token_oracle
    n = max(int(w["N"] * (frac if w["sweep"] else 1.0)), 1)   # (a build of the whole stream takes seconds)
    L, k = w["L"], w["k"]
    vocab, toks, offs = make_tokens(w, 0, n)
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, n)
    rl = np.full(n, L * 1000 + 100, np.int64)
    t = time.perf_counter()
    o = token_oracle.Sweep(toks, offs, vocab.two_v, gs, gs + 899, rl) if w["sweep"] else token_oracle.Sweep(toks, offs, vocab.two_v)
    try:
        o.build(k)
        if w["sweep"]:
            o.filter(3, 1)
            o.correct_reads()
            o.adopt_corrected()
            o.build(k)
            o.remove_short_linear_paths(k)
            o.correct_reads()
            o.adopt_corrected()
            o.build(k)
        spent = time.perf_counter() - t
    finally:
        o.close()
    what = "full sweep" if w["sweep"] else "build"
    return {"value": n * (L - k + 1) / spent, "unit": "gene-mers/s", "cores": 1, "kind": "port",
            "sample": f"the first {n} reads ({100.0 * n / w['N']:.3g} %) of the same stream, {what}, sequential C "
                      f"restatement of the reference (oracle/token_sweep.c, integer tokens; pinned to the Python oracle), "
                      f"{spent:.1f} s; `cpu_baseline_python` is the same algorithm at the reference's own cost model "
                      f"(sha256 + pickle per call)"}


def cpu_baseline_ncore(w, per_worker=2500, max_workers=64):
    """N host cores, read-sharded: the model of build_multiprocessed_graph (graph_utils.py:105-124) WITHOUT
    its merge (which the reference does sequentially and gets wrong for edge coverages)
    — an upper bound of what N processes of the reference's Python could build."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, max_workers))
    build_only = dict(w, sweep=False)
    jobs = [(build_only, i * per_worker, per_worker) for i in range(workers)]
    t = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t
    done = sum(r[0] for r in res)
    return {"value": done / wall, "unit": "gene-mers/s", "cores": workers, "host_cores": cores, "kind": "port",
            "sample": f"{workers} processes x {per_worker} reads of the same stream, build only, no merge "
                      f"(upper bound of graph_utils.py:105-124), wall {wall:.1f} s incl. process start-up"}


def cpu_baseline_cfg4(w, budget_s=10.0):
    """BASELINE configs[3] on one host core with the pure-Python oracle (the reference's cost model): build AND
    assign_reads_to_genes on the first reads of the same stream, sized to ~budget_s of CPU work from a pilot."""
    import threading
    synth, GeneMerGraph, driver, values = _oracle()
    L, k, n_amr = w["L"], w["k"], w["n_amr"]
    genes = [f"amr{j}" for j in range(n_amr)]
    box = {}

    def once(n):
        ids, sts = synth.block_reads(w["seed"], 0, n, L, w["V"], w["err"], n_amr=n_amr)
        reads = synth.to_read_dict(ids, sts, synth.gene_names(w["V"], n_amr), first=0)
        pos = synth.positions_for(reads)
        t0 = time.perf_counter()
        g = GeneMerGraph(reads, k, pos)
        t1 = time.perf_counter()
        clustered, _ = g.assign_reads_to_genes(genes, 1, {}, None)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, sum(len(d) for comp in clustered.values() for d in comp.values())

    def run():
        try:
            sys.setrecursionlimit(2_000_000)
            values.CACHE_HASHES = False
            b, c_, _ = once(200)
            n = max(200, min(int(w["N"] * 0.01), int(budget_s / max((b + c_) / 200, 1e-6))))
            box["out"] = (n,) + once(n)
        except BaseException as err:  # noqa: BLE001
            box["err"] = err
        finally:
            values.CACHE_HASHES = True

    old = threading.stack_size(1 << 30)   # (the reference labels components by recursive depth-first search)
    try:
        t = threading.Thread(target=run)
        t.start()
        t.join()
    finally:
        threading.stack_size(old)
    if "err" in box:
        raise box["err"]
    n, build_s, cluster_s, alleles = box["out"]
    return {"value": n * (L - k + 1) / (build_s + cluster_s), "unit": "gene-mers/s", "cores": 1, "kind": "port",
            "build_s": round(build_s, 2), "assign_reads_to_genes_s": round(cluster_s, 2), "alleles_found": alleles,
            "sample": f"the first {n} reads ({100.0 * n / w['N']:.2g} %) of the same stream, GeneMerGraph(...) + "
                      f"assign_reads_to_genes for the {n_amr} planted genes, pure-Python oracle (the reference's cost model); "
                      f"the depth is {n / w['N']:.2g}x the workload's, so most planted copies are seen by a handful of reads "
                      "or none: the clustering of the full workload is far more work per read than this sample's"}


# ---------------------------------------------------------------------------------------------
def run_cfg4(args, w, rank, world, local_rank):
    """BASELINE configs[3]: build + read-path clustering through amira_amd.GeneMerGraph (the reference's
    API: GeneMerGraph(reads, k, positions).assign_reads_to_genes(genes, 1, {}, None)).  Reads are handed
    over tokenised (amira_amd.io.TokenizedReads is the {read: [genes]} mapping the API takes, decoded
    lazily), so the step is H2D + device build + clustering."""
    import torch
    torch.cuda.set_device(local_rank)
    from amira_amd import GeneMerGraph, synth
    from amira_amd.io import TokenizedPositions, TokenizedReads
    N, L, k = w["N"], w["L"], w["k"]
    vocab, toks, offs = make_tokens(w, rank * N, (rank + 1) * N)
    read_ids = synth.read_names(rank * N, (rank + 1) * N)
    reads = TokenizedReads(vocab, toks, offs, read_ids)
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
    positions = TokenizedPositions(read_ids, offs, gs, gs + 899)
    genes = [f"amr{j}" for j in range(w["n_amr"])]
    n_windows = N * (L - k + 1)
    stage = {"build_s": [], "cluster_s": [], "device_build_ms": []}
    info = {}
    dev_stages = {}   # device stage -> [ms, launches] of the instrumented step (HIP events on the engine's stream)

    def note(engine):
        for name, ms in engine.timings():
            e = dev_stages.setdefault(name, [0.0, 0])
            e[0] += ms
            e[1] += 1

    def step(record):
        from amira_amd import Engine
        t0 = time.perf_counter()
        g = GeneMerGraph(reads, k, positions)
        t1 = time.perf_counter()
        dev_ms = sum(m for _, m in g._engine.timings()) if record else 0.0
        plain_match = Engine.match_patterns
        if record:   # the device searches of the clustering (k_match), call by call
            note(g._engine)

            def timed_match(self, which, patterns):
                out = plain_match(self, which, patterns)
                note(self)
                return out
            Engine.match_patterns = timed_match
        g._engine.set_timing(bool(record))
        try:
            clustered, path_reads = g.assign_reads_to_genes(genes, 1, {}, None)
        finally:
            Engine.match_patterns = plain_match
            g._engine.set_timing(True)
        t2 = time.perf_counter()
        if record:
            stage["build_s"].append(t1 - t0)
            stage["cluster_s"].append(t2 - t1)
            stage["device_build_ms"].append(dev_ms)
            info["alleles"] = {gene: sorted(len(v) for v in d.values())
                               for comp in clustered.values() for gene, d in comp.items()}
            info["nodes"] = g.get_total_number_of_nodes()
        g.close()

    for _ in range(args.warmup):
        step(False)
    step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {
        "metric": "gene-mers/s to GeneMerGraph + read-path clusters", "value": world * n_windows * args.steps / dt,
        "unit": "gene-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic", "reads_per_s": world * N * args.steps / dt,
        "config": {"workload": w["desc"], "reads_per_gpu": N, "genes_per_read": L, "k": k, "vocab": w["V"],
                   "genes_of_interest": genes, "nodes": info.get("nodes"),
                   "alleles_found": sum(len(v) for v in info.get("alleles", {}).values()),
                   "allele_sizes": info.get("alleles")},
        "stages_s_per_step": {"build_incl_h2d_and_view": round(float(np.mean(stage["build_s"])), 4),
                              "device_build_ms": round(float(np.mean(stage["device_build_ms"])), 3),
                              "assign_reads_to_genes": round(float(np.mean(stage["cluster_s"])), 4)},
        "roofline": None,
    }
    # the device side of the step: the build's table passes and the batched sub-list searches of the clustering
    # (k_match: every pass reads the gene calls of all reads once — SURVEY 8(d)'s B_cluster term, 4 L bytes per read
    # and pass); the dominant one is priced against the HBM roofline, the rest listed
    T = N * L
    algo = {"match_count": 4.0 * T, "match_fill": 4.0 * T,
            "node_upsert": stage_bytes("node_upsert", k, L, n_windows, N, 0),
            "edge_upsert": stage_bytes("edge_upsert", k, L, n_windows, N, 0)}
    priced = {s_: v for s_, v in dev_stages.items() if s_ in algo and v[1] > 0}
    if priced:
        groups = {"k_match": ["match_count", "match_fill"], "k_nodes_m": ["node_upsert"], "k_edges_v": ["edge_upsert"]}
        per = {}
        for kern, names in groups.items():
            ms = sum(priced[n_][0] for n_ in names if n_ in priced)
            launches = sum(priced[n_][1] for n_ in names if n_ in priced)
            if launches:
                per[kern] = {"ms_per_step": round(ms, 3), "launches_per_step": launches,
                             "avg_launch_ms": ms / launches, "algorithmic_bytes_per_launch": algo[names[0]],
                             "achieved_GBs": algo[names[0]] / (ms / launches * 1e-3) / 1e9}
        dom = max(per, key=lambda kern: per[kern]["ms_per_step"])
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": per[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": per[dom]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": per[dom]["algorithmic_bytes_per_launch"],
                           "avg_launch_ms": per[dom]["avg_launch_ms"], "launches_per_step": per[dom]["launches_per_step"],
                           "device_ms_per_step": round(sum(v[0] for v in dev_stages.values()), 3),
                           "kernels": per,
                           "note": "the step is host-bound: the device works for device_ms_per_step of ms_per_step; "
                                   "k_match passes read the gene calls of ALL reads per call (4 L bytes per read)"}
    return out


def run_e2e_pipelined(w, vocab, toks, offs, k, n_windows, local_rank, counts, steps=4, lanes=2):
    """SURVEY 8(d)'s region as a stream of batches: `lanes` engines, each fed by its own host thread, so that one
    batch's upload, another's sweep and a third's download share the device — PCIe is full duplex and the sweep
    needs neither direction.  Every lane does exactly what `e2e` does per step (pinned host CSR + positions -> H2D ->
    build, filter, correct, build, clip, correct, build -> D2H of the corrected calls with positions and of the final
    graph); the rate is batches completed per second over all lanes."""
    import threading
    import torch
    from amira_amd import Engine
    N, L = w["N"], w["L"]
    T = len(toks)
    dev = torch.device("cuda", local_rank)
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    pinned = lambda n, dt_: torch.empty(n, dtype=dt_).pin_memory().numpy()
    h_toks, h_offs = pin(toks), pin(offs)
    h_gs = pin(np.tile(np.arange(L, dtype=np.int32) * 1000, N))   # read coordinates fit 32 bits (amg_set_positions32)
    h_ge = pin(h_gs.numpy() + 899)
    h_rl = pin(np.full(N, L * 1000 + 100, np.int64))
    cap_t, cap_r = T + T // 8 + 1024, N + 1024
    cap_d, cap_e = max(counts["n_nodes"] * 2, T // 8) + 1024, max(counts["n_edges"] * 2, T // 4) + 1024

    class Lane:
        def __init__(self):
            self.eng = Engine(local_rank)
            self.eng.set_timing(False)
            self.d_toks = torch.empty(T, dtype=torch.int32, device=dev)
            self.d_offs = torch.empty(N + 1, dtype=torch.int64, device=dev)
            self.d_gs = torch.empty(T, dtype=torch.int32, device=dev)
            self.d_ge = torch.empty(T, dtype=torch.int32, device=dev)
            self.d_rl = torch.empty(N, dtype=torch.int64, device=dev)
            self.side = torch.cuda.Stream(device=dev)
            self.ev_reads, self.ev_pos = torch.cuda.Event(), torch.cuda.Event()
            self.buf = {"tokens": pinned(cap_d * k, torch.int32), "coverage": pinned(cap_d, torch.int32).view(np.uint32),
                        "first_token": pinned(cap_d, torch.int64), "first_dir": pinned(cap_d, torch.int8),
                        "component": pinned(cap_d, torch.int32), "alive": pinned(cap_d, torch.uint8),
                        "src": pinned(cap_e, torch.int32), "tgt": pinned(cap_e, torch.int32),
                        "sdir": pinned(cap_e, torch.int8), "tdir": pinned(cap_e, torch.int8),
                        "ecoverage": pinned(cap_e, torch.int32).view(np.uint32), "ealive": pinned(cap_e, torch.uint8),
                        "tok_node": pinned(cap_t, torch.int32), "tok_dir": pinned(cap_t, torch.int8),
                        "c_tokens": pinned(cap_t, torch.int32), "c_read_offsets": pinned(cap_r, torch.int64),
                        "c_orig_read": pinned(cap_r, torch.int32), "c_changed": pinned(cap_r, torch.uint8),
                        "c_gene_start": pinned(cap_t, torch.int64), "c_gene_end": pinned(cap_t, torch.int64),
                        "c_gene_start32": pinned(cap_t, torch.int32), "c_gene_end32": pinned(cap_t, torch.int32)}

        def step(self):
            eng = self.eng
            with torch.cuda.stream(self.side):
                self.d_toks.copy_(h_toks, non_blocking=True)
                self.d_offs.copy_(h_offs, non_blocking=True)
                self.ev_reads.record(self.side)
                self.d_gs.copy_(h_gs, non_blocking=True)
                self.d_ge.copy_(h_ge, non_blocking=True)
                self.d_rl.copy_(h_rl, non_blocking=True)
                self.ev_pos.record(self.side)
            self.ev_reads.synchronize()
            eng.set_reads_device(self.d_toks.data_ptr(), self.d_offs.data_ptr(), N, vocab.two_v, borrow=True)
            eng.build(k)
            self.ev_pos.synchronize()
            eng.set_positions32_device(self.d_gs.data_ptr(), self.d_ge.data_ptr(), self.d_rl.data_ptr())
            eng.filter(3, 1)
            eng.correct_reads()
            eng.adopt_corrected()
            eng.build(k)
            eng.remove_short_linear_paths(k)
            n_out = eng.correct_reads()
            eng.corrected(*n_out, True, buf=self.buf, pos32=True)
            eng.adopt_corrected()
            eng.build(k)
            eng.nodes(self.buf)
            eng.edges(self.buf)
            eng.read_nodes(self.buf)

    ls = [Lane() for _ in range(lanes)]
    for lane in ls:
        lane.step()
    torch.cuda.synchronize()
    errors = []

    def work(lane):
        try:
            torch.cuda.set_device(local_rank)
            for _ in range(steps):
                lane.step()
        except BaseException as err:  # noqa: BLE001
            errors.append(err)

    threads = [threading.Thread(target=work, args=(lane,)) for lane in ls]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps * lanes)
    for lane in ls:
        lane.eng.close()
    if errors:
        raise errors[0]
    return {"value": n_windows / dt, "unit": "gene-mers/s", "ms_per_step": dt * 1e3, "steps": steps * lanes, "lanes": lanes,
            "what": f"the same region as a stream of batches on {lanes} engines fed by {lanes} host threads: uploads, "
                    "sweeps and downloads of different batches overlap (PCIe is full duplex); batches completed per "
                    "second over all lanes"}


def run_api_e2e(w, vocab, toks, offs, k, n_windows, steps=5):
    """graph_utils.cleaning_sweep's call sequence on the workload's stream through the Python drop-in: GeneMerGraph(...)
    [+ filter_graph fused], correct_reads, GeneMerGraph, remove_short_linear_paths, correct_reads, GeneMerGraph — host
    arrays in (TokenizedReads / TokenizedPositions / ReadLengths), array-backed mappings out; what a correction produced
    stays on the device for the next build."""
    from amira_amd import graph_utils as gu, synth
    from amira_amd.io import ReadLengths, TokenizedPositions, TokenizedReads
    N, L = w["N"], w["L"]
    ids = synth.read_names(0, N)
    gs = np.tile(np.arange(L, dtype=np.int32) * 1000, N)   # (read coordinates fit 32 bits: uploaded as they are)
    ge = gs + 899
    lengths = ReadLengths(ids, np.full(N, L * 1000 + 100, np.int64))
    stages = {}

    def timed(name, fn):
        t = time.perf_counter()
        r = fn()
        stages[name] = stages.get(name, 0.0) + time.perf_counter() - t
        return r

    def sweep():
        reads, pos = TokenizedReads(vocab, toks, offs, ids), TokenizedPositions(ids, offs, gs, ge)
        g = timed("build_filtered_graph", lambda: gu.build_filtered_graph(reads, k, pos, 3, 1))
        r1, p1 = timed("correct_reads", lambda: g.correct_reads(lengths))
        g2 = timed("build_graph", lambda: gu.build_multiprocessed_graph(r1, k, 1, p1))
        timed("remove_short_linear_paths", lambda: g2.remove_short_linear_paths(k, _lazy_hashes=True))
        r2, p2 = timed("correct_reads", lambda: g2.correct_reads(lengths))
        g3 = timed("build_graph", lambda: gu.build_multiprocessed_graph(r2, k, 1, p2))
        n = g3.get_total_number_of_nodes()
        for x in (g, g2, g3):
            x.close()
        return n, len(r2)

    for _ in range(3):   # the three graphs of a sweep take their engines from a pool in turn: after three sweeps every
        sweep()          # pooled engine has grown its buffers to the largest role (the first build)
    stages.clear()
    # the cyclic collector is paused for the timed sweeps, as GeneMerGraph.assign_reads_to_genes does for itself: a
    # generation-2 pass walks every million-entry id list alive in this process (~100 ms) and finds nothing — the
    # mappings hold arrays and flat lists, no cycles
    import gc
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            nodes, reads_left = sweep()
            if os.environ.get("AMG_API_TRACE"):
                print(f"[api_e2e] sweep {1e3 * (time.perf_counter() - t1):.1f} ms {stages}", file=sys.stderr, flush=True)
        dt = (time.perf_counter() - t0) / steps
    finally:
        if gc_was_on:
            gc.enable()
    return {"value": n_windows / dt, "unit": "gene-mers/s", "ms_per_step": dt * 1e3, "steps": steps,
            "final_nodes": nodes, "reads_left": reads_left,
            "stages_ms_per_step": {n: round(v * 1e3 / steps, 1) for n, v in stages.items()},
            "what": "graph_utils.cleaning_sweep's calls through amira_amd.GeneMerGraph with array-backed mappings "
                    "(amira_amd.io.TokenizedReads / TokenizedPositions / ReadLengths) in and out; the corrected reads go "
                    "from a correction to the next build on the device (amg_set_reads_from_corrected), the first build "
                    "uploads its host arrays"}


def run_front_end(w, vocab, toks, offs, k, n_windows):
    """SURVEY 8 row f2 at the workload's size: gene-call JSON + gene-position JSON -> CSR arrays (amg_calls_load_json /
    amg_calls_load_positions_json: the file read and parsed in pieces by the host's cores, one sha256 per distinct gene)
    and CSR -> JSON (amg_calls_write_json / amg_calls_write_positions_json), then `json_e2e`: both files in ->
    graph_utils.cleaning_sweep through the array-backed API -> corrected calls and positions out as JSON."""
    import tempfile
    from amira_amd import graph_utils as gu, synth
    from amira_amd.io import (ReadLengths, TokenizedPositions, load_gene_calls, write_gene_calls, write_gene_positions)
    N, L = w["N"], w["L"]
    ids = synth.read_names(0, N)
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
    ge = gs + 899
    out = {"reads": N, "host_threads": os.cpu_count()}
    # the files live in memory where the box has a tmpfs with room for them (3.2 GB written, twice): on a disk-backed
    # /tmp the writers are throttled by the kernel's dirty-page limits as soon as earlier runs have left gigabytes of
    # unwritten pages behind (measured: 2.6 GB/s alone, 1.4 GB/s at the end of a long session)
    base = os.environ.get("AMG_BENCH_TMP")
    if not base:
        base = os.environ.get("TMPDIR", "/tmp")
        try:
            import shutil
            if shutil.disk_usage("/dev/shm").free > (12 << 30):
                base = "/dev/shm"
        except OSError:
            pass
    out["tmp_dir"] = base
    with tempfile.TemporaryDirectory(dir=base) as d:
        cj, pj, cj2, pj2 = (os.path.join(d, n) for n in ("calls.json", "positions.json", "corrected.json", "corrected_positions.json"))

        def timed(fn):
            t = time.perf_counter()
            r = fn()
            return r, time.perf_counter() - t

        _, t_wc = timed(lambda: write_gene_calls(cj, vocab, toks, offs, ids))
        _, t_wp = timed(lambda: write_gene_positions(pj, gs, ge, offs, ids))
        sz_c, sz_p = os.path.getsize(cj), os.path.getsize(pj)
        cold = []
        for rep in range(2):   # (the first call of a process pays for the page faults of a gigabyte of fresh heap)
            (reads, lgs, lge), t_l = timed(lambda: load_gene_calls(cj, pj))
            cold.append(t_l)
        assert np.array_equal(reads.tokens, toks) and np.array_equal(lgs, gs)
        _, t_lc = timed(lambda: load_gene_calls(cj))
        out.update({
            "calls_json_bytes": sz_c, "positions_json_bytes": sz_p,
            "load_calls_s": round(t_lc, 3), "load_calls_GBs": round(sz_c / t_lc / 1e9, 3),
            "load_calls_and_positions_s": round(cold[1], 3),
            "load_calls_and_positions_GBs": round((sz_c + sz_p) / cold[1] / 1e9, 3),
            "load_calls_and_positions_first_call_s": round(cold[0], 3),
            "load_gene_mers_per_s": n_windows / cold[1],
            "write_calls_s": round(t_wc, 3), "write_calls_GBs": round(sz_c / t_wc / 1e9, 3),
            "write_positions_s": round(t_wp, 3), "write_positions_GBs": round(sz_p / t_wp / 1e9, 3)})
        lengths = ReadLengths(reads.read_ids, np.full(N, L * 1000 + 100, np.int64))

        from amira_amd.pre_processing import process_pandora_json
        from amira_amd.result_utils import write_pandora_gene_calls
        wanted = [vocab.names[i] for i in range(0, vocab.V, max(vocab.V // 40, 1))] + ["not_in_the_reads"]

        def whole():
            # the reference's own entry points on both sides of the sweep (pre_processing.py:44-63,
            # result_utils.py:1260-1264): both files in, the genes of interest the reads contain, both files out
            r, genes, p = process_pandora_json(cj, wanted, pj)
            g, r2, p2 = gu.cleaning_sweep(r, p, k, ReadLengths(r.read_ids, lengths.lengths), 3)
            write_pandora_gene_calls(d, p2, r2, cj2, pj2)
            n = g.get_total_number_of_nodes()
            g.close()
            return n, len(genes)

        whole()
        (nodes, n_genes), t_all = timed(whole)
        out["json_e2e"] = {"value": n_windows / t_all, "unit": "gene-mers/s", "s_per_step": round(t_all, 3),
                           "final_nodes": nodes, "genes_of_interest_in_reads": n_genes,
                           "what": "process_pandora_json (gene calls + positions JSON in, genes of interest reduced to "
                                   "those in the reads) -> graph_utils.cleaning_sweep (array-backed API, device sweep) -> "
                                   "write_pandora_gene_calls (corrected calls + positions JSON out, the two files side by "
                                   "side), second of two runs"}
    return out


def _bubble_inputs(seed, N, L, V, err):
    """gene calls with compact positions (60-base genes every 80 bases) and nucleotide reads that agree with them: every
    gene name owns one pseudo-random sequence, laid down (reverse-complemented for '-') at its position on each read,
    random filler in between — reads through the same genes share their k-mers (the shape tests/golden/procedures.py
    synth_fastq gives the bubble-popping goldens)"""
    from amira_amd import synth
    ids, sts = synth.loop_reads(seed, N, L, V, err, 0)
    calls = synth.to_read_dict(ids, sts, synth.gene_names(V, 0))
    rng = np.random.default_rng(seed + 1)
    comp = np.zeros(256, np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    gene_seq = {}
    pos, fq = {}, {}
    for rid, genes in calls.items():
        pos[rid] = [(80 * i, 80 * i + 59) for i in range(len(genes))]
        seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 80 * len(genes) + 40)].copy()
        for g, (a, b) in zip(genes, pos[rid]):
            piece = gene_seq.get(g[1:])
            if piece is None:
                piece = gene_seq[g[1:]] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 60)]
            seq[a:b + 1] = piece if g[0] == "+" else comp[piece][::-1]
        text = seq.tobytes().decode()
        fq[rid] = {"sequence": text, "quality": "I" * len(text)}
    return calls, pos, fq


def run_bubbles(local_rank, with_cpu=True):
    """SURVEY 8 row f1 (construct_graph.py:2148-2250 correct_low_coverage_paths, graph_utils.py:127-181): the device
    sketch kernel priced against the HBM roofline on a large batch, and the whole driver iterative_bubble_popping on a
    small read set with nucleotide reads, the pure-Python oracle (its own MinHash restatement) beside it."""
    import ctypes as C
    import tempfile
    from amira_amd import Engine, _ffi, graph_utils as gu
    out = {}
    # ---- k_minhash on one batch: 2^27 bases in 1 000-base segments, the node sketch's parameters (ksize 11, scaled 10)
    n_seg, seg_len, ksize, scaled = 1 << 17, 1000, 11, 10
    rng = np.random.default_rng(11)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_seg * seg_len, dtype=np.uint8)]
    offs = np.arange(n_seg + 1, dtype=np.int64) * seg_len
    sets = (np.arange(n_seg, dtype=np.int32) >> 4)
    cap = len(bases) // scaled * 2
    o_set, o_hash = np.empty(cap, np.int32), np.empty(cap, np.uint64)
    eng = Engine(local_rank)
    try:
        eng.set_timing(True)
        n = C.c_int64(0)
        best_ms, wall = None, None
        for rep in range(3):
            t = time.perf_counter()
            _ffi.check(_ffi.lib.amg_minhash(eng._h, _ffi.ptr(bases), _ffi.ptr(offs), _ffi.ptr(sets), n_seg, ksize, scaled,
                                            _ffi.ptr(o_set), _ffi.ptr(o_hash), cap, C.byref(n)))
            dt = time.perf_counter() - t
            ms = dict(eng.timings()).get("minhash")
            if ms is not None and (best_ms is None or ms < best_ms):
                best_ms, wall = ms, dt
        kept = int(n.value)
        alg = len(bases) + 12 * kept   # 1 B per base in, {set id 4 B, hash 8 B} per kept hash out
        out["minhash_kernel"] = {
            "bases": int(len(bases)), "segments": n_seg, "ksize": ksize, "scaled": scaled, "kept_hashes": kept,
            "kernel_ms": round(best_ms, 4), "call_ms_with_pcie": round(wall * 1e3, 2),
            "roofline": {"bound": "hbm", "kernel": "k_minhash", "algorithmic_bytes": alg,
                         "achieved": alg / (best_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (best_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "bases_per_s": len(bases) / (best_ms * 1e-3),
                         "note": "one thread hashes one k-mer start (MurmurHash3_x64_128 over the canonical 11-mer, the k-mer "
                                 "taken, reverse-complemented and compared as 64-bit words: ~ten 64-bit multiplies per "
                                 "byte read): the kernel is bound by instruction issue, not by bytes — the fraction says "
                                 "how far from the HBM line it sits, not that bytes are wasted"}}
    finally:
        eng.close()
    # ---- the driver on reads with nucleotide sequences: three cleaning iterations with bubble popping, at a real size
    # (a cfg-3-style read set: 150 x coverage of a 20 000-gene genome, 2 % wrong genes) and, next to the pure-Python
    # oracle, on the first reads of the same stream
    import contextlib
    import io
    import tempfile as _tmp
    from amira_amd import bubble_popping as bp
    N, L, V, k, err = int(os.environ.get("AMG_BENCH_BUBBLE_READS", "50000")), 60, 20000, 5, 0.02
    calls, pos, fq = _bubble_inputs(4242, N, L, V, err)
    n_windows = sum(max(0, len(v) - k + 1) for v in calls.values())
    reads_t, pos_t = gu._tokenized(calls, pos)
    lap = {}

    def timed(name, fn):
        def wrapper(*a, **kw):
            t = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                lap[name] = lap.get(name, 0.0) + time.perf_counter() - t
        return wrapper

    def drive(mod, reads, positions, fastq, kk):
        short, short_pos = {}, {}
        with _tmp.TemporaryDirectory() as tmp:
            t = time.perf_counter()
            out_reads, _ = mod.iterative_bubble_popping(reads, positions, 3, kk, 1, short, short_pos, fastq, tmp, 3, set(), 2)
            return time.perf_counter() - t, out_reads

    with contextlib.redirect_stderr(io.StringIO()):
        drive(gu, reads_t, pos_t, fq, k)                      # (first call: the bases go up, buffers are made)
        patched = {"correct_low_coverage_paths": gu.GeneMerGraph.correct_low_coverage_paths,
                   "_junction_paths_on_device": gu.GeneMerGraph._junction_paths_on_device,
                   "_path_overlaps_on_device": gu.GeneMerGraph._path_overlaps_on_device,
                   "filter_paths_between_bubble_starts": gu.GeneMerGraph.filter_paths_between_bubble_starts,
                   "correct_bubble_paths": gu.GeneMerGraph.correct_bubble_paths}
        seq_fn = bp._sequences_for
        try:
            for name, fn in patched.items():
                setattr(gu.GeneMerGraph, name, timed(name, fn))
            bp._sequences_for = timed("_sequences_for", seq_fn)
            bp.release_sequences()   # (a cleaning run uploads its reads' bases once: inside the clock)
            t_arr, out_arr = drive(gu, reads_t, pos_t, fq, k)
        finally:
            bp._sequences_for = seq_fn
            for name, fn in patched.items():
                setattr(gu.GeneMerGraph, name, fn)
        bp.release_sequences()
        t_dict, out_dict = drive(gu, {r: list(v) for r, v in calls.items()}, {r: list(v) for r, v in pos.items()}, fq, k)
    genes_arr = int(out_arr.settled().read_offsets[-1])
    out["iterative_bubble_popping"] = {
        "reads": N, "genes_per_read": L, "vocab": V, "k": k, "error_rate": err, "gene_mers": n_windows,
        "bases": sum(len(v["sequence"]) for v in fq.values()),
        "s_per_call": round(t_arr, 3), "gene_mers_per_s": n_windows / t_arr, "genes_out": genes_arr,
        "input": "array-backed mappings (amira_amd.io.TokenizedReads / TokenizedPositions: what the drop-in's loader hands on)",
        "stages_s_per_call": {"bubble_popping (3 calls of correct_low_coverage_paths)": round(lap.get("correct_low_coverage_paths", 0.0), 3),
                              "  paths between junctions (device search + hashing the nodes' names)": round(lap.get("_junction_paths_on_device", 0.0), 3),
                              "  path filter": round(lap.get("filter_paths_between_bubble_starts", 0.0), 3),
                              "  sketches and overlaps (device)": round(lap.get("_path_overlaps_on_device", 0.0), 3),
                              "    of which the reads' bases to the device, once (PCIe)": round(lap.get("_sequences_for", 0.0), 3),
                              "  operations + rewriting the reads": round(lap.get("correct_bubble_paths", 0.0), 3),
                              "cleaning sweeps (9 builds, 6 corrections, 3 clips) and the rest": round(t_arr - lap.get("correct_low_coverage_paths", 0.0), 3)},
        "from_dicts": {"s_per_call": round(t_dict, 3), "gene_mers_per_s": n_windows / t_dict,
                       "same_result": sum(len(v) for v in out_dict.values()) == genes_arr,
                       "note": "dicts in, dicts out as the reference's driver is called: tokenising once and spelling "
                               "2.9 M gene names and position pairs back into lists is the difference"}}
    bp.release_sequences()
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from amira_oracle import driver as odriver
        n_s, L_s, V_s, k_s, err_s = 500, 30, 150, 3, 0.04   # (the pure-Python oracle takes ~25 s at this size)
        calls_s, pos_s, fq_s = _bubble_inputs(4242, n_s, L_s, V_s, err_s)
        windows_s = sum(max(0, len(v) - k_s + 1) for v in calls_s.values())
        copy = lambda: ({r: list(v) for r, v in calls_s.items()}, {r: list(v) for r, v in pos_s.items()})  # noqa: E731
        with contextlib.redirect_stderr(io.StringIO()):
            t_dev, small_dev = drive(gu, *copy(), fq_s, k_s)
            t_cpu, small_cpu = drive(odriver, *copy(), fq_s, k_s)
        bp.release_sequences()
        out["iterative_bubble_popping"]["cpu_baseline"] = {
            "s_per_call": round(t_cpu, 3), "gene_mers_per_s": windows_s / t_cpu, "cores": 1, "kind": "port",
            "same_result": {r: list(v) for r, v in small_cpu.items()} == {r: list(v) for r, v in small_dev.items()},
            "product_on_the_same_sample_s": round(t_dev, 3),
            "sample": f"the same call on {n_s} reads x {L_s} genes (k = {k_s}) through the pure-Python oracle (oracle/amira_oracle, its "
                      "own restatement of sourmash's sketch): the oracle searches once per pair of junctions, the full "
                      "size would take hours"}
    return out


def run_multi_k(w, vocab, toks, offs, local_rank):
    """SURVEY 8 row f3 (graph_utils.py:258-296 choose_kmer_size): the seven graphs k = 3, 5, ..., 15 of the workload's
    reads — ONE amg_build_multi call (the reads on the device once, every graph by the ordinary build on an engine of
    its own) against seven amg_build calls on engines that each hold a copy of the reads; which key scheme each k took.
    (The shared passes of rounds 2-5, one staged tile for the node passes of all k, bought nothing and are gone.)"""
    from amira_amd import Engine
    ks = list(range(3, 16, 2))
    engines = [Engine(local_rank) for _ in ks]
    try:
        for e in engines:
            e.set_reads(toks, offs, vocab.two_v)
            e.set_timing(False)

        def sync():
            for e in engines:
                e.sync()

        def timed(fn, reps=3):
            fn()
            sync()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            return (time.perf_counter() - t) / reps * 1e3

        many = timed(lambda: Engine.build_multi(engines, ks))
        nodes_multi = [e.counts()["n_nodes"] for e in engines]
        for e in engines:   # (build_multi left the others borrowing the first one's reads)
            e.set_reads(toks, offs, vocab.two_v)
        singles = [timed(lambda i=i: engines[i].build(ks[i])) for i in range(len(ks))]
        nodes_single = [e.counts()["n_nodes"] for e in engines]
        schemes = [engines[i].counts()["exact_keys"] for i in range(len(ks))]
        return {"ks": ks, "build_many_ms": round(many, 3), "seven_builds_ms": round(sum(singles), 3),
                "single_build_ms": [round(x, 3) for x in singles],
                "key_scheme": {"exact tuple": [k_ for k_, s_ in zip(ks, schemes) if s_ == 1],
                               "94-bit fingerprint, verified": [k_ for k_, s_ in zip(ks, schemes) if s_ == 2],
                               "32-byte slots": [k_ for k_, s_ in zip(ks, schemes) if s_ == 0]},
                "nodes": nodes_single, "same_graph_sizes": nodes_single == nodes_multi}
    finally:
        for e in engines:
            e.close()


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start one rank per GPU as fresh child processes
    (torch.distributed.run, rendezvous on 127.0.0.1) BEFORE this process makes any GPU call — counting the devices
    does not initialise HIP — and hand rank 0's JSON line on as this process's own last line.  A box with fewer than
    N GPUs fails here, loudly, instead of measuring something else."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and os.environ.get("AMG_SHARE_GPU") != "1":   # (AMG_SHARE_GPU=1 + AMG_DIST_BACKEND=gloo: functional test)
        print(f"bench.py: --gpus {n} requested but this machine has {have} GPU(s); refusing to report a "
              f"{n}-GPU number from fewer devices", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n}-rank run failed (exit code {proc.returncode})", file=sys.stderr, flush=True)
        return proc.returncode or 1
    if json.loads(line).get("n_gpus") != n:
        print(f"bench.py: the ranks reported n_gpus != {n}", file=sys.stderr, flush=True)
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)   # (a fresh box runs its first steps 3 % slower)
    ap.add_argument("--workload", default="cfg3-sweep", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="cpu_baseline_python on SURVEY 8(d)'s full 1 %% read sample (~100 s of one core) instead of the "
                         "~12 s sample of the default line")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the extra `cfg4` object (read-path clustering) of the default line")
    ap.add_argument("--no-fused-line", action="store_true", help="skip the extra `fused_first_filter` measurement")
    ap.add_argument("--no-merge", action="store_true", help="N > 1: independent shards, no table merge")
    ap.add_argument("--fused-filter", action="store_true",
                    help="sweep: first build and filter_graph(3,1) as one device pass (amg_build_filtered), as the "
                         "merged multi-GPU path always does; not the default so that `value` stays the sweep as the "
                         "reference spells it")
    ap.add_argument("--force-merge", action="store_true",
                    help="run the merged (multi-GPU) code path even at world size 1 (self-test)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) "
                 f"or call bench.py --gpus {args.gpus} without a launcher and it starts them itself")

    # CPU yardsticks first (rank 0, N = 1), before this process touches the GPU: the N-core line
    # starts worker processes
    cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != "cfg4":
        cpu["cpu_baseline"] = cpu_baseline_c(w)
        cpu["cpu_baseline_python"] = cpu_baseline(w, budget_s=1e9) if args.full_cpu_baseline else cpu_baseline(w)
        cpu["cpu_baseline_ncore"] = cpu_baseline_ncore(w)

    cpu4 = None
    if (rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "cfg3-sweep" and not args.no_cfg4
            and not args.no_e2e):
        cpu4 = cpu_baseline_cfg4(WORKLOADS["cfg4"])
    if args.workload == "cfg4":
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = {"cpu_baseline": cpu_baseline_cfg4(w)}
        out = run_cfg4(args, w, rank, world, local_rank)
        out.update(cpu)
        print(json.dumps(out), flush=True)
        return

    import torch
    shared_gpu = os.environ.get("AMG_SHARE_GPU") == "1" and torch.cuda.device_count() < world
    if shared_gpu:   # functional test of the N > 1 path on a box with fewer GPUs (backend gloo): never a scaling figure
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_merge:
        import torch.distributed as dist
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ.pop("NCCL_DEBUG")  # the RCCL version banner goes to stdout at exit, after the JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        backend = os.environ.get("AMG_DIST_BACKEND", "nccl")   # "nccl" is RCCL; "gloo" only for functional tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from amira_amd import Engine
    from amira_amd.dist import dist_build
    merge = (world > 1 or args.force_merge) and not args.no_merge

    def build(fused_filter=None):
        if merge:
            # first build of the sweep: filter_graph(3,1) is fused into the merge so that the
            # low-coverage nodes (90 % of an uncorrected graph) are never replicated
            dist_build(eng, k, None, *(fused_filter or (1, 1)))
        elif fused_filter and args.fused_filter:
            eng.build_filtered(k, *fused_filter)
        else:
            eng.build(k)

    # weak scaling: rank r holds reads [r N, (r+1) N) of the global stream
    N, L, k = w["N"], w["L"], w["k"]
    vocab, toks, offs = make_tokens(w, rank * N, (rank + 1) * N)
    dev = torch.device("cuda", local_rank)
    d_toks = torch.from_numpy(toks).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    d_gs = d_ge = d_rl = None
    if w["sweep"]:
        d_gs = (torch.arange(L, dtype=torch.int64, device=dev) * 1000).repeat(N)
        d_ge = d_gs + 899
        d_rl = torch.full((N,), L * 1000 + 100, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng = Engine(local_rank)
    n_windows = N * (L - k + 1)
    stage_ms, stage_each, info = {}, {}, {}

    def tally():
        for name, ms in eng.timings():  # HIP events recorded on the engine's own stream
            stage_ms.setdefault(name, [0.0, 0])
            stage_ms[name][0] += ms
            stage_ms[name][1] += 1
            stage_each.setdefault(name, []).append(ms)

    def sweep_after_first_build(record, readback=None, moved=None, full=True):
        if not (merge or args.fused_filter):
            eng.filter(3, 1)
            if record:
                tally()
        if record and "marked_reads" not in info:
            info["marked_reads"] = eng.counts()["n_reads_to_correct"]
        eng.correct_reads()
        if record:
            tally()
        eng.adopt_corrected()
        build()
        if record:
            tally()
        eng.remove_short_linear_paths(k, want_ids=False)   # (device-resident region: the list of removed ids stays there)
        if record:
            tally()
        n_out = eng.correct_reads()
        if record:
            tally()
        if readback is not None and full:
            # the corrected calls + positions the reference hands on (graph_utils.py:165) as the drop-in hands them on:
            # contiguous corrected position arrays, every read's (Engine.corrected = what amira_amd.io.DeviceCorrected
            # fetches for GeneMerGraph.correct_reads' mappings)
            eng.corrected(*n_out, True, buf=readback, pos32=True)
        elif readback is not None:
            # the boundary's lighter form: 32-bit positions, and only those the correction PRODUCED — an untouched or
            # trimmed read's positions are a slice of the arrays the caller already has, named by offset
            # (amg_get_corrected32); putting the two together is left to the caller
            got = eng.corrected32(*n_out, buf=readback)
            if moved is not None:
                moved["new_positions"] = len(got["new_start"])
        eng.adopt_corrected()
        build()
        if record:
            tally()

    def step(record):
        # per-stage HIP events (two per stage, ~5 us of stream idle each) only in the instrumented step: the timed
        # steps do not pay for the measurement
        eng.set_timing(record)
        # inputs are resident in HBM and handed over as borrowed device pointers (no copy, no PCIe)
        eng.set_reads_device(d_toks.data_ptr(), d_offs.data_ptr(), N, vocab.two_v, borrow=True)
        if w["sweep"]:
            eng.set_positions_device(d_gs.data_ptr(), d_ge.data_ptr(), d_rl.data_ptr(), borrow=True)
        build((3, 1) if w["sweep"] else None)
        if record:
            tally()
        if w["sweep"]:
            sweep_after_first_build(record)
        # component ids + per-node edge lists of the step's final graph (made on demand otherwise): the
        # step leaves behind everything GeneMerGraph.__init__ would
        eng.finalize()
        if record:
            tally()

    for _ in range(args.warmup):
        step(False)
    # one instrumented step outside the timed region (per-stage HIP-event times, counts)
    step(True)
    counts = eng.counts()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the same sweep with its first two steps as one device pass (amg_build_filtered): reported beside `value`
    fused_line = None
    if w["sweep"] and world == 1 and not merge and not args.fused_filter and not args.no_fused_line:
        args.fused_filter = True
        for _ in range(max(args.warmup, 1)):
            step(False)
        eng.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(False)
        eng.sync()
        df = (time.perf_counter() - t1) / args.steps
        args.fused_filter = False
        fused_line = {"value": n_windows / df, "unit": "gene-mers/s", "ms_per_step": df * 1e3, "steps": args.steps,
                      "what": "the same step with build 1 and filter_graph(3,1) as ONE device pass (amg_build_filtered): "
                              "nodes below the threshold are never ranked, stored or joined by edges; everything the "
                              "rest of the sweep reads is identical (tests/test_gpu_filtered.py, full size in "
                              "tests/test_gpu_fullsize.py); `value` above is the sweep as the reference spells it"}

    # ---- SURVEY 8(d)'s timed region: host CSR -> host graph arrays (N = 1)
    e2e = None
    if world == 1 and not merge and not args.no_e2e:
        pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
        h_toks, h_offs = pin(toks), pin(offs)
        T = len(toks)
        h_gs = h_ge = h_rl = None
        d_gs32 = d_ge32 = None
        if w["sweep"]:
            # positions cross the boundary as int32 (amg_set_positions32 / amg_get_corrected32): read coordinates fit,
            # and they are 16 of the 20 bytes per gene the region moves each way
            h_gs = pin(np.tile(np.arange(L, dtype=np.int32) * 1000, N))
            h_ge = pin(h_gs.numpy() + 899)
            h_rl = pin(np.full(N, L * 1000 + 100, np.int64))
            d_gs32 = torch.empty(T, dtype=torch.int32, device=dev)
            d_ge32 = torch.empty(T, dtype=torch.int32, device=dev)
        cap_t, cap_r = T + T // 8 + 1024, N + 1024
        cap_d, cap_e = max(counts["n_nodes"] * 2, T // 8) + 1024, max(counts["n_edges"] * 2, T // 4) + 1024
        pinned = lambda n, dt_: torch.empty(n, dtype=dt_).pin_memory().numpy()
        buf = {"tokens": pinned(cap_d * k, torch.int32), "coverage": pinned(cap_d, torch.int32).view(np.uint32),
               "first_token": pinned(cap_d, torch.int64), "first_dir": pinned(cap_d, torch.int8),
               "component": pinned(cap_d, torch.int32), "alive": pinned(cap_d, torch.uint8),
               "src": pinned(cap_e, torch.int32), "tgt": pinned(cap_e, torch.int32), "sdir": pinned(cap_e, torch.int8),
               "tdir": pinned(cap_e, torch.int8), "ecoverage": pinned(cap_e, torch.int32).view(np.uint32),
               "ealive": pinned(cap_e, torch.uint8), "tok_node": pinned(cap_t, torch.int32),
               "tok_dir": pinned(cap_t, torch.int8)}
        if w["sweep"]:
            buf.update({"c_tokens": pinned(cap_t, torch.int32), "c_read_offsets": pinned(cap_r, torch.int64),
                        "c_orig_read": pinned(cap_r, torch.int32), "c_changed": pinned(cap_r, torch.uint8),
                        "c_pos_src": pinned(cap_r, torch.int64),
                        "c_new_start": pinned(cap_t, torch.int32), "c_new_end": pinned(cap_t, torch.int32),
                        "c_gene_start": pinned(cap_t, torch.int64), "c_gene_end": pinned(cap_t, torch.int64),
                        "c_gene_start32": pinned(cap_t, torch.int32), "c_gene_end32": pinned(cap_t, torch.int32)})
        side = torch.cuda.Stream(device=dev)
        moved = {}
        ev_reads, ev_pos = torch.cuda.Event(), torch.cuda.Event()

        def e2e_step(full=True):
            with torch.cuda.stream(side):
                d_toks.copy_(h_toks, non_blocking=True)
                d_offs.copy_(h_offs, non_blocking=True)
                ev_reads.record(side)
                if w["sweep"]:   # twice the bytes of the genes: uploaded while the first build runs
                    d_gs32.copy_(h_gs, non_blocking=True)
                    d_ge32.copy_(h_ge, non_blocking=True)
                    d_rl.copy_(h_rl, non_blocking=True)
                    ev_pos.record(side)
            ev_reads.synchronize()
            eng.set_reads_device(d_toks.data_ptr(), d_offs.data_ptr(), N, vocab.two_v, borrow=True)
            build()
            if w["sweep"]:
                ev_pos.synchronize()
                eng.set_positions32_device(d_gs32.data_ptr(), d_ge32.data_ptr(), d_rl.data_ptr())
                sweep_after_first_build(False, readback=buf, moved=moved, full=full)
            eng.nodes(buf)
            eng.edges(buf)
            eng.read_nodes(buf)

        def timed_e2e(full):
            e2e_step(full)
            torch.cuda.synchronize()
            n = max(2, min(args.steps, 5))
            t0 = time.perf_counter()
            for _ in range(n):
                e2e_step(full)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, n

        de, n_e2e = timed_e2e(True)
        h2d = T * 4 + (N + 1) * 8 + (T * 8 + N * 8 if w["sweep"] else 0)
        c_fin = eng.counts()
        d2h_graph = c_fin["n_nodes"] * (4 * k + 4 + 8 + 1 + 4 + 1) + c_fin["n_edges"] * 14 + c_fin["n_tokens"] * 5
        d2h = d2h_graph + (c_fin["n_tokens"] * (4 + 8) + c_fin["n_reads"] * 13 if w["sweep"] else 0)
        e2e = {"value": n_windows / de, "unit": "gene-mers/s", "ms_per_step": de * 1e3, "steps": n_e2e,
               "timed_region": "SURVEY 8(d): host CSR arrays -> host graph arrays, H2D and D2H inside the clock",
               "h2d_bytes": int(h2d), "d2h_bytes": int(d2h),
               "region": "pinned host CSR (genes, offsets, gene positions as int32, read lengths) -> H2D -> the step -> D2H "
                         "of the final graph (node + edge arrays, node id and direction per window)"
                         + (" and of the corrected calls with EVERY read's corrected positions as contiguous arrays, int32 "
                            "because they fit (Engine.corrected(pos32=True): what the drop-in's correct_reads mappings "
                            "fetch, amira_amd.io.DeviceCorrected)" if w["sweep"] else "")
                         + "; position upload overlapped with the first build on a second stream"}
        if w["sweep"]:
            dd, n_dd = timed_e2e(False)
            d2h32 = d2h_graph + c_fin["n_tokens"] * 4 + c_fin["n_reads"] * 21 + moved.get("new_positions", 0) * 8
            e2e["delta_positions32"] = {
                "value": n_windows / dd, "unit": "gene-mers/s", "ms_per_step": dd * 1e3, "steps": n_dd, "d2h_bytes": int(d2h32),
                "what": "the same region with the corrected positions in the boundary's lighter form (amg_get_corrected32: "
                        "int32, only the positions the corrections produced + an offset per read into the caller's own "
                        "arrays; the gather into contiguous arrays is left to the caller, outside the clock) — the round-4 "
                        "`e2e` figure"}

    if e2e is not None and w["sweep"]:
        try:
            e2e["pipelined"] = run_e2e_pipelined(w, vocab, toks, offs, k, n_windows, local_rank, counts,
                                                 lanes=int(os.environ.get("AMG_E2E_LANES", "2")))
        except Exception as err:  # noqa: BLE001  (an extra figure: never costs the line)
            e2e["pipelined"] = {"error": repr(err)}

    # ---- the same sweep through the reference-shaped Python API (amira_amd.graph_utils / GeneMerGraph), inputs as
    # array-backed mappings (amira_amd.io): what a caller of the drop-in pays per cleaning iteration, PCIe included
    api_e2e = None
    if w["sweep"] and world == 1 and not merge and not args.no_e2e:
        api_e2e = run_api_e2e(w, vocab, toks, offs, k, n_windows)

    front_end = multi_k = bubbles = None
    if w["sweep"] and world == 1 and not merge and not args.no_e2e and rank == 0:
        for name, fn in (("front_end", lambda: run_front_end(w, vocab, toks, offs, k, n_windows)),
                         ("multi_k", lambda: run_multi_k(w, vocab, toks, offs, local_rank)),
                         ("bubbles", lambda: run_bubbles(local_rank, with_cpu=not args.no_cpu_baseline))):
            try:
                got = fn()
            except Exception as err:  # noqa: BLE001  (extra figures: never cost the line)
                got = {"error": repr(err)}
            if name == "front_end":
                front_end = got
            elif name == "multi_k":
                multi_k = got
            else:
                bubbles = got

    out = None
    if rank == 0:
        stage_avg = {n: v[0] / v[1] for n, v in stage_ms.items()}     # ms per launch
        stage_tot = {n: v[0] for n, v in stage_ms.items()}            # ms per step
        n_gapped = info.get("marked_reads", 0)
        cands = {s: stage_bytes(s, k, L, n_windows, N, n_gapped) for s in stage_tot}
        # dominant kernel = the stage with the largest time per step, full stop; `largest_stages` lists the top four
        ranked = sorted((s for s in cands if cands[s]), key=lambda s: -stage_tot[s])
        dom = ranked[0]
        achieved = cands[dom] / (stage_avg[dom] * 1e-3) / 1e9
        per_kernel = {s: {"ms_per_step": round(stage_tot[s], 3), "avg_launch_ms": round(stage_avg[s], 4),
                          "achieved_GBs": round(cands[s] / (stage_avg[s] * 1e-3) / 1e9, 1)} for s in ranked[:4]}
        exact = bool(counts.get("exact_keys"))
        buckets = os.environ.get("AMG_NODE_BUCKETS", "1") != "0" and k in (3, 5, 7) and not os.environ.get("AMG_X_GENERIC_K")
        kernel_of = {"node_upsert": ("k_nodes_m" if buckets else "k_nodes_v") if exact else "k_node_upsert",
                     "edge_upsert": "k_edges_v" if exact else "k_edges",
                     "node_count": "k_count_ids",
                     "edge_count": "k_count_ids", "correct_positions": "k_corr_nw_fast",
                     "correct_gapped": "k_corr_gapped_lean + k_corr_gapped_fast"}
        build_ms = sum(stage_tot.get(n, 0.0) for n in ("read_stats", "table_clear", "graph_upsert", "node_table_clear",
                                                        "node_upsert_head", "node_upsert", "node_rank", "node_filter", "edge_table_clear", "edge_upsert_head",
                                                        "edge_upsert",
                                                        "edge_rank", "node_count", "edge_count", "edge_emit",
                                                        "components", "adjacency"))
        n_builds = max(stage_ms.get("graph_upsert", stage_ms.get("node_upsert", [0, 1]))[1], 1)
        survey_b = 4.0 * L / (L - k + 1) + 5 + (4 * k + 8) + 20.0 * (L - k) / (L - k + 1)
        # both table passes together: their algorithmic bytes over their time (per step)
        table_passes = None
        tp = [x for x in ("node_upsert", "node_upsert_head", "edge_upsert", "edge_upsert_head", "graph_upsert") if x in stage_tot]
        if tp:
            tp_bytes = sum((cands.get(x) or 0.0) * stage_ms[x][1] for x in tp)
            tp_ms = sum(stage_tot[x] for x in tp)
            table_passes = {"stages": tp, "algorithmic_bytes_per_step": tp_bytes, "ms_per_step": round(tp_ms, 4),
                            "achieved": tp_bytes / (tp_ms * 1e-3) / 1e9, "unit": "GB/s",
                            "frac": tp_bytes / (tp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the whole step by SURVEY 8(d)'s figure: B_sweep = B_build + 12 + B_build per input gene-mer (126 B at k = 5)
        whole_sweep = None
        if w["sweep"]:
            b_sweep = 2 * survey_b + 12.0
            ach = b_sweep * n_windows / (dt / args.steps) / 1e9
            whole_sweep = {"algorithmic_bytes_per_gene_mer": b_sweep, "ms_per_step": dt * 1e3 / args.steps,
                           "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this
        # process, so the per-launch FETCH_SIZE + WRITE_SIZE of the last committed
        # `rocprofv3 --pmc` passes over this same command (profiles/) is reported, or null
        traffic, traffic_note = None, None
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sweep_pmc_summary.json")))
        pmc_path = pmc_files[-1] if pmc_files else ""
        if w["sweep"] and world == 1 and pmc_path:
            try:
                pmc = json.load(open(pmc_path))
                # (the short head launch of a table pass is a template instance of its own: the main launch is the
                # instance that moves the most)
                row = max((r for r in pmc["kernels"] if r["kernel"].split("<")[0] == kernel_of[dom]),
                          key=lambda r: r.get("FETCH_SIZE_KB_mean", 0) + r.get("WRITE_SIZE_KB_mean", 0))
                n = min(len(row["FETCH_SIZE_KB_per_launch"]), len(row["WRITE_SIZE_KB_per_launch"]))
                # gfx950: FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes (MI355X_MICROARCH.md,
                # HBM): every load of the table passes is such a load (token stream, slot probes), so the read side is
                # doubled; WRITE_SIZE is exact for the 16-byte-per-lane stores
                traffic = sum((2.0 * row["FETCH_SIZE_KB_per_launch"][i] + row["WRITE_SIZE_KB_per_launch"][i]) * 1024.0
                              for i in range(n)) / n
                traffic_note = ("mean over the launches of one sweep, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 counts the "
                                "128-byte requests of 16-byte-per-lane loads at 64 bytes; calibrated for streams, assumed for "
                                "the slot probes), separate --pmc passes, from profiles/" + os.path.basename(pmc_path))
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "gene-mers/s to corrected GeneMerGraph" if w["sweep"] else "gene-mers/s to GeneMerGraph (build + coverage)",
            "value": world * n_windows * args.steps / dt, "unit": "gene-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "reads_per_s": world * N * args.steps / dt,
            "config": {"workload": w["desc"] + ("; first build and filter_graph(3,1) as one device pass (--fused-filter)"
                                                 if (args.fused_filter and w["sweep"] and not merge) else ""),
                       "timed_region": "device-resident: inputs in HBM when the clock starts, nothing read back "
                                       "(`e2e` is SURVEY 8(d)'s host-to-host region)",
                       "reads_per_gpu": N, "genes_per_read": L, "k": k,
                       "vocab": w["V"], "error_rate": w["err"], "gene_mers_per_gpu": n_windows,
                       "final_nodes": counts["n_nodes"], "final_edges": counts["n_edges"],
                       "shared_gpu_functional_test_only": True if shared_gpu else None,
                       "multi_gpu": ("n/a" if not (world > 1 or merge) else
                                     "read shards + key-owner table merge per build (RCCL all-to-all + all-gather)"
                                     if merge else "independent read shards, no table merge")},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": cands[dom],
                         "avg_launch_ms": stage_avg[dom], "launches_per_step": stage_ms[dom][1],
                         # every launch of the step on its own (a cleaning sweep: the first build's pass over uncorrected
                         # reads creates 5.4 M keys, the rebuild's finds nearly all of its keys; the third graph of the
                         # sweep is made from the second one's live part and has no table pass at all)
                         "launch_ms": [round(x, 4) for x in stage_each.get(dom, [])],
                         "launch_frac": [round(cands[dom] / (x * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) for x in stage_each.get(dom, []) if x > 0],
                         "largest_stages": per_kernel,
                         "table_passes": table_passes,
                         "whole_sweep": whole_sweep,
                         "whole_build": {"algorithmic_bytes_per_gene_mer": survey_b,
                                         "ms_per_build": build_ms / n_builds,
                                         "achieved": survey_b * n_windows / (build_ms / n_builds * 1e-3) / 1e9,
                                         "unit": "GB/s"}},
            "stages_ms_per_step": {n: round(v, 3) for n, v in stage_tot.items()},
        }
        if fused_line is not None:
            out["fused_first_filter"] = fused_line
        if e2e is not None:
            out["e2e"] = e2e
        if api_e2e is not None:
            out["api_e2e"] = api_e2e
        if front_end is not None:
            out["front_end"] = front_end
        if multi_k is not None:
            out["multi_k"] = multi_k
        if bubbles is not None:
            out["bubbles"] = bubbles
        out.update(cpu)
    # ---- BASELINE configs[3] beside it: build + read-path clustering through the Python API (its own engine)
    if rank == 0 and world == 1 and not merge and w["sweep"] and not args.no_cfg4 and not args.no_e2e:
        eng.close()
        eng = None
        c4 = run_cfg4(argparse.Namespace(steps=2, warmup=1), WORKLOADS["cfg4"], 0, 1, local_rank)
        out["cfg4"] = {key: c4[key] for key in ("metric", "value", "unit", "ms_per_step", "steps", "reads_per_s",
                                                 "config", "stages_s_per_step", "roofline")}
        if cpu4 is not None:
            out["cfg4"]["cpu_baseline"] = cpu4
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()  # RCCL prints its version banner here: keep the JSON line last
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
