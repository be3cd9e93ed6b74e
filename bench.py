#!/usr/bin/env python3
"""bench.py — gene-mer graph hot path on MI355X (driver contract: see the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg3-sweep]

A "step" is one pass of the hot path over the resident synthetic read set:
  cfg2        100 k reads x 40 genes, k=5, 5 k-gene vocabulary: graph build + coverage
  cfg3        1 M reads x 60 genes, k=5, 20 k-gene vocabulary: graph build
  cfg3-sweep  cfg3 + error-correction sweep (build -> filter(3,1) -> correct -> build ->
              clip(k) -> correct -> build)
Inputs (CSR tokens) are resident in HBM before the timed region.  Prints ONE JSON line.
"""
import argparse
import ctypes
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
                            "error-correction sweep"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes_per_window(k, L):
    """SURVEY.md section 8(d), split per kernel (tokens int32, slot/node id int32, dir int8,
    counters uint32, first-seen uint64)."""
    tok = 4.0 * L / (L - k + 1)
    node_kernel = tok + 5 + (4 * k + 8)            # token read + (slot, dir) write + key/ctr RMW
    edge_kernel = (4 + 1) + 4 + (12 + 8) * (L - k) / (L - k + 1)  # slot,dir read + id write + edge RMW
    return node_kernel, edge_kernel


def make_tokens(w, lo, hi):
    from amira_amd import synth
    from amira_amd.tokens import Vocabulary
    ids, sts = synth.block_reads(w["seed"], lo, hi, w["L"], w["V"], w["err"])
    vocab = Vocabulary(synth.gene_names(w["V"]))
    rank = np.array([vocab.rank[n] for n in synth.gene_names(w["V"])], dtype=np.int64)
    r = rank[ids]
    toks = np.where(sts == 1, vocab.V + r, vocab.V - 1 - r).astype(np.int32)
    offs = (np.arange(hi - lo + 1, dtype=np.int64) * w["L"])
    return vocab, toks.reshape(-1), offs


def cpu_baseline(w, budget_s=20.0):
    """Pure-Python restatement of the reference (oracle/, same sha256+pickle work per
    gene-mer as construct_gene.py:5-10) on a bounded sample of the same workload, 1 core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from amira_amd import synth
    from amira_oracle import GeneMerGraph, values
    values.CACHE_HASHES = False
    n = 250
    done_windows, spent, total_reads = 0, 0.0, 0
    while spent < budget_s and total_reads < w["N"]:
        ids, sts = synth.block_reads(w["seed"], total_reads, total_reads + n, w["L"], w["V"], w["err"])
        reads = synth.to_read_dict(ids, sts, synth.gene_names(w["V"]), first=total_reads)
        t = time.perf_counter()
        GeneMerGraph(reads, w["k"])
        spent += time.perf_counter() - t
        done_windows += n * (w["L"] - w["k"] + 1)
        total_reads += n
    values.CACHE_HASHES = True
    return {"value": done_windows / spent, "unit": "gene-mers/s", "cores": 1, "kind": "port",
            "sample": f"{total_reads} reads of the same workload, build only, pure-Python oracle "
                      f"with per-call sha256+pickle (reference cost model), {spent:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from amira_amd import Engine

    # weak scaling: every rank holds N reads of the global stream (rank r: reads [rN, (r+1)N))
    N = w["N"]
    vocab, toks, offs = make_tokens(w, rank * N, (rank + 1) * N)
    eng = Engine(local_rank)
    eng.set_reads(toks, offs, vocab.two_v)  # H2D happens here, outside the timed region
    n_windows = N * (w["L"] - w["k"] + 1)

    def step():
        eng.build(w["k"])

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    eng.sync()
    stage_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for name, ms in eng.timings():   # HIP events recorded on the engine's own stream
            stage_ms[name] = stage_ms.get(name, 0.0) + ms
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    counts = eng.counts()
    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = world * n_windows * args.steps / dt
        node_b, edge_b = algorithmic_bytes_per_window(w["k"], w["L"])
        stage_avg = {k_: v / args.steps for k_, v in stage_ms.items()}
        dom = max(("node_upsert", "edge_upsert"), key=lambda s: stage_avg.get(s, 0.0))
        per_launch = (node_b if dom == "node_upsert" else edge_b) * n_windows
        achieved = per_launch / (stage_avg[dom] * 1e-3) / 1e9
        out = {
            "metric": "gene-mers/s to GeneMerGraph (build + coverage)", "value": value,
            "unit": "gene-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "reads_per_s": world * N * args.steps / dt,
            "config": {"workload": w["desc"], "reads_per_gpu": N, "genes_per_read": w["L"],
                       "k": w["k"], "vocab": w["V"], "error_rate": w["err"],
                       "gene_mers_per_gpu": n_windows, "nodes": counts["n_nodes"],
                       "edges": counts["n_edges"],
                       "multi_gpu": "independent read shards, no table merge (round 1)" if world > 1 else "n/a"},
            "roofline": {"bound": "hbm", "kernel": "k_node_upsert" if dom == "node_upsert" else "k_edges",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": per_launch,
                         "avg_launch_ms": stage_avg[dom]},
            "stages_ms": stage_avg,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
