#!/usr/bin/env python3
"""bench.py — read -> corrected gene-mer graph hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3-sweep|cfg3|cfg2]

A "step" is one pass of the hot path over one batch of synthetic gene calls whose CSR
token arrays (and gene positions) are already resident in HBM:
  cfg3-sweep (default) 1 M reads x 60 genes, k=5, 20 k-gene vocabulary, 2 % substitutions:
             build -> filter_graph(3,1) -> correct_reads -> build ->
             remove_short_linear_paths(5) -> correct_reads -> build
             (the cleaning sweep of graph_utils.py:145-166; BASELINE.json configs[2])
  cfg3       the first build of that sweep only
  cfg2       100 k reads x 40 genes, k=5, 5 k-gene vocabulary: build + coverage (configs[1])
Prints ONE JSON line (driver contract) with `roofline` and `cpu_baseline` objects.
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
    ids, sts = synth.block_reads(w["seed"], lo, hi, w["L"], w["V"], w["err"])
    vocab = Vocabulary(synth.gene_names(w["V"]))
    rank = np.array([vocab.rank[n] for n in synth.gene_names(w["V"])], dtype=np.int64)
    r = rank[ids]
    toks = np.where(sts == 1, vocab.V + r, vocab.V - 1 - r).astype(np.int32)
    offs = (np.arange(hi - lo + 1, dtype=np.int64) * w["L"])
    return vocab, toks.reshape(-1), offs


def cpu_baseline(w, budget_s=20.0):
    """Pure-Python restatement of the reference (oracle/: same sha256 + pickle work per
    gene-mer as construct_gene.py:5-10) on a bounded sample of the same workload, 1 core —
    the reference pipeline always builds with cores=1."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from amira_amd import synth
    from amira_oracle import GeneMerGraph, driver, values
    values.CACHE_HASHES = False
    L, k = w["L"], w["k"]
    try:
        if not w["sweep"]:
            n, done, spent, first = 250, 0, 0.0, 0
            while spent < budget_s and first < w["N"]:
                ids, sts = synth.block_reads(w["seed"], first, first + n, L, w["V"], w["err"])
                reads = synth.to_read_dict(ids, sts, synth.gene_names(w["V"]), first=first)
                t = time.perf_counter()
                GeneMerGraph(reads, k)
                spent += time.perf_counter() - t
                done += n * (L - k + 1)
                first += n
            sample = (f"{first} reads of the same stream, build only, pure-Python oracle with "
                      f"per-call sha256+pickle (reference cost model), {spent:.1f} s")
        else:
            # a sweep needs depth to leave anything after filter_graph(3,1): same L, k, error
            # rate, vocabulary scaled down so 1 200 reads give ~140x depth
            n, V = 1200, 500
            ids, sts = synth.block_reads(w["seed"], 0, n, L, V, w["err"])
            reads = synth.to_read_dict(ids, sts, synth.gene_names(V))
            pos = synth.positions_for(reads)
            fq = driver.FakeFastq(synth.fake_fastq_lengths(reads))
            t = time.perf_counter()
            driver.correction_sweep(reads, pos, k, fq, 3)
            spent = time.perf_counter() - t
            done = n * (L - k + 1)
            sample = (f"{n} reads x {L} genes, {V}-gene vocab (depth-preserving down-scale of the "
                      f"workload), full sweep, pure-Python oracle with per-call sha256+pickle, {spent:.1f} s")
    finally:
        values.CACHE_HASHES = True
    return {"value": done / spent, "unit": "gene-mers/s", "cores": 1, "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3-sweep", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-merge", action="store_true", help="N > 1: independent shards, no table merge")
    ap.add_argument("--force-merge", action="store_true",
                    help="run the merged (multi-GPU) code path even at world size 1 (self-test)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_merge:
        import torch.distributed as dist
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ.pop("NCCL_DEBUG")  # the RCCL version banner goes to stdout at exit, after the JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    from amira_amd import Engine
    from amira_amd.dist import dist_build
    merge = (world > 1 or args.force_merge) and not args.no_merge

    def build(fused_filter=None):
        if merge:
            # first build of the sweep: filter_graph(3,1) is fused into the merge so that the
            # low-coverage nodes (90 % of an uncorrected graph) are never replicated
            dist_build(eng, k, None, *(fused_filter or (1, 1)))
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
    stage_ms, info = {}, {}

    def tally():
        for name, ms in eng.timings():  # HIP events recorded on the engine's own stream
            stage_ms.setdefault(name, [0.0, 0])
            stage_ms[name][0] += ms
            stage_ms[name][1] += 1

    def step(record):
        # inputs are resident in HBM and handed over as borrowed device pointers (no copy, no PCIe)
        eng.set_reads_device(d_toks.data_ptr(), d_offs.data_ptr(), N, vocab.two_v, borrow=True)
        if w["sweep"]:
            eng.set_positions_device(d_gs.data_ptr(), d_ge.data_ptr(), d_rl.data_ptr(), borrow=True)
        build((3, 1) if w["sweep"] else None)
        if record:
            tally()
        if not w["sweep"]:
            return
        if not merge:
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
        eng.remove_short_linear_paths(k)
        if record:
            tally()
        eng.correct_reads()
        if record:
            tally()
        eng.adopt_corrected()
        build()
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

    out = None
    if rank == 0:
        stage_avg = {n: v[0] / v[1] for n, v in stage_ms.items()}     # ms per launch
        stage_tot = {n: v[0] for n, v in stage_ms.items()}            # ms per step
        n_gapped = info.get("marked_reads", 0)
        cands = {s: stage_bytes(s, k, L, n_windows, N, n_gapped) for s in stage_tot}
        # dominant kernel = the stage with the largest time per step; the two table passes run within
        # a few per cent of each other, so stages within 3 % of the top are ranked by the bytes they move
        ranked = sorted((s for s in cands if cands[s]), key=lambda s: -stage_tot[s])
        top = [s for s in ranked if stage_tot[s] >= 0.97 * stage_tot[ranked[0]]]
        dom = max(top, key=lambda s: cands[s])
        achieved = cands[dom] / (stage_avg[dom] * 1e-3) / 1e9
        per_kernel = {s: {"ms_per_step": round(stage_tot[s], 3), "avg_launch_ms": round(stage_avg[s], 4),
                          "achieved_GBs": round(cands[s] / (stage_avg[s] * 1e-3) / 1e9, 1)} for s in ranked[:4]}
        exact = bool(counts.get("exact_keys"))
        kernel_of = {"graph_upsert": "k_graph_x", "node_upsert": "k_nodes_x" if exact else "k_node_upsert",
                     "edge_upsert": "k_edges_x" if exact else "k_edges", "node_count": "k_count_ids",
                     "edge_count": "k_count_ids", "correct_positions": "k_corr_nw_fast",
                     "correct_gapped": "k_corr_gapped_fast"}
        build_ms = sum(stage_tot.get(n, 0.0) for n in ("read_stats", "table_clear", "graph_upsert", "node_table_clear", "node_upsert", "node_rank",
                                                        "edge_table_clear", "edge_upsert", "edge_rank", "node_count",
                                                        "edge_count", "edge_emit", "components", "adjacency"))
        n_builds = max(stage_ms.get("graph_upsert", stage_ms.get("node_upsert", [0, 1]))[1], 1)
        survey_b = 4.0 * L / (L - k + 1) + 5 + (4 * k + 8) + 20.0 * (L - k) / (L - k + 1)
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
                row = next(r for r in pmc["kernels"] if r["kernel"].split("<")[0] == kernel_of[dom])
                n = min(len(row["FETCH_SIZE_KB_per_launch"]), len(row["WRITE_SIZE_KB_per_launch"]))
                traffic = sum((row["FETCH_SIZE_KB_per_launch"][i] + row["WRITE_SIZE_KB_per_launch"][i]) * 1024.0
                              for i in range(n)) / n
                traffic_note = ("mean over the launches of one sweep, (FETCH_SIZE + WRITE_SIZE) x 1024, separate "
                                "--pmc passes, from profiles/" + os.path.basename(pmc_path) + "; not corrected for the "
                                "gfx950 FETCH_SIZE under-count of wide coalesced streams")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "gene-mers/s to corrected GeneMerGraph" if w["sweep"] else "gene-mers/s to GeneMerGraph (build + coverage)",
            "value": world * n_windows * args.steps / dt, "unit": "gene-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "reads_per_s": world * N * args.steps / dt,
            "config": {"workload": w["desc"], "reads_per_gpu": N, "genes_per_read": L, "k": k,
                       "vocab": w["V"], "error_rate": w["err"], "gene_mers_per_gpu": n_windows,
                       "final_nodes": counts["n_nodes"], "final_edges": counts["n_edges"],
                       "multi_gpu": ("n/a" if not (world > 1 or merge) else
                                     "read shards + key-owner table merge per build (RCCL all-to-all + all-gather)"
                                     if merge else "independent read shards, no table merge")},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": cands[dom],
                         "avg_launch_ms": stage_avg[dom], "launches_per_step": stage_ms[dom][1],
                         "largest_stages": per_kernel,
                         "whole_build": {"algorithmic_bytes_per_gene_mer": survey_b,
                                         "ms_per_build": build_ms / n_builds,
                                         "achieved": survey_b * n_windows / (build_ms / n_builds * 1e-3) / 1e9,
                                         "unit": "GB/s"}},
            "stages_ms_per_step": {n: round(v, 3) for n, v in stage_tot.items()},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()  # RCCL prints its version banner here: keep the JSON line last
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
