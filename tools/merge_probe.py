"""Where the merged (multi-GPU) sweep spends its time at world size 1 (RCCL self-exchange):
wall time per Engine / exchange call, accumulated over one sweep.  python tools/merge_probe.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.pop("NCCL_DEBUG", None)
import torch, torch.distributed as dist
import bench
from amira_amd import Engine
import amira_amd.dist as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = bench.WORKLOADS["cfg3-sweep"]; N, L, k = w["N"], w["L"], w["k"]
vocab, toks, offs = bench.make_tokens(w, 0, N)
dev = torch.device("cuda", 0)
d_toks = torch.from_numpy(toks).to(dev); d_offs = torch.from_numpy(offs).to(dev)
d_gs = (torch.arange(L, dtype=torch.int64, device=dev) * 1000).repeat(N); d_ge = d_gs + 899
d_rl = torch.full((N,), L * 1000 + 100, dtype=torch.int64, device=dev)
eng = Engine(0)
acc = collections.defaultdict(float)
def timed(name, fn):
    def wrap(*a, **kw):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(*a, **kw); torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t
        return r
    return wrap
for m in ("dist_nodes_local", "dist_edges_local", "dist_pack", "dist_reduce", "dist_global",
          "correct_reads", "adopt_corrected", "remove_short_linear_paths", "set_reads_device", "set_positions_device"):
    setattr(Engine, m, timed(m, getattr(Engine, m)))
D.exchange_a2a = timed("exchange_a2a", D.exchange_a2a); D.exchange_ag = timed("exchange_ag", D.exchange_ag)
def step():
    eng.set_reads_device(d_toks.data_ptr(), d_offs.data_ptr(), N, vocab.two_v)
    eng.set_positions_device(d_gs.data_ptr(), d_ge.data_ptr(), d_rl.data_ptr())
    D.dist_build(eng, k, None, 3, 1); eng.correct_reads(); eng.adopt_corrected()
    D.dist_build(eng, k, None); eng.remove_short_linear_paths(k); eng.correct_reads(); eng.adopt_corrected()
    D.dist_build(eng, k, None)
step(); acc.clear()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); tot = (time.perf_counter() - t) / 3
print("ms per sweep (with per-call syncs):", round(tot * 1e3, 2))
for n, v in sorted(acc.items(), key=lambda x: -x[1]): print(f"  {n:28s} {v / 3 * 1e3:7.3f} ms")
print("  unaccounted", round((tot - sum(acc.values()) / 3) * 1e3, 3))
eng.close(); dist.destroy_process_group()
