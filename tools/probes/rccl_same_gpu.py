"""probe: will RCCL take two ranks on ONE GPU (amg_dist_init from two processes, both on device 0)?"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, path):
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=2)
    from amira_amd import Engine
    box = [Engine.dist_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    e = Engine(0)
    try:
        e.dist_init(box[0], rank, 2)
        print(rank, "communicator made", flush=True)
    except Exception as err:  # noqa: BLE001
        print(rank, "refused:", err, flush=True)
    dist.barrier()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    d = tempfile.mkdtemp()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, os.path.join(d, "rv"))) for r in range(2)]
    [p.start() for p in ps]
    [p.join(120) for p in ps]
    for p in ps:
        if p.is_alive():
            p.terminate()
