"""Multi-GPU plumbing for the frame-parallel path (one process per GPU, torch.distributed).

Frames are independent: every rank runs the whole pipeline on its own
shard, no tensor ever crosses xGMI.  The only collective is the throughput bookkeeping:
MAX of the elapsed time and SUM of the processed frames (16 bytes, latency-bound), over
RCCL on GPUs (backend "nccl") or gloo in the CPU tests.
"""
import os


def shard_sources(num_sources, frames_per_rank, rank):
    """Which source frame each of this rank's buffers replays (round-robin, rank-rotated)."""
    return [(i + rank) % num_sources for i in range(frames_per_rank)]


def init(backend, device=None):
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def aggregate(elapsed_s, frames_done, device=None):
    """Whole-job view: (max elapsed over ranks, total frames over ranks)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(frames_done)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    n = torch.tensor([int(frames_done)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
