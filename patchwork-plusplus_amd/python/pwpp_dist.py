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


def barrier(device=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if device is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])  # (RCCL: name the rank's own GPU, or the barrier guesses one from the rank)
        else:
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


def gather_values(value, device=None):
    """One float of every rank, in rank order (the per-GPU frames/s of BASELINE.json configs[3])."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


def describe(backend, device=None):
    """What the N > 1 record shows of the process group: backend, world size, RCCL version, and every rank's GPU
    (index and PCI bus id, all-gathered in rank order) -- evidence that RCCL saw N ranks on N different devices."""
    import torch
    import torch.distributed as dist
    info = {"backend": backend, "world_size": 1, "rccl_version": None, "devices": []}
    try:
        v = torch.cuda.nccl.version()  # ("nccl" IS RCCL on ROCm)
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass
    idx, bus = -1, -1
    if device is not None and device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(idx)
        bus = int(getattr(pr, "pci_bus_id", -1))
        info["device_name"] = pr.name
        info["hbm_gb"] = pr.total_memory / 1e9
    dev = device if backend == "nccl" else None
    if dist.is_available() and dist.is_initialized():
        info["world_size"] = dist.get_world_size()
        info["backend"] = dist.get_backend()
    ids = gather_values(idx, dev)
    buses = gather_values(bus, dev)
    info["devices"] = [{"rank": r, "index": int(i), "pci_bus_id": int(b)} for r, (i, b) in enumerate(zip(ids, buses))]
    return info


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
