"""The RCCL plumbing of bench.py with a process group of one rank (all a one-GPU box can run): init with device_id, the
barrier that names the device, MAX / SUM all-reduce, all-gather.  python tools/rccl_single_rank.py"""
import os, sys, torch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
sys.path.insert(0, "patchwork-plusplus_amd/python")
import torch.distributed as dist
import pwpp_dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)   # what pwpp_dist.init does for WORLD_SIZE > 1
pwpp_dist.barrier(dev)
print("aggregate", pwpp_dist.aggregate(1.5, 7, dev))
print("gather", pwpp_dist.gather_values(3.25, dev))
pwpp_dist.barrier(dev)
pwpp_dist.finalize()
print("rccl single-rank ok")
