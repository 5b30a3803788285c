import torch, time
torch.cuda.init()
for mb, n in ((2, 64), (8, 16), (32, 4), (128, 1)):
    src=[torch.empty(mb*1024*1024//4, dtype=torch.float32).pin_memory() for _ in range(n)]
    dst=[torch.empty_like(s, device='cuda') for s in src]
    for _ in range(2):
        for s,d in zip(src,dst): d.copy_(s, non_blocking=True)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(10):
        for s,d in zip(src,dst): d.copy_(s, non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("H2D %3d x %3d MB pinned: %.1f GB/s"%(n,mb,10*n*mb/1024/dt))
    t0=time.perf_counter()
    for _ in range(10):
        for s,d in zip(src,dst): s.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("D2H %3d x %3d MB pinned: %.1f GB/s"%(n,mb,10*n*mb/1024/dt))
