"""What a plain device-to-device copy reaches on this GPU (bytes read + bytes written per second): the practical
roofline of the streaming kernels that read and write in equal parts (k_czm_bin_scatter, k_emit)."""
import torch, time
torch.cuda.init()
for gb in (0.5, 2.0, 4.0):
    n = int(gb * 1e9 / 4)
    a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    a.fill_(1.0)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("copy of %.1f GB: %.3f ms -> %.2f TB/s (read + write)" % (gb, dt * 1e3, 2 * n * 4 / dt / 1e12))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s = a.sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("   read-only reduction: %.3f ms -> %.2f TB/s" % (dt * 1e3, n * 4 / dt / 1e12))
