import torch, time
for mb in (1, 8, 25, 64, 256):
    src = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    n = max(4, 2048 // mb)
    t = time.perf_counter()
    for _ in range(n): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print(f"H2D {mb} MB: {mb * n / 1024 / el:.1f} GiB/s ({1e3 * el / n:.3f} ms per copy)")
