import torch
for mb in (128, 378, 512, 1024):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"device copy of {mb} MiB: {us:.1f} us -> {2 * mb * 1.048576 / us * 1e-3 * 1e3:.0f} GB/s (read + write)")
    # write only (fill) and read only (sum)
    for _ in range(3): b.fill_(1.0)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): b.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"   fill of {mb} MiB: {us:.1f} us -> {mb * 1.048576 / us:.2f} TB/s (write)")
