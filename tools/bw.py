import torch, time
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"copy {mb} MB: {ms*1e3:.1f} us -> {2*mb/1024/ (ms/1e3) / 1e3:.2f} TB/s (read+write)")
    e0.record()
    for _ in range(20): a.zero_()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"fill {mb} MB: {ms*1e3:.1f} us -> {mb/1024/(ms/1e3)/1e3:.2f} TB/s (write)")
    e0.record()
    for _ in range(20): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"sum {mb} MB: {ms*1e3:.1f} us -> {mb/1024/(ms/1e3)/1e3:.2f} TB/s (read)")
