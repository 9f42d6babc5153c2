import torch, time
dev = torch.device("cuda", 0)
for mb in (632, 2048):
    n = mb * (1 << 20) // 8
    a = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device=dev); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print("D2D copy %d MB: best %.3f ms = %.0f GB/s read+write (median %.3f ms)" % (mb, t, 2 * n * 8 / t / 1e6, sorted(ts)[5]))
    # also a += 1 style read-modify-write and fill
    ts = []
    for _ in range(10):
        e0.record(); b.fill_(3); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("   fill: %.0f GB/s" % (n * 8 / min(ts) / 1e6))
    ts = []
    for _ in range(10):
        e0.record(); s = a.sum(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("   read (sum): %.0f GB/s" % (n * 8 / min(ts) / 1e6))
