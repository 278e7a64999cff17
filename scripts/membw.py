import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (26, 105, 420, 1680):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x))
    ms2 = t(lambda: y.fill_(1.0))
    ms3 = t(lambda: torch.mul(x, 2.0, out=y))
    print("%5d MB: copy %.3f ms %.2f TB/s | fill %.3f ms %.2f TB/s | mul %.3f ms %.2f TB/s" % (mb, ms, 2*mb*1.048576e-3/ms, ms2, mb*1.048576e-3/ms2, ms3, 2*mb*1.048576e-3/ms3))
