import torch, time
dev = torch.device("cuda:0")
n = 1 << 29   # 512 Mi doubles = 4 GiB
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
b = n * 8
ms = t(lambda: x.fill_(1.0)); print("fill   (write only)        %.3f ms  %.0f GB/s" % (ms, b / ms / 1e6))
ms = t(lambda: y.copy_(x)); print("copy   (1 read : 1 write)   %.3f ms  %.0f GB/s" % (ms, 2 * b / ms / 1e6))
ms = t(lambda: x.sum()); print("sum    (read only)         %.3f ms  %.0f GB/s" % (ms, b / ms / 1e6))
ms = t(lambda: torch.add(x, y, out=y)); print("add    (2 reads : 1 write)  %.3f ms  %.0f GB/s" % (ms, 3 * b / ms / 1e6))
