import torch, time
dev = torch.device("cuda", 0)
n = 369_000_000 // 4
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, fn, bytes_ in (("fill (write only)", lambda: x.fill_(1.0), n * 4), ("zero_ (memset)", lambda: x.zero_(), n * 4),
                         ("copy (read + write)", lambda: y.copy_(x), 2 * n * 4), ("sum (read only)", lambda: x.sum(), n * 4),
                         ("add_ (read + write same)", lambda: x.add_(1.0), 2 * n * 4)):
    ms = t(fn)
    print("%-28s %.4f ms  %.2f TB/s" % (name, ms, bytes_ / ms / 1e9))
