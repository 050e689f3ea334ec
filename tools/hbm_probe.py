"""Development tool: achievable HBM read / write bandwidth on this box with plain torch ops (4 GiB buffers)."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = timeit(lambda: a.sum()); print(f"read  (sum)      : {4 * n / t / 1e6:8.0f} GB/s")
t = timeit(lambda: b.fill_(1.0)); print(f"write (fill)     : {4 * n / t / 1e6:8.0f} GB/s")
t = timeit(lambda: b.copy_(a)); print(f"copy  (r+w)      : {8 * n / t / 1e6:8.0f} GB/s")
t = timeit(lambda: torch.add(a, 1.0, out=b)); print(f"add   (r+w)      : {8 * n / t / 1e6:8.0f} GB/s")
