"""dev tool: on-box HBM bandwidth reference figures (device-to-device copy and a read-only reduction) next to the
8 TB/s vendor peak that bench.py's roofline uses."""
import time
import torch
assert torch.cuda.is_available()
n = 1 << 30  # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
t = timeit(lambda: b.copy_(a))
print(f"device-to-device copy of 4 GiB: {t*1e3:.2f} ms -> {2*4*n/t/1e12:.2f} TB/s (read + write)")
t = timeit(lambda: a.sum())
print(f"read-only reduction over 4 GiB: {t*1e3:.2f} ms -> {4*n/t/1e12:.2f} TB/s")
t = timeit(lambda: torch.add(a, b, out=b))
print(f"triad-like a+b->b over 2x4 GiB: {t*1e3:.2f} ms -> {3*4*n/t/1e12:.2f} TB/s")
