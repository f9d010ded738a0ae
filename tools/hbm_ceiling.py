"""Measure this box's practical HBM ceilings (pure write stream, copy) with plain torch kernels, to put
the roofline fractions of the write-bound kernels in context."""
import torch, time
dev = torch.device("cuda:0")
n = 1 << 30                      # 8 GiB of f64
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
tf = t(lambda: x.fill_(1.5))
tz = t(lambda: x.zero_())
tc = t(lambda: y.copy_(x))
print(f"fill_  8 GiB: {tf*1e3:.3f} ms  {n*8/tf/1e12:.2f} TB/s write")
print(f"zero_  8 GiB: {tz*1e3:.3f} ms  {n*8/tz/1e12:.2f} TB/s write")
print(f"copy_  8 GiB: {tc*1e3:.3f} ms  {2*n*8/tc/1e12:.2f} TB/s read+write")
