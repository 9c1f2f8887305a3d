"""Cross-check of the library's HBM stream probe (rgbdslam_profile_hbm_stream) against torch's own elementwise kernels."""
import torch, time
n = (2 << 30) // 4
a = torch.zeros(n, device="cuda"); b = torch.ones(n, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
GB = n * 4 / 1e9
print("torch copy_", 2 * GB / t(lambda: a.copy_(b)))
print("torch fill_", GB / t(lambda: a.fill_(1.5)))
print("torch zero_", GB / t(lambda: a.zero_()))
print("torch sum  ", GB / t(lambda: b.sum()))
print("torch add  (2r+1w)", 3 * GB / t(lambda: torch.add(a, b, out=a)))
