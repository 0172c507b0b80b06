"""Calibration: what does a plain streaming copy of a conv-sized tensor achieve on this box (torch copy kernels, HIP events)?"""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for shape in ((64, 8, 256, 256), (64, 8, 128, 128), (64, 16, 64, 64), (32, 3, 256, 256)):
    x = torch.randn(*shape, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
    mb = x.numel() * 4 / 1e6
    us = t(lambda: y.copy_(x))
    print(f"copy {shape}: {us:.1f} us, {2 * mb / us * 1e-3:.2f} TB/s (read+write {2 * mb:.0f} MB)")
    us = t(lambda: torch.add(x, y, out=z))
    print(f"add  {shape}: {us:.1f} us, {3 * mb / us * 1e-3:.2f} TB/s")
    us = t(lambda: x.sum())
    print(f"sum  {shape}: {us:.1f} us, {mb / us * 1e-3:.2f} TB/s (read only)")
