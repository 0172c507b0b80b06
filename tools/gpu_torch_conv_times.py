import torch, time, torch.nn.functional as F
dev = torch.device("cuda:0")
def bench(name, fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)
B = 32
x = torch.randn(B, 6, 256, 256, device=dev)
for k, co in ((3, 4), (7, 2), (15, 2)):
    w = torch.randn(co, 6, k, k, device=dev, requires_grad=True)
    b = torch.zeros(co, device=dev, requires_grad=True)
    y = F.conv2d(x, w, b, padding=k // 2)
    gy = torch.randn_like(y)
    bench(f"crossembed k{k} fwd", lambda: F.conv2d(x, w, b, padding=k // 2))
    def bw():
        y = F.conv2d(x, w, b, padding=k // 2); y.backward(gy)
    bench(f"crossembed k{k} fwd+wgrad", bw)
for (ci, co, S) in ((8, 8, 256), (8, 16, 128)):
    xx = torch.randn(B, ci, S, S, device=dev, requires_grad=True)
    w = torch.randn(co, ci, 4, 4, device=dev, requires_grad=True)
    gy = torch.randn(B, co, S // 2, S // 2, device=dev)
    bench(f"down k4s2 {ci}->{co} @{S} fwd", lambda: F.conv2d(xx, w, None, stride=2, padding=1))
    def bw():
        y = F.conv2d(xx, w, None, stride=2, padding=1); y.backward(gy)
    bench(f"down k4s2 {ci}->{co} @{S} fwd+bwd", bw)
for (ci, co, S) in ((16, 8, 256), (32, 16, 128)):
    xx = torch.randn(B, ci, S, S, device=dev, requires_grad=True)
    w = torch.randn(co, ci, 1, 1, device=dev, requires_grad=True)
    gy = torch.randn(B, co, S, S, device=dev)
    def bw():
        y = F.conv2d(xx, w, None); y.backward(gy)
    bench(f"res 1x1 {ci}->{co} @{S} fwd+bwd", bw)
