"""How fast does this box stream activation-sized tensors?  (context for the roofline numbers)"""
import torch, time
dev = "cuda:0"
B, C, T = 16, 64, 16000
a = torch.randn(B, C, T, device=dev); b = torch.randn(B, C, T, device=dev); c = torch.empty_like(a)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
mb = a.numel() * 4 / 1e6
t = timeit(lambda: torch.mul(a, b, out=c)); print(f"mul (2 reads + 1 write of {mb:.0f} MB): {t*1e6:.1f} us = {3*mb/t/1e6:.2f} TB/s")
t = timeit(lambda: c.copy_(a)); print(f"copy: {t*1e6:.1f} us = {2*mb/t/1e6:.2f} TB/s")
big = torch.randn(30, B, C, T, device=dev)
t = timeit(lambda: big.sum(), 5); print(f"sum over {30*mb:.0f} MB: {t*1e6:.1f} us = {30*mb/t/1e6:.2f} TB/s")
# row-piece pattern: gather (b, c, 256-column pieces) in a permuted order
x = a.view(B, C, T // 64, 64)
perm = torch.randperm(T // 64, device=dev)
t = timeit(lambda: x.index_select(2, perm)); print(f"index_select of 256-B pieces: {t*1e6:.1f} us = {2*mb/t/1e6:.2f} TB/s")
