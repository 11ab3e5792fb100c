import torch, time
dev = torch.device("cuda:0")
for mb in (1.5, 6, 24, 96):
    n = int(mb * 1e6)
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True); h.fill_(3)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("H2D pinned %5.1f MB: %.3f ms  %.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
# 16 slices of 1.5 MB over 4 streams
n = int(1.5e6); hs = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(16)]; d = torch.empty(16 * n, dtype=torch.uint8, device=dev)
ss = [torch.cuda.Stream() for _ in range(4)]
def go():
    for i, h in enumerate(hs):
        with torch.cuda.stream(ss[i % 4]): d[i * n:(i + 1) * n].copy_(h, non_blocking=True)
for _ in range(3): go()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): go()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("16 x 1.5 MB over 4 streams: %.3f ms  %.1f GB/s" % (dt * 1e3, 16 * n / dt / 1e9))
# one lone copy after idle
import time as _t
for _ in range(3):
    _t.sleep(0.05); torch.cuda.synchronize(); t0 = time.perf_counter(); go(); torch.cuda.synchronize(); print("lone 24 MB in slices after 50 ms idle: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
