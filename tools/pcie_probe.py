"""PCIe rates of the box the bench runs on: pinned H2D, D2H, both at once, and both beside a busy GPU (what the pcie_inclusive
leg of bench.py can reach at best)."""
import time
import torch

dev = torch.device("cuda:0")
nb = 400 << 20
h_in = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
h_out = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(nb, dtype=torch.uint8, device=dev)
d_b = torch.zeros(nb, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_a.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_out.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for _ in range(2):
    run(True, True, 1)
a, b, c = run(True, False), run(False, True), run(True, True)
print("400 MiB pinned: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  both at once %.2f ms (%.1f GB/s summed)" % (
    a * 1e3, nb / a / 1e9, b * 1e3, nb / b / 1e9, c * 1e3, 2 * nb / c / 1e9))
