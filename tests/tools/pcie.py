"""PCIe link of the box: pinned host <-> device copy rates, one direction at a time and both at once (two streams).
Context for the host-buffer column of bench.py (value_host): at C2 a pair moves 6.0 KB up and 4.4 KB down."""
import time, torch
n = 256 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def up():
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
def down():
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
def both():
    up(); down()
print(f"H2D {n / t(up) / 1e9:.1f} GB/s, D2H {n / t(down) / 1e9:.1f} GB/s, both at once {2 * n / t(both) / 1e9:.1f} GB/s in sum")
