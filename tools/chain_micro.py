"""Per-op cost of the row-chain kernel: chains of n identical ops, 64 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd  # noqa
from melo_gan_amd import ops
from _timeit import timeit

rows = 64
x = torch.randn(rows, 256, device="cuda"); y = torch.empty(rows, 256, device="cuda")
w = torch.randn(256, 256, device="cuda") * 0.05; b = torch.randn(256, device="cuda")
a3 = torch.randn(rows, 256, 256, device="cuda")

def chain(build):
    def run():
        ch = ops.Chain(rows); build(ch); ch.launch()
    return run

for n in (1, 2, 4, 8):
    print(f"{n} x LOAD               : {timeit(chain(lambda ch: [ch.load(i % 6, x) for i in range(n)])):6.1f} us", flush=True)
for n in (1, 2, 4, 8):
    print(f"LOAD + {n} x LIN_FWD 256 : {timeit(chain(lambda ch: [ch.load(0, x)] + [ch.linear_fwd(i % 6, (i + 1) % 6, w, b) for i in range(n)])):6.1f} us", flush=True)
for n in (1, 2, 4, 8):
    print(f"LOAD + {n} x LIN_DGRAD   : {timeit(chain(lambda ch: [ch.load(0, x)] + [ch.linear_dgrad(i % 6, (i + 1) % 6, w) for i in range(n)])):6.1f} us", flush=True)
print(f"MEAN_T 256x256          : {timeit(chain(lambda ch: ch.mean_t(0, a3))):6.1f} us", flush=True)
print(f"LOAD + STORE            : {timeit(chain(lambda ch: [ch.load(0, x), ch.store(0, y)])):6.1f} us", flush=True)
y2 = torch.empty(rows, 256, device="cuda")
print(f"linear_fwd launch 64x256x256: {timeit(lambda: ops.linear_fwd(x, w, y2, bias=b)):6.1f} us", flush=True)
