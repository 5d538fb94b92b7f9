"""hipGraph-replay timing helper of the tools/ benches (no Python launch overhead in the numbers)."""
import torch
from melo_gan_amd import ops


def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = ops.Graph(); g.begin()
        for _ in range(reps):
            fn()
        g.end()
        g.launch(); torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(); g.launch(); g.launch(); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_ms(e1) / (2 * reps) * 1e3
