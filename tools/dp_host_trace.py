#!/usr/bin/env python3
"""Where does the HOST spend its time in a data-parallel step?  1-rank RCCL group on one GPU (the N > 1 control path),
every engine sub-step and collective of DataParallel.step timed on the host (no device sync inside the loop).
    MELO_DP_MODE=gather|allreduce|overlap python tools/dp_host_trace.py"""
import collections
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import melo_gan_amd  # noqa: E402,F401
from melo_gan_amd.gan.config import default_ed_cfg, default_gan_cfg  # noqa: E402
from melo_gan_amd.gan.dp import DataParallel  # noqa: E402
from melo_gan_amd.gan.engine import GanEngine  # noqa: E402

B, T, C = 64, 256, 128
cfg = default_gan_cfg(B, T, C)
eng = GanEngine(cfg, default_ed_cfg(C), "cuda:0", B)
eng.init_weights(seed=42)
dp = DataParallel(eng, 1, dist, force_collectives=True)
batch = ((torch.rand(B, T, C) * 2 - 1).cuda(), torch.randn(B, 6).cuda(), torch.zeros(B, cfg["LATENT_DIM"]).cuda(),
         torch.randint(0, 4, (B,)).cuda())
eng.seed(1)
acc = collections.OrderedDict()


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w


run0 = eng.run
eng.run = lambda name, g=True: timed("run:" + name, run0)(name, g)
for nm in ("allreduce_d", "allreduce_g", "allreduce_g_rest", "gather_p2", "_wait"):
    setattr(dp, nm, timed(nm, getattr(dp, nm)))
N = 200
with torch.cuda.stream(eng.stream):
    for _ in range(10):
        eng.set_batch(*batch)
        dp.step(True)
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(N):
        eng.set_batch(*batch)
        dp.step(True)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
print(f"mode {dp.mode}: host loop {1e6 * host / N:.0f} us/step, with final sync {1e6 * total / N:.0f} us/step")
for k, v in acc.items():
    print(f"  {k:28s} {1e6 * v / N:8.1f} us/step")
dist.destroy_process_group()
