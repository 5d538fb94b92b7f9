"""Per-step HIP-event times of the first 40 replays behind W warm-up steps (argv[1], default 5): the clock ramp a short timed
window would see (bench.py --settle-steps)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import melo_gan_amd  # noqa
from melo_gan_amd import ops
from melo_gan_amd.gan.engine import GanEngine
from melo_gan_amd.gan.dp import DataParallel
from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg
B, T, C = 64, 256, 128
cfg, ed_cfg = default_gan_cfg(B, T, C), default_ed_cfg(C)
eng = GanEngine(cfg, ed_cfg, "cuda:0", B); eng.init_weights(seed=42)
dp = DataParallel(eng, 1, None)
g = torch.Generator().manual_seed(42)
pool = [((torch.rand(B, T, C, generator=g) * 2 - 1).cuda(), torch.randn(B, 6, generator=g).cuda(), torch.zeros(B, cfg["LATENT_DIM"]).cuda(), torch.randint(0, 4, (B,), generator=g).cuda()) for _ in range(4)]
eng.seed(1234); eng.bind_batches(*[torch.cat([b[j] for b in pool]) for j in range(4)])
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with torch.cuda.stream(eng.stream):
    for i in range(W):
        dp.step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [ops.Event() for _ in range(41)]
    evs[0].record()
    host = []
    for i in range(40):
        h0 = time.perf_counter(); dp.step(True); host.append((time.perf_counter() - h0) * 1e3); evs[i + 1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print("wall per step over 40: %.4f ms; first 20: see events" % (el / 40 * 1e3))
per = [evs[i].elapsed_ms(evs[i + 1]) for i in range(40)]
print("gpu  :", " ".join("%.3f" % p for p in per))
print("host :", " ".join("%.3f" % p for p in host))
print("first20 sum %.4f  last20 sum %.4f" % (sum(per[:20]) / 20, sum(per[20:]) / 20))
