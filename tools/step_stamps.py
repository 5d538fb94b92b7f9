"""Where a production step's time goes WITHOUT a profiler attached: MELO_STAMPS=1 puts one-lane nodes into the step's graph(s)
that write the device's 100-MHz clock (mg_stamp); this runs the bench's step loop and prints the stamps of the steps' medians.
usage: [MELO_ED_FLOW=split|ingraph|fork2] python tools/step_stamps.py [steps]"""
import os, sys, time
os.environ["MELO_STAMPS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import melo_gan_amd  # noqa: F401
from melo_gan_amd.gan.engine import GanEngine
from melo_gan_amd.gan.dp import DataParallel
from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, T, C = 64, 256, 128
cfg, ed_cfg = default_gan_cfg(B, T, C), default_ed_cfg(C)
eng = GanEngine(cfg, ed_cfg, "cuda:0", B)
eng.init_weights(seed=42)
dp = DataParallel(eng, 1, None)
g = torch.Generator().manual_seed(42)
pool = []
for _ in range(4):
    pool.append(((torch.rand(B, T, C, generator=g) * 2 - 1).cuda(), torch.randn(B, 6, generator=g).cuda(),
                 torch.zeros(B, cfg["LATENT_DIM"]).cuda(), torch.randint(0, 4, (B,), generator=g).cuda()))
eng.seed(1234)
eng.bind_batches(*[torch.cat([b[j] for b in pool]) for j in range(4)])
torch.cuda.set_stream(eng.stream)


def step():
    dp.step(True)


for _ in range(10):
    step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(K):
    step()
    torch.cuda.synchronize()          # per-step sync: stamps of THIS step (costs the host overlap; see the free-running line)
    rec.append(eng.stamps.cpu().clone())
dt_sync = (time.perf_counter() - t0) / K * 1e3
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
dt_free = (time.perf_counter() - t0) / K * 1e3
free_last = eng.stamps.cpu().clone()
r = torch.stack(rec).double()
names = ["start", "fork", "ed_first", "ed_last", "main_at_join", "after_join", "end"]
rel = (r - r[:, :1]) / 100.0          # us
med = rel.median(0).values
print(f"flow={os.environ.get('MELO_ED_FLOW', 'default')} pad={os.environ.get('MELO_ED_PAD', '')}  per-step-sync {dt_sync:.4f} ms  free-running {dt_free:.4f} ms/step")
for i, n in enumerate(names):
    print(f"  {n:13s} {med[i]:8.1f} us   (free-running last step: {(free_last[i] - free_last[0]).item() / 100.0:8.1f})")
print(f"  branch: starts {med[2] - med[1]:.1f} us after the fork, runs {med[3] - med[2]:.1f} us; main reaches the join {med[4] - med[1]:.1f} us after the fork "
      f"and waits {med[5] - med[4]:.1f} us; tail {med[6] - med[5]:.1f} us")
