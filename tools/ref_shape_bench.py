"""Step time at the reference's own default shape (B=32, T=512, C=4; SURVEY 8 'ref') -- not the headline metric."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd
from melo_gan_amd.gan.engine import GanEngine
from melo_gan_amd.gan.dp import DataParallel
from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg
B, T, C = 32, 512, 4
cfg, ed_cfg = default_gan_cfg(B, T, C), default_ed_cfg(C)
eng = GanEngine(cfg, ed_cfg, "cuda", B)
eng.init_weights(42)
dp = DataParallel(eng, 1, None)
g = torch.Generator().manual_seed(0)
batch = ((torch.rand(B, T, C, generator=g) * 2 - 1).cuda(), torch.randn(B, 6, generator=g).cuda(), torch.zeros(B, cfg["LATENT_DIM"]).cuda(), torch.randint(0, 4, (B,), generator=g).cuda())
with torch.cuda.stream(eng.stream):
    for _ in range(5):
        eng.set_batch(*batch); dp.step(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        eng.set_batch(*batch); dp.step(True)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"ref shape B={B} T={T} C={C}: {el*10:.3f} ms/step, {B*100/el:.0f} samples/s")
