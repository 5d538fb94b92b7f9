"""Host-resident (streamed) data plane at the cfg2 batch shape: batches/s the host gather + async H2D sustains beside
the training step, and the step time with it (PCIe-inclusive rate of DESIGN section 7)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd
from melo_gan_amd.gan.engine import GanEngine
from melo_gan_amd.gan.dp import DataParallel
from melo_gan_amd.gan.dataset import GANDataset
from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg
B, T, C = 64, 256, 128
cfg, ed_cfg = default_gan_cfg(B, T, C), default_ed_cfg(C)
eng = GanEngine(cfg, ed_cfg, "cuda", B); eng.init_weights(42)
dp = DataParallel(eng, 1, None)
for resident in (True, False):
    ds = GANDataset.synthetic(64 * 40, T, C, cfg["LATENT_DIM"], resident=resident)
    g = torch.Generator().manual_seed(0)
    with torch.cuda.stream(eng.stream):
        for ep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
            for notes, numeric, latent, emot in ds.batches(B, g):
                eng.set_batch(notes, numeric, latent, emot); dp.step(True); n += 1
            torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"resident={resident}: {1e3 * el / n:.3f} ms/step, {B * n / el:.0f} samples/s (threads {torch.get_num_threads()})")
