"""The emotion discriminator's three-tap layers: direct window GEMM (conv_wgemm_kernel) against minimal filtering
(wino3_kernel), each as a 20-launch hipGraph replayed 20 times.  usage: python tools/wino_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import melo_gan_amd  # noqa: F401
from melo_gan_amd import ops

B, T = 64, 256
torch.manual_seed(0)


from _timeit import timeit as timed  # noqa: E402

for name, ci, co in (("conv1", 64, 128), ("conv2", 128, 256), ("conv3", 256, 256)):
    x = torch.randn(B, T, ci).cuda()
    w = (torch.randn(co, ci, 3) / (3 * ci) ** 0.5).cuda()
    wcnk = w.permute(1, 0, 2).contiguous()          # the engine's forward layout (c, n, k)
    sc, sh = torch.rand(co).cuda() + 0.5, torch.randn(co).cuda()
    a, z = torch.empty(B, T, co).cuda(), torch.empty(B, T, co).cuda()
    wt_f = ops.wino3_weights(w, co, ci, 3 * ci, 3)
    wt_d = ops.wino3_weights(w, ci, co, 3, 3 * ci, flip=True)
    dy, zp, gs = torch.randn(B, T, co).cuda(), torch.randn(B, T, ci).cuda(), torch.rand(ci).cuda() + 0.5
    dx = torch.empty(B, T, ci).cuda()
    gf = 2.0 * B * T * ci * co * 3 / 1e9
    for pad in (0, 42000):
        with ops.conv_lds_pad(pad):
            t_d = timed(lambda: ops.conv_gather(x, wcnk, a, co, 3, 1, 3, co * 3, scale=sc, shift=sh, zout=z, act=ops.ACT_GELU))
            t_w = timed(lambda: ops.conv_wino3(x, wt_f, a, scale=sc, shift=sh, zout=z, act=ops.ACT_GELU))
            b_d = timed(lambda: ops.conv1d_dgrad(dy, w, dx, 1, gref=zp, gact=ops.ACT_GELU, gscale=gs))
            b_w = timed(lambda: ops.conv_wino3(dy, wt_d, dx, gref=zp, gact=ops.ACT_GELU, gscale=gs))
        print(f"{name} {ci:3d}->{co:3d} pad={pad:5d}  fwd direct {t_d:6.1f} us ({gf / t_d * 1e3:6.1f} TF)  wino {t_w:6.1f} us ({gf / t_w * 1e3:6.1f} TF alg.)"
              f" | dgrad direct {b_d:6.1f} us  wino {b_w:6.1f} us ({gf / b_w * 1e3:6.1f} TF alg.)", flush=True)
