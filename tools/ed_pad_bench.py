"""The stride-1 window GEMMs of the emotion discriminator (cfg2 shapes) at 3 / 2 / 1 resident workgroups per CU (extra LDS per
workgroup: ops.conv_lds_pad), forward and data-gradient, alone on the chip (hipGraph replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd  # noqa
from melo_gan_amd import ops
from _timeit import timeit

B, T = 64, 256
for (ci, co, k) in ((128, 64, 5), (64, 128, 3), (128, 256, 3), (256, 256, 3)):
    x = torch.randn(B, T, ci, device="cuda"); w = torch.randn(co, ci, k, device="cuda") * 0.05
    wt = w.permute(1, 0, 2).contiguous()
    y = torch.empty(B, T, co, device="cuda"); z = torch.empty_like(y); dx = torch.empty_like(x)
    sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
    fl = 2.0 * B * T * ci * co * k
    line = f"{ci:3d}->{co:3d} k{k}: "
    for pad in (0, 14000, 42000):
        def fwd():
            with ops.conv_lds_pad(pad):
                ops.conv_gather(x, wt, y, co, k, 1, k, co * k, scale=sc, shift=sh, zout=z, act=ops.ACT_GELU)
        def dgr():
            with ops.conv_lds_pad(pad):
                ops.conv1d_dgrad(y, w, dx, 1, gref=x, gact=ops.ACT_GELU, gscale=torch.ones(ci, device="cuda"))
        a, b = timeit(fwd), timeit(dgr)
        line += f" pad {pad:5d}: fwd {a:6.1f} us {fl / a / 1e6:6.1f} TF  dgrad {b:6.1f} us {fl / b / 1e6:6.1f} TF |"
    print(line, flush=True)
