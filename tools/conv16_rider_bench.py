"""What the conv16 riders cost: the generator's three deconvolutions at the fused step's 2B = 128 rows, plain / with the BatchNorm
partial statistics / (deconv.6) with the gradient penalty's interpolate, each as a 20-launch hipGraph."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import melo_gan_amd  # noqa: F401
from melo_gan_amd import ops
from _timeit import timeit as timed

torch.manual_seed(0)
B = 128
for name, L, ci, co in (("deconv.0", 32, 256, 128), ("deconv.3", 64, 128, 64), ("deconv.6", 128, 64, 128)):
    x = torch.randn(B, L, ci).cuda()
    w = (torch.randn(ci, co, 5) * 0.05).cuda()          # ConvTranspose1d weight (Cin, Cout, 5)
    wq = torch.zeros(co * ci * 5).cuda()
    ops.wq_relayout(w, wq, co, ci, 5, 5, co * 5)
    y = torch.empty(B, 2 * L, co).cuda()
    bias = torch.randn(co).cuda()
    _, rows = ops.conv16_plan(B, L, co, True)
    part = torch.zeros(3 * rows * co).cuda()
    t0 = timed(lambda: ops.conv16(x, wq, y, co, True, bias=bias))
    t1 = timed(lambda: ops.conv16(x, wq, y, co, True, bias=bias, stats=part))
    line = f"{name} {ci:3d}->{co:3d} L={L:3d}: plain {t0:5.1f} us   + statistics {t1:5.1f} us"
    if name == "deconv.6":
        real, alpha, out = torch.randn(64, 2 * L, co).cuda(), torch.rand(64).cuda(), torch.empty(64, 2 * L, co).cuda()
        t2 = timed(lambda: ops.conv16(x, wq, y, co, True, bias=bias, mix=(real, alpha, out, 64)))
        line += f"   + interpolate {t2:5.1f} us"
    print(line, flush=True)
