"""conv16 (16x16 MFMA tiles, no split-K) against the 64x64-tile window GEMM (+ its split-K finish launch) on the cfg2
stride-2 layers, per launch via hipGraph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd  # noqa
from melo_gan_amd import ops
from layer_bench import timeit, R  # noqa

def wq_of(w, N, Cc, sn, sc):
    wq = torch.empty(N * Cc * 5, device="cuda"); ops.wq_relayout(w, wq, N, Cc, 5, sn, sc); return wq

def conv(tag, nb, T, Cin, Cout):
    x = R(nb, T, Cin); w = R(Cout, Cin, 5) * 0.05; Tout = (T - 1) // 2 + 1
    y = R(nb, Tout, Cout); dx = torch.empty_like(x); b = R(Cout)
    wf, wd = wq_of(w, Cout, Cin, Cin * 5, 5), wq_of(w, Cin, Cout, 5, Cin * 5)
    fl = 2.0 * nb * Tout * Cout * Cin * 5
    for name, old, new in (("fwd", lambda: ops.conv1d_fwd(x, w, y, 2, bias=b, act=ops.ACT_LRELU), lambda: ops.conv16(x, wf, y, Cout, False, bias=b, act=ops.ACT_LRELU)),
                           ("dgrad", lambda: ops.conv1d_dgrad(y, w, dx, 2), lambda: ops.conv16(y, wd, dx, Cin, True))):
        a, c = timeit(old), timeit(new)
        print(f"{tag:22s} {name:6s} nb={nb:3d} T={T:3d} {Cin:3d}->{Cout:3d}: old {a:6.1f} us {fl/a/1e6:6.1f} TF | conv16 {c:6.1f} us {fl/c/1e6:6.1f} TF", flush=True)

def convT(tag, nb, L, Cin, Cout):
    x = R(nb, L, Cin); w = R(Cin, Cout, 5) * 0.05; y = R(nb, 2 * L, Cout); dx = torch.empty_like(x); b = R(Cout)
    wf, wd = wq_of(w, Cout, Cin, 5, Cout * 5), wq_of(w, Cin, Cout, Cout * 5, 5)
    fl = 2.0 * nb * L * Cout * Cin * 5
    for name, old, new in (("fwd", lambda: ops.convT1d_fwd(x, w, y, bias=b), lambda: ops.conv16(x, wf, y, Cout, True, bias=b)),
                           ("dgrad", lambda: ops.convT1d_dgrad(y, w, dx), lambda: ops.conv16(y, wd, dx, Cin, False))):
        a, c = timeit(old), timeit(new)
        print(f"{tag:22s} {name:6s} nb={nb:3d} L={L:3d} {Cin:3d}->{Cout:3d}: old {a:6.1f} us {fl/a/1e6:6.1f} TF | conv16 {c:6.1f} us {fl/c/1e6:6.1f} TF", flush=True)

B = 64
for nb in (3 * B, B):
    conv(f"critic conv.0 ({nb})", nb, 256, 128, 64)
    conv(f"critic conv.2 ({nb})", nb, 128, 64, 128)
    conv(f"critic conv.4 ({nb})", nb, 64, 128, 256)
for nb in (2 * B, B):
    convT(f"gen deconv.0 ({nb})", nb, 32, 256, 128)
    convT(f"gen deconv.3 ({nb})", nb, 64, 128, 64)
    convT(f"gen deconv.6 ({nb})", nb, 128, 64, 128)
