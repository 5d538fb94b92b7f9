"""Per-layer timing of the cfg2 launches via hipGraph replay (no Python launch overhead in the numbers).  Every line takes
the PRODUCT route of its layer: stride-2 forward / data-gradient launches run on conv16 (WQ weights) exactly where the
engine's _conv5s2 puts them, the rest on the window GEMM / thin kernels behind ops.conv1d_* (tools/conv16_bench.py has the
old-vs-new comparison)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd
from melo_gan_amd import _lib
if os.environ.get("MELO_LIB_VARIANT"):      # a tools/build_variant.sh build
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libmelogan_" + os.environ["MELO_LIB_VARIANT"] + ".so")
from melo_gan_amd import ops

def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        g = ops.Graph(); g.begin()
        for _ in range(reps): fn()
        g.end()
        g.launch(); torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(); g.launch(); g.launch(); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_ms(e1) / (2 * reps) * 1e3

def R(*s): return torch.randn(*s, device='cuda')
B = 64
rows = []
def wq_of(w, N, Cc, sn, sc):
    wq = torch.empty(N * Cc * 5, device="cuda"); ops.wq_relayout(w, wq, N, Cc, 5, sn, sc); return wq
def conv(tag, nb, T, Cin, Cout, K, stride):
    x = R(nb, T, Cin); w = R(Cout, Cin, K) * 0.05
    Tout = (T + 2 * (K // 2) - K) // stride + 1
    y = R(nb, Tout, Cout); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(Cout, device='cuda')
    fl = 2.0 * nb * Tout * Cout * Cin * K
    fwd, dgr = (lambda: ops.conv1d_fwd(x, w, y, stride)), (lambda: ops.conv1d_dgrad(y, w, dx, stride))
    if stride == 2 and ops.conv16_supported(nb, T, Cin, Cout, False):
        wf = wq_of(w, Cout, Cin, Cin * 5, 5); fwd = lambda: ops.conv16(x, wf, y, Cout, False)
    if stride == 2 and ops.conv16_supported(nb, Tout, Cout, Cin, True, T):
        wd = wq_of(w, Cin, Cout, 5, Cin * 5); dgr = lambda: ops.conv16(y, wd, dx, Cin, True, odd=(T % 2 == 1))
    for name, fn in (("fwd", fwd), ("dgrad", dgr),
                     ("wgrad", lambda: ops.conv1d_wgrad(x, y, dw, stride, db=db))):
        us = timeit(fn); print(f"{tag:26s} {name:6s} nb={nb:3d} T={T:3d} {Cin:3d}->{Cout:3d} K={K} s={stride}: {us:7.1f} us {fl/us/1e6:6.1f} TF", flush=True)
def convT(tag, nb, T, Cin, Cout):
    x = R(nb, T, Cin); w = R(Cin, Cout, 5) * 0.05; y = R(nb, 2 * T, Cout); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(Cout, device='cuda')
    fl = 2.0 * nb * T * Cout * Cin * 5
    fwd, dgr = (lambda: ops.convT1d_fwd(x, w, y)), (lambda: ops.convT1d_dgrad(y, w, dx))
    if ops.conv16_supported(nb, T, Cin, Cout, True, 2 * T):
        wf = wq_of(w, Cout, Cin, 5, Cout * 5); fwd = lambda: ops.conv16(x, wf, y, Cout, True)
    if ops.conv16_supported(nb, 2 * T, Cout, Cin, False):
        wd = wq_of(w, Cin, Cout, Cout * 5, 5); dgr = lambda: ops.conv16(y, wd, dx, Cin, False)
    for name, fn in (("fwd", fwd), ("dgrad", dgr),
                     ("wgrad", lambda: ops.convT1d_wgrad(x, y, dw, db=db))):
        us = timeit(fn); print(f"{tag:26s} {name:6s} nb={nb:3d} T={T:3d} {Cin:3d}->{Cout:3d} K=5 T2: {us:7.1f} us {fl/us/1e6:6.1f} TF", flush=True)
def lin(tag, nb, Cin, Cout):
    x = R(nb, Cin); w = R(Cout, Cin) * 0.05; y = R(nb, Cout); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(Cout, device='cuda')
    fl = 2.0 * nb * Cout * Cin
    for name, fn in (("fwd", lambda: ops.linear_fwd(x, w, y)), ("dgrad", lambda: ops.linear_dgrad(y, w, dx)),
                     ("wgrad", lambda: ops.linear_wgrad(x, y, dw, db=db))):
        us = timeit(fn); print(f"{tag:26s} {name:6s} nb={nb:3d} {Cin:4d}->{Cout:4d}: {us:7.1f} us {fl/us/1e6:6.2f} TF", flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "conv"):
    conv("critic conv.0 (3B)", 3 * B, 256, 128, 64, 5, 2)
    conv("critic conv.2 (3B)", 3 * B, 128, 64, 128, 5, 2)
    conv("critic conv.4 (3B)", 3 * B, 64, 128, 256, 5, 2)
    conv("critic conv.0 (B)", B, 256, 128, 64, 5, 2)
    conv("critic conv.4 (B)", B, 64, 128, 256, 5, 2)
    convT("gen deconv.0", B, 32, 256, 128)
    convT("gen deconv.3", B, 64, 128, 64)
    convT("gen deconv.6", B, 128, 64, 128)
    conv("ED conv0 k5", B, 256, 128, 64, 5, 1)
    conv("ED conv1 k3", B, 256, 64, 128, 3, 1)
    conv("ED conv2 k3", B, 256, 128, 256, 3, 1)
    conv("ED conv3 k3", B, 256, 256, 256, 3, 1)
if which in ("all", "lin"):
    lin("gen pre.0", B, 512, 512)
    lin("gen pre.2", B, 512, 8192)
    lin("critic fc", 3 * B, 256, 256)
    lin("E_num", B, 256, 128)
