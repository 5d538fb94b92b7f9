"""In-kernel phase stamps of the weight-gradient kernel (tools/build_variant.sh wstamp wgrad_mfma -DMG_STAMPS)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, melo_gan_amd
from melo_gan_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libmelogan_" + os.environ.get("MELO_LIB_VARIANT", "wstamp") + ".so")
from melo_gan_amd import ops
lib = _lib.load()
lib.mg_dbg_set_wstamps.argtypes = [C.c_void_p]; lib.mg_dbg_set_wstamps.restype = C.c_int

def run(tag, nb, T, Cin, Cout, K, stride, convT=False):
    if convT:
        x = torch.randn(nb, T, Cin, device='cuda'); y = torch.randn(nb, 2 * T, Cout, device='cuda'); dw = torch.empty(Cin, Cout, 5, device='cuda')
        f = lambda: ops.convT1d_wgrad(x, y, dw)
    else:
        x = torch.randn(nb, T, Cin, device='cuda'); Tout = (T + 2 * (K // 2) - K) // stride + 1
        y = torch.randn(nb, Tout, Cout, device='cuda'); dw = torch.empty(Cout, Cin, K, device='cuda')
        f = lambda: ops.conv1d_wgrad(x, y, dw, stride)
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device='cuda')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): f()
        torch.cuda.synchronize()
        assert lib.mg_dbg_set_wstamps(stamps.data_ptr()) == 0
        f(); torch.cuda.synchronize()
        lib.mg_dbg_set_wstamps(None)
    st = stamps.cpu().numpy().reshape(-1, 8); st = st[st[:, 0] > 0]
    t0 = st[:, 0].min(); rel = (st - t0) * 0.01
    print(f"\n{tag}: {len(st)} WGs")
    for k, n in enumerate(["entry", "loop start", "loop end", "end"]):
        print(f"  {n:11s} min {rel[:, k].min():7.2f} med {np.median(rel[:, k]):7.2f} max {rel[:, k].max():7.2f}")
    print(f"  per WG: prologue {np.median(rel[:,1]-rel[:,0]):.2f}  loop {np.median(rel[:,2]-rel[:,1]):.2f}  epilogue {np.median(rel[:,3]-rel[:,2]):.2f} us")

run("critic conv.4 (3B) wgrad", 192, 64, 128, 256, 5, 2)
run("critic conv.0 (3B) wgrad", 192, 256, 128, 64, 5, 2)
run("gen deconv.0 wgrad", 64, 32, 256, 128, 5, 2, convT=True)
