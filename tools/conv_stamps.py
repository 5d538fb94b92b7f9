"""In-kernel phase stamps of the window-GEMM kernel (debug build: tools/build_stamp.sh [TAG] [-DMG_EXP_...]).
Prints, per layer shape, when workgroups enter, have their first chunk staged, leave the chunk loop and finish,
the shader clock over a workgroup's life and how workgroups spread over the CUs.  Run on the GPU box:
    bash tools/build_stamp.sh && python tools/conv_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, melo_gan_amd
from melo_gan_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libmelogan_stamp" + os.environ.get("STAMP_TAG", "") + ".so")
from melo_gan_amd import ops
lib = _lib.load()
lib.mg_dbg_set_stamps.argtypes = [C.c_void_p]; lib.mg_dbg_set_stamps.restype = C.c_int

def run(B, T, Cin, Cout, K, stride, mode="fwd"):
    x = torch.randn(B, T, Cin, device='cuda')
    if mode == "fwd":
        w = torch.randn(Cout, Cin, K, device='cuda') * 0.05
        Tout = (T + 2 * (K // 2) - K) // stride + 1
        y = torch.empty(B, Tout, Cout, device='cuda')
        f = lambda: ops.conv1d_fwd(x, w, y, stride)
    else:
        w = torch.randn(Cin, Cout, K, device='cuda') * 0.05
        y = torch.empty(B, 2 * T, Cout, device='cuda')
        f = lambda: ops.convT1d_fwd(x, w, y)
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device='cuda')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(); 
        for _ in range(20): f()
        e1.record(); us = e0.elapsed_ms(e1) / 20 * 1e3
        torch.cuda.synchronize()
        assert lib.mg_dbg_set_stamps(stamps.data_ptr()) == 0
        f(); torch.cuda.synchronize()
        lib.mg_dbg_set_stamps(None)
    st = stamps.cpu().numpy().reshape(-1, 16)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    rel = (st - t0) * 0.01   # us
    rel[st == 0] = np.nan
    rel[:, 15] = np.nan; rel[:, 14] = np.nan
    names = ["entry", "ld_issued", "first_ready", "loop_end", "end", "it0 compute", "it0 barrier1", "it0 store", "it0 end",
             "it1 compute", "it1 barrier1", "it1 store", "it1 end", "it2 end", "it3 end", "-"]
    print(f"\n{mode} B={B} T={T} Cin={Cin} Cout={Cout} K={K} s={stride}: {us:.1f} us/launch, {len(st)} WGs")
    for k in [0, 1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 3, 4]:
        col = rel[:, k]
        if np.all(np.isnan(col)): continue
        print(f"  {names[k]:12s} min {np.nanmin(col):7.2f}  med {np.nanmedian(col):7.2f}  max {np.nanmax(col):7.2f}")
    hw = st[:, 14] & 0xffffffff; xcc = (st[:, 14] >> 32) & 0xf
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf))
    import collections
    cnt = collections.Counter(cu.tolist())
    print(f"  distinct CUs used {len(cnt)}, WGs per CU histogram {sorted(collections.Counter(cnt.values()).items())}")
    ghz = st[:, 15] / ((st[:, 4] - st[:, 0]) * 10.0)
    print(f"  shader clock over WG life: med {np.median(ghz):.2f} GHz (min {ghz.min():.2f} max {ghz.max():.2f})")
    d = rel[:, 4] - rel[:, 0]
    print(f"  WG lifetime  min {d.min():7.2f}  med {np.median(d):7.2f}  max {d.max():7.2f};  epilogue med {np.median(rel[:,4]-rel[:,3]):.2f}")

run(64, 256, 128, 64, 3, 1)
run(64, 256, 256, 256, 3, 1)
run(64, 32, 256, 128, 5, 2, "convT")
run(64, 128, 64, 128, 5, 2, "convT")
