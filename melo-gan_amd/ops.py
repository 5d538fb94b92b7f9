"""Thin, shape-checked Python wrappers over the C-ABI of libmelogan_hip.so.

Every function takes fp32 CUDA (HIP) tensors, validates the shapes the kernel and its grid
assume ON THE HOST (a wrong shape must never reach the GPU), and enqueues on PyTorch's
current stream.  Activations are channels-last (B, T, C); weights keep the reference's
state_dict layouts.  Nothing here falls back to PyTorch math.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from ._lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_TANH, Epilogue  # noqa: F401

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: Tensor, name: str, shape=None, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/HIP tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def epilogue(out_shape, N, bias=None, scale=None, shift=None, zout=None, act=ACT_NONE, gref=None,
             gact=ACT_NONE, emul=None, gscale=None, accumulate=False) -> Epilogue:
    for nm, v in (("bias", bias), ("scale", scale), ("shift", shift), ("gscale", gscale)):
        if v is not None:
            _chk(v, nm, (N,))
    for nm, v in (("zout", zout), ("gref", gref), ("emul", emul)):
        if v is not None:
            _chk(v, nm)
            if v.numel() != _numel(out_shape):
                raise ValueError(f"{nm}: numel {v.numel()} != output numel {_numel(out_shape)}")
    if (scale is None) != (shift is None):
        raise ValueError("scale and shift come in pairs")
    return Epilogue(_p(bias), _p(scale), _p(shift), _p(zout), act, _p(gref), gact, _p(emul), _p(gscale),
                    1 if accumulate else 0)


# Optional per-launch observer (bench.py's roofline leg): called as hook(symbol, flops) and must return a
# context manager that brackets the launch (e.g. with HIP events).  None => zero overhead.
_launch_hook = None


def set_launch_hook(hook):
    global _launch_hook
    _launch_hook = hook


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _observe(symbol_fn, flops, launch=None):
    """`launch` (optional) re-issues exactly this launch -- a profiling hook may keep it to replay the launch later."""
    if _launch_hook is None:
        return _NullCtx()
    return _launch_hook(symbol_fn(), flops, launch)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


# ---------------------------------------------------------------------------------------
# window GEMMs
# ---------------------------------------------------------------------------------------
_THIN_SYMBOLS = {1: "thin_in_kernel%.0s", 2: "thin_out_kernel<%s,4>", 3: "thin_out_kernel<%s,8>"}


def _conv_work(lib, B, Tout, N, Cin, device):
    """Split-K scratch for small-output convolutions (only they can use it: <= 8 MB of output)."""
    if Cin < 64 or B * Tout * N > (1 << 21):
        return None
    return workspace(lib.mg_conv_workspace_bytes(B, Tout, N), device, "conv")


class conv_lds_pad:
    """with ops.conv_lds_pad(nbytes): the window-GEMM launches inside take `nbytes` more LDS per workgroup -- fewer resident
    workgroups per CU -- for a branch that runs beside another stream's critical path (mg_conv_set_lds_pad)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)

    def __enter__(self):
        L.check(L.load().mg_conv_set_lds_pad(self.nbytes), "mg_conv_set_lds_pad")

    def __exit__(self, *a):
        L.load().mg_conv_set_lds_pad(0)
        return False


def conv_gather(x: Tensor, w: Tensor, y: Tensor, N: int, K: int, stride: int, w_sn: int, w_sc: int,
                flip: bool = False, **epi) -> Tensor:
    """Generic gather window-GEMM (see mg_conv1d_gather).  x: (B, Tin, Cin); y: (B, Ty, N) with
    Ty >= Tout (rows beyond Tout are left untouched -- the generator's zero-pad branch)."""
    _chk(x, "x")
    _chk(w, "w")
    _chk(y, "y")
    if x.dim() != 3 or y.dim() != 3:
        raise ValueError("x and y must be (B, T, C)")
    B, Tin, Cin = x.shape
    pad = (K - 1) // 2
    Tout = (Tin + 2 * pad - K) // stride + 1
    if y.shape[0] != B or y.shape[2] != N or y.shape[1] < Tout:
        raise ValueError(f"y: expected (B={B}, >={Tout}, {N}), got {tuple(y.shape)}")
    need = (N - 1) * w_sn + (Cin - 1) * w_sc + K
    if w.numel() < need:
        raise ValueError(f"w: numel {w.numel()} < {need} implied by strides")
    e = epilogue((B, Tout, N), N, **epi)
    if y.shape[1] != Tout and (e.zout or e.gref or e.emul):
        raise ValueError("padded y cannot be combined with elementwise epilogue tensors")
    lib = L.load()
    def sym():
        thin = lib.mg_conv_thin_route(_p(x), Tin * Cin, Cin, N, K, stride, 0)
        if thin:
            return _THIN_SYMBOLS[thin] % "false"
        return "conv_wgemm_kernel<%d,%d,false,%s,%s>" % (            # (S, K, TR2, NCK weight layout, tile)
            stride, K, "true" if w_sc < w_sn else "false",
            {22: "2,2", 12: "1,2", 11: "1,1"}[lib.mg_conv_tile_config(B * Tout, N, 0)])
    work = _conv_work(lib, B, Tout, N, Cin, x.device)
    def launch():
        return lib.mg_conv1d_gather(_p(x), _p(w), _p(y), B, Tin, Cin, N, K, stride, 1 if flip else 0, w_sn, w_sc,
                                    Tin * Cin, y.shape[1] * N, C.byref(e), _p(work),
                                    work.numel() if work is not None else 0, _stream())
    with _observe(sym, 2.0 * B * Tout * N * Cin * K, launch):
        rc = launch()
    L.check(rc, "mg_conv1d_gather")
    return y


def wino3_supported(B: int, T: int, Cin: int, N: int) -> bool:
    return bool(L.load().mg_conv1d_wino3_supported(B, T, Cin, N))


def wino3_weights(w: Tensor, N: int, Cin: int, w_sn: int, w_sc: int, flip: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """The F(2,3) filter transform of a three-tap weight, (Cin/4, 4, N, 4) (mg_wino3_weights); `out`: an existing image to
    refresh (a captured graph keeps reading the same buffer)."""
    _chk(w, "w")
    if w.numel() < (N - 1) * w_sn + (Cin - 1) * w_sc + 3:
        raise ValueError("wino3_weights: w smaller than its strides imply")
    wt = out if out is not None else torch.empty(Cin // 4, 4, N, 4, device=w.device, dtype=torch.float32)
    _chk(wt, "wt", (Cin // 4, 4, N, 4))
    L.check(L.load().mg_wino3_weights(_p(w), _p(wt), N, Cin, w_sn, w_sc, 1 if flip else 0, _stream()), "mg_wino3_weights")
    return wt


def wino3_weights_multi(jobs):
    """jobs = [(w, wt, N, Cin, w_sn, w_sc, flip), ...]: the filter transforms of several layers in one launch (wt: existing
    (Cin/4, 4, N, 4) images)."""
    if not 0 < len(jobs) <= L.MAX_WINO_WJOBS:
        raise ValueError(f"wino3_weights_multi: 1..{L.MAX_WINO_WJOBS} jobs")
    arr = (L.WinoWJob * len(jobs))()
    for a, (w, wt, N, Cin, w_sn, w_sc, flip) in zip(arr, jobs):
        _chk(w, "w")
        _chk(wt, "wt", (Cin // 4, 4, N, 4))
        if w.numel() < (N - 1) * w_sn + (Cin - 1) * w_sc + 3:
            raise ValueError("wino3_weights_multi: w smaller than its strides imply")
        a.w, a.wt, a.N, a.Cin, a.w_sn, a.w_sc, a.flip = w.data_ptr(), wt.data_ptr(), N, Cin, w_sn, w_sc, 1 if flip else 0
    L.check(L.load().mg_wino3_weights_multi(arr, len(jobs), _stream()), "mg_wino3_weights_multi")


def conv_wino3(x: Tensor, wt: Tensor, y: Tensor, **epi) -> Tensor:
    """Stride-1 three-tap convolution (padding 1) through minimal filtering (mg_conv1d_wino3).  x: (B, T, Cin);
    wt: wino3_weights(...) (Cin/4, 4, N, 4); y: (B, T, N)."""
    _chk(x, "x")
    _chk(wt, "wt")
    _chk(y, "y")
    B, T, Cin = x.shape
    N = wt.shape[2]
    if tuple(wt.shape) != (Cin // 4, 4, N, 4) or tuple(y.shape) != (B, T, N):
        raise ValueError(f"conv_wino3: wt {tuple(wt.shape)} / y {tuple(y.shape)} do not fit x {tuple(x.shape)}")
    e = epilogue((B, T, N), N, **epi)
    lib = L.load()
    def launch():
        return lib.mg_conv1d_wino3(_p(x), _p(wt), _p(y), B, T, Cin, N, C.byref(e), _stream())
    with _observe(lambda: "wino3_kernel", 2.0 * B * T * N * Cin * 3, launch):      # the direct form's FLOPs (algorithmic)
        rc = launch()
    L.check(rc, "mg_conv1d_wino3")
    return y


def conv_scatter2(x: Tensor, w: Tensor, y: Tensor, N: int, w_sn: int, w_sc: int, odd: bool = False, **epi) -> Tensor:
    """Stride-2 K=5 transposed window-GEMM (see mg_conv1d_scatter2).  x: (B, Tin, Cin); y: (B, Ty>=Tout, N),
    Tout = 2*Tin, or 2*Tin-1 with odd=True (dgrad of a stride-2 conv over an odd-length input)."""
    _chk(x, "x")
    _chk(w, "w")
    _chk(y, "y")
    B, Tin, Cin = x.shape
    Tout = 2 * Tin - (1 if odd else 0)
    if y.dim() != 3 or y.shape[0] != B or y.shape[2] != N or y.shape[1] < Tout:
        raise ValueError(f"y: expected (B={B}, >={Tout}, {N}), got {tuple(y.shape)}")
    need = (N - 1) * w_sn + (Cin - 1) * w_sc + 5
    if w.numel() < need:
        raise ValueError(f"w: numel {w.numel()} < {need} implied by strides")
    e = epilogue((B, Tout, N), N, **epi)
    if y.shape[1] != Tout and (e.zout or e.gref or e.emul):
        raise ValueError("padded y cannot be combined with elementwise epilogue tensors")
    lib = L.load()
    def sym():
        thin = lib.mg_conv_thin_route(_p(x), Tin * Cin, Cin, N, 5, 2, 1)
        if thin:
            return _THIN_SYMBOLS[thin] % "true"
        return "conv_wgemm_kernel<2,5,true,%s,%s>" % (
            "true" if w_sc < w_sn else "false", "1,2" if lib.mg_conv_tile_config(B * Tin, N, 1) == 12 else "1,1")
    work = _conv_work(lib, B, Tout, N, Cin, x.device)
    with _observe(sym, 2.0 * B * Tin * N * Cin * 5):
        rc = lib.mg_conv1d_scatter2(_p(x), _p(w), _p(y), B, Tin, Cin, N, Tout, w_sn, w_sc, Tin * Cin, y.shape[1] * N,
                                    C.byref(e), _p(work), work.numel() if work is not None else 0, _stream())
    L.check(rc, "mg_conv1d_scatter2")
    return y


def conv1d_fwd(x, w, y, stride, **epi):
    """nn.Conv1d(Cin, Cout, K, stride, padding=K//2) forward; w: (Cout, Cin, K)."""
    Cout, Cin, K = w.shape
    if x.shape[2] != Cin:
        raise ValueError("conv1d_fwd: channel mismatch")
    return conv_gather(x, w, y, Cout, K, stride, Cin * K, K, **epi)


def conv1d_dgrad(dy, w, dx, stride, **epi):
    """Data gradient of nn.Conv1d; w: (Cout, Cin, K); dy: (B, Tout, Cout) -> dx: (B, Tin, Cin)."""
    Cout, Cin, K = w.shape
    if dy.shape[2] != Cout:
        raise ValueError("conv1d_dgrad: channel mismatch")
    if stride == 1:
        return conv_gather(dy, w, dx, Cin, K, 1, K, Cin * K, flip=True, **epi)
    if K != 5:
        raise ValueError("stride-2 dgrad needs K=5")
    odd = dx.shape[1] == 2 * dy.shape[1] - 1
    return conv_scatter2(dy, w, dx, Cin, K, Cin * K, odd=odd, **epi)


def convT1d_fwd(x, w, y, **epi):
    """nn.ConvTranspose1d(Cin, Cout, 5, stride=2, padding=2, output_padding=1) forward; w: (Cin, Cout, 5)."""
    Cin, Cout, K = w.shape
    if x.shape[2] != Cin or K != 5:
        raise ValueError("convT1d_fwd: shape mismatch")
    return conv_scatter2(x, w, y, Cout, K, Cout * K, **epi)


def convT1d_dgrad(dy, w, dx, **epi):
    """Data gradient of the stride-2 ConvTranspose1d: dx[u,ci] = sum dy[2u+k-2,co] w[ci,co,k]."""
    Cin, Cout, K = w.shape
    if dy.shape[2] != Cout:
        raise ValueError("convT1d_dgrad: channel mismatch")
    return conv_gather(dy, w, dx, Cin, K, 2, Cout * K, K, **epi)


def wq_relayout(w: Tensor, wq: Tensor, N: int, Cc: int, K: int, w_sn: int, w_sc: int) -> Tensor:
    """wq[((c/4)*K + k)*N + n][c%4] = w[n*w_sn + c*w_sc + k]: the weight layout of conv16 (see mg_conv16)."""
    _chk(w, "w")
    _chk(wq, "wq")
    if wq.numel() != N * Cc * K or w.numel() < (N - 1) * w_sn + (Cc - 1) * w_sc + K or Cc % 4:
        raise ValueError("wq_relayout: size mismatch")
    L.check(L.load().mg_wq_relayout(_p(w), _p(wq), N, Cc, K, w_sn, w_sc, _stream()), "mg_wq_relayout")
    return wq


def conv16_supported(B: int, Tin: int, Cin: int, N: int, transposed: bool, Tout: int = 0) -> bool:
    return bool(L.load().mg_conv16_supported(B, Tin, Cin, N, 1 if transposed else 0, Tout))


def conv16_poolable(B: int, Tin: int, Cin: int, N: int) -> bool:
    return bool(L.load().mg_conv16_poolable(B, Tin, Cin, N))


def conv16_pool(x: Tensor, wq: Tensor, y: Tensor, N: int, pool: Tensor, scale: float, **epi) -> Tensor:
    """conv16 (gather form) that also writes pool[b][n] = scale * sum_t y[b][t][n] (mg_conv16_pool)."""
    return conv16(x, wq, y, N, False, pool=(pool, scale), **epi)


def conv16_plan(B: int, Tin: int, N: int, transposed: bool):
    """(batch rows one tile spans, partial-statistics rows a launch writes) of mg_conv16's tiling for a shape."""
    return _conv16_plan(B, Tin, N, transposed)[:2]


def _conv16_plan(B, Tin, N, transposed):
    tb, rows, bm = C.c_int(), C.c_int(), C.c_int()
    L.check(L.load().mg_conv16_plan(B, Tin, N, 1 if transposed else 0, C.byref(tb), C.byref(rows), C.byref(bm)), "mg_conv16_plan")
    return tb.value, rows.value, bm.value


def conv16(x: Tensor, wq: Tensor, y: Tensor, N: int, transposed: bool, odd: bool = False, stats=None, pool=None,
           perm: bool = False, mix=None, bnb=None, **epi) -> Tensor:
    """Stride-2 K=5 window GEMM on 16x16 MFMA tiles with WQ-layout weights (mg_conv16_ex).  transposed=False: the gather
    form (Conv1d forward / ConvTranspose1d data-gradient), True: the scatter form (ConvTranspose1d forward / Conv1d
    data-gradient; odd: Tout = 2*Tin - 1).  y: (B, Ty >= Tout, N).  Riders of the same launch:
      stats: a float buffer that receives per-column partial statistics (sum, centred sum of squares, count) of the stored
             values (size 3 * part_rows * N from conv16_plan) for bn_train_fwd_parts;
      pool = (tensor (B, N), scale): the temporal mean of the output (gather form, conv16_poolable shapes);
      perm: y is (B, N, Tout) -- i.e. the (B, N*Tout) matrix a Linear produced and the reference views as (B, N, L);
      mix = (real, alpha, out, rows): out[b] = alpha[b] * real[b] + (1 - alpha[b]) * y[b] for b < rows (tensors laid out
            like y): the gradient penalty's interpolate;
      bnb = (a, z, mean, invstd, part, act): y is the gradient reaching a train-mode BatchNorm + ReLU / LeakyReLU layer whose
            forward tensors are a, z (like y): part (2 * part_rows * N float64, conv16_plan) receives the two column sums its
            backward needs (bn_train_bwd_parts then is one launch)."""
    _chk(x, "x")
    _chk(wq, "wq")
    _chk(y, "y")
    B, Tin, Cin = x.shape
    Tout = (2 * Tin - (1 if odd else 0)) if transposed else (Tin + 4 - 5) // 2 + 1
    if perm:
        if tuple(y.shape) != (B, N, Tout):
            raise ValueError(f"y (perm): expected {(B, N, Tout)}, got {tuple(y.shape)}")
        Ty = Tout
    else:
        if y.dim() != 3 or y.shape[0] != B or y.shape[2] != N or y.shape[1] < Tout:
            raise ValueError(f"y: expected (B={B}, >={Tout}, {N}), got {tuple(y.shape)}")
        Ty = y.shape[1]
    if wq.numel() != N * Cin * 5:
        raise ValueError(f"wq: numel {wq.numel()} != {N * Cin * 5}")
    e = epilogue((B, Tout, N), N, **epi)
    if Ty != Tout and (e.zout or e.gref or e.emul):
        raise ValueError("padded y cannot be combined with elementwise epilogue tensors")
    lib = L.load()
    if not lib.mg_conv16_supported(B, Tin, Cin, N, 1 if transposed else 0, Tout):
        raise ValueError(f"conv16: unsupported shape B={B} Tin={Tin} Cin={Cin} N={N}")
    ex = L.Conv16Extra()
    if stats is not None:
        part = _chk(stats, "stats")
        if part.numel() < 3 * conv16_plan(B, Tin, N, transposed)[1] * N:
            raise ValueError("conv16: partial-statistics buffer too small (conv16_plan)")
        ex.part = _p(part)
    if pool is not None:
        pt, scale = pool
        _chk(pt, "pool", (B, N))
        if transposed or not lib.mg_conv16_poolable(B, Tin, Cin, N):
            raise ValueError(f"conv16: shape B={B} Tin={Tin} Cin={Cin} N={N} is not poolable")
        ex.pool, ex.pool_scale = _p(pt), float(scale)
    ex.y_perm = 1 if perm else 0
    if bnb is not None:
        ba, bz, bmean, binv, bpart, bact = bnb
        if perm or Ty != Tout:
            raise ValueError("conv16: bnb needs the plain, dense output")
        _chk(ba, "bnb a", (B, Tout, N))
        _chk(bz, "bnb z", (B, Tout, N))
        _chk(bmean, "bnb mean", (N,))
        _chk(binv, "bnb invstd", (N,))
        _chk(bpart, "bnb part", dtype=torch.float64)
        if bpart.numel() < 2 * conv16_plan(B, Tin, N, transposed)[1] * N:
            raise ValueError("conv16: bnb partial buffer too small (2 * part_rows * N doubles)")
        ex.bnb_a, ex.bnb_z, ex.bnb_mean, ex.bnb_invstd, ex.bnb_part, ex.bnb_act = _p(ba), _p(bz), _p(bmean), _p(binv), _p(bpart), int(bact)
    if mix is not None:
        real, alpha, out, rows = mix
        if perm or not 0 < rows <= B:
            raise ValueError("conv16: mix needs the plain output order and 0 < rows <= B")
        for nm, t in (("mix real", real), ("mix out", out)):
            _chk(t, nm)
            if t.dim() != 3 or t.shape[0] < rows or tuple(t.shape[1:]) != (Ty, N):
                raise ValueError(f"conv16: {nm} must be (>= {rows}, {Ty}, {N}), got {tuple(t.shape)}")
        _chk(alpha, "mix alpha")
        if alpha.numel() < rows:
            raise ValueError("conv16: mix alpha has fewer entries than rows")
        ex.mix_real, ex.mix_alpha, ex.mix_out, ex.mix_rows = _p(real), _p(alpha), _p(out), int(rows)

    def launch():
        return lib.mg_conv16_ex(_p(x), _p(wq), _p(y), B, Tin, Cin, N, 1 if transposed else 0, Tout, Tin * Cin, Ty * N, C.byref(e),
                                C.byref(ex), _stream())
    rid = "true" if (stats is not None or pool is not None or perm or mix is not None) else "false"
    sym = lambda: "conv16_kernel<%s,%d,%s>" % ("true" if transposed else "false", _conv16_plan(B, Tin, N, transposed)[2] // 32, rid)  # noqa: E731
    with _observe(sym, 2.0 * B * (Tin if transposed else Tout) * N * Cin * 5, launch):
        rc = launch()
    L.check(rc, "mg_conv16")
    return y


SKINNY_MAX_ROWS = 512      # Linear layers with at most this many rows use the skinny-GEMM kernel


def _linear(x, w, y, K, N, w_sn, w_sc, epi, perm_L=0):
    M = x.shape[0]
    e = epilogue((M, N), N, **epi)
    lib = L.load()
    need = lib.mg_linear_workspace_bytes(M, N, K)
    work = workspace(need, x.device, "linear") if need else None
    with _observe(lambda: "linear_skinny_kernel", 2.0 * M * N * K):
        rc = lib.mg_linear_perm(_p(x), _p(w), _p(y), M, K, N, w_sn, w_sc, C.byref(e), perm_L, _p(work),
                                work.numel() if work is not None else 0, _stream())
    L.check(rc, "mg_linear")
    return y


def linear_fwd(x, w, y, perm_L: int = 0, **epi):
    """nn.Linear forward; x: (B, in), w: (out, in), y: (B, out).  perm_L > 0: y is (B, perm_L, out / perm_L) -- the
    channels-last tensor behind the reference's view(B, C, L) + permute (mg_linear_perm); elementwise epilogue tensors are
    laid out like y."""
    _chk(x, "x")
    _chk(w, "w")
    _chk(y, "y")
    out_f, in_f = w.shape
    B = x.shape[0]
    if perm_L:
        if B > SKINNY_MAX_ROWS or out_f % perm_L or x.dim() != 2 or x.shape[1] != in_f or tuple(y.shape) != (B, perm_L, out_f // perm_L):
            raise ValueError(f"linear_fwd(perm_L={perm_L}): shape mismatch x{tuple(x.shape)} w{tuple(w.shape)} y{tuple(y.shape)}")
        return _linear(x, w, y, in_f, out_f, in_f, 1, epi, perm_L)
    if x.dim() != 2 or x.shape[1] != in_f or tuple(y.shape) != (B, out_f):
        raise ValueError(f"linear_fwd: shape mismatch x{tuple(x.shape)} w{tuple(w.shape)} y{tuple(y.shape)}")
    if B <= SKINNY_MAX_ROWS:
        return _linear(x, w, y, in_f, out_f, in_f, 1, epi)
    conv_gather(x.view(B, 1, in_f), w, y.view(B, 1, out_f), out_f, 1, 1, in_f, 1, **epi)
    return y


def linear_dgrad(dy, w, dx, **epi):
    """dx = dy @ w; dy: (B, out), w: (out, in), dx: (B, in)."""
    _chk(dy, "dy")
    _chk(w, "w")
    _chk(dx, "dx")
    out_f, in_f = w.shape
    B = dy.shape[0]
    if dy.dim() != 2 or dy.shape[1] != out_f or tuple(dx.shape) != (B, in_f):
        raise ValueError("linear_dgrad: shape mismatch")
    if B <= SKINNY_MAX_ROWS:
        return _linear(dy, w, dx, out_f, in_f, 1, in_f, epi)
    conv_gather(dy.view(B, 1, out_f), w, dx.view(B, 1, in_f), in_f, 1, 1, 1, in_f, **epi)
    return dx


# ---------------------------------------------------------------------------------------
# row chains: a sample's small layer stack in one launch (csrc/row_chain.hip)
# ---------------------------------------------------------------------------------------
class Chain:
    """Builder for mg_row_chain: ops over LDS vector slots (L.CHAIN_SLOTS slots of up to L.CHAIN_MAX_VEC floats), one
    workgroup per row.  Row tensors are 2-D float32 device tensors with unit column stride (column blocks of wider
    matrices are fine: the row stride is taken from the tensor).  launch() enqueues on the current stream."""

    def __init__(self, rows: int):
        self.rows, self.ops, self._keep = int(rows), [], []

    @staticmethod
    def supported(*dims) -> bool:
        """Every vector length of the chain fits a slot."""
        return all(0 < int(d) <= L.CHAIN_MAX_VEC for d in dims)

    @staticmethod
    def weights_ok(*ws) -> bool:
        """Weight matrices a chain can read: contiguous (N, K); rows 16-byte aligned wherever the 16-byte path applies."""
        for w in ws:
            if w.dim() != 2 or not w.is_contiguous() or w.dtype != torch.float32:
                return False
            if w.shape[1] >= 16 and w.shape[1] % 4 == 0 and w.data_ptr() % 16:
                return False
        return True

    def _rows(self, t, name, n, rows=None, dtype=torch.float32):
        rows = self.rows if rows is None else rows
        if (not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1 or
                t.shape[0] < rows or t.shape[1] < n or (t.shape[0] > 1 and t.stride(0) < t.shape[1])):
            raise ValueError(f"chain {name}: expected a (>= {rows}, >= {n}) float32 device tensor with unit column stride")
        self._keep.append(t)
        return C.c_void_p(t.data_ptr()), t.stride(0)

    def _vec(self, t, name, n):
        _chk(t, name)
        if t.numel() < n:
            raise ValueError(f"chain {name}: needs {n} elements")
        self._keep.append(t)
        return C.c_void_p(t.data_ptr())

    def _op(self, kind, a=0, b=0, n0=0, n1=0):
        if len(self.ops) >= L.CHAIN_MAX_OPS:
            raise ValueError(f"chain: more than {L.CHAIN_MAX_OPS} ops")
        for s_ in (a, b):
            if not 0 <= s_ < L.CHAIN_SLOTS:
                raise ValueError("chain: slot out of range")
        for n in (n0, n1):
            if n > L.CHAIN_MAX_VEC:
                raise ValueError(f"chain: vector of {n} floats exceeds a slot ({L.CHAIN_MAX_VEC})")
        o = L.ChainOp()
        o.kind, o.a, o.b, o.n0, o.n1 = kind, a, b, n0, n1
        self.ops.append(o)
        return o

    def load(self, slot, t, n=None, accumulate=False, mod=0, at=0):
        """slot[at : at + n] (+)= row of t (row r % mod if mod)."""
        n = t.shape[1] if n is None else n
        if at < 0 or at + n > L.CHAIN_MAX_VEC:
            raise ValueError("chain load: destination range exceeds the slot")
        o = self._op(L.CH_LOAD, b=slot, n0=n)
        o.n1 = int(at)
        o.p0, o.ld0 = self._rows(t, "load", n, rows=mod or None)
        o.i0, o.i1 = (1 if accumulate else 0), int(mod)
        return self

    def copy(self, src, dst, n, src_at=0, dst_at=0):
        """slot dst[dst_at : dst_at + n] = slot src[src_at : src_at + n]."""
        if src == dst or min(src_at, dst_at) < 0 or max(src_at, dst_at) + n > L.CHAIN_MAX_VEC:
            raise ValueError("chain copy: bad ranges")
        o = self._op(L.CH_COPY, a=src, b=dst, n0=n)
        o.n1, o.i1 = int(dst_at), int(src_at)
        return self

    def store(self, slot, t, n=None):
        n = t.shape[1] if n is None else n
        o = self._op(L.CH_STORE, a=slot, n0=n)
        o.q0, o.lq0 = self._rows(t, "store", n)
        return self

    def mean_t(self, slot, x, out=None):
        """slot = mean over time of x (rows, T, C) contiguous."""
        _chk(x, "mean_t x")
        if x.dim() != 3 or x.shape[0] < self.rows:
            raise ValueError("chain mean_t: x must be (>= rows, T, C)")
        self._keep.append(x)
        o = self._op(L.CH_MEAN_T, b=slot, n0=x.shape[2])
        o.p0, o.ld0, o.i0 = C.c_void_p(x.data_ptr()), x.shape[2], x.shape[1]
        if out is not None:
            o.q0, o.lq0 = self._rows(out, "mean_t out", x.shape[2])
        return self

    def layernorm(self, a, b, gamma, beta, xhat=None, y=None, eps=1e-5):
        D = gamma.numel()
        o = self._op(L.CH_LAYERNORM, a=a, b=b, n0=D)
        o.p0, o.p1, o.f0 = self._vec(gamma, "gamma", D), self._vec(beta, "beta", D), float(eps)
        if xhat is not None:
            o.q0, o.lq0 = self._rows(xhat, "xhat", D)
        if y is not None:
            o.q1, o.lq1 = self._rows(y, "ln out", D)
        return self

    def linear_fwd(self, a, b, w, bias=None, act=ACT_NONE, mask=None, zout=None, out=None):
        """slot b = act(slot a @ w.T + bias) * mask;  w: (N, K) contiguous."""
        _chk(w, "w")
        N, K = w.shape
        self._keep.append(w)
        o = self._op(L.CH_LIN_FWD, a=a, b=b, n0=K, n1=N)
        o.p0, o.ld0, o.act = C.c_void_p(w.data_ptr()), K, act
        if bias is not None:
            o.p1 = self._vec(bias, "bias", N)
        if mask is not None:
            o.p2, o.ld2 = self._rows(mask, "mask", N)
        if zout is not None:
            o.q0, o.lq0 = self._rows(zout, "zout", N)
        if out is not None:
            o.q1, o.lq1 = self._rows(out, "out", N)
        return self

    def linear_dgrad(self, a, b, w, gref=None, gact=ACT_NONE, mask=None, out=None):
        """slot b = (slot a @ w) * act'(gref) * mask;  w: (OUT, IN) contiguous."""
        _chk(w, "w")
        OUT, IN = w.shape
        self._keep.append(w)
        o = self._op(L.CH_LIN_DGRAD, a=a, b=b, n0=OUT, n1=IN)
        o.p0, o.ld0, o.act = C.c_void_p(w.data_ptr()), IN, gact
        if gref is not None:
            o.p1, o.ld1 = self._rows(gref, "gref", IN)
        if mask is not None:
            o.p2, o.ld2 = self._rows(mask, "mask", IN)
        if out is not None:
            o.q1, o.lq1 = self._rows(out, "out", IN)
        return self

    def softmax_ce(self, a, b, target, loss_rows, scale, n_classes):
        """Per row: loss_rows[r] = CE(logits in slot a, target[r]); slot b = scale * (softmax - onehot)."""
        _chk(target, "target", dtype=torch.int64)
        _chk(loss_rows, "loss_rows")
        if target.numel() < self.rows or loss_rows.numel() < self.rows or not 0 < n_classes <= 32:
            raise ValueError("chain softmax_ce: sizes")
        self._keep += [target, loss_rows]
        o = self._op(L.CH_SOFTMAX_CE, a=a, b=b, n0=n_classes)
        o.t0, o.q0, o.f0 = C.c_void_p(target.data_ptr()), C.c_void_p(loss_rows.data_ptr()), float(scale)
        return self

    def dhead(self, a, b, w, bias, emb, ds, s, demb=None):
        """Critic head on slot a = f (F): s[r] = f . w[:F] + emb[r % Be] . w[F:] + bias; slot b = ds[r] * w[:F] * lrelu'(f);
        demb rows <- ds[r] * w[F:] (needs one embedding row per sample)."""
        _chk(w, "w")
        _chk(bias, "bias")
        _chk(ds, "ds")
        _chk(s, "s")
        E = emb.shape[1] if emb is not None else 0
        F = w.numel() - E
        if F <= 0 or ds.numel() < self.rows or s.numel() < self.rows:
            raise ValueError("chain dhead: sizes")
        self._keep += [w, bias, ds, s]
        o = self._op(L.CH_DHEAD, a=a, b=b, n0=F, n1=E)
        o.p0, o.p1, o.p3, o.q0 = C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(ds.data_ptr()), C.c_void_p(s.data_ptr())
        if emb is not None:
            Be = emb.shape[0]
            o.p2, o.ld2 = self._rows(emb, "emb", E, rows=Be)
            o.i1 = Be
            if demb is not None:
                if Be != self.rows:
                    raise ValueError("chain dhead: the embedding gradient needs one embedding row per sample")
                o.q1, o.lq1 = self._rows(demb, "demb", E)
        return self

    def launch(self):
        arr = (L.ChainOp * len(self.ops))(*self.ops)
        flops = 0.0
        for o in self.ops:
            if o.kind in (L.CH_LIN_FWD, L.CH_LIN_DGRAD):
                flops += 2.0 * self.rows * o.n0 * o.n1
        with _observe(lambda: "row_chain_kernel", flops):
            rc = L.load().mg_row_chain(arr, len(self.ops), self.rows, _stream())
        L.check(rc, "mg_row_chain")


def mean_scaled(src, out, scale=1.0):
    _chk(src, "src")
    _chk(out, "out")
    L.check(L.load().mg_mean_scaled(_p(src), _p(out), src.numel(), float(scale), _stream()), "mg_mean_scaled")


def _sn_jobs(layers, bwd):
    if not 0 < len(layers) <= L.MAX_SN_JOBS:
        raise ValueError(f"spectral_norm: 1..{L.MAX_SN_JOBS} layers per launch")
    arr = (L.SnJob * len(layers))()
    for a, ly in zip(arr, layers):
        w, we, u, v, sg = ly["w_orig"], ly["w_eff"], ly["u"], ly["v"], ly["sigma"]
        rows = w.shape[0]
        cols = w.numel() // rows
        _chk(w, "w_orig")
        _chk(we, "w_eff", tuple(w.shape))
        _chk(u, "u", (rows,))
        _chk(v, "v", (cols,))
        _chk(sg, "sigma", (1,))
        a.w_orig, a.w_eff, a.u, a.v, a.sigma, a.rows, a.cols = w.data_ptr(), we.data_ptr(), u.data_ptr(), v.data_ptr(), sg.data_ptr(), rows, cols
        a.dw = None
        if bwd:
            _chk(ly["dw"], "dw", tuple(w.shape))
            a.dw = ly["dw"].data_ptr()
    return arr


def spectral_norm_fwd(layers, train: bool, eps: float = 1e-12):
    """torch.nn.utils.spectral_norm's weight computation for several layers in one launch (mg_spectral_norm_fwd): layers =
    dicts of w_orig (out, ...), w_eff (same shape), u (out), v (rest), sigma (1).  train: one power iteration first (u, v
    updated in place), as the module does in training mode."""
    arr = _sn_jobs(layers, False)
    L.check(L.load().mg_spectral_norm_fwd(arr, len(layers), 1 if train else 0, float(eps), _stream()), "mg_spectral_norm_fwd")


def spectral_norm_bwd(layers):
    """layers[i]["dw"] holds the gradient w.r.t. w_eff and leaves holding the gradient w.r.t. w_orig (mg_spectral_norm_bwd)."""
    arr = _sn_jobs(layers, True)
    L.check(L.load().mg_spectral_norm_bwd(arr, len(layers), _stream()), "mg_spectral_norm_bwd")


def stamp(buf, i: int):
    """buf[i] (int64 device tensor) = the device clock when this node runs (100 MHz ticks)."""
    L.check(L.load().mg_stamp(buf.data_ptr() + 8 * int(i), _stream()), "mg_stamp")


# ---------------------------------------------------------------------------------------
# weight gradients
# ---------------------------------------------------------------------------------------
_ws_cache = {}
_ws_retired = []      # superseded scratch buffers: never freed, a captured hipGraph may still hold their address


def workspace(nbytes: int, device, tag: str = "default") -> Tensor:
    """Grow-only scratch buffer per (device, tag, current stream) -- launches that may run concurrently on
    different streams never share scratch.  A buffer that has to grow is REPLACED, and the old one is kept alive for
    the life of the process: a hipGraph captured earlier has its address baked in and would otherwise replay into
    memory the caching allocator has handed to someone else (sub-steps are captured one by one, each after its own
    eager warm-up, so a later sub-step may ask for more than the graphs captured before it saw).  Growth is geometric,
    so the retired buffers of a tag add up to less than its final size."""
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        size = max(nbytes, 1 << 20, 2 * buf.numel() if buf is not None else 0)
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def wgrad(small: Tensor, large: Tensor, out: Tensor, K: int, stride: int,
          small2: Optional[Tensor] = None, large2: Optional[Tensor] = None, work: Optional[Tensor] = None,
          bias_out: Optional[Tensor] = None, bias_from: int = 0):
    """out[a][b][k] = sum S[bt,u,a] * L[bt,u*stride+k-(K-1)/2,b] over one or two (S, L) segments; optionally the
    bias gradient of segment 0 in the same launch (bias_from 1: column sums of S, 2: of L)."""
    _chk(small, "small")
    _chk(large, "large")
    _chk(out, "out")
    nb0, Ts, A = small.shape
    if large.shape[0] != nb0:
        raise ValueError("wgrad: batch mismatch")
    Tl, Bc = large.shape[1], large.shape[2]
    pad = (K - 1) // 2
    if (Tl + 2 * pad - K) // stride + 1 != Ts and not (stride == 2 and Tl == 2 * Ts):
        raise ValueError(f"wgrad: Ts={Ts} inconsistent with Tl={Tl}, K={K}, stride={stride}")
    if out.numel() != A * Bc * K:
        raise ValueError(f"wgrad: out numel {out.numel()} != {A * Bc * K}")
    nb1 = 0
    if small2 is not None:
        _chk(small2, "small2")
        _chk(large2, "large2")
        nb1 = small2.shape[0]
        if tuple(small2.shape[1:]) != (Ts, A) or tuple(large2.shape) != (nb1, Tl, Bc):
            raise ValueError("wgrad: segment 1 shape mismatch")
    if bias_out is not None:
        _chk(bias_out, "bias_out", (A if bias_from == 1 else Bc,))
        if bias_from not in (1, 2):
            raise ValueError("wgrad: bias_from must be 1 (S) or 2 (L) when bias_out is given")
    elif bias_from:
        raise ValueError("wgrad: bias_from without bias_out")
    lib = L.load()
    need = lib.mg_wgrad_workspace_bytes(A, Bc, K, nb0 + nb1, Ts)
    if work is None:
        work = workspace(need, small.device, "wgrad")
    if work.numel() * work.element_size() < need:
        raise ValueError("wgrad: workspace too small")
    with _observe(lambda: f"wgrad_multi_kernel<{stride},{K}>", 2.0 * (nb0 + nb1) * Ts * A * Bc * K):
        rc = lib.mg_wgrad(_p(small), _p(large), nb0, _p(small2), _p(large2), nb1, _p(out), _p(bias_out), bias_from,
                          Ts, Tl, A, Bc, K, stride, _p(work), work.numel() * work.element_size(), _stream())
    L.check(rc, "mg_wgrad")
    return out


def _wgrad_job(small, large, out, K, stride, small2=None, large2=None, bias_out=None, bias_from=0):
    """Argument checks of wgrad(), as a tuple for wgrad_multi()."""
    _chk(small, "small")
    _chk(large, "large")
    _chk(out, "out")
    nb0, Ts, A = small.shape
    Tl, Bc = large.shape[1], large.shape[2]
    pad = (K - 1) // 2
    if large.shape[0] != nb0 or ((Tl + 2 * pad - K) // stride + 1 != Ts and not (stride == 2 and Tl == 2 * Ts)):
        raise ValueError(f"wgrad: S {tuple(small.shape)} inconsistent with L {tuple(large.shape)}, K={K}, stride={stride}")
    if out.numel() != A * Bc * K:
        raise ValueError(f"wgrad: out numel {out.numel()} != {A * Bc * K}")
    nb1 = 0
    if small2 is not None:
        _chk(small2, "small2")
        _chk(large2, "large2")
        nb1 = small2.shape[0]
        if tuple(small2.shape[1:]) != (Ts, A) or tuple(large2.shape) != (nb1, Tl, Bc):
            raise ValueError("wgrad: segment 1 shape mismatch")
    if bias_out is not None:
        if bias_from not in (1, 2):
            raise ValueError("wgrad: bias_from must be 1 (S) or 2 (L) when bias_out is given")
        _chk(bias_out, "bias_out", (A if bias_from == 1 else Bc,))
    elif bias_from:
        raise ValueError("wgrad: bias_from without bias_out")
    return (small, large, nb0, small2, large2, nb1, out, bias_out, bias_from, Ts, Tl, A, Bc, K, stride)


def wgrad_multi(jobs, tag=""):
    """`tag` names the slab workspace: two wgrad_multi calls that may run CONCURRENTLY (different streams) need different tags.
    Independent weight gradients of one (K, stride), given as _wgrad_job tuples (conv1d_wgrad / convT1d_wgrad /
    linear_wgrad with defer=True return them), in ONE launch (+ one reduce launch if any of them is split over the
    batch): the small layers' gradients are each a few workgroups at the launch floor.  More than MG_MAX_WGRAD_JOBS
    jobs, or jobs of different (K, stride), go out as several launches."""
    lib = L.load()
    groups = {}
    for j in jobs:
        groups.setdefault((j[13], j[14]), []).append(j)
    for (K, stride), js in groups.items():
        for i in range(0, len(js), L.MAX_WGRAD_JOBS):
            part = js[i:i + L.MAX_WGRAD_JOBS]
            arr = (L.WgradJob * len(part))()
            need, flops = 0, 0.0
            for a, (s0, l0, nb0, s1, l1, nb1, out, bo, bf, Ts, Tl, A, Bc, _, _) in zip(arr, part):
                a.s0, a.l0, a.nb0 = s0.data_ptr(), l0.data_ptr(), nb0
                a.s1, a.l1, a.nb1 = (None if s1 is None else s1.data_ptr()), (None if l1 is None else l1.data_ptr()), nb1
                a.out, a.bias_out, a.bias_from = out.data_ptr(), (None if bo is None else bo.data_ptr()), bf
                a.Ts, a.Tl, a.A, a.Bc = Ts, Tl, A, Bc
                need += (lib.mg_wgrad_workspace_bytes(A, Bc, K, nb0 + nb1, Ts) + 255) & ~255
                flops += 2.0 * (nb0 + nb1) * Ts * A * Bc * K
            work = workspace(need, part[0][0].device, "wgrad_multi" + tag)
            with _observe(lambda: f"wgrad_multi_kernel<{stride},{K}>", flops):
                rc = lib.mg_wgrad_multi(arr, len(part), K, stride, _p(work), work.numel() * work.element_size(), _stream())
            L.check(rc, "mg_wgrad_multi")


def conv1d_wgrad(x, dy, dw, stride, x2=None, dy2=None, db=None, defer=False):
    """dw (Cout, Cin, K) for nn.Conv1d: S = dy, L = x; db (Cout) = column sums of dy (segment 0) if given.
    defer=True: nothing is launched, the job is returned for wgrad_multi()."""
    K = dw.shape[2]
    if defer:
        return _wgrad_job(dy, x, dw, K, stride, dy2, x2, db, 1 if db is not None else 0)
    return wgrad(dy, x, dw, K, stride, dy2, x2, bias_out=db, bias_from=1 if db is not None else 0)


def convT1d_wgrad(x, dy, dw, db=None, defer=False):
    """dw (Cin, Cout, 5) for the stride-2 ConvTranspose1d: S = x, L = dy; db (Cout) = column sums of dy if given."""
    if defer:
        return _wgrad_job(x, dy, dw, 5, 2, None, None, db, 2 if db is not None else 0)
    return wgrad(x, dy, dw, 5, 2, bias_out=db, bias_from=2 if db is not None else 0)


def linear_wgrad(x, dy, dw, x2=None, dy2=None, db=None, defer=False):
    """dw (out, in) = dy^T @ x (+ second segment); db (out) = column sums of dy (segment 0) if given."""
    B = x.shape[0]
    s2 = l2 = None
    if x2 is not None:
        s2, l2 = dy2.view(dy2.shape[0], 1, -1), x2.view(x2.shape[0], 1, -1)
    if defer:
        return _wgrad_job(dy.view(B, 1, -1), x.view(B, 1, -1), dw, 1, 1, s2, l2, db, 1 if db is not None else 0)
    return wgrad(dy.view(B, 1, -1), x.view(B, 1, -1), dw, 1, 1, s2, l2, bias_out=db, bias_from=1 if db is not None else 0)


# ---------------------------------------------------------------------------------------
# bf16-storage variant of the frozen emotion-discriminator branch (secondary configuration; see include/melo_gan_hip.h)
# ---------------------------------------------------------------------------------------
BF16 = torch.bfloat16


def wb_relayout(w: Tensor, wb: Tensor, N: int, Cc: int, K: int, w_sn: int, w_sc: int, flip: bool = False) -> Tensor:
    """wb[k][n][c] = bf16(W(n, c, flip ? K-1-k : k)), W(n,c,k) = w[n*w_sn + c*w_sc + k]: the weight image of conv_s1_bf16."""
    _chk(w, "w")
    _chk(wb, "wb", (K, N, Cc), BF16)
    if w.numel() < (N - 1) * w_sn + (Cc - 1) * w_sc + K:
        raise ValueError("wb_relayout: w smaller than its strides imply")
    L.check(L.load().mg_wb_relayout(_p(w), _p(wb), N, Cc, K, w_sn, w_sc, 1 if flip else 0, _stream()), "mg_wb_relayout")
    return wb


def conv_s1_bf16_supported(B: int, T: int, Cin: int, N: int, K: int) -> bool:
    return bool(L.load().mg_conv1d_s1_bf16_supported(B, T, Cin, N, K))


def conv_s1_bf16(x: Tensor, wb: Tensor, y: Tensor, scale=None, shift=None, zout=None, act=ACT_NONE, gref=None,
                 gact=ACT_NONE, gscale=None, accumulate=False) -> Tensor:
    """Stride-1 K-tap convolution, bf16 storage / fp32 accumulate (mg_conv1d_s1_bf16).  x: (B, T, Cin) bf16 or fp32;
    wb: (K, N, Cin) bf16 (wb_relayout); y: (B, T, N) bf16 or fp32; zout / gref: (B, T, N) bf16."""
    if x.dtype not in (BF16, torch.float32) or y.dtype not in (BF16, torch.float32):
        raise ValueError("conv_s1_bf16: x and y must be bf16 or fp32")
    _chk(x, "x", dtype=x.dtype)
    _chk(y, "y", dtype=y.dtype)
    if x.dim() != 3 or y.dim() != 3:
        raise ValueError("x and y must be (B, T, C)")
    B, T, Cin = x.shape
    K, N = wb.shape[0], wb.shape[1]
    _chk(wb, "wb", (K, N, Cin), BF16)
    if tuple(y.shape) != (B, T, N):
        raise ValueError(f"y: expected {(B, T, N)}, got {tuple(y.shape)}")
    lib = L.load()
    if not lib.mg_conv1d_s1_bf16_supported(B, T, Cin, N, K):
        raise ValueError(f"conv_s1_bf16: unsupported shape B={B} T={T} Cin={Cin} N={N} K={K} (T % 128, Cin % 32, N % 64)")
    e = L.EpilogueBf16()
    for nm, v in (("scale", scale), ("shift", shift), ("gscale", gscale)):
        if v is not None:
            _chk(v, nm, (N,))
    for nm, v in (("zout", zout), ("gref", gref)):
        if v is not None:
            _chk(v, nm, (B, T, N), BF16)
    if accumulate and y.dtype != torch.float32:
        raise ValueError("conv_s1_bf16: accumulate needs an fp32 output")
    e.scale, e.shift, e.zout, e.act = _p(scale), _p(shift), _p(zout), act
    e.gref, e.gact, e.gscale, e.accumulate = _p(gref), gact, _p(gscale), 1 if accumulate else 0
    def launch():
        return lib.mg_conv1d_s1_bf16(_p(x), 1 if x.dtype == torch.float32 else 0, _p(wb), _p(y),
                                     1 if y.dtype == torch.float32 else 0, B, T, Cin, N, K, C.byref(e), _stream())
    with _observe(lambda: "conv_bf16_kernel<%d,%s,%s>" % (K, "true" if x.dtype == torch.float32 else "false",
                                                          "true" if y.dtype == torch.float32 else "false"),
                  2.0 * B * T * N * Cin * K, launch):
        rc = launch()
    L.check(rc, "mg_conv1d_s1_bf16")
    return y


def meanT_fwd_bf16(a: Tensor, h: Tensor) -> Tensor:
    _chk(a, "a", dtype=BF16)
    B, T, Cc = a.shape
    _chk(h, "h", (B, Cc))
    L.check(L.load().mg_meanT_fwd_bf16(_p(a), _p(h), B, T, Cc, _stream()), "mg_meanT_fwd_bf16")
    return h


def meanT_bwd_bf16(dh: Tensor, dz: Tensor, gref: Tensor, gact=ACT_NONE, gscale=None) -> Tensor:
    _chk(dz, "dz", dtype=BF16)
    B, T, Cc = dz.shape
    _chk(dh, "dh", (B, Cc))
    _chk(gref, "gref", (B, T, Cc), BF16)
    if gscale is not None:
        _chk(gscale, "gscale", (Cc,))
    L.check(L.load().mg_meanT_bwd_bf16(_p(dh), _p(dz), _p(gref), gact, _p(gscale), B, T, Cc, _stream()), "mg_meanT_bwd_bf16")
    return dz


# ---------------------------------------------------------------------------------------
# reductions / normalisation
# ---------------------------------------------------------------------------------------
def colsum(x: Tensor, out: Tensor, sumsq: Optional[Tensor] = None):
    """out[c] = sum over all leading dims of x[..., c]."""
    _chk(x, "x")
    Cc = x.shape[-1]
    R = x.numel() // Cc
    _chk(out, "out", (Cc,))
    if sumsq is not None:
        _chk(sumsq, "sumsq", (Cc,))
    lib = L.load()
    need = lib.mg_colsum_workspace_bytes(Cc)
    work = workspace(need, x.device, "colsum")
    L.check(lib.mg_colsum(_p(x), R, Cc, _p(out), _p(sumsq), _p(work), work.numel(), _stream()), "mg_colsum")
    return out


def bn_train_fwd(z, a, gamma, beta, running_mean, running_var, save_mean, save_invstd, act=ACT_RELU,
                 momentum=0.1, eps=1e-5, groups=1):
    """groups > 1: z / a stack `groups` independent batches along dim 0 (statistics per group, save_* of shape
    (groups, C), running statistics updated group after group)."""
    _chk(z, "z")
    _chk(a, "a", z.shape)
    Cc = z.shape[-1]
    R = z.numel() // Cc
    if groups < 1 or z.shape[0] % groups:
        raise ValueError(f"bn_train_fwd: {z.shape[0]} batch rows do not split into {groups} groups")
    for nm, v in (("gamma", gamma), ("beta", beta)):
        _chk(v, nm, (Cc,))
    for nm, v in (("save_mean", save_mean), ("save_invstd", save_invstd)):
        _chk(v, nm, (Cc,) if groups == 1 else (groups, Cc))
    if running_mean is not None:
        _chk(running_mean, "running_mean", (Cc,))
        _chk(running_var, "running_var", (Cc,))
    lib = L.load()
    work = workspace(lib.mg_bn_groups_workspace_bytes(Cc, groups), z.device, "bn")
    L.check(lib.mg_bn_train_fwd_groups(_p(z), _p(a), R // groups, Cc, groups, _p(gamma), _p(beta), _p(running_mean),
                                       _p(running_var), momentum, eps, _p(save_mean), _p(save_invstd), act, _p(work),
                                       work.numel(), _stream()), "mg_bn_train_fwd")
    return a


def bn_train_fwd_parts(part, part_rows, groups, z, a, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                       act=ACT_RELU, momentum=0.1, eps=1e-5):
    """bn_train_fwd from the partial column statistics the producing conv16 launch left in `part` (part_rows rows in
    all, part_rows / groups per group): statistics, running statistics and the apply pass in ONE launch."""
    _chk(z, "z")
    _chk(a, "a", z.shape)
    _chk(part, "part")
    Cc = z.shape[-1]
    R = z.numel() // Cc
    if groups < 1 or z.shape[0] % groups or part_rows % groups or part.numel() < 3 * part_rows * Cc:
        raise ValueError("bn_train_fwd_parts: groups / partial rows do not fit")
    for nm, v in (("gamma", gamma), ("beta", beta)):
        _chk(v, nm, (Cc,))
    for nm, v in (("save_mean", save_mean), ("save_invstd", save_invstd)):
        _chk(v, nm, (Cc,) if groups == 1 else (groups, Cc))
    if running_mean is not None:
        _chk(running_mean, "running_mean", (Cc,))
        _chk(running_var, "running_var", (Cc,))
    L.check(L.load().mg_bn_train_fwd_parts(_p(part), part_rows // groups, groups, _p(z), _p(a), R // groups, Cc, _p(gamma),
                                           _p(beta), _p(running_mean), _p(running_var), momentum, eps, _p(save_mean),
                                           _p(save_invstd), act, _stream()), "mg_bn_train_fwd_parts")
    return a


def bn_train_bwd(da, a, z, dz, gamma, save_mean, save_invstd, dgamma, dbeta, act=ACT_RELU, beta=None):
    """beta is needed for act=ACT_GELU only (the derivative is taken at the BN output, recomputed from z)."""
    _chk(da, "da", z.shape)
    _chk(a, "a", z.shape)
    _chk(z, "z")
    _chk(dz, "dz", z.shape)
    Cc = z.shape[-1]
    R = z.numel() // Cc
    for nm, v in (("gamma", gamma), ("save_mean", save_mean), ("save_invstd", save_invstd), ("dgamma", dgamma),
                  ("dbeta", dbeta)):
        _chk(v, nm, (Cc,))
    lib = L.load()
    work = workspace(lib.mg_bn_workspace_bytes(Cc), z.device, "bn")
    if beta is not None:
        _chk(beta, "beta", (Cc,))
    L.check(lib.mg_bn_train_bwd(_p(da), _p(a), _p(z), _p(dz), R, Cc, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                _p(dgamma), _p(dbeta), act, _p(work), work.numel(), _stream()), "mg_bn_train_bwd")
    return dz


def bn_train_bwd_parts(part, part_rows, da, a, z, dz, gamma, save_mean, save_invstd, dgamma, dbeta, act=ACT_RELU):
    """bn_train_bwd behind a conv16 launch that left the two column sums in `part` (conv16(bnb=...)): ONE launch."""
    _chk(part, "part", dtype=torch.float64)
    _chk(da, "da", z.shape)
    _chk(a, "a", z.shape)
    _chk(z, "z")
    _chk(dz, "dz", z.shape)
    Cc = z.shape[-1]
    R = z.numel() // Cc
    if part.numel() < 2 * part_rows * Cc:
        raise ValueError("bn_train_bwd_parts: partial buffer too small")
    for nm, v in (("gamma", gamma), ("save_mean", save_mean), ("save_invstd", save_invstd), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(v, nm, (Cc,))
    L.check(L.load().mg_bn_train_bwd_parts(_p(part), part_rows, _p(da), _p(a), _p(z), _p(dz), R, Cc, _p(gamma), None, _p(save_mean),
                                           _p(save_invstd), _p(dgamma), _p(dbeta), act, _stream()), "mg_bn_train_bwd_parts")
    return dz


def bn_eval_fwd(z, a, gamma, beta, running_mean, running_var, act=ACT_RELU, eps=1e-5):
    _chk(z, "z")
    _chk(a, "a", z.shape)
    Cc = z.shape[-1]
    for nm, v in (("gamma", gamma), ("beta", beta), ("running_mean", running_mean), ("running_var", running_var)):
        _chk(v, nm, (Cc,))
    L.check(L.load().mg_bn_eval_fwd(_p(z), _p(a), z.numel() // Cc, Cc, _p(gamma), _p(beta), _p(running_mean),
                                    _p(running_var), eps, act, _stream()), "mg_bn_eval_fwd")
    return a


def bn_fold(gamma, beta, running_mean, running_var, conv_bias, scale, shift, eps=1e-5):
    Cc = gamma.numel()
    for nm, v in (("gamma", gamma), ("beta", beta), ("running_mean", running_mean), ("running_var", running_var),
                  ("scale", scale), ("shift", shift)):
        _chk(v, nm, (Cc,))
    if conv_bias is not None:
        _chk(conv_bias, "conv_bias", (Cc,))
    L.check(L.load().mg_bn_fold(_p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(conv_bias), eps,
                                _p(scale), _p(shift), Cc, _stream()), "mg_bn_fold")


def meanT_fwd(a, h):
    _chk(a, "a")
    B, T, Cc = a.shape
    _chk(h, "h", (B, Cc))
    L.check(L.load().mg_meanT_fwd(_p(a), _p(h), B, T, Cc, _stream()), "mg_meanT_fwd")
    return h


def meanT_bwd(dh, dz, gref=None, gact=ACT_NONE, gscale=None, mean=None):
    """mean = (src, out, scale): out[0] = scale * mean(src) rides in the same launch (mg_meanT_bwd_mean)."""
    _chk(dz, "dz")
    B, T, Cc = dz.shape
    _chk(dh, "dh", (B, Cc))
    if gref is not None:
        _chk(gref, "gref", dz.shape)
    if gscale is not None:
        _chk(gscale, "gscale", (Cc,))
    msrc = mout = None
    mn, msc = 0, 0.0
    if mean is not None:
        msrc, mout, msc = mean
        _chk(msrc, "mean src")
        _chk(mout, "mean out")
        mn = msrc.numel()
        if mn < 1 or mout.numel() < 1:
            raise ValueError("meanT_bwd: empty mean rider")
    L.check(L.load().mg_meanT_bwd_mean(_p(dh), _p(dz), B, T, Cc, _p(gref), gact, _p(gscale), _p(msrc), _p(mout), mn, float(msc),
                                       _stream()), "mg_meanT_bwd")
    return dz


def layernorm_fwd(x, y, xhat, gamma, beta, eps=1e-5):
    _chk(x, "x")
    B, D = x.shape
    _chk(y, "y", (B, D))
    if xhat is not None:
        _chk(xhat, "xhat", (B, D))
    _chk(gamma, "gamma", (D,))
    _chk(beta, "beta", (D,))
    L.check(L.load().mg_layernorm_fwd(_p(x), _p(y), _p(xhat), B, D, _p(gamma), _p(beta), eps, _stream()),
            "mg_layernorm_fwd")
    return y


def layernorm_bwd_params(dy, xhat, dgamma, dbeta):
    _chk(dy, "dy")
    B, D = dy.shape
    _chk(xhat, "xhat", (B, D))
    _chk(dgamma, "dgamma", (D,))
    _chk(dbeta, "dbeta", (D,))
    L.check(L.load().mg_layernorm_bwd_params(_p(dy), _p(xhat), _p(dgamma), _p(dbeta), B, D, _stream()),
            "mg_layernorm_bwd_params")


# ---------------------------------------------------------------------------------------
# critic head, GP, losses
# ---------------------------------------------------------------------------------------
def dhead_fwd(f, emb, w, bias, s):
    _chk(f, "f")
    B, F = f.shape
    Be, E = (emb.shape if emb is not None else (0, 0))
    if emb is not None:
        _chk(emb, "emb")
        if B % Be:
            raise ValueError("dhead_fwd: B must be a multiple of emb rows")
    if w.numel() != F + E:
        raise ValueError("dhead_fwd: weight size mismatch")
    _chk(w, "w")
    _chk(bias, "bias")
    _chk(s, "s", (B,))
    L.check(L.load().mg_dhead_fwd(_p(f), _p(emb), _p(w), _p(bias), _p(s), B, Be, F, E, _stream()), "mg_dhead_fwd")
    return s


def dhead_fwd_bwd(ds, f, emb, w, bias, s, dU, demb=None, nb_emb=0):
    """dhead_fwd + dhead_bwd in one launch (ds is a constant of the step)."""
    _chk(f, "f")
    B, F = f.shape
    _chk(emb, "emb")
    Be, E = emb.shape
    if B % Be or w.numel() != F + E:
        raise ValueError("dhead_fwd_bwd: shape mismatch")
    _chk(ds, "ds", (B,))
    _chk(w, "w")
    _chk(bias, "bias")
    _chk(s, "s", (B,))
    _chk(dU, "dU", (B, F))
    if demb is not None:
        _chk(demb, "demb", (Be, E))
    L.check(L.load().mg_dhead_fwd_bwd(_p(ds), _p(f), _p(emb), _p(w), _p(bias), _p(s), _p(dU), _p(demb), B, Be, F, E, nb_emb,
                                      _stream()), "mg_dhead_fwd_bwd")


def dhead_bwd(ds, f, w, dU, demb=None, nb_emb=0):
    _chk(f, "f")
    B, F = f.shape
    _chk(ds, "ds", (B,))
    _chk(dU, "dU", (B, F))
    _chk(w, "w")
    Be = E = 0
    if demb is not None:
        _chk(demb, "demb")
        Be, E = demb.shape
        if w.numel() != F + E or nb_emb > B or nb_emb % Be:
            raise ValueError("dhead_bwd: emb shape mismatch")
    L.check(L.load().mg_dhead_bwd(_p(ds), _p(f), _p(w), _p(dU), _p(demb), B, Be, F, E, nb_emb, _stream()),
            "mg_dhead_bwd")


def dhead_wgrad(ds, f, emb, gf, dw, dbias, nb, ng, loss=None):
    """loss = (s, norms, lambda_gp, out, gp_out, nb): the critic's loss scalars (wgan_d_loss with norms) ride in the same launch."""
    _chk(f, "f")
    F = f.shape[1]
    Be, E = (emb.shape if emb is not None else (0, 0))
    if nb > f.shape[0] or nb > ds.numel() or (gf is not None and (ng > gf.shape[0] or gf.shape[1] != F)):
        raise ValueError("dhead_wgrad: row counts exceed tensors")
    _chk(dw, "dw")
    if dw.numel() != F + E:
        raise ValueError("dhead_wgrad: dw size mismatch")
    _chk(dbias, "dbias")
    ls = ln = lo = lg = None
    lam, lnb = 0.0, 0
    if loss is not None:
        ls, ln, lam, lo, lg, lnb = loss
        _chk(ls, "loss s")
        _chk(ln, "loss norms", (lnb,))
        _chk(lg, "gp_out")
        if ls.numel() < 2 * lnb or lo.numel() < 3 or lnb < 1:
            raise ValueError("dhead_wgrad: loss rider sizes")
    L.check(L.load().mg_dhead_wgrad_loss(_p(ds), _p(f), _p(emb), _p(gf), _p(dw), _p(dbias), nb, ng if gf is not None else 0,
                                         Be, F, E, _p(ls), _p(ln), float(lam), _p(lo), _p(lg), lnb, _stream()), "mg_dhead_wgrad")


def gp_interp(real, fake, alpha, xhat):
    _chk(real, "real")
    _chk(fake, "fake", real.shape)
    _chk(xhat, "xhat", real.shape)
    B = real.shape[0]
    _chk(alpha, "alpha")
    if alpha.numel() != B:
        raise ValueError("gp_interp: alpha must have B elements")
    L.check(L.load().mg_gp_interp(_p(real), _p(fake), _p(alpha), _p(xhat), B, real.numel() // B, _stream()),
            "mg_gp_interp")
    return xhat


def gp_penalty(g, gbar, norms, gp, coef):
    _chk(g, "g")
    B = g.shape[0]
    if gbar is not None:
        _chk(gbar, "gbar", g.shape)
    _chk(norms, "norms", (B,))
    if gp is not None:           # None: the mean is left to wgan_d_loss(..., norms=...)
        _chk(gp, "gp")
    L.check(L.load().mg_gp_penalty(_p(g), _p(gbar), _p(norms), _p(gp), coef, B, g.numel() // B, _stream()),
            "mg_gp_penalty")


def wgan_d_loss(s, gp, lambda_gp, out, nb, norms=None):
    """out = {loss_d, mean_real, mean_fake}.  With `norms` (per-sample gradient norms) the penalty mean((norm-1)^2) is
    computed here and written to gp; otherwise gp is read."""
    _chk(s, "s")
    if s.numel() < 2 * nb or out.numel() < 3:
        raise ValueError("wgan_d_loss: sizes")
    if norms is not None:
        _chk(norms, "norms", (nb,))
        L.check(L.load().mg_wgan_d_loss_gp(_p(s), _p(norms), lambda_gp, _p(out), _p(gp), nb, _stream()), "mg_wgan_d_loss_gp")
        return
    L.check(L.load().mg_wgan_d_loss(_p(s), _p(gp), lambda_gp, _p(out), nb, _stream()), "mg_wgan_d_loss")


def softmax_ce(logits, target, loss, dlogits, coef=1.0):
    _chk(logits, "logits")
    B, Cc = logits.shape
    _chk(target, "target", (B,), torch.int64)
    if dlogits is not None:
        _chk(dlogits, "dlogits", (B, Cc))
    L.check(L.load().mg_softmax_ce(_p(logits), _p(target), _p(loss), _p(dlogits), coef, B, Cc, _stream()),
            "mg_softmax_ce")


def neg_mean(s, out):
    _chk(s, "s")
    L.check(L.load().mg_neg_mean(_p(s), _p(out), s.numel(), _stream()), "mg_neg_mean")


# ---------------------------------------------------------------------------------------
# elementwise / optimiser
# ---------------------------------------------------------------------------------------
def fill(x, v):
    _chk(x, "x")
    L.check(L.load().mg_fill(_p(x), v, x.numel(), _stream()), "mg_fill")


def axpby(x, y, a=1.0, b=0.0):
    _chk(x, "x")
    _chk(y, "y")
    if x.numel() != y.numel():
        raise ValueError("axpby: size mismatch")
    L.check(L.load().mg_axpby(_p(x), _p(y), a, b, x.numel(), _stream()), "mg_axpby")


def copy_cols(src, soff, dst, doff, ncols, accumulate=False):
    _chk(src, "src")
    _chk(dst, "dst")
    if src.dim() != 2 or dst.dim() != 2 or src.shape[0] != dst.shape[0]:
        raise ValueError("copy_cols: need 2-D tensors with equal rows")
    if soff + ncols > src.shape[1] or doff + ncols > dst.shape[1]:
        raise ValueError("copy_cols: column range out of bounds")
    L.check(L.load().mg_copy_cols(_p(src), src.shape[1], soff, _p(dst), dst.shape[1], doff, src.shape[0], ncols,
                                  1 if accumulate else 0, _stream()), "mg_copy_cols")


def stage_rows(jobs, n_rows):
    """Batch staging in ONE launch: for every (src, dst, idx[, rows]) in `jobs`, dst[r] = src[idx[r] if idx is not None else r]
    for r < rows (default n_rows, the launch's row count).  src/dst: contiguous device tensors of one dtype whose rows (dim 0) have equal byte size; idx: int64
    device tensor of n_rows entries (clamped into the source on the device)."""
    if not 0 < len(jobs) <= L.MAX_STAGE_JOBS:
        raise ValueError(f"stage_rows: 1..{L.MAX_STAGE_JOBS} jobs")
    arr = (L.StageJob * len(jobs))()
    for a, job in zip(arr, jobs):
        src, dst, idx = job[:3]
        jrows = job[3] if len(job) > 3 else n_rows
        if not 0 < jrows <= n_rows:
            raise ValueError("stage_rows: a job's row count must be in 1..n_rows")
        if not isinstance(src, torch.Tensor) or not src.is_cuda or not src.is_contiguous() or src.dim() < 1:
            raise ValueError("stage_rows: src must be a contiguous device tensor")
        # dst: dense, or a column block of a wider 2-D matrix (rows contiguous, a row pitch between them)
        dense = isinstance(dst, torch.Tensor) and dst.is_cuda and dst.dim() >= 1 and dst.is_contiguous()
        block = (isinstance(dst, torch.Tensor) and dst.is_cuda and dst.dim() == 2 and dst.stride(1) == 1 and
                 dst.stride(0) >= dst.shape[1])
        if not (dense or block):
            raise ValueError("stage_rows: dst must be a contiguous device tensor or a column block of one")
        if src.dtype != dst.dtype or src.shape[1:] != dst.shape[1:] or dst.shape[0] < jrows:
            raise ValueError(f"stage_rows: row mismatch {tuple(src.shape)} {src.dtype} -> {tuple(dst.shape)} {dst.dtype}")
        if idx is not None:
            _chk(idx, "idx", (jrows,), torch.int64)
        elif src.shape[0] < jrows:
            raise ValueError("stage_rows: source has fewer rows than the batch")
        row_bytes = src.element_size() * (src[0].numel() if src.dim() > 1 else 1)
        a.src, a.dst, a.idx = src.data_ptr(), dst.data_ptr(), (None if idx is None else idx.data_ptr())
        a.row_bytes, a.src_rows = row_bytes, src.shape[0]
        a.dst_pitch = 0 if dense else dst.stride(0) * dst.element_size()
        a.rows = 0 if jrows == n_rows else jrows
    L.check(L.load().mg_stage_rows(arr, len(jobs), n_rows, _stream()), "mg_stage_rows")


def stage_rows_cursor(jobs, n_rows, order, order_len, counter, base):
    """stage_rows with the source rows picked on the device (mg_stage_rows_cursor): for every (src, dst) in `jobs`,
    dst[r] = src[order[p]] (order None: src[p]) with p = ((counter - base) * n_rows + r) % order_len.  counter / base: int64
    device scalars (1,)."""
    if not 0 < len(jobs) <= L.MAX_STAGE_JOBS:
        raise ValueError(f"stage_rows_cursor: 1..{L.MAX_STAGE_JOBS} jobs")
    _chk(counter, "counter", (1,), torch.int64)
    _chk(base, "base", (1,), torch.int64)
    if order is not None:
        _chk(order, "order", dtype=torch.int64)
        if order.numel() < order_len:
            raise ValueError("stage_rows_cursor: order shorter than order_len")
    arr = _cursor_jobs(jobs, n_rows, order, order_len)
    L.check(L.load().mg_stage_rows_cursor(arr, len(jobs), n_rows, _p(order), int(order_len), _p(counter), _p(base), _stream()),
            "mg_stage_rows_cursor")


def _cursor_jobs(jobs, n_rows, order, order_len):
    arr = (L.StageJob * len(jobs))()
    for a, (src, dst) in zip(arr, jobs):
        if not (isinstance(src, torch.Tensor) and src.is_cuda and src.is_contiguous() and src.dim() >= 1):
            raise ValueError("stage_rows_cursor: src must be a contiguous device tensor")
        if not (isinstance(dst, torch.Tensor) and dst.is_cuda and dst.is_contiguous() and dst.dim() >= 1):
            raise ValueError("stage_rows_cursor: dst must be a contiguous device tensor")
        if src.dtype != dst.dtype or src.shape[1:] != dst.shape[1:] or dst.shape[0] < n_rows:
            raise ValueError(f"stage_rows_cursor: row mismatch {tuple(src.shape)} {src.dtype} -> {tuple(dst.shape)} {dst.dtype}")
        if order is None and src.shape[0] < order_len:
            raise ValueError("stage_rows_cursor: without an order the source must hold order_len rows")
        a.src, a.dst, a.idx = src.data_ptr(), dst.data_ptr(), None
        a.row_bytes, a.src_rows = src.element_size() * (src[0].numel() if src.dim() > 1 else 1), src.shape[0]
        a.dst_pitch, a.rows = 0, 0
    return arr


def transpose_bcl_blc(x, y, gref=None, gact=ACT_NONE):
    """y[b, l, c] = x[b, c, l] (* act'(gref[b, l, c]) if gref is given: the activation backward behind the view)."""
    _chk(x, "x")
    B, Cc, Ln = x.shape
    _chk(y, "y", (B, Ln, Cc))
    if gref is not None:
        _chk(gref, "gref")
        if gref.numel() != y.numel():
            raise ValueError("transpose_bcl_blc: gref size mismatch")
    L.check(L.load().mg_transpose_bcl_blc(_p(x), _p(y), B, Cc, Ln, _p(gref), gact, _stream()), "mg_transpose_bcl_blc")
    return y


def act_bwd(dy, dx, gref=None, gact=ACT_NONE, emul=None):
    _chk(dy, "dy")
    _chk(dx, "dx")
    n = dy.numel()
    for v in (dx, gref, emul):
        if v is not None and v.numel() != n:
            raise ValueError("act_bwd: size mismatch")
    L.check(L.load().mg_act_bwd(_p(dy), _p(gref), gact, _p(emul), _p(dx), n, _stream()), "mg_act_bwd")


def rng_fill(normal, uniform, mask0, mask1, p_drop, seed, step_counter, tick_state=None, betas=None, tick_state2=None, stage=None):
    """normal ~ N(0,1), uniform ~ U(0,1), mask* = keep-mask/(1-p_drop); any of them may be None.  Plain: draw, then
    advance step_counter (two launches).  With tick_state (an optimiser's Adam state) and betas: ONE launch that
    draws and advances that Adam state instead; adam_flat(..., ticked_rng_step=step_counter) later advances the counter."""
    for t in (normal, uniform, mask0, mask1):
        if t is not None:
            _chk(t, "rng tensor")
    _chk(step_counter, "step_counter", (1,), torch.int64)
    n = lambda t: 0 if t is None else t.numel()  # noqa: E731
    if stage is not None and tick_state2 is None:
        raise ValueError("rng_fill(stage=...): only with the fused draw (tick_state and tick_state2)")
    if tick_state2 is not None:          # one draw in front of two updates: both Adam states advance here
        _chk(tick_state, "tick_state", (4,), torch.float64)
        _chk(tick_state2, "tick_state2", (4,), torch.float64)
        if stage is not None:
            # stage = (jobs, n_rows, order, order_len, base): stage_rows_cursor with counter = step_counter, riding in this launch
            jobs, n_rows, order, order_len, base = stage
            if not 0 < len(jobs) <= L.MAX_STAGE_JOBS:
                raise ValueError(f"rng_fill(stage=...): 1..{L.MAX_STAGE_JOBS} jobs")
            _chk(base, "base", (1,), torch.int64)
            if order is not None:
                _chk(order, "order", dtype=torch.int64)
                if order.numel() < order_len:
                    raise ValueError("rng_fill(stage=...): order shorter than order_len")
            arr = _cursor_jobs(jobs, n_rows, order, order_len)
            L.check(L.load().mg_rng_fill_tick2_stage(_p(normal), n(normal), _p(uniform), n(uniform), _p(mask0), n(mask0), _p(mask1),
                                                     n(mask1), p_drop, seed & 0xFFFFFFFFFFFFFFFF, _p(step_counter), _p(tick_state),
                                                     _p(tick_state2), betas[0], betas[1], arr, len(jobs), n_rows, _p(order),
                                                     int(order_len), _p(base), _stream()), "mg_rng_fill_tick2_stage")
            return
        L.check(L.load().mg_rng_fill_tick2(_p(normal), n(normal), _p(uniform), n(uniform), _p(mask0), n(mask0), _p(mask1),
                                           n(mask1), p_drop, seed & 0xFFFFFFFFFFFFFFFF, _p(step_counter), _p(tick_state),
                                           _p(tick_state2), betas[0], betas[1], _stream()), "mg_rng_fill_tick2")
        return
    if tick_state is not None:
        _chk(tick_state, "tick_state", (4,), torch.float64)
        L.check(L.load().mg_rng_fill_tick(_p(normal), n(normal), _p(uniform), n(uniform), _p(mask0), n(mask0), _p(mask1),
                                          n(mask1), p_drop, seed & 0xFFFFFFFFFFFFFFFF, _p(step_counter), _p(tick_state),
                                          betas[0], betas[1], _stream()), "mg_rng_fill_tick")
        return
    L.check(L.load().mg_rng_fill(_p(normal), n(normal), _p(uniform), n(uniform), _p(mask0), n(mask0), _p(mask1),
                                 n(mask1), p_drop, seed & 0xFFFFFFFFFFFFFFFF, _p(step_counter), _stream()), "mg_rng_fill")


def wq_table(entries):
    """ctypes table for adam_flat(wq=...): entries = [(start, N, Cc, K, w_sn, w_sc, dst tensor), ...] (mg_wq_entry)."""
    if len(entries) > L.MAX_WQ_ENTRIES:
        raise ValueError(f"wq_table: at most {L.MAX_WQ_ENTRIES} entries")
    arr = (L.WqEntry * max(1, len(entries)))()
    for a, (start, N, Cc, K, sn, sc, dst) in zip(arr, entries):
        _chk(dst, "wq dst")
        if dst.numel() != N * Cc * K:
            raise ValueError("wq_table: dst size mismatch")
        a.start, a.N, a.Cc, a.K, a.w_sn, a.w_sc, a.dst = start, N, Cc, K, sn, sc, dst.data_ptr()
    return arr, len(entries), [e[6] for e in entries]          # keeps the destinations alive


def adam_flat(p, g, m, v, state, lr, beta1, beta2, eps=1e-8, weight_decay=0.0, grad_scale=1.0, gs_dev=None,
              ticked_rng_step=None, ticked=False, wq=None):
    """Fused flat Adam/AdamW.  ticked_rng_step: `state` was already advanced by rng_fill(tick_state=state); apply the
    update only and advance that Philox step counter (one launch instead of two).  ticked=True without a counter: the
    state was advanced by the draw, and another update advances the counter (rng_fill(tick_state2=...))."""
    n = p.numel()
    for nm, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, nm)
        if t.numel() != n:
            raise ValueError("adam_flat: size mismatch")
    _chk(state, "state", (4,), torch.float64)
    if wq is not None:             # the update also refreshes WQ-layout weight copies (wq_table)
        if ticked_rng_step is not None:
            _chk(ticked_rng_step, "ticked_rng_step", (1,), torch.int64)
        L.check(L.load().mg_adam_flat_wq(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, _p(state),
                                         grad_scale, _p(gs_dev), 1 if (ticked or ticked_rng_step is not None) else 0,
                                         _p(ticked_rng_step), wq[0], wq[1], _stream()), "mg_adam_flat_wq")
        return
    if ticked_rng_step is not None or ticked:
        if ticked_rng_step is not None:
            _chk(ticked_rng_step, "ticked_rng_step", (1,), torch.int64)
        L.check(L.load().mg_adam_flat_ticked(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, _p(state),
                                             grad_scale, _p(gs_dev), _p(ticked_rng_step), _stream()), "mg_adam_flat_ticked")
        return
    L.check(L.load().mg_adam_flat(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, _p(state),
                                  grad_scale, _p(gs_dev), _stream()), "mg_adam_flat")


def grad_norm_clip(g, max_norm, out):
    _chk(g, "g")
    _chk(out, "out")
    lib = L.load()
    work = workspace(lib.mg_grad_norm_workspace_bytes(g.numel()), g.device, "gnorm")
    L.check(lib.mg_grad_norm_clip(_p(g), g.numel(), max_norm, _p(out), _p(work), work.numel(), _stream()),
            "mg_grad_norm_clip")


def reparam_fwd(mu, logvar, eps, z):
    for t in (mu, logvar, eps, z):
        _chk(t, "t", mu.shape)
    L.check(L.load().mg_reparam_fwd(_p(mu), _p(logvar), _p(eps), _p(z), mu.numel(), _stream()), "mg_reparam_fwd")


def reparam_bwd(dz, logvar, eps, dmu_kld, dlv_kld, dmu, dlv):
    for t in (dz, logvar, eps, dmu_kld, dlv_kld, dmu, dlv):
        _chk(t, "t", dz.shape)
    L.check(L.load().mg_reparam_bwd(_p(dz), _p(logvar), _p(eps), _p(dmu_kld), _p(dlv_kld), _p(dmu), _p(dlv),
                                    dz.numel(), _stream()), "mg_reparam_bwd")


def vae_loss(recon, x, mu, logvar, beta, out, drecon=None, dmu=None, dlv=None):
    _chk(recon, "recon")
    _chk(x, "x", recon.shape)
    _chk(mu, "mu")
    _chk(logvar, "logvar", mu.shape)
    lib = L.load()
    work = workspace(lib.mg_vae_loss_workspace_bytes(), recon.device, "vae_loss")
    L.check(lib.mg_vae_loss(_p(recon), _p(x), x.numel(), _p(mu), _p(logvar), mu.numel(), beta, _p(out),
                            _p(drecon), _p(dmu), _p(dlv), _p(work), work.numel(), _stream()), "mg_vae_loss")


# ---------------------------------------------------------------------------------------
# hipGraph capture + events
# ---------------------------------------------------------------------------------------
# Graphs and events still alive at interpreter exit: destroyed from an atexit handler, i.e. while the HIP runtime is still
# up.  Left to __del__ during interpreter shutdown they may be destroyed after the runtime's own static teardown has begun --
# a segmentation fault at exit (seen once in ~60 bench runs: the JSON line was out, the exit code was not 0).
import atexit
import weakref

_live_handles = weakref.WeakSet()


def _release_all():
    for obj in list(_live_handles):
        try:
            obj.release()
        except Exception:
            pass


atexit.register(_release_all)


class Graph:
    """hipGraph captured from whatever is enqueued on PyTorch's current stream between begin() and end(), instantiated
    `execs` times (MELO_GRAPH_EXECS, default 1): launch() goes round the executables.  (Measured: alternating between two or
    three executables of the forked step graph is SLOWER than replaying one -- 0.879 / 0.897 / 0.906 ms per step -- so the
    default stays 1; the knob remains for experiments.)"""

    def __init__(self, execs: Optional[int] = None):
        import os
        self.n = max(1, min(8, int(execs if execs is not None else os.environ.get("MELO_GRAPH_EXECS", "1"))))
        self.handles = (C.c_void_p * self.n)()
        self._next = 0
        _live_handles.add(self)

    def begin(self):
        L.check(L.load().mg_graph_begin(_stream()), "mg_graph_begin")

    def end(self):
        L.check(L.load().mg_graph_end_n(_stream(), self.handles, self.n), "mg_graph_end_n")
        self.kernel_nodes = L.load().mg_graph_last_kernel_nodes()      # launches one replay stands for

    def launch(self):
        h = self.handles[self._next]
        self._next = (self._next + 1) % self.n
        L.check(L.load().mg_graph_launch(h, _stream()), "mg_graph_launch")

    def release(self):
        for i in range(self.n):
            if self.handles[i]:
                L.load().mg_graph_destroy(self.handles[i])
                self.handles[i] = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        L.check(L.load().mg_event_create(C.byref(self.h)), "mg_event_create")
        _live_handles.add(self)

    def record(self):
        L.check(L.load().mg_event_record(self.h, _stream()), "mg_event_record")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        L.check(L.load().mg_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "mg_event_elapsed_ms")
        return ms.value

    def release(self):
        if self.h:
            L.load().mg_event_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
