"""Seeded shape sweep of the window-GEMM / weight-gradient kernels against PyTorch's own GPU convolutions (an independent
implementation -- MIOpen / rocBLAS -- on the same device): ragged time lengths, channel counts that are and are not
multiples of the 16-channel chunk and the 64-column tile, batch tails, both strides, with and without split-K."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-4):
    err = (a.detach().double() - b.detach().double()).norm() / (b.detach().double().norm() + 1e-30)
    assert float(err) < tol, float(err)


def cases(seed, n):
    g = torch.Generator().manual_seed(seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]  # noqa: E731
    out = []
    for _ in range(n):
        out.append(dict(B=pick([1, 2, 3, 5, 8, 17, 64]), T=pick([4, 7, 16, 20, 31, 32, 48, 64, 100, 128, 256]),
                        Cin=pick([4, 16, 24, 32, 64, 96, 128, 256]), Cout=pick([4, 32, 64, 96, 128, 192, 256]),
                        K=pick([3, 5]), stride=pick([1, 2])))
    return out


@pytest.mark.parametrize("c", cases(1234, 40), ids=lambda c: "B{B}_T{T}_{Cin}to{Cout}_k{K}_s{stride}".format(**c))
def test_conv1d_fwd_dgrad_wgrad_vs_torch_gpu(c):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    B, T, Cin, Cout, K, stride = c["B"], c["T"], c["Cin"], c["Cout"], c["K"], c["stride"]
    if stride == 2 and K != 5:
        K = 5
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B, T, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, K, device="cuda", generator=g) / math.sqrt(Cin * K)
    bias = torch.randn(Cout, device="cuda", generator=g) * 0.1
    pad = K // 2
    Tout = (T + 2 * pad - K) // stride + 1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv1d(xr.permute(0, 2, 1), wr, bias, stride, pad).permute(0, 2, 1).contiguous()
    dy = torch.randn(B, Tout, Cout, device="cuda", generator=g)
    dxr, dwr = torch.autograd.grad(yr, (xr, wr), dy)
    y = torch.empty(B, Tout, Cout, device="cuda")
    ops.conv1d_fwd(x, w, y, stride, bias=bias)
    close(y, yr)
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.conv1d_dgrad(dy, w, dx, stride)
    close(dx, dxr)
    dw, db = torch.empty_like(w), torch.empty(Cout, device="cuda")
    ops.conv1d_wgrad(x, dy, dw, stride, db=db)
    close(dw, dwr)
    close(db, dy.sum((0, 1)))


@pytest.mark.parametrize("c", cases(99, 24), ids=lambda c: "B{B}_T{T}_{Cin}to{Cout}".format(**c))
def test_convT1d_fwd_dgrad_wgrad_vs_torch_gpu(c):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    B, T, Cin, Cout = c["B"], c["T"], c["Cin"], c["Cout"]
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(B, T, Cin, device="cuda", generator=g)
    w = torch.randn(Cin, Cout, 5, device="cuda", generator=g) / math.sqrt(Cin * 2.5)
    bias = torch.randn(Cout, device="cuda", generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose1d(xr.permute(0, 2, 1), wr, bias, 2, 2, 1).permute(0, 2, 1).contiguous()
    dy = torch.randn(B, 2 * T, Cout, device="cuda", generator=g)
    dxr, dwr = torch.autograd.grad(yr, (xr, wr), dy)
    y = torch.empty(B, 2 * T, Cout, device="cuda")
    ops.convT1d_fwd(x, w, y, bias=bias)
    close(y, yr)
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.convT1d_dgrad(dy, w, dx)
    close(dx, dxr)
    dw, db = torch.empty_like(w), torch.empty(Cout, device="cuda")
    ops.convT1d_wgrad(x, dy, dw, db=db)
    close(dw, dwr)
    close(db, dy.sum((0, 1)))


def lin_cases(seed, n):
    g = torch.Generator().manual_seed(seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]  # noqa: E731
    return [dict(M=pick([1, 4, 7, 64, 65, 192, 300, 513, 600, 1024]), K=pick([6, 8, 64, 100, 128, 256, 384, 512, 1000]),
                 N=pick([1, 4, 31, 64, 128, 130, 256, 512, 2048]), act=pick([0, 1, 2, 3, 4]), epi=pick(["bias", "gref", "emul+acc", "zout"]))
            for _ in range(n)]


@pytest.mark.parametrize("c", lin_cases(5, 40), ids=lambda c: "M{M}_K{K}_N{N}_act{act}_{epi}".format(**c))
def test_linear_epilogues_vs_torch_gpu(c):
    """Linear forward / data-gradient through both kernels (skinny GEMM for <= 512 rows, K=1 window GEMM above) with the
    fused epilogue pieces, against torch GPU matmuls and torch activations."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    M, K, N, act, epi = c["M"], c["K"], c["N"], c["act"], c["epi"]
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    actf = [lambda t: t, torch.relu, lambda t: F.leaky_relu(t, 0.2), F.gelu, torch.tanh][act]
    z = x @ w.t() + bias
    if epi == "bias":
        y = torch.empty(M, N, device="cuda")
        ops.linear_fwd(x, w, y, bias=bias, act=act)
        close(y, actf(z))
    elif epi == "zout":
        y, zo = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        ops.linear_fwd(x, w, y, bias=bias, act=act, zout=zo)
        close(zo, z)
        close(y, actf(z))
    elif epi == "gref":          # data-gradient with the activation derivative of the producer layer fused in
        dy = torch.randn(M, N, device="cuda", generator=g)
        ref = torch.randn(M, K, device="cuda", generator=g)
        rr = ref.clone().requires_grad_(True)
        deriv = torch.autograd.grad(actf(rr).sum(), rr)[0] if act != 4 else (1 - ref * ref)   # tanh': from the output
        dx = torch.empty(M, K, device="cuda")
        ops.linear_dgrad(dy, w, dx, gref=ref, gact=act)
        close(dx, (dy @ w) * deriv)
    else:
        mask = (torch.rand(M, N, device="cuda", generator=g) > 0.3).float() * 1.25
        y0 = torch.randn(M, N, device="cuda", generator=g)
        y = y0.clone()
        ops.linear_fwd(x, w, y, bias=bias, act=act, emul=mask, accumulate=True)
        close(y, y0 + actf(z) * mask)
