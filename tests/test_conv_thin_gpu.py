"""The 4-channel first / last layers (NOTE_DIM = 4, config/gan_config.yaml:43-44) run on the VALU kernels of
csrc/conv_thin.hip behind mg_conv1d_gather / mg_conv1d_scatter2.  Checked against PyTorch's CPU convolutions in fp64
(an independent implementation), every fused epilogue the engines use on these layers, ragged lengths and batch tails,
and bit-for-bit agreement of repeated launches."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import importlib
    return importlib.import_module("melo-gan_amd.ops")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def dev(t):
    return t.float().cuda().contiguous()


def close(got, ref, tol=3e-6):
    ref = ref.double()
    err = (got.cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30)
    assert float(err) < tol, float(err)


# ---- thin reduction side: Conv1d forward (stride 1 / 2) and ConvTranspose1d data gradient ---------------------------
@pytest.mark.parametrize("B,T,Cin,Cout,K,stride", [
    (3, 512, 4, 64, 5, 2), (2, 20, 4, 64, 5, 2), (5, 21, 4, 64, 5, 2), (1, 5, 4, 32, 5, 2), (3, 64, 4, 64, 5, 1),
    (2, 33, 4, 96, 5, 1), (2, 16, 8, 128, 5, 2), (2, 16, 3, 6, 5, 1), (2, 17, 4, 64, 3, 1), (33, 4, 4, 64, 5, 2),
])
def test_thin_in_conv_fwd_epilogues(ops, B, T, Cin, Cout, K, stride):
    x, w, b = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, K, seed=2, scale=1 / math.sqrt(Cin * K)), rnd(Cout, seed=3, scale=0.1)
    z = F.conv1d(x.permute(0, 2, 1), w, b, stride, K // 2).permute(0, 2, 1).contiguous()
    Tout = z.shape[1]
    xd, wd, bd = dev(x), dev(w), dev(b)
    # bias + LeakyReLU (critic conv.0, src/gan/models.py:129-131)
    y = torch.empty(B, Tout, Cout, device="cuda")
    ops.conv1d_fwd(xd, wd, y, stride, bias=bd, act=ops.ACT_LRELU)
    close(y, F.leaky_relu(z, 0.2))
    y2 = torch.empty_like(y)
    ops.conv1d_fwd(xd, wd, y2, stride, bias=bd, act=ops.ACT_LRELU)
    assert torch.equal(y, y2)
    # folded BatchNorm + saved pre-activation + GELU (emotion discriminator conv0, eval mode)
    sc, sh = rnd(Cout, seed=4).abs() + 0.5, rnd(Cout, seed=5)
    zo = torch.empty_like(y)
    ops.conv1d_fwd(xd, wd, y, stride, scale=dev(sc), shift=dev(sh), zout=zo, act=ops.ACT_GELU)
    zz = (z - b) * sc + sh
    close(zo, zz)
    close(y, F.gelu(zz))
    # tangent pass: no bias, multiplied by LeakyReLU'(saved activation) (gradient penalty double backward)
    ref = rnd(B, Tout, Cout, seed=6)
    ops.conv1d_fwd(xd, wd, y, stride, gref=dev(ref), gact=ops.ACT_LRELU)
    close(y, (z - b) * torch.where(ref > 0, 1.0, 0.2))
    # accumulate
    base = rnd(B, Tout, Cout, seed=7)
    y = dev(base)
    ops.conv1d_fwd(xd, wd, y, stride, accumulate=True)
    close(y, base + (z - b))


@pytest.mark.parametrize("B,T,Cin,Cout", [(3, 256, 64, 4), (2, 9, 32, 4), (5, 16, 128, 8), (1, 4, 16, 3)])
def test_thin_in_convT_dgrad(ops, B, T, Cin, Cout):
    """d/dx of ConvTranspose1d(Cin -> Cout = 4): a stride-2 gather over the 4-channel gradient."""
    w, dy = rnd(Cin, Cout, 5, seed=2, scale=0.2), rnd(B, 2 * T, Cout, seed=3)
    x = rnd(B, Cin, T, seed=1).requires_grad_(True)
    yr = F.conv_transpose1d(x, w, None, 2, 2, 1)
    dx_ref, = torch.autograd.grad(yr, x, dy.permute(0, 2, 1))
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.convT1d_dgrad(dev(dy), dev(w), dx)
    close(dx, dx_ref.permute(0, 2, 1))


# ---- thin output side: ConvTranspose1d forward, Conv1d data gradients ------------------------------------------------
@pytest.mark.parametrize("B,T,Cin,Cout", [(3, 256, 64, 4), (2, 9, 32, 4), (5, 16, 128, 8), (2, 8, 256, 4), (1, 4, 16, 3),
                                          (37, 5, 64, 4)])
def test_thin_out_convT_fwd(ops, B, T, Cin, Cout):
    x, w, b = rnd(B, T, Cin, seed=1), rnd(Cin, Cout, 5, seed=2, scale=1 / math.sqrt(Cin * 2.5)), rnd(Cout, seed=3, scale=0.1)
    yr = F.conv_transpose1d(x.permute(0, 2, 1), w, b, 2, 2, 1).permute(0, 2, 1).contiguous()
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty(B, 2 * T, Cout, device="cuda")
    ops.convT1d_fwd(xd, wd, y, bias=bd)                      # generator deconv.6 (src/gan/models.py:67-70)
    close(y, yr)
    ops.convT1d_fwd(xd, wd, y, bias=bd, act=ops.ACT_TANH)    # VAE decoder's last layer (src/ae/model.py)
    close(y, torch.tanh(yr))
    y2 = torch.empty_like(y)
    ops.convT1d_fwd(xd, wd, y2, bias=bd, act=ops.ACT_TANH)
    assert torch.equal(y, y2)
    # zero-pad branch: 2*T rows into a longer buffer
    ypad = torch.zeros(B, 2 * T + 3, Cout, device="cuda")
    ops.convT1d_fwd(xd, wd, ypad, bias=bd)
    close(ypad[:, :2 * T], yr)
    assert float(ypad[:, 2 * T:].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,Cin,Cout,K,stride", [
    (3, 512, 4, 64, 5, 2), (2, 20, 4, 64, 5, 2), (5, 21, 4, 64, 5, 2), (2, 7, 4, 32, 5, 2), (3, 64, 4, 64, 5, 1),
    (2, 33, 4, 128, 5, 1), (2, 16, 8, 256, 5, 2), (2, 17, 4, 64, 3, 1), (2, 16, 3, 16, 5, 1),
])
def test_thin_out_conv_dgrad(ops, B, T, Cin, Cout, K, stride):
    w = rnd(Cout, Cin, K, seed=2, scale=1 / math.sqrt(Cout * K))
    x = rnd(B, Cin, T, seed=1).requires_grad_(True)
    z = F.conv1d(x, w, None, stride, K // 2)
    dz = rnd(B, z.shape[2], Cout, seed=3)
    dx_ref, = torch.autograd.grad(z, x, dz.permute(0, 2, 1))
    dx_ref = dx_ref.permute(0, 2, 1)
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.conv1d_dgrad(dev(dz), dev(w), dx, stride)
    close(dx, dx_ref)
    base = rnd(B, T, Cin, seed=4)
    dx = dev(base)
    ops.conv1d_dgrad(dev(dz), dev(w), dx, stride, accumulate=True)   # dnotes += critic branch (engine g_backward)
    close(dx, base + dx_ref)


def test_thin_kernels_are_the_ones_launched(ops):
    """The route is the product path, not a fallback: the launch observer must report the thin kernels' symbols."""
    seen = []
    x, w = dev(rnd(2, 32, 4, seed=1)), dev(rnd(64, 4, 5, seed=2))
    y = torch.empty(2, 16, 64, device="cuda")
    import contextlib

    def hook(sym, flops, launch=None):
        seen.append(sym)
        return contextlib.nullcontext()

    ops.set_launch_hook(hook)
    try:
        ops.conv1d_fwd(x, w, y, 2)
        dx = torch.empty(2, 32, 4, device="cuda")
        ops.conv1d_dgrad(y, w, dx, 2)
    finally:
        ops.set_launch_hook(None)
    assert any("thin_in" in s for s in seen) and any("thin_out" in s for s in seen), seen
