"""mg_conv1d_wino3 (csrc/conv_wino.hip): the stride-1 three-tap convolution by minimal filtering F(2,3), against fp64 torch
convolutions and against the direct window-GEMM kernel it replaces in the frozen emotion discriminator's branch
(src/emotion_discriminator/ed_model.py:24-46 inside src/gan/train_gan.py:228-236)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    return ops


def _ref_fwd(x, w):
    # x (B,T,Cin), w (N,Cin,3) -> (B,T,N), fp64
    return F.conv1d(x.double().transpose(1, 2), w.double(), padding=1).transpose(1, 2)


def _rel(a, ref):
    return ((a.double() - ref).norm() / ref.norm()).item()


@pytest.mark.parametrize("B,T,Cin,N", [(2, 256, 64, 128), (3, 128, 128, 256), (2, 256, 256, 256), (1, 2, 16, 64), (2, 130, 32, 64),
                                       (5, 6, 16, 64), (1, 384, 48, 192)])
def test_forward_matches_fp64_and_the_direct_kernel(B, T, Cin, N):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + T + Cin + N)
    x = torch.randn(B, T, Cin, generator=g).cuda()
    w = (torch.randn(N, Cin, 3, generator=g) / (3 * Cin) ** 0.5).cuda()
    assert ops.wino3_supported(B, T, Cin, N)
    wt = ops.wino3_weights(w, N, Cin, 3 * Cin, 3)
    y = torch.full((B, T, N), float("nan"), device="cuda")
    ops.conv_wino3(x, wt, y)
    ref = _ref_fwd(x, w)
    yd = torch.empty_like(y)
    ops.conv_gather(x, w, yd, N, 3, 1, 3 * Cin, 3)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    e_w, e_d = _rel(y, ref), _rel(yd, ref)
    # fp32 rounding of a length-3*Cin dot product; the transformed form may cost a small factor over the direct one
    assert e_w < 3e-6, (e_w, e_d)
    assert e_w < 6 * e_d + 1e-7, (e_w, e_d)
    assert (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_data_gradient_and_fused_epilogues():
    """Backward of Conv1d(k=3, padding=1) w.r.t. its input = the flipped correlation over the output gradient; with the branch's
    epilogues: forward scale / shift -> z stored -> GELU; backward GELU'(z_prev) * gscale."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    B, T, Cin, N = 3, 256, 128, 256
    x = torch.randn(B, T, Cin, generator=g).cuda()
    w = (torch.randn(N, Cin, 3, generator=g) / (3 * Cin) ** 0.5).cuda()
    scale, shift = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    wt_f = ops.wino3_weights(w, N, Cin, 3 * Cin, 3)
    a, z = torch.empty(B, T, N, device="cuda"), torch.empty(B, T, N, device="cuda")
    ops.conv_wino3(x, wt_f, a, scale=scale, shift=shift, zout=z, act=ops.ACT_GELU)
    zr = _ref_fwd(x, w) * scale.double() + shift.double()
    ar = F.gelu(zr)
    torch.cuda.synchronize()
    assert _rel(z, zr) < 3e-6 and _rel(a, ar) < 3e-6
    # data gradient: dx[b,t,c] = sum_{k,n} dy[b, t-k+1, n] w[n,c,k], times GELU'(zprev) * gscale
    dy = torch.randn(B, T, N, generator=g).cuda()
    zprev = torch.randn(B, T, Cin, generator=g).cuda()
    gscale = (torch.rand(Cin, generator=g) + 0.5).cuda()
    wt_d = ops.wino3_weights(w, Cin, N, 3, 3 * Cin, flip=True)
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.conv_wino3(dy, wt_d, dx, gref=zprev, gact=ops.ACT_GELU, gscale=gscale)
    dxd = torch.empty_like(dx)
    ops.conv1d_dgrad(dy, w, dxd, 1, gref=zprev, gact=ops.ACT_GELU, gscale=gscale)
    zp = zprev.double().requires_grad_(True)
    xin = F.gelu(zp) * 1.0
    out = F.conv1d(xin.transpose(1, 2), w.double(), padding=1).transpose(1, 2)
    (out * dy.double()).sum().backward()
    ref = zp.grad * gscale.double()
    torch.cuda.synchronize()
    assert _rel(dx, ref) < 3e-6, _rel(dx, ref)
    assert _rel(dx, dxd.double()) < 3e-6
    # accumulate
    y2 = torch.ones(B, T, Cin, device="cuda")
    ops.conv_wino3(dy, wt_d, y2, accumulate=True)
    y3 = torch.empty_like(y2)
    ops.conv_wino3(dy, wt_d, y3)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3 + 1.0) or _rel(y2, (y3 + 1.0).double()) < 1e-7


def test_rejects_unsupported_shapes():
    ops = _ops()
    assert not ops.wino3_supported(2, 255, 64, 64)
    assert not ops.wino3_supported(2, 256, 24, 64)
    assert not ops.wino3_supported(2, 256, 64, 96)
    x = torch.zeros(2, 255, 64, device="cuda")
    wt = torch.zeros(16, 4, 64, 4, device="cuda")
    y = torch.zeros(2, 255, 64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv_wino3(x, wt, y)


def test_occupancy_cap_does_not_change_the_result():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 256, 64, generator=g).cuda()
    w = torch.randn(128, 64, 3, generator=g).cuda()
    wt = ops.wino3_weights(w, 128, 64, 192, 3)
    y0, y1 = torch.empty(4, 256, 128, device="cuda"), torch.empty(4, 256, 128, device="cuda")
    ops.conv_wino3(x, wt, y0)
    with ops.conv_lds_pad(42000):
        ops.conv_wino3(x, wt, y1)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


def test_multi_filter_transform_equals_the_single_launches():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    jobs, want = [], []
    for (co, ci) in ((128, 64), (256, 128), (256, 256)):
        w = torch.randn(co, ci, 3, generator=g).cuda()
        for flip in (False, True):
            N, Cin, sn, sc = (co, ci, 3 * ci, 3) if not flip else (ci, co, 3, 3 * ci)
            img = torch.full((Cin // 4, 4, N, 4), float("nan"), device="cuda")
            jobs.append((w, img, N, Cin, sn, sc, flip))
            want.append(ops.wino3_weights(w, N, Cin, sn, sc, flip))
    ops.wino3_weights_multi(jobs)
    torch.cuda.synchronize()
    for (_, img, *_), ref in zip(jobs, want):
        assert torch.equal(img, ref)
