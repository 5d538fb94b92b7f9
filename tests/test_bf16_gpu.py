"""bf16-storage / fp32-accumulate variant of the frozen emotion-discriminator branch (secondary configuration).

Kernel level: the result must equal an fp32 convolution of the SAME bf16-rounded operands up to accumulation order and the
final rounding of a bf16 output (rtol 2^-8 on bf16 outputs, 1e-4 on fp32 outputs) -- i.e. the only precision given up is
storage.  Engine level: the branch's loss and input gradient against the fp32 engine at stated bf16 bounds."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    import importlib
    return importlib.import_module("melo-gan_amd.ops")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def gelu_grad(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("B,T,Cin,Cout,K,x32", [(2, 128, 64, 128, 3, False), (3, 256, 128, 64, 5, True), (2, 128, 256, 256, 3, False),
                                                 (1, 384, 32, 64, 5, False), (64, 256, 128, 256, 3, False)])
def test_conv_bf16_forward_epilogue(ops, B, T, Cin, Cout, K, x32):
    x = rnd(B, T, Cin, seed=1)
    w = rnd(Cout, Cin, K, seed=2, scale=1 / math.sqrt(Cin * K))
    sc, sh = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    xq, wq = x.to(BF).float(), w.to(BF).float()                   # what the kernel multiplies
    z = F.conv1d(xq.double().permute(0, 2, 1), wq.double(), None, 1, K // 2).permute(0, 2, 1) * sc.double() + sh.double()
    a = F.gelu(z)
    wb = torch.empty(K, Cout, Cin, dtype=BF, device="cuda")
    ops.wb_relayout(w.cuda(), wb, Cout, Cin, K, Cin * K, K)
    assert torch.equal(wb.float().cpu(), wq.permute(2, 0, 1))
    xd = x.cuda() if x32 else x.to(BF).cuda()
    y = torch.empty(B, T, Cout, dtype=BF, device="cuda")
    zo = torch.empty_like(y)
    ops.conv_s1_bf16(xd, wb, y, scale=sc.cuda(), shift=sh.cuda(), zout=zo, act=ops.ACT_GELU)
    torch.testing.assert_close(zo.float().cpu().double(), z, rtol=2 ** -8, atol=2e-3)
    torch.testing.assert_close(y.float().cpu().double(), a, rtol=2 ** -8, atol=2e-3)
    y2 = torch.empty_like(y)
    ops.conv_s1_bf16(xd, wb, y2, scale=sc.cuda(), shift=sh.cuda(), act=ops.ACT_GELU)
    assert torch.equal(y, y2)                                     # reproducible, and zout does not disturb y
    # fp32 output: only accumulation order differs from the reference
    yf = torch.empty(B, T, Cout, device="cuda")
    ops.conv_s1_bf16(xd, wb, yf)
    ref = F.conv1d(xq.double().permute(0, 2, 1), wq.double(), None, 1, K // 2).permute(0, 2, 1)
    torch.testing.assert_close(yf.cpu().double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,T,Cin,Cout,K", [(2, 128, 64, 128, 3), (2, 256, 128, 64, 5), (2, 128, 256, 256, 3)])
def test_conv_bf16_data_gradient(ops, B, T, Cin, Cout, K):
    """dx = conv^T(dz) * GELU'(z_prev) * gscale with flipped taps; last layer: fp32 output accumulated into dnotes."""
    w = rnd(Cout, Cin, K, seed=2, scale=1 / math.sqrt(Cout * K))
    dz = rnd(B, T, Cout, seed=3)
    zprev, gs = rnd(B, T, Cin, seed=4), rnd(Cin, seed=5).abs() + 0.5
    wq, dzq, zq = w.to(BF).float(), dz.to(BF).float(), zprev.to(BF).float()
    x = torch.zeros(B, Cin, T, dtype=torch.float64, requires_grad=True)
    out = F.conv1d(x, wq.double(), None, 1, K // 2)
    dx, = torch.autograd.grad(out, x, dzq.double().permute(0, 2, 1))
    dx = dx.permute(0, 2, 1)
    wbd = torch.empty(K, Cin, Cout, dtype=BF, device="cuda")
    ops.wb_relayout(w.cuda(), wbd, Cin, Cout, K, K, Cin * K, flip=True)
    y = torch.empty(B, T, Cin, dtype=BF, device="cuda")
    ops.conv_s1_bf16(dz.to(BF).cuda(), wbd, y, gref=zprev.to(BF).cuda(), gact=ops.ACT_GELU, gscale=gs.cuda())
    ref = dx * gelu_grad(zq.double()) * gs.double()
    torch.testing.assert_close(y.float().cpu().double(), ref, rtol=2 ** -8, atol=2e-3 * float(ref.abs().max()))
    base = rnd(B, T, Cin, seed=6)
    yf = base.clone().cuda()
    ops.conv_s1_bf16(dz.to(BF).cuda(), wbd, yf, accumulate=True)
    torch.testing.assert_close(yf.cpu().double(), base.double() + dx, rtol=1e-4, atol=1e-4 * float(dx.abs().max()))


def test_meanT_bf16(ops):
    B, T, C = 5, 256, 256
    a, dh, z, gs = rnd(B, T, C, seed=1), rnd(B, C, seed=2), rnd(B, T, C, seed=3), rnd(C, seed=4)
    h = torch.empty(B, C, device="cuda")
    ops.meanT_fwd_bf16(a.to(BF).cuda(), h)
    torch.testing.assert_close(h.cpu(), a.to(BF).float().mean(dim=1), rtol=1e-5, atol=1e-6)
    dz = torch.empty(B, T, C, dtype=BF, device="cuda")
    ops.meanT_bwd_bf16(dh.cuda(), dz, z.to(BF).cuda(), ops.ACT_GELU, gs.cuda())
    ref = dh[:, None, :] / T * gelu_grad(z.to(BF).float()) * gs
    torch.testing.assert_close(dz.float().cpu(), ref, rtol=2 ** -8, atol=1e-6)


def test_conv_bf16_rejects_unsupported_shapes(ops):
    wb = torch.empty(3, 64, 32, dtype=BF, device="cuda")
    with pytest.raises(ValueError):
        ops.conv_s1_bf16(torch.empty(1, 100, 32, dtype=BF, device="cuda"), wb, torch.empty(1, 100, 64, dtype=BF, device="cuda"))
    assert not ops.conv_s1_bf16_supported(1, 128, 4, 64, 5) and ops.conv_s1_bf16_supported(64, 256, 128, 64, 5)


def _engine_pair(B=8, T=128, C=128):
    """Two engines with the same weights and batch: fp32 (the product path) and the bf16-stored emotion discriminator."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.engine import GanEngine
    from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg
    cfg, ed_cfg = default_gan_cfg(B, T, C), default_ed_cfg(C)
    g = torch.Generator().manual_seed(3)
    batch = ((torch.rand(B, T, C, generator=g) * 2 - 1).cuda(), torch.randn(B, 6, generator=g).cuda(),
             torch.zeros(B, cfg["LATENT_DIM"]).cuda(), torch.randint(0, 4, (B,), generator=g).cuda())
    engs = []
    for dt in ("fp32", "bf16"):
        e = GanEngine(cfg, ed_cfg, "cuda", B, ed_dtype=dt)
        e.init_weights(seed=7)
        # BatchNorm running statistics of a "pre-trained" discriminator (non-trivial folded scale / shift)
        gg = torch.Generator().manual_seed(11)
        for k, v in e.EDbuf.items():
            if k.endswith("running_mean"):
                v.copy_((torch.randn(v.shape, generator=gg) * 0.1).cuda())
            elif k.endswith("running_var"):
                v.copy_((torch.rand(v.shape, generator=gg) + 0.5).cuda())
        e.params_changed()
        e.seed(99)
        e.set_batch(*batch)
        engs.append(e)
    return engs


def test_engine_bf16_ed_branch_against_fp32():
    """Forward logits / cross-entropy and the gradient w.r.t. the generated notes of the bf16-stored branch against the fp32
    branch on the same generated batch.  Stated bounds: logits and loss 1e-2 relative, gradient 3e-2 of its norm (three
    bf16 roundings per layer across four layers, forward and backward)."""
    e32, e16 = _engine_pair()
    outs = []
    for e in (e32, e16):
        with torch.cuda.stream(e.stream):
            e.run("dg_forward_d_backward_rng", False)      # generator pass: `notes` of the generator-step half
            e.dnotes.zero_()
            e.run("g_ed_branch", False)
        torch.cuda.synchronize()
        outs.append((e.notes.clone(), e.logits.clone(), e.emo.clone(), e.dnotes.clone()))
    (n32, l32, c32, d32), (n16, l16, c16, d16) = outs
    assert torch.equal(n32, n16)                           # everything up to the branch is the fp32 path in both
    assert float((l16 - l32).abs().max()) <= 1e-2 * float(l32.abs().max()) + 1e-3
    assert abs(float(c16) - float(c32)) <= 1e-2 * abs(float(c32)) + 1e-4
    rel = float((d16 - d32).norm() / d32.norm())
    assert rel < 3e-2, rel
    assert e16.ed_a[0].dtype == torch.bfloat16 and e16.ed_z[-1].dtype == torch.bfloat16


def test_engine_bf16_full_step_runs_under_graphs_and_tracks_fp32():
    """Three full (critic + generator) steps, hipGraph replay: the secondary configuration trains -- losses finite and
    within 2 % of the fp32 engine's after the same three batches (the branch only contributes lambda_emo * CE)."""
    e32, e16 = _engine_pair()
    from melo_gan_amd.gan.dp import DataParallel
    res = []
    for e in (e32, e16):
        dp = DataParallel(e, 1, None)
        with torch.cuda.stream(e.stream):
            for _ in range(3):
                dp.step(True)
        torch.cuda.synchronize()
        res.append((float(e.loss_d_out[0]), float(e.adv), float(e.emo)))
    for a, b in zip(res[0], res[1]):
        assert math.isfinite(b) and abs(a - b) <= 2e-2 * abs(a) + 2e-3, (res[0], res[1])
