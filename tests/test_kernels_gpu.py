"""Per-kernel parity of the HIP library (through its C-ABI) against plain PyTorch-CPU fp32
references of the same op (the ops the reference's nn.Modules dispatch).  GPU only."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5      # single fwd/bwd fp32 tolerance (hard part 3)


@pytest.fixture(scope="module")
def ops():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(t):
    return t.cuda().contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close(got, ref, rtol=RTOL, atol=ATOL):
    torch.testing.assert_close(got.cpu(), ref, rtol=rtol, atol=atol)


CONV_CASES = [
    # B, T, Cin, Cout, K, stride
    (3, 64, 128, 64, 5, 2),
    (2, 32, 64, 128, 5, 2),
    (2, 16, 128, 256, 5, 2),
    (3, 20, 4, 64, 5, 2),      # odd lengths 20 -> 10
    (3, 5, 128, 256, 5, 2),    # 5 -> 3
    (2, 32, 128, 64, 5, 1),
    (2, 32, 64, 128, 3, 1),
    (2, 48, 128, 256, 3, 1),
    (3, 20, 4, 64, 5, 1),
    (64, 256, 128, 64, 5, 2),  # headline D conv.0 shape (big-tile config)
    (16, 256, 256, 256, 3, 1),  # ED conv3 shape (big tile)
]


@pytest.mark.parametrize("B,T,Cin,Cout,K,stride", CONV_CASES)
def test_conv1d_fwd_dgrad_wgrad(ops, B, T, Cin, Cout, K, stride):
    x = rnd(B, T, Cin, seed=1)
    w = rnd(Cout, Cin, K, seed=2, scale=1.0 / math.sqrt(Cin * K))
    b = rnd(Cout, seed=3, scale=0.1)
    xr = x.permute(0, 2, 1).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    zr = F.conv1d(xr, wr, b, stride, K // 2)
    yr = F.leaky_relu(zr, 0.2)
    Tout = zr.shape[2]
    dy = rnd(B, Tout, Cout, seed=4)
    dz_ref = torch.autograd.grad(yr, zr, dy.permute(0, 2, 1), retain_graph=True)[0]
    dx_ref, dw_ref = torch.autograd.grad(zr, (xr, wr), dz_ref)
    # forward with fused bias + LeakyReLU
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty(B, Tout, Cout, device="cuda")
    ops.conv1d_fwd(xd, wd, y, stride, bias=bd, act=ops.ACT_LRELU)
    assert_close(y, yr.detach().permute(0, 2, 1).contiguous())
    # dz = dy * lrelu'(y) via act_bwd, then dgrad and wgrad
    dz = torch.empty_like(y)
    ops.act_bwd(dev(dy), dz, gref=y, gact=ops.ACT_LRELU)
    assert_close(dz, dz_ref.permute(0, 2, 1).contiguous())
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.conv1d_dgrad(dz, wd, dx, stride)
    assert_close(dx, dx_ref.permute(0, 2, 1).contiguous(), atol=1e-4 if Cout * K > 1000 else ATOL)
    dw, dbf = torch.empty(Cout, Cin, K, device="cuda"), torch.empty(Cout, device="cuda")
    ops.conv1d_wgrad(xd, dz, dw, stride, db=dbf)             # bias gradient fused into the wgrad launch
    scale = dw_ref.abs().max().item()
    assert_close(dw, dw_ref, rtol=1e-4, atol=2e-5 * max(1.0, scale))
    db = torch.empty(Cout, device="cuda")
    ops.colsum(dz, db)
    assert_close(db, dz_ref.sum(dim=(0, 2)), rtol=1e-4, atol=1e-4)
    assert_close(dbf, dz_ref.sum(dim=(0, 2)), rtol=1e-4, atol=1e-4)


CONVT_CASES = [(3, 8, 256, 128), (2, 16, 128, 64), (2, 32, 64, 128), (3, 4, 64, 4), (64, 32, 256, 128), (64, 128, 64, 128)]


@pytest.mark.parametrize("B,T,Cin,Cout", CONVT_CASES)
def test_convT1d_fwd_dgrad_wgrad(ops, B, T, Cin, Cout):
    x = rnd(B, T, Cin, seed=5)
    w = rnd(Cin, Cout, 5, seed=6, scale=1.0 / math.sqrt(Cin * 2.5))
    b = rnd(Cout, seed=7, scale=0.1)
    xr = x.permute(0, 2, 1).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv_transpose1d(xr, wr, b, 2, 2, 1)
    dy = rnd(B, 2 * T, Cout, seed=8)
    dx_ref, dw_ref = torch.autograd.grad(yr, (xr, wr), dy.permute(0, 2, 1))
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    y = torch.empty(B, 2 * T, Cout, device="cuda")
    ops.convT1d_fwd(xd, wd, y, bias=bd)
    assert_close(y, yr.detach().permute(0, 2, 1).contiguous())
    dx = torch.empty(B, T, Cin, device="cuda")
    ops.convT1d_dgrad(dyd, wd, dx)
    assert_close(dx, dx_ref.permute(0, 2, 1).contiguous(), atol=1e-4)
    dw, db = torch.empty(Cin, Cout, 5, device="cuda"), torch.empty(Cout, device="cuda")
    ops.convT1d_wgrad(xd, dyd, dw, db=db)
    assert_close(dw, dw_ref, rtol=1e-4, atol=2e-5 * max(1.0, dw_ref.abs().max().item()))
    assert_close(db, dy.sum(dim=(0, 1)), rtol=1e-4, atol=1e-4)


def test_convT1d_padded_output(ops):
    """Generator zero-pad branch (src/gan/models.py:76-81): write 2*Tin rows into a longer buffer."""
    B, T, Cin, Cout, Tpad = 2, 8, 64, 4, 20
    x, w = rnd(B, T, Cin, seed=1), rnd(Cin, Cout, 5, seed=2, scale=0.1)
    yr = F.conv_transpose1d(x.permute(0, 2, 1), w, None, 2, 2, 1).permute(0, 2, 1)
    y = torch.zeros(B, Tpad, Cout, device="cuda")
    ops.convT1d_fwd(dev(x), dev(w), y)
    assert_close(y[:, :16], yr.contiguous())
    assert float(y[:, 16:].abs().max()) == 0.0


# rows > 512 take the K=1 window-GEMM instantiation (64-channel chunks) instead of the skinny kernel
@pytest.mark.parametrize("B,inf,outf", [(64, 256, 512), (5, 6, 256), (64, 512, 8192), (7, 384, 1), (192, 256, 256), (64, 128, 4),
                                        (1024, 256, 256), (600, 192, 320), (777, 100, 130)])
def test_linear_fwd_dgrad_wgrad(ops, B, inf, outf):
    x, w, b = rnd(B, inf, seed=1), rnd(outf, inf, seed=2, scale=1 / math.sqrt(inf)), rnd(outf, seed=3, scale=0.1)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    zr = F.linear(xr, wr, b)
    yr = F.gelu(zr)
    dy = rnd(B, outf, seed=4)
    dz_ref = torch.autograd.grad(yr, zr, dy, retain_graph=True)[0]
    dx_ref, dw_ref = torch.autograd.grad(zr, (xr, wr), dz_ref)
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty(B, outf, device="cuda")
    z = torch.empty(B, outf, device="cuda")
    ops.linear_fwd(xd, wd, y, bias=bd, zout=z, act=ops.ACT_GELU)
    assert_close(z, zr.detach())
    assert_close(y, yr.detach())
    dz = torch.empty_like(y)
    ops.act_bwd(dev(dy), dz, gref=z, gact=ops.ACT_GELU)
    assert_close(dz, dz_ref)
    dx = torch.empty(B, inf, device="cuda")
    ops.linear_dgrad(dz, wd, dx)
    assert_close(dx, dx_ref, atol=1e-4)
    dw, db = torch.empty(outf, inf, device="cuda"), torch.empty(outf, device="cuda")
    ops.linear_wgrad(xd, dz, dw, db=db)
    assert_close(dw, dw_ref, atol=1e-4)
    assert_close(db, dz_ref.sum(dim=0), rtol=1e-4, atol=1e-4)


def test_epilogue_chain_and_accumulate(ops):
    """scale/shift (folded eval BN), zout, GELU, act-grad multiply, per-channel gscale, accumulate."""
    B, T, Cin, Cout = 2, 16, 64, 128
    x, w = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, 3, seed=2, scale=0.1)
    sc, sh = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    zr = F.conv1d(x.permute(0, 2, 1), w, None, 1, 1).permute(0, 2, 1) * sc + sh
    y = torch.empty(B, T, Cout, device="cuda")
    z = torch.empty_like(y)
    ops.conv1d_fwd(dev(x), dev(w), y, 1, scale=dev(sc), shift=dev(sh), zout=z, act=ops.ACT_GELU)
    assert_close(z, zr.contiguous())
    assert_close(y, F.gelu(zr).contiguous())
    # dgrad with fused gelu'(zprev)*gscale and accumulation onto an existing tensor
    dy = rnd(B, T, Cout, seed=5)
    zprev, gsc, base = rnd(B, T, Cin, seed=6), rnd(Cin, seed=7), rnd(B, T, Cin, seed=8)
    xr = x.permute(0, 2, 1).clone().requires_grad_(True)
    dx_ref = torch.autograd.grad(F.conv1d(xr, w, None, 1, 1), xr, dy.permute(0, 2, 1))[0].permute(0, 2, 1)
    zp = zprev.clone().requires_grad_(True)
    gelu_g = torch.autograd.grad(F.gelu(zp).sum(), zp)[0]
    ref = base + dx_ref * gelu_g * gsc
    out = dev(base)
    ops.conv1d_dgrad(dev(dy), dev(w), out, 1, gref=dev(zprev), gact=ops.ACT_GELU, gscale=dev(gsc), accumulate=True)
    assert_close(out, ref.contiguous(), atol=1e-4)


def test_wgrad_two_segments(ops):
    B0, B1, T, Cin, Cout = 5, 3, 32, 64, 128
    x0, x1 = rnd(B0, T, Cin, seed=1), rnd(B1, T, Cin, seed=2)
    d0, d1 = rnd(B0, T // 2, Cout, seed=3), rnd(B1, T // 2, Cout, seed=4)
    w = torch.zeros(Cout, Cin, 5, requires_grad=True)
    xa = torch.cat([x0, x1]).permute(0, 2, 1)
    da = torch.cat([d0, d1]).permute(0, 2, 1)
    ref = torch.autograd.grad(F.conv1d(xa, w, None, 2, 2), w, da)[0]
    dw = torch.empty(Cout, Cin, 5, device="cuda")
    ops.conv1d_wgrad(dev(x0), dev(d0), dw, 2, dev(x1), dev(d1))
    assert_close(dw, ref, atol=1e-4)


@pytest.mark.parametrize("R,C", [(4096, 128), (8192, 64), (37, 4), (64 * 256, 128)])
def test_bn_train_fwd_bwd(ops, R, C):
    z = rnd(R, C, seed=1) * 2 + 0.3
    gamma, beta = rnd(C, seed=2).abs() + 0.5, rnd(C, seed=3)
    rm, rv = rnd(C, seed=4) * 0.1, rnd(C, seed=5).abs() + 0.5
    zr = z.t().reshape(1, C, R).clone().requires_grad_(True)     # (N=1, C, L=R)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    ar = F.relu(F.batch_norm(zr, rm_r, rv_r, gr, br, True, 0.1, 1e-5))
    da = rnd(R, C, seed=6)
    dz_ref, dg_ref, db_ref = torch.autograd.grad(ar, (zr, gr, br), da.t().reshape(1, C, R))
    zd = dev(z)
    a = torch.empty_like(zd)
    rmd, rvd = dev(rm), dev(rv)
    sm, si = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_train_fwd(zd, a, dev(gamma), dev(beta), rmd, rvd, sm, si)
    assert_close(a, ar.detach()[0].t().contiguous())
    assert_close(rmd, rm_r)
    assert_close(rvd, rv_r)
    dz, dg, db = torch.empty_like(zd), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_train_bwd(dev(da), a, zd, dz, dev(gamma), sm, si, dg, db)
    assert_close(dz, dz_ref[0].t().contiguous(), atol=2e-5)
    assert_close(dg, dg_ref, rtol=1e-4, atol=1e-3)
    assert_close(db, db_ref, rtol=1e-4, atol=1e-3)


def test_bn_eval_and_fold(ops):
    R, C = 512, 64
    z = rnd(R, C, seed=1)
    gamma, beta, rm, rv = rnd(C, seed=2), rnd(C, seed=3), rnd(C, seed=4), rnd(C, seed=5).abs() + 0.1
    ref = F.gelu(F.batch_norm(z.t().reshape(1, C, R), rm, rv, gamma, beta, False, 0.1, 1e-5))[0].t().contiguous()
    a = torch.empty(R, C, device="cuda")
    ops.bn_eval_fwd(dev(z), a, dev(gamma), dev(beta), dev(rm), dev(rv), act=ops.ACT_GELU)
    assert_close(a, ref)
    cb = rnd(C, seed=6)
    sc, sh = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_fold(dev(gamma), dev(beta), dev(rm), dev(rv), dev(cb), sc, sh)
    s_ref = gamma / torch.sqrt(rv + 1e-5)
    assert_close(sc, s_ref)
    assert_close(sh, beta + (cb - rm) * s_ref)


def test_meanT_layernorm_transpose(ops):
    a = rnd(5, 37, 96, seed=1)
    h = torch.empty(5, 96, device="cuda")
    ops.meanT_fwd(dev(a), h)
    assert_close(h, a.mean(dim=1))
    dh = rnd(5, 96, seed=2)
    dz = torch.empty(5, 37, 96, device="cuda")
    ops.meanT_bwd(dev(dh), dz, gref=dev(a), gact=ops.ACT_LRELU)
    assert_close(dz, (dh[:, None, :] / 37) * torch.where(a > 0, 1.0, 0.2))
    x, g, b = rnd(9, 6, seed=3), rnd(6, seed=4), rnd(6, seed=5)
    y, xh = torch.empty(9, 6, device="cuda"), torch.empty(9, 6, device="cuda")
    ops.layernorm_fwd(dev(x), y, xh, dev(g), dev(b))
    assert_close(y, F.layer_norm(x, (6,), g, b, 1e-5))
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    dy = rnd(9, 6, seed=6)
    dg_ref, db_ref = torch.autograd.grad(F.layer_norm(x, (6,), gr, br, 1e-5), (gr, br), dy)
    dg, db = torch.empty(6, device="cuda"), torch.empty(6, device="cuda")
    ops.layernorm_bwd_params(dev(dy), xh, dg, db)
    assert_close(dg, dg_ref)
    assert_close(db, db_ref)
    t = rnd(3, 70, 45, seed=7)
    out = torch.empty(3, 45, 70, device="cuda")
    ops.transpose_bcl_blc(dev(t), out)
    assert_close(out, t.permute(0, 2, 1).contiguous(), rtol=0, atol=0)


def test_gp_losses_adam(ops):
    B, n = 6, 4096
    real, fake, alpha = rnd(B, 32, 128, seed=1), rnd(B, 32, 128, seed=2), torch.rand(B, 1, 1)
    xh = torch.empty(B, 32, 128, device="cuda")
    ops.gp_interp(dev(real), dev(fake), dev(alpha), xh)
    assert_close(xh, alpha * real + (1 - alpha) * fake)
    g = (rnd(B, 32, 128, seed=3) * 0.01).requires_grad_(True)
    gpr = ((g.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
    gbar_ref = torch.autograd.grad(10.0 * gpr, g)[0]
    gbar, norms, gp = torch.empty(B, 32, 128, device="cuda"), torch.empty(B, device="cuda"), torch.empty(1, device="cuda")
    ops.gp_penalty(dev(g.detach()), gbar, norms, gp, 10.0)
    assert_close(gp, gpr.detach().reshape(1))
    assert_close(gbar, gbar_ref, rtol=1e-4, atol=1e-6)
    logits, tgt = rnd(7, 4, seed=4), torch.tensor([0, 3, 1, 2, 2, 0, 1])
    lr = logits.clone().requires_grad_(True)
    ce = F.cross_entropy(lr, tgt)
    dl_ref = torch.autograd.grad(5.0 * ce, lr)[0]
    loss, dl = torch.empty(1, device="cuda"), torch.empty(7, 4, device="cuda")
    ops.softmax_ce(dev(logits), tgt.cuda(), loss, dl, 5.0)
    assert_close(loss, ce.detach().reshape(1))
    assert_close(dl, dl_ref)
    # Adam vs torch.optim.Adam for 3 steps
    p0 = rnd(1000, seed=5)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.5, 0.9))
    p, m, v = dev(p0), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    st = torch.zeros(4, dtype=torch.float64, device="cuda")
    for i in range(3):
        gi = rnd(1000, seed=10 + i)
        pr.grad = gi.clone()
        opt.step()
        ops.adam_flat(p, dev(gi), m, v, st, 2e-4, 0.5, 0.9)
    assert_close(p, pr.detach(), rtol=1e-5, atol=1e-7)


def test_dhead(ops):
    B, Be, Fd, E = 12, 4, 256, 128
    f, emb, w, b = rnd(B, Fd, seed=1), rnd(Be, E, seed=2), rnd(Fd + E, seed=3, scale=0.1), rnd(1, seed=4)
    cat = torch.cat([f, emb.repeat(B // Be, 1)], dim=1)
    s = torch.empty(B, device="cuda")
    ops.dhead_fwd(dev(f), dev(emb), dev(w), dev(b), s)
    assert_close(s, cat @ w + b)
    ds = rnd(B, seed=5)
    dU, demb = torch.empty(B, Fd, device="cuda"), torch.empty(Be, E, device="cuda")
    ops.dhead_bwd(dev(ds), dev(f), dev(w), dU, demb, nb_emb=B)
    assert_close(dU, ds[:, None] * w[None, :Fd] * torch.where(f > 0, 1.0, 0.2))
    assert_close(demb, (ds[:, None] * w[None, Fd:]).reshape(B // Be, Be, E).sum(0))
    gf = rnd(5, Fd, seed=6)
    dw, db = torch.empty(Fd + E, device="cuda"), torch.empty(1, device="cuda")
    ops.dhead_wgrad(dev(ds), dev(f), dev(emb), dev(gf), dw, db, 8, 5)
    ref = (ds[:8, None] * cat[:8]).sum(0)
    ref[:Fd] += gf.sum(0)
    assert_close(dw, ref, atol=1e-4)
    assert_close(db, ds[:8].sum().reshape(1))


def test_rng_fill(ops):
    n = 1 << 18
    normal, uni = torch.empty(n, device="cuda"), torch.empty(1001, device="cuda")
    m0, m1 = torch.empty(64, 256, device="cuda"), torch.empty(64, 128, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.rng_fill(normal, uni, m0, m1, 0.2, 1234, ctr)
    assert int(ctr.item()) == 1
    assert abs(normal.mean().item()) < 0.01 and abs(normal.std().item() - 1.0) < 0.01
    assert abs((normal ** 4).mean().item() - 3.0) < 0.1                   # kurtosis of N(0,1)
    assert 0.0 < uni.min().item() and uni.max().item() < 1.0 and abs(uni.mean().item() - 0.5) < 0.05
    vals = torch.unique(m0).cpu()
    assert torch.allclose(vals, torch.tensor([0.0, 1.25]))
    assert abs((m0 > 0).float().mean().item() - 0.8) < 0.02 and abs((m1 > 0).float().mean().item() - 0.8) < 0.03
    # same (seed, step) => same numbers; advancing the step or changing the seed => different numbers
    a, b, c = torch.empty(4096, device="cuda"), torch.empty(4096, device="cuda"), torch.empty(4096, device="cuda")
    c0 = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.rng_fill(a, None, None, None, 0.2, 7, c0)
    ops.rng_fill(b, None, None, None, 0.2, 7, c0)
    c0.zero_()
    ops.rng_fill(c, None, None, None, 0.2, 7, c0)
    assert torch.equal(a, c) and not torch.equal(a, b)
    assert abs(torch.corrcoef(torch.stack([a, b]))[0, 1].item()) < 0.05


def test_tick_free_draw_and_update_match_the_plain_pair(ops):
    """rng_fill(tick_state=...) + adam_flat(ticked_rng_step=...) (one launch each) == rng_fill + adam_flat (two each):
    same draws for the same (seed, step), same Adam state and parameters after the update, counter advanced once."""
    n = 4096 + 3
    torch.manual_seed(0)
    p0, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    betas = (0.5, 0.9)

    def run(ticked, steps=3):
        p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        state = torch.zeros(4, dtype=torch.float64, device="cuda")
        ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
        draws = []
        for _ in range(steps):
            x = torch.empty(1001, device="cuda")
            if ticked:
                ops.rng_fill(x, None, None, None, 0.2, 99, ctr, tick_state=state, betas=betas)
                ops.adam_flat(p, g, m, v, state, 1e-3, *betas, ticked_rng_step=ctr)
            else:
                ops.rng_fill(x, None, None, None, 0.2, 99, ctr)
                ops.adam_flat(p, g, m, v, state, 1e-3, *betas)
            draws.append(x)
        return p, m, v, state, ctr, draws

    a, b = run(False), run(True)
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    for x, y in zip(a[5], b[5]):
        assert torch.equal(x, y)
    assert int(b[4].item()) == 3 and float(b[3][0].item()) == 3.0


def test_wgrad_multi_equals_separate_launches():
    """mg_wgrad_multi: several weight gradients per launch equal one launch each -- bit for bit where the reduction is
    not split over workgroups (the Linear layers), to fp32 summation-order noise where it is (the multi launch plans the
    slice counts of its jobs TOGETHER, so a job's slices differ from its stand-alone plan) -- and the multi launch is
    itself bitwise reproducible: Linear layers, stride-2 K=5 conv / convT gradients with fused bias sums and batch
    splits, a two-segment job, ragged channel counts, and more jobs than one launch holds."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    g = torch.Generator().manual_seed(11)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).cuda()  # noqa: E731
    jobs, want = [], []

    def both(fn, *a, outs, **k):
        fn(*a, **k)
        want.append([o.clone() for o in outs])
        for o in outs:
            o.fill_(float("nan"))
        jobs.append((fn(*a, defer=True, **k), outs))

    for (B, i, o) in [(64, 64, 512), (64, 512, 64), (64, 256, 512), (64, 6, 256), (64, 256, 128), (64, 128, 128),
                      (192, 256, 256), (64, 40, 72), (64, 128, 4), (30, 100, 36)]:
        x, dy, dw, db = r(B, i), r(B, o), torch.empty(o, i).cuda(), torch.empty(o).cuda()
        both(ops.linear_wgrad, x, dy, dw, db=db, outs=[dw, db])
    x, dy, x2, dy2, dw = r(16, 24), r(16, 40), r(8, 24), r(8, 40), torch.empty(40, 24).cuda()
    both(ops.linear_wgrad, x, dy, dw, x2=x2, dy2=dy2, outs=[dw])
    for (B, T, ci, co) in [(64, 32, 256, 128), (64, 64, 128, 64), (64, 128, 64, 128), (5, 16, 20, 12)]:
        x, dy, dw, db = r(B, T, ci), r(B, 2 * T, co), torch.empty(ci, co, 5).cuda(), torch.empty(co).cuda()
        both(ops.convT1d_wgrad, x, dy, dw, db=db, outs=[dw, db])
    for (B, T, ci, co) in [(192, 64, 128, 256), (24, 32, 64, 128)]:
        x, dy, dw, db = r(B, T, ci), r(B, T // 2, co), torch.empty(co, ci, 5).cuda(), torch.empty(co).cuda()
        both(ops.conv1d_wgrad, x, dy, dw, 2, db=db, outs=[dw, db])
    ops.wgrad_multi([j for j, _ in jobs])
    first = [[o.clone() for o in outs] for _, outs in jobs]
    for (job, outs), ws in zip(jobs, want):
        for o, w in zip(outs, ws):
            if job[13] == 1:                     # K = 1: one slice either way
                assert torch.equal(o, w)
            else:
                torch.testing.assert_close(o, w, rtol=2e-5, atol=2e-5 * float(w.abs().max()))
    for _, outs in jobs:
        for o in outs:
            o.fill_(float("nan"))
    ops.wgrad_multi([j for j, _ in jobs])
    for (_, outs), fs in zip(jobs, first):
        for o, f in zip(outs, fs):
            assert torch.equal(o, f)


def test_transpose_with_activation_backward():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(5, 37, 70, generator=g) * 2 - 1).cuda()
    ref = (torch.rand(5, 70, 37, generator=g) * 2 - 1).cuda()
    y = torch.empty(5, 70, 37).cuda()
    ops.transpose_bcl_blc(x, y)
    assert torch.equal(y, x.transpose(1, 2).contiguous())
    ops.transpose_bcl_blc(x, y, gref=ref, gact=ops.ACT_RELU)
    assert torch.equal(y, x.transpose(1, 2) * (ref > 0).float())


def test_dhead_fwd_bwd_equals_separate_launches():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    g = torch.Generator().manual_seed(9)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).cuda()  # noqa: E731
    for B, Be, with_demb in ((192, 64, False), (64, 64, True), (6, 3, True)):
        F, E = 256, 128
        f, emb, w, bias, ds = r(B, F), r(Be, E), r(F + E), r(1), r(B)
        s0, dU0, de0 = torch.empty(B).cuda(), torch.empty(B, F).cuda(), (torch.empty(Be, E).cuda() if with_demb else None)
        s1, dU1, de1 = torch.empty(B).cuda(), torch.empty(B, F).cuda(), (torch.empty(Be, E).cuda() if with_demb else None)
        ops.dhead_fwd(f, emb, w, bias, s0)
        ops.dhead_bwd(ds, f, w, dU0, de0, nb_emb=B if with_demb else 0)
        ops.dhead_fwd_bwd(ds, f, emb, w, bias, s1, dU1, de1, nb_emb=B if with_demb else 0)
        assert torch.equal(s0, s1) and torch.equal(dU0, dU1)
        if with_demb:
            assert torch.equal(de0, de1)
