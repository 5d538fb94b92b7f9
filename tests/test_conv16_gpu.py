"""conv16 (csrc/conv16_mfma.hip: the stride-2 five-tap window GEMMs on 16x16x4 MFMA tiles, WQ-layout weights) against a
plain PyTorch fp32 reference of the same op, over the layer shapes of the critic / generator and ragged ones, with the
fused epilogue's pieces; and bit-for-bit run-to-run reproducibility (no split-K, no atomics)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def wq_of(ops, w, N, Cc, sn, sc):
    wq = torch.empty(N * Cc * 5, device="cuda")
    ops.wq_relayout(w.cuda().contiguous(), wq, N, Cc, 5, sn, sc)
    return wq


def close(got, ref, tol=2e-5):
    err = float((got.cpu().double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
    assert err < tol, err


# (B, T, Cin, Cout): critic conv.0 / conv.2 / conv.4 at B and 3B rows, plus ragged batch / odd lengths / short rolls
CONV_SHAPES = [(64, 256, 128, 64), (64, 128, 64, 128), (64, 64, 128, 256), (192, 64, 128, 256), (3, 20, 32, 32),
               (5, 33, 32, 64), (2, 8, 32, 32), (7, 100, 64, 96)]


@pytest.mark.parametrize("B,T,Cin,Cout", CONV_SHAPES)
def test_conv_s2_forward_and_data_gradient(ops, B, T, Cin, Cout):
    x, w, bias = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, 5, seed=2, scale=0.05), rnd(Cout, seed=3)
    Tout = (T - 1) // 2 + 1
    ref = F.conv1d(x.transpose(1, 2), w, bias, stride=2, padding=2).transpose(1, 2)
    assert ops.conv16_supported(B, T, Cin, Cout, False)
    y = torch.full((B, Tout, Cout), float("nan"), device="cuda")
    ops.conv16(x.cuda(), wq_of(ops, w, Cout, Cin, Cin * 5, 5), y, Cout, False, bias=bias.cuda())
    close(y, ref)
    # LeakyReLU + saved pre-activation, as the critic's forward uses it
    z = torch.empty_like(y)
    ops.conv16(x.cuda(), wq_of(ops, w, Cout, Cin, Cin * 5, 5), y, Cout, False, bias=bias.cuda(), zout=z, act=ops.ACT_LRELU)
    close(z, ref)
    close(y, F.leaky_relu(ref, 0.2))
    # data gradient (the transposed form: n = Cin, c = Cout), times the LeakyReLU mask of a reference tensor
    dy = rnd(B, Tout, Cout, seed=4)
    xr = x.clone().requires_grad_(True)
    F.conv1d(xr.transpose(1, 2), w, None, stride=2, padding=2).transpose(1, 2).backward(dy)
    assert ops.conv16_supported(B, Tout, Cout, Cin, True, T)
    dx = torch.full((B, T, Cin), float("nan"), device="cuda")
    wq_d = wq_of(ops, w, Cin, Cout, 5, Cin * 5)
    ops.conv16(dy.cuda(), wq_d, dx, Cin, True, odd=(T % 2 == 1))
    close(dx, xr.grad)
    gref = rnd(B, T, Cin, seed=5)
    ops.conv16(dy.cuda(), wq_d, dx, Cin, True, odd=(T % 2 == 1), gref=gref.cuda(), gact=ops.ACT_LRELU)
    close(dx, xr.grad * torch.where(gref > 0, 1.0, 0.2))
    # accumulate into an existing gradient
    base = rnd(B, T, Cin, seed=6)
    dx.copy_(base)
    ops.conv16(dy.cuda(), wq_d, dx, Cin, True, odd=(T % 2 == 1), accumulate=True)
    close(dx, xr.grad + base)


# (B, L, Cin, Cout): generator deconv.0 / .3 / .6 at B and 2B rows, ragged ones
CONVT_SHAPES = [(64, 32, 256, 128), (128, 64, 128, 64), (128, 128, 64, 128), (3, 5, 32, 32), (4, 12, 32, 64)]


@pytest.mark.parametrize("B,L,Cin,Cout", CONVT_SHAPES)
def test_convT_s2_forward_and_data_gradient(ops, B, L, Cin, Cout):
    x, w, bias = rnd(B, L, Cin, seed=1), rnd(Cin, Cout, 5, seed=2, scale=0.05), rnd(Cout, seed=3)
    xr = x.clone().requires_grad_(True)
    ref = F.conv_transpose1d(xr.transpose(1, 2), w, bias, stride=2, padding=2, output_padding=1).transpose(1, 2)
    y = torch.full((B, 2 * L, Cout), float("nan"), device="cuda")
    ops.conv16(x.cuda(), wq_of(ops, w, Cout, Cin, 5, Cout * 5), y, Cout, True, bias=bias.cuda())
    close(y, ref.detach())
    y2 = torch.full((B, 2 * L, Cout), float("nan"), device="cuda")
    ops.conv16(x.cuda(), wq_of(ops, w, Cout, Cin, 5, Cout * 5), y2, Cout, True, bias=bias.cuda())
    assert torch.equal(y, y2)                                                   # run-to-run: identical bits
    # zero-padded tail (the generator's T % 8 != 0 branch): rows beyond 2L stay untouched
    ypad = torch.full((B, 2 * L + 3, Cout), 7.0, device="cuda")
    ops.conv16(x.cuda(), wq_of(ops, w, Cout, Cin, 5, Cout * 5), ypad, Cout, True, bias=bias.cuda())
    assert torch.equal(ypad[:, :2 * L], y) and bool((ypad[:, 2 * L:] == 7.0).all())
    # data gradient (gather form: n = Cin, c = Cout)
    dy = rnd(B, 2 * L, Cout, seed=4)
    ref.backward(dy)
    dx = torch.full((B, L, Cin), float("nan"), device="cuda")
    ops.conv16(dy.cuda(), wq_of(ops, w, Cin, Cout, Cout * 5, 5), dx, Cin, False)
    close(dx, xr.grad)


def test_unsupported_shapes_are_refused(ops):
    assert not ops.conv16_supported(4, 32, 4, 64, False)          # Cin % 16
    assert not ops.conv16_supported(4, 32, 64, 4, False)          # N % 32
    assert not ops.conv16_supported(4, 4, 64, 64, False)          # two output positions per roll
    with pytest.raises(ValueError):
        ops.conv16(torch.zeros(4, 32, 4, device="cuda"), torch.zeros(64 * 4 * 5, device="cuda"), torch.zeros(4, 16, 64, device="cuda"), 64, False)


@pytest.mark.parametrize("B,L,Cin,Cout,groups", [(128, 32, 256, 128, 2), (64, 64, 128, 64, 1), (8, 12, 32, 64, 2)])
def test_batchnorm_from_the_convolutions_partial_statistics(ops, B, L, Cin, Cout, groups):
    """conv16 leaves per-column partial sums of what it stores; BatchNorm forward finishes them without a reduction pass
    over the tensor.  Against the stand-alone three-launch BatchNorm kernels (themselves pinned to the reference in
    test_kernels_gpu.py / test_engine_gpu.py) and against torch's batch_norm."""
    x, w, bias = rnd(B, L, Cin, seed=1), rnd(Cin, Cout, 5, seed=2, scale=0.05), rnd(Cout, seed=3)
    gamma, beta = rnd(Cout, seed=4).abs() + 0.5, rnd(Cout, seed=5)
    wq = wq_of(ops, w, Cout, Cin, 5, Cout * 5)
    tb, rows = ops.conv16_plan(B, L, Cout, True)
    assert (B // groups) % tb == 0
    part = torch.full((3 * rows * Cout,), float("nan"), device="cuda")
    z = torch.empty(B, 2 * L, Cout, device="cuda")
    ops.conv16(x.cuda(), wq, z, Cout, True, bias=bias.cuda(), stats=part)
    a1, a2 = torch.empty_like(z), torch.empty_like(z)
    rm1, rv1, rm2, rv2 = (torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda"),
                          torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda"))
    shp = (Cout,) if groups == 1 else (groups, Cout)
    m1, i1, m2, i2 = (torch.empty(shp, device="cuda") for _ in range(4))
    ops.bn_train_fwd_parts(part, rows, groups, z, a1, gamma.cuda(), beta.cuda(), rm1, rv1, m1, i1, ops.ACT_RELU)
    ops.bn_train_fwd(z, a2, gamma.cuda(), beta.cuda(), rm2, rv2, m2, i2, ops.ACT_RELU, groups=groups)
    for got, want in ((a1, a2), (m1, m2), (i1, i2), (rm1, rm2), (rv1, rv2)):
        close(got, want.cpu(), 2e-6)
    zg = z.cpu().view(groups, B // groups, 2 * L, Cout)
    ref = torch.stack([F.relu(F.batch_norm(zg[g].reshape(-1, Cout), None, None, gamma, beta, True, 0.1, 1e-5)) for g in range(groups)])
    close(a1, ref.view(B, 2 * L, Cout), 2e-5)


@pytest.mark.parametrize("B,Tin,Cin,N", [(192, 64, 128, 256), (64, 64, 128, 256), (3, 64, 32, 64), (5, 63, 64, 96)])
def test_conv16_pool_is_the_temporal_mean_of_the_output(B, Tin, Cin, N):
    """mg_conv16_pool: AdaptiveAvgPool1d(1) of the activated output from the convolution's own launch (critic conv.4 ->
    pooling, src/gan/models.py:141-148) equals the mean over time of what the launch stored, for full and ragged batch
    tiles, with the bias + LeakyReLU and the tangent-pass epilogues; shapes that do not qualify are refused."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    g = torch.Generator().manual_seed(B + Tin)
    x = torch.randn(B, Tin, Cin, generator=g).cuda()
    w = (torch.randn(N, Cin, 5, generator=g) * 0.05).cuda()
    b = torch.randn(N, generator=g).cuda()
    wq = torch.empty(N * Cin * 5, device="cuda")
    ops.wq_relayout(w, wq, N, Cin, 5, Cin * 5, 5)
    Tout = (Tin - 1) // 2 + 1
    assert ops.conv16_poolable(B, Tin, Cin, N) == (Tout == 32)
    y, pool = torch.empty(B, Tout, N, device="cuda"), torch.full((B, N), float("nan"), device="cuda")
    if Tout != 32:
        with pytest.raises((RuntimeError, ValueError)):
            ops.conv16_pool(x, wq, y, N, pool, 1.0 / Tout, bias=b, act=ops.ACT_LRELU)
        return
    ops.conv16_pool(x, wq, y, N, pool, 1.0 / Tout, bias=b, act=ops.ACT_LRELU)
    y2 = torch.empty_like(y)
    ops.conv16(x, wq, y2, N, False, bias=b, act=ops.ACT_LRELU)
    assert torch.equal(y, y2)
    torch.testing.assert_close(pool, y.double().mean(dim=1).float(), rtol=1e-5, atol=1e-6)
    ref = torch.randn(B, Tout, N, generator=g).cuda()
    ops.conv16_pool(x, wq, y, N, pool, 1.0 / Tout, gref=ref, gact=ops.ACT_LRELU)
    torch.testing.assert_close(pool, y.double().mean(dim=1).float(), rtol=1e-5, atol=1e-6)


def test_batchnorm_statistics_keep_their_digits_under_a_large_channel_offset(ops):
    """The partial statistics are centred per wave and combined by the parallel-variance rule: a channel whose mean is
    1000x its deviation (bias 300, deviation ~0.3) still gets invstd / running_var to fp32 accuracy.  (E[x^2] - mean^2
    from fp32 sums of squares loses ~1e-6 * mean^2 / var = 1 of the variance here.)"""
    B, L, Cin, Cout = 64, 32, 64, 64
    x, w = rnd(B, L, Cin, seed=1), rnd(Cin, Cout, 5, seed=2, scale=0.02)
    bias = torch.full((Cout,), 300.0)
    bias[::2] = -75.0
    gamma, beta = torch.ones(Cout), torch.zeros(Cout)
    wq = wq_of(ops, w, Cout, Cin, 5, Cout * 5)
    tb, rows = ops.conv16_plan(B, L, Cout, True)
    part = torch.full((3 * rows * Cout,), float("nan"), device="cuda")
    z = torch.empty(B, 2 * L, Cout, device="cuda")
    ops.conv16(x.cuda(), wq, z, Cout, True, bias=bias.cuda(), stats=part)
    a = torch.empty_like(z)
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    m, i = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    ops.bn_train_fwd_parts(part, rows, 1, z, a, gamma.cuda(), beta.cuda(), rm, rv, m, i, ops.ACT_NONE)
    z64 = z.cpu().double().view(-1, Cout)
    mean, var = z64.mean(0), z64.var(0, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 100
    torch.testing.assert_close(m.cpu().double(), mean, rtol=1e-6, atol=0)
    torch.testing.assert_close(i.cpu().double(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-5, atol=0)
    torch.testing.assert_close(rv.cpu().double(), 0.9 + 0.1 * z64.var(0, unbiased=True), rtol=2e-5, atol=0)
    ref = (z64 - mean) / torch.sqrt(var + 1e-5)
    assert float((a.cpu().double().view(-1, Cout) - ref).abs().max()) < 2e-3      # z itself carries 300 * 2^-24 of rounding per element


@pytest.mark.parametrize("B,Tin,Cin,N", [(64, 64, 128, 256), (3, 16, 32, 64)])
def test_permuted_output_order_equals_a_transpose(ops, B, Tin, Cin, N):
    """perm: the gather form writes (B, N, Tout) -- decoder.pre.2's (B, 256*L) output order behind view(B, 256, L)
    (src/gan/models.py:70) -- with the elementwise epilogue operand still in the dense (B, Tout, N) order."""
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, Tin, Cin, generator=g).cuda()
    w = (torch.randn(N, Cin, 5, generator=g) * 0.05).cuda()
    wq = wq_of(ops, w.cpu(), N, Cin, Cin * 5, 5)
    Tout = (Tin - 1) // 2 + 1
    gref = torch.randn(B, Tout, N, generator=g).cuda()
    dense = torch.empty(B, Tout, N, device="cuda")
    ops.conv16(x, wq, dense, N, False, gref=gref, gact=ops.ACT_RELU)
    perm = torch.full((B, N, Tout), float("nan"), device="cuda")
    ops.conv16(x, wq, perm, N, False, perm=True, gref=gref, gact=ops.ACT_RELU)
    assert torch.equal(perm, dense.permute(0, 2, 1).contiguous())


def test_gradient_penalty_interpolate_rides_in_the_producing_launch(ops):
    """mix: the launch that stores the fake batch also writes x_hat = alpha * real + (1 - alpha) * fake for the first
    mix_rows samples (src/gan/utils.py:76-79); equal to mg_gp_interp of the stored values; other rows untouched."""
    B, L, Cin, N, rows = 8, 16, 64, 128, 4
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, L, Cin, generator=g).cuda()
    w = torch.randn(Cin, N, 5, generator=g) * 0.05
    wq = wq_of(ops, w, N, Cin, 5, N * 5)
    bias = torch.randn(N, generator=g).cuda()
    real = torch.randn(rows, 2 * L, N, generator=g).cuda()
    alpha = torch.rand(rows, generator=g).cuda()
    y = torch.empty(B, 2 * L, N, device="cuda")
    out = torch.full((B, 2 * L, N), 7.0, device="cuda")
    ops.conv16(x, wq, y, N, True, bias=bias, mix=(real, alpha, out, rows))
    y2 = torch.empty_like(y)
    ops.conv16(x, wq, y2, N, True, bias=bias)
    assert torch.equal(y, y2)
    want = torch.empty(rows, 2 * L, N, device="cuda")
    ops.gp_interp(real, y[:rows].contiguous(), alpha, want)
    assert torch.equal(out[:rows], want) and bool((out[rows:] == 7.0).all())
    a = alpha.view(-1, 1, 1)
    torch.testing.assert_close(out[:rows], a * real + (1 - a) * y[:rows], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M,K,Cc,Lp", [(128, 512, 256, 32), (64, 512, 256, 32), (3, 40, 8, 4)])
def test_linear_with_permuted_output_columns(ops, M, K, Cc, Lp):
    """mg_linear_perm: the Linear output lands as (M, L, C) = view(M, C, L).permute(0, 2, 1) of the reference order
    (src/gan/models.py:70-73), bias and activation included, elementwise operands in the stored order."""
    g = torch.Generator().manual_seed(M)
    N = Cc * Lp
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = torch.empty(M, N, device="cuda")
    ops.linear_fwd(x, w, ref, bias=b, act=ops.ACT_RELU)
    y = torch.full((M, Lp, Cc), float("nan"), device="cuda")
    z = torch.full((M, Lp, Cc), float("nan"), device="cuda")
    ops.linear_fwd(x, w, y, perm_L=Lp, bias=b, act=ops.ACT_RELU, zout=z)
    if M >= 128:
        torch.testing.assert_close(y, ref.view(M, Cc, Lp).permute(0, 2, 1).contiguous(), rtol=1e-5, atol=1e-5)
    else:
        assert torch.equal(y, ref.view(M, Cc, Lp).permute(0, 2, 1).contiguous())
    ops.linear_fwd(x, w, ref, bias=b)
    if M >= 128:      # the permuted call took the 64x64-tile window GEMM (mg_conv_linear_perm), the plain one the skinny kernel
        assert ((z.double() - ref.view(M, Cc, Lp).permute(0, 2, 1).double()).norm() / ref.double().norm()).item() < 2e-6
    else:
        assert torch.equal(z, ref.view(M, Cc, Lp).permute(0, 2, 1).contiguous())
    want = (x.double() @ w.double().t() + b.double()).view(M, Cc, Lp).permute(0, 2, 1)
    assert ((z.double() - want).norm() / want.norm()).item() < 2e-6


@pytest.mark.parametrize("B,L,Cin,N", [(64, 128, 128, 64), (64, 64, 64, 128), (3, 16, 16, 32)])
def test_batchnorm_backward_sums_ride_in_the_data_gradient_launch(ops, B, L, Cin, N):
    """conv16(bnb=...): the launch that produces the gradient reaching a train-mode BatchNorm + ReLU layer also leaves the two
    column sums of that layer's backward; bn_train_bwd_parts (one launch) then equals bn_train_bwd (reduction + apply) -- and
    autograd in fp64."""
    g = torch.Generator().manual_seed(B + L + Cin + N)
    # gather form: y (B, L/2... ) -- use the generator's data-gradient shape: x (B, L, Cin) -> y (B, L//2, N) via conv16 gather
    x = torch.randn(B, L, Cin, generator=g).cuda()
    w = (torch.randn(N, Cin, 5, generator=g) * 0.05).cuda()
    wq = torch.zeros(N * Cin * 5, device="cuda")
    ops.wq_relayout(w, wq, N, Cin, 5, Cin * 5, 5)
    Tout = L // 2
    z = torch.randn(B, Tout, N, generator=g).cuda()
    gamma, beta = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    a = torch.empty_like(z)
    mean, invstd = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.bn_train_fwd(z, a, gamma, beta, None, None, mean, invstd, ops.ACT_RELU, 0.1, 1e-5)
    _, rows = ops.conv16_plan(B, L, N, False)
    part = torch.full((2 * rows * N,), float("nan"), device="cuda", dtype=torch.float64)
    dy = torch.empty(B, Tout, N, device="cuda")
    ops.conv16(x, wq, dy, N, False, bnb=(a, z, mean, invstd, part, ops.ACT_RELU))
    dy0 = torch.empty_like(dy)
    ops.conv16(x, wq, dy0, N, False)
    assert torch.equal(dy, dy0)                                  # the rider does not touch the output
    dz1, dg1, db1 = torch.empty_like(z), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.bn_train_bwd_parts(part, rows, dy, a, z, dz1, gamma, mean, invstd, dg1, db1, ops.ACT_RELU)
    dz2, dg2, db2 = torch.empty_like(z), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.bn_train_bwd(dy, a, z, dz2, gamma, mean, invstd, dg2, db2, ops.ACT_RELU)
    torch.cuda.synchronize()
    rel = lambda p_, q_: ((p_.double() - q_.double()).norm() / (q_.double().norm() + 1e-30)).item()  # noqa: E731
    assert rel(dz1, dz2) < 2e-6 and rel(dg1, dg2) < 2e-6 and rel(db1, db2) < 2e-6
    z64 = z.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yy = torch.relu(torch.nn.functional.batch_norm(z64.reshape(-1, N), None, None, g64, b64, True, 0.1, 1e-5)).reshape(B, Tout, N)
    (yy * dy.double()).sum().backward()
    assert rel(dz1, z64.grad) < 2e-5 and rel(dg1, g64.grad) < 2e-5 and rel(db1, b64.grad) < 2e-5
