"""BASELINE.json's full size (cfg2: B=64, T=256, C=128) on the GPU: one full D-step + G-step against the oracle
(which finishes such a step in well under a second on the box's host cores), linearity of the window GEMM, and
bitwise agreement of hipGraph replay with eager launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402

B, T, C = 64, 256, 128


def rel_err(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_update_matches(P, old, P_ref, grads_ref, lr, skip=()):
    """First Adam step at full size, element by element.  The update is -lr * g / (|g| + eps): wherever the gradient
    is well-conditioned ours and the oracle's must agree to a fraction of lr; elsewhere it only has to stay within
    the +-lr any Adam step makes.  Well-conditioned = |g| at least 30x the per-element error the gradient checks above
    tolerate (3e-3 relative L2, i.e. 3e-3 * rms(g) per element): smaller elements may legitimately differ in sign."""
    for k, v_ref in P_ref.items():
        if k in skip:
            continue
        upd, upd_ref = P[k].detach().cpu() - old[k], v_ref - old[k]
        gref = grads_ref[k]
        mask = gref.abs() >= 30 * 3e-3 * gref.pow(2).mean().sqrt()
        assert float(mask.float().mean()) > 0.2, (k, float(mask.float().mean()))
        worst = float((upd - upd_ref)[mask].abs().max())
        assert worst <= 1e-3 * lr, (k, worst / lr)
        assert float(upd.abs().max()) <= 1.001 * lr, k


@pytest.fixture(scope="module")
def setup():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.engine import GanEngine
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "weights_init", seed=3)
    # a critic away from initialisation (x8): non-trivial LeakyReLU masks and a penalty far from 1
    for k in S.PD:
        S.PD[k].mul_(8.0 if k.endswith("weight") else 1.0)
    eng = GanEngine(cfg, ed_cfg, "cuda", B)
    eng.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
    batch = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, 42)
    real, numeric, latent, emot = batch
    eng.set_batch(real.cuda(), numeric.cuda(), latent.cuda(), emot.cuda())
    R = O.step_randoms(B, cfg["NOISE_DIM"], seed=9)
    return S, eng, cfg, batch, R


def test_full_size_step_matches_oracle(setup):
    S, eng, cfg, (real, numeric, latent, emot), R = setup
    torch.set_num_threads(16)
    eng.set_randoms(R["noise_d"].cuda(), [m.cuda() for m in R["dm_d"]], R["alpha"].cuda())
    eng.d_backward()
    rd = O.d_step(S, real, latent, numeric, R["noise_d"], R["alpha"], R["dm_d"])
    assert abs(eng.loss_d_out[0].item() - rd["loss_d"].item()) <= 2e-4 * abs(rd["loss_d"].item())
    assert abs(eng.gp.item() - rd["gp"].item()) <= 2e-4 * abs(rd["gp"].item())
    torch.testing.assert_close(eng.fake_d.cpu(), rd["fake"], rtol=1e-3, atol=1e-5)
    for k, g in rd["grads"].items():
        got = eng.D.g[k]
        if k == "real_fake.bias":
            continue
        if k == "real_fake.weight":
            got, g = got[:, :256], g[:, :256]
        assert rel_err(got, g) < 1e-3, (k, rel_err(got, g))
    d_old = {k: v.detach().cpu().clone() for k, v in eng.D.p.items()}
    eng.d_update()
    assert_update_matches(eng.D.p, d_old, S.PD, rd["grads"], eng.lr_d, skip=("real_fake.bias", "real_fake.weight"))
    with torch.no_grad():
        for k, v in S.PD.items():
            eng.D.p[k].copy_(v)                          # teacher forcing (tests/test_engine_gpu.py)
        eng.params_changed()
    eng.set_randoms(R["noise_g"].cuda(), [m.cuda() for m in R["dm_g"]])
    eng.g_backward()
    d64 = lambda P: type(P)((k, v.double().clone()) for k, v in P.items())  # noqa: E731
    S64 = O.GanState(S.cfg, S.ed_cfg, d64(S.PE), d64(S.PG), d64(S.BG), d64(S.PD), d64(S.PED), d64(S.BED))
    rg64 = O.g_step(S64, latent.double(), numeric.double(), emot, R["noise_g"].double(), [m.double() for m in R["dm_g"]])
    rg = O.g_step(S, latent, numeric, emot, R["noise_g"], R["dm_g"])
    assert abs(eng.adv.item() - rg["loss_g_adv"].item()) <= 2e-4 * max(1.0, abs(rg["loss_g_adv"].item()))
    assert abs(eng.emo.item() - rg["loss_g_emo"].item()) <= 2e-4
    torch.testing.assert_close(eng.notes.cpu(), rg["fake"], rtol=1e-3, atol=1e-5)
    for k, g in rg["grads"].items():
        if k in ("G.decoder.deconv.0.bias", "G.decoder.deconv.3.bias"):
            continue
        # two train-mode BatchNorms amplify fp32 rounding: judge against the fp64 truth, relative to the
        # reference's own fp32 error (same criterion as tests/test_engine_gpu.py)
        # + 3e-3: at this size a handful of the 10^6 LeakyReLU/ReLU pre-activations lie within fp32 rounding of the
        # kink, and ONE flipped mask (observed: z = -1.1e-6 in fp64, +1.3e-6 here) already moves the L2 error of the
        # critic's input gradient to 1e-3 -- a different, equally valid subgradient, not an arithmetic error.
        e_mine, e_ref = rel_err(eng.GE.g[k], rg64["grads"][k]), rel_err(g, rg64["grads"][k])
        assert e_mine <= 8 * e_ref + 3e-3, (k, e_mine, e_ref)
    ge_old = {k: v.detach().cpu().clone() for k, v in eng.GE.p.items()}
    eng.g_update()
    assert_update_matches(eng.GE.p, ge_old, S.PGE, rg["grads"], eng.lr_g,
                          skip=("G.decoder.deconv.0.bias", "G.decoder.deconv.3.bias"))


# (No finite-difference test of the critic gradient: the WGAN-GP penalty of a LeakyReLU critic is DIScontinuous in
#  theta -- grad_x D is piecewise constant -- so at this size any usable step crosses hundreds of kinks; the
#  hand-derived double backward is instead pinned against autograd's create_graph path, above and at small shapes.)


def test_window_gemm_linearity_and_graph_replay(setup):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    S, eng, cfg, batch, R = setup
    gen = torch.Generator(device="cuda").manual_seed(1)
    x1 = torch.randn(B, T, C, device="cuda", generator=gen)
    x2 = torch.randn(B, T, C, device="cuda", generator=gen)
    w = eng.ED.p["encoder.conv.0.net.0.weight"]
    y1, y2, y3 = (torch.empty(B, T, 64, device="cuda") for _ in range(3))
    ops.conv1d_fwd(x1, w, y1, 1)
    ops.conv1d_fwd(x2, w, y2, 1)
    ops.conv1d_fwd(2.0 * x1 - 3.0 * x2, w, y3, 1)
    assert rel_err(y3, 2.0 * y1 - 3.0 * y2) < 1e-5
    # graph replay == eager, bit for bit, at full size
    from melo_gan_amd.gan.engine import GanEngine
    e2 = GanEngine(cfg, O.default_ed_cfg(C), "cuda", B)
    for e in (eng, e2):
        e.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
        e.D.m.zero_(); e.D.v.zero_(); e.D.state.zero_(); e.GE.m.zero_(); e.GE.v.zero_(); e.GE.state.zero_()
        e.seed(77)
    e2.set_batch(*(t.cuda() for t in batch))
    with torch.cuda.stream(e2.stream):
        for _ in range(3):
            for e, graph in ((eng, False), (e2, True)):
                e.run("d_backward_rng", graph)
                e.run("d_update", graph)
                e.run("g_backward_rng", graph)
                e.run("g_update", graph)
        torch.cuda.synchronize()
    assert torch.equal(eng.D.data, e2.D.data) and torch.equal(eng.GE.data, e2.GE.data)
    assert torch.isfinite(eng.GE.data).all() and torch.isfinite(eng.loss_d_out).all()
