"""The flows that bench.py and the trainers actually replay, at the sizes that are benchmarked, against the oracle.

 * cfg2 (B=64, T=256, C=128): the production sequence of `dg_step_rng` -- ONE 2B-row generator pass
   (`mg_bn_train_fwd_groups` / conv16 statistics epilogue at 128 rows, conv.4 with the fused temporal mean, the 3B-row
   tile plans) + critic step + generator step -- with injected randoms: teacher-forced at the critic update (every
   quantity judged from identical inputs), and the `split` / `ingraph` side-stream flows free-running (the engine's own
   critic update feeds its generator step) against the oracle's two consecutive sub-steps.
 * cfg4 (B=256, T=256, C=4): one VAE step against `O.ae_step`.
 * f-2 at the cfg2 shape (B=64, T=256, C=128): one emotion-discriminator pre-training step against `O.ed_step`.

Reference: src/gan/train_gan.py:183-251, src/ae/train_ae.py:110-122, src/emotion_discriminator/train_ed.py:51-82.
Updates are checked element by element (`assert_update_matches`), not by a relative L2 bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402

B, T, C = 64, 256, 128
NOISE_PARAMS_G = ("G.decoder.deconv.0.bias", "G.decoder.deconv.3.bias")


def rel_err(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_update_matches(P, old, P_ref, grads_ref, lr, skip=(), what="", wd=0.0, min_frac=0.2):
    """First Adam(W) step, element by element.  The update is -lr * g / (|g| + eps) (- lr * wd * p): wherever the
    gradient is well-conditioned ours and the oracle's must agree to 1e-3 of lr; elsewhere it only has to stay within
    the +-lr any first Adam step makes.  Well-conditioned = |g| at least 30x the per-element error the gradient checks
    tolerate (3e-3 relative L2, i.e. 3e-3 * rms(g) per element): smaller elements may legitimately differ in sign."""
    for k, v_ref in P_ref.items():
        if k in skip:
            continue
        upd, upd_ref = P[k].detach().cpu() - old[k], v_ref - old[k]
        gref = grads_ref[k]
        mask = gref.abs() >= 30 * 3e-3 * gref.pow(2).mean().sqrt()
        assert float(mask.float().mean()) > min_frac, (what, k, float(mask.float().mean()))
        worst = float((upd - upd_ref)[mask].abs().max())
        assert worst <= 1e-3 * lr, (what, k, worst / lr)
        assert float(upd.abs().max()) <= (1.001 + wd * float(old[k].abs().max())) * lr, (what, k)


def d64(P):
    return type(P)((k, v.double().clone()) for k, v in P.items())


def as_f64(S):
    return O.GanState(S.cfg, S.ed_cfg, d64(S.PE), d64(S.PG), d64(S.BG), d64(S.PD), d64(S.PED), d64(S.BED))


def clone_state(S):
    c = lambda P: type(P)((k, v.clone()) for k, v in P.items())  # noqa: E731
    return O.GanState(S.cfg, S.ed_cfg, c(S.PE), c(S.PG), c(S.BG), c(S.PD), c(S.PED), c(S.BED))


def fresh(seed_state=3, d_scale=8.0):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.engine import GanEngine
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "weights_init", seed=seed_state)
    for k in S.PD:      # a critic away from initialisation: non-trivial LeakyReLU masks, a penalty far from 1
        S.PD[k].mul_(d_scale if k.endswith("weight") else 1.0)
    eng = GanEngine(cfg, ed_cfg, "cuda", B)
    eng.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
    batch = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, 42)
    eng.set_batch(*(t.cuda() for t in batch))
    R = O.step_randoms(B, cfg["NOISE_DIM"], seed=9)
    eng.set_randoms(R["noise_d"].cuda(), [m.cuda() for m in R["dm_d"]], R["alpha"].cuda())      # critic-step half
    eng.set_randoms(R["noise_g"].cuda(), [m.cuda() for m in R["dm_g"]])                         # generator-step half
    return S, eng, cfg, batch, R


def check_d_side(eng, rd, tol_loss=2e-4):
    assert abs(eng.loss_d_out[0].item() - rd["loss_d"].item()) <= tol_loss * abs(rd["loss_d"].item())
    assert abs(eng.gp.item() - rd["gp"].item()) <= tol_loss * abs(rd["gp"].item())
    torch.testing.assert_close(eng.fake_d.cpu(), rd["fake"], rtol=1e-3, atol=1e-5)
    for k, g in rd["grads"].items():
        got = eng.D.g[k]
        if k == "real_fake.bias":
            continue                                 # +1/B and -1/B cancel: rounding noise on both sides
        if k == "real_fake.weight":
            got, g = got[:, :256], g[:, :256]        # the embedding half cancels the same way
        assert rel_err(got, g) < 1e-3, (k, rel_err(got, g))


def test_fused_production_flow_full_size_matches_oracle():
    """dg_step_rng's launch sequence (minus the draw: the randoms are the oracle's), teacher-forced at the critic update."""
    S, eng, cfg, (real, numeric, latent, emot), R = fresh()
    torch.set_num_threads(16)
    bg_old = {k: v.clone() for k, v in S.BG.items()}
    eng.dg_forward()                       # ONE pass over 2B rows: both fake batches, both embeddings
    eng.d_backward(forward=False)
    rd = O.d_step(S, real, latent, numeric, R["noise_d"], R["alpha"], R["dm_d"])
    check_d_side(eng, rd)
    torch.testing.assert_close(eng.emb_d.cpu(), rd["emb"], rtol=1e-4, atol=1e-6)
    d_old = {k: v.detach().cpu().clone() for k, v in eng.D.p.items()}
    d_old_ref = {k: d_old[k] for k in d_old}
    eng.d_update()
    assert_update_matches(eng.D.p, d_old_ref, S.PD, rd["grads"], eng.lr_d, skip=("real_fake.bias", "real_fake.weight"), what="D")
    with torch.no_grad():                  # teacher forcing (tests/test_engine_gpu.py): the generator step sees the oracle's critic
        for k, v in S.PD.items():
            eng.D.p[k].copy_(v)
        eng.params_changed()
    eng.g_backward_a2()
    eng.g_backward_b()
    S64 = as_f64(S)
    rg64 = O.g_step(S64, latent.double(), numeric.double(), emot, R["noise_g"].double(), [m.double() for m in R["dm_g"]])
    rg = O.g_step(S, latent, numeric, emot, R["noise_g"], R["dm_g"])
    assert abs(eng.adv.item() - rg["loss_g_adv"].item()) <= 2e-4 * max(1.0, abs(rg["loss_g_adv"].item()))
    assert abs(eng.emo.item() - rg["loss_g_emo"].item()) <= 2e-4
    torch.testing.assert_close(eng.notes.cpu(), rg["fake"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(eng.logits.cpu(), rg["logits"], rtol=1e-3, atol=1e-5)
    for k, g in rg["grads"].items():
        if k in NOISE_PARAMS_G:
            continue
        # two train-mode BatchNorms amplify fp32 rounding: judged against the fp64 truth relative to the reference's own
        # fp32 error; + 3e-3 for the handful of LeakyReLU / ReLU pre-activations within rounding of the kink
        # (tests/test_fullsize_gpu.py explains the criterion)
        e_mine, e_ref = rel_err(eng.GE.g[k], rg64["grads"][k]), rel_err(g, rg64["grads"][k])
        assert e_mine <= 8 * e_ref + 3e-3, (k, e_mine, e_ref)
    # running statistics moved twice (critic-step half first), as two consecutive forward calls move them
    for k, v in S.BG.items():
        if k.endswith("num_batches_tracked"):
            continue
        assert rel_err(eng.Gbuf[k], v) < 1e-4 and rel_err(bg_old[k], v) > 1e-3, k
    assert eng.num_batches_tracked == 2
    ge_old = {k: v.detach().cpu().clone() for k, v in eng.GE.p.items()}
    eng.g_update()
    assert_update_matches(eng.GE.p, ge_old, S.PGE, rg["grads"], eng.lr_g, skip=NOISE_PARAMS_G, what="GE")


@pytest.mark.parametrize("flow", ["split", "ingraph"])
def test_side_stream_flows_full_size_match_oracle(flow):
    """The split flow (emotion branch on the side stream beside the critic step) and the in-graph fork, launched as the
    production step launches them but without the draw, FREE-RUNNING: the engine's own critic update feeds its generator
    step.  Against the oracle's d_step followed by g_step.  The generator step then starts from a critic that differs from
    the oracle's in the elements whose first Adam step is a coin toss (|g| at rounding level: +-lr either way, in the
    reference as well), hence the looser bounds on what depends on the updated critic: adv 1e-3, gradients + 1e-2."""
    S, eng, cfg, (real, numeric, latent, emot), R = fresh()
    torch.set_num_threads(16)
    d_old = {k: v.detach().cpu().clone() for k, v in eng.D.p.items()}
    ge_old = {k: v.detach().cpu().clone() for k, v in eng.GE.p.items()}
    side = eng.ed_side
    assert side is not None
    with torch.cuda.stream(eng.stream):
        cur = torch.cuda.current_stream()
        if flow == "split":
            eng.dg_forward()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                eng.g_ed_branch_side()
            eng.d_step_g_critic_front()
            cur.wait_stream(side)
            eng.g_finish()
        else:
            eng.dg_fork_step_rng(draw=False)
        torch.cuda.synchronize()
    rd = O.d_step(S, real, latent, numeric, R["noise_d"], R["alpha"], R["dm_d"])
    check_d_side(eng, rd)
    assert_update_matches(eng.D.p, d_old, S.PD, rd["grads"], eng.lr_d, skip=("real_fake.bias", "real_fake.weight"), what="D")
    S64 = as_f64(S)
    rg64 = O.g_step(S64, latent.double(), numeric.double(), emot, R["noise_g"].double(), [m.double() for m in R["dm_g"]])
    rg = O.g_step(S, latent, numeric, emot, R["noise_g"], R["dm_g"])
    assert abs(eng.adv.item() - rg["loss_g_adv"].item()) <= 1e-3 * max(1.0, abs(rg["loss_g_adv"].item()))
    assert abs(eng.emo.item() - rg["loss_g_emo"].item()) <= 2e-4
    torch.testing.assert_close(eng.notes.cpu(), rg["fake"], rtol=1e-3, atol=1e-5)
    for k, g in rg["grads"].items():
        if k in NOISE_PARAMS_G:
            continue
        e_mine, e_ref = rel_err(eng.GE.g[k], rg64["grads"][k]), rel_err(g, rg64["grads"][k])
        assert e_mine <= 8 * e_ref + 1e-2, (k, e_mine, e_ref)
    # the update: elementwise wherever the gradient is well-conditioned; "well-conditioned" has to clear the 1e-2 above
    for k, v_ref in S.PGE.items():
        if k in NOISE_PARAMS_G:
            continue
        upd, upd_ref = eng.GE.p[k].detach().cpu() - ge_old[k], v_ref - ge_old[k]
        gref = rg["grads"][k]
        mask = gref.abs() >= 30 * 1e-2 * gref.pow(2).mean().sqrt()
        if float(mask.float().mean()) > 0.02:
            assert float((upd - upd_ref)[mask].abs().max()) <= 1e-3 * eng.lr_g, (k, float((upd - upd_ref)[mask].abs().max()) / eng.lr_g)
        assert float(upd.abs().max()) <= 1.001 * eng.lr_g, k


def test_production_graphs_full_size_equal_eager_and_each_other(monkeypatch):
    """dg_step_rng replayed == eager, and the default flow (the forked graph: emotion branch on the side stream) and the
    split flow's four graphs == the one graph, bit for bit, at cfg2."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.engine import GanEngine
    from melo_gan_amd.gan.dp import DataParallel
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "weights_init", seed=3)
    batch = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, 42)
    engs = [GanEngine(cfg, ed_cfg, "cuda", B) for _ in range(4)]
    for e in engs:
        e.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
        e.seed(77)
    e_eager, e_graph, e_fork, e_split = engs
    dp_fork = DataParallel(e_fork, 1, None)
    monkeypatch.setenv("MELO_ED_FLOW", "split")
    dp_split = DataParallel(e_split, 1, None)
    monkeypatch.delenv("MELO_ED_FLOW")
    assert dp_fork._ed_flow == "ingraph" and dp_split._ed_flow == "split"
    with torch.cuda.stream(e_graph.stream):
        for e in engs:
            e.set_batch(*(t.cuda() for t in batch))
        for _ in range(4):
            e_eager.run("dg_step_rng", False)
            e_graph.run("dg_step_rng", True)
            dp_fork.step(True)
            dp_split.step(True)
        torch.cuda.synchronize()
    assert "dg_fork_step_rng" in e_fork._graphs                          # the forked graph ran
    assert any(k.startswith("g_finish") for k in e_split._graphs)        # the split flow ran
    for o in (e_graph, e_fork, e_split):
        assert torch.equal(e_eager.D.data, o.D.data) and torch.equal(e_eager.GE.data, o.GE.data)
        assert torch.equal(e_eager.notes, o.notes) and torch.equal(e_eager.loss_d_out, o.loss_d_out)
    assert torch.isfinite(e_eager.GE.data).all() and torch.isfinite(e_eager.loss_d_out).all()


# ----------------------------------------------------------------------------------------------------------------------
# cfg4: the VAE step at the size bench.py --workload ae times (B=256, T=256, C=4)
# ----------------------------------------------------------------------------------------------------------------------
VAE_PRE_BN_BIAS = {"encoder.conv.0.bias", "encoder.conv.3.bias", "encoder.conv.6.bias",
                   "decoder.deconv.0.bias", "decoder.deconv.3.bias"}


def test_vae_step_cfg4_size_matches_oracle():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.ae.engine import VaeEngine
    Bv, Tv, L = 256, 256, 8
    torch.set_num_threads(16)
    spec, bufs = O.vae_spec(Tv, L)
    gen = torch.Generator().manual_seed(11)
    P = type(spec)()
    for k, s in spec.items():               # torch-default-like init (train_ae.py uses no weights_init), BatchNorm affine 1 / 0
        if len(s) == 1:
            P[k] = torch.ones(s) if (k.endswith("weight")) else torch.zeros(s)
        else:
            fan_in = s[1] * (s[2] if len(s) == 3 else 1)
            P[k] = (torch.rand(s, generator=gen) * 2 - 1) / fan_in ** 0.5
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    eng = VaeEngine(dict(MAX_NOTES=Tv, LATENT_DIM=L, BATCH_SIZE=Bv, LR=1e-4, WEIGHT_DECAY=1e-5), "cuda", Bv)
    assert list(eng.P.spec) == list(spec)
    eng.load_state(P, Bf)
    opt = O.AdamState(P, 1e-4, (0.9, 0.999), 1e-8, weight_decay=1e-5, decoupled=True)
    x = torch.rand(Bv, Tv, 4, generator=gen) * 2 - 1
    eps = torch.randn(Bv, L, generator=gen)
    eng.x.copy_(x.cuda())
    eng.eps.copy_(eps.cuda())
    eng.forward(True)
    eng.backward(10.0)
    P64 = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
    rec64, _, mu64, lv64 = O.vae_fwd(P64, {k: v.double().clone() for k, v in Bf.items()}, x.double(), eps.double(), Tv, True)
    l64, _, _ = O.vae_loss(rec64, x.double(), mu64, lv64, 10.0)
    g64 = dict(zip(P64, torch.autograd.grad(l64, list(P64.values()))))
    old = {k: v.clone() for k, v in P.items()}
    r = O.ae_step(P, Bf, opt, x, eps, 10.0, Tv)
    loss = eng.loss.cpu()
    assert abs(loss[0].item() - r["loss"].item()) <= 2e-4 * abs(r["loss"].item())
    torch.testing.assert_close(eng.recon.cpu(), r["recon"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(eng.mu.cpu(), r["mu"], rtol=1e-3, atol=1e-5)
    gtot = torch.sqrt(sum(v.double().pow(2).sum() for v in r["grads"].values()))
    for k in spec:
        if k in VAE_PRE_BN_BIAS:
            continue
        e_mine, e_ref = rel_err(eng.P.g[k], g64[k]), rel_err(r["grads"][k], g64[k])
        assert e_mine <= 8 * e_ref + 1e-4, (k, e_mine, e_ref)
    eng.update()
    assert abs(eng.gnorm[0].item() - float(gtot)) <= 1e-3 * float(gtot)
    # clip_grad_norm_(1.0) rescales every gradient by the same factor: Adam's first step is invariant to it
    assert_update_matches(eng.P.p, old, P, r["grads"], 1e-4, skip=VAE_PRE_BN_BIAS, what="VAE", wd=1e-5, min_frac=0.05)
    for k in Bf:
        if not k.endswith("num_batches_tracked"):
            assert rel_err(eng.buf[k], Bf[k]) < 1e-4, k


# ----------------------------------------------------------------------------------------------------------------------
# f-2: the emotion-discriminator pre-training step at the cfg2 shape (what bench.py --workload ed times)
# ----------------------------------------------------------------------------------------------------------------------
def test_ed_train_step_cfg2_size_matches_oracle():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    torch.set_num_threads(16)
    ed_cfg = dict(O.default_ed_cfg(C), dropout=0.2)
    spec, bufs = O.emotion_disc_spec(ed_cfg)
    gen = torch.Generator().manual_seed(21)
    P = type(spec)()
    for k, s in spec.items():
        if len(s) == 1:
            P[k] = torch.ones(s) if (k.endswith("weight") and ".net.1." in k) else torch.zeros(s)
        else:
            fan_in = s[1] * (s[2] if len(s) == 3 else 1)
            P[k] = (torch.rand(s, generator=gen) * 2 - 1) / fan_in ** 0.5
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    wd = 0.01
    cfg = dict(ed_cfg, batch_size=B, max_notes=T, optimizer=dict(name="AdamW", lr=2e-4, betas=[0.5, 0.999], weight_decay=wd))
    eng = EdEngine(cfg, "cuda", B, T)
    assert list(eng.P.spec) == list(spec)
    eng.load_state(P, Bf)
    opt = O.AdamState(P, 2e-4, (0.5, 0.999), 1e-8, weight_decay=wd, decoupled=True)
    x = torch.rand(B, T, C, generator=gen) * 2 - 1
    y = torch.randint(0, 4, (B,), generator=gen)
    mh = tuple(ed_cfg.get("mlp_hidden", (256, 128)))
    dm = [(torch.rand(B, h, generator=gen) >= 0.2).float() / 0.8 for h in mh]
    eng.set_batch(x.cuda(), y.cuda())
    eng.set_masks([m.cuda() for m in dm])
    eng.backward()
    P64 = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
    B64 = {k: v.double().clone() for k, v in Bf.items()}
    l64 = torch.nn.functional.cross_entropy(O.emotion_disc_fwd(P64, B64, x.double(), ed_cfg, True, [m.double() for m in dm]), y)
    g64 = dict(zip(P64, torch.autograd.grad(l64, list(P64.values()))))
    old = {k: v.clone() for k, v in P.items()}
    r = O.ed_step(P, Bf, opt, x, y, ed_cfg, dm)
    assert abs(eng.loss.item() - r["loss"].item()) <= 2e-4 * abs(r["loss"].item())
    torch.testing.assert_close(eng.logits.cpu(), r["logits"], rtol=1e-3, atol=1e-5)
    pre_bn_bias = tuple(f"encoder.conv.{i}.net.0.bias" for i in range(4))
    for k in spec:
        if k in pre_bn_bias:
            continue
        e_mine, e_ref = rel_err(eng.P.g[k], g64[k]), rel_err(r["grads"][k], g64[k])
        assert e_mine <= 8 * e_ref + 3e-4, (k, e_mine, e_ref)
    eng.update()
    assert_update_matches(eng.P.p, old, P, r["grads"], 2e-4, skip=pre_bn_bias, what="ED", wd=wd, min_frac=0.05)
    for k in Bf:
        if not k.endswith("num_batches_tracked"):
            assert rel_err(eng.buf[k], Bf[k]) < 1e-4, k
