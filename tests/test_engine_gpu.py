"""Full-step parity of the HIP engine against the oracle and the reference-generated golden
fixtures: critic step (incl. the hand-derived gradient-penalty double backward), generator
step (through the frozen emotion discriminator), Adam, BatchNorm running stats, eval-mode
generation.  GPU only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402  (the checker)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# Parameters whose gradient is mathematically zero (rounding noise amplified by Adam to +-lr
# per step, in the reference as well): conv biases feeding a train-mode BatchNorm, and the
# embedding half / bias of the critic head in the D-step (the +1/B and -1/B terms cancel).
NOISE_PARAMS_G = ("G.decoder.deconv.0.bias", "G.decoder.deconv.3.bias")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def engine_mod():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan import engine
    return engine


def make(engine_mod, g, use_graph=False):
    B, T, C = int(g["B"]), int(g["T"]), int(g["C"])
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    cfg["INTEGRATION_MODE"] = str(g["mode"])
    ed_cfg["input_mode"] = str(g["ed_mode"])
    S = O.build_gan_state(cfg, ed_cfg, "closed_form", d_scale=float(g["d_scale"]))
    eng = engine_mod.GanEngine(cfg, ed_cfg, "cuda", B)
    eng.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
    real, numeric, latent, emot = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, int(g["seed"]))
    if cfg["INTEGRATION_MODE"] == "conditioning":
        latent = O.closed_form((B, cfg["LATENT_DIM"]), 9.0, 0.5)
    eng.set_batch(real.cuda(), numeric.cuda(), latent.cuda(), emot.cuda())
    return S, eng, cfg, (real, numeric, latent, emot)


def as_f64(S, cfg_):
    """fp64 copy of the oracle state: the 'exact' answer both fp32 implementations are judged against."""
    d = lambda P: type(P)((k, v.double().clone()) for k, v in P.items())  # noqa: E731
    return O.GanState(S.cfg, S.ed_cfg, d(S.PE), d(S.PG), d(S.BG), d(S.PD), d(S.PED), d(S.BED))


def grad_ok(got, ref32, ref64, slack=4.0, floor=2e-5):
    """HIP fp32 result may deviate from the fp64 truth at most `slack` x as much as the reference's own
    fp32 arithmetic (PyTorch-CPU) does, plus a small floor."""
    if float(ref64.abs().max()) < 1e-8 and float(got.detach().abs().max().cpu()) < 1e-8:
        return True, (0.0, 0.0)      # exactly-cancelling gradient (+1/B and -1/B terms): any summation order's residue
    e_mine, e_ref = rel_err(got, ref64), rel_err(ref32, ref64)
    return e_mine <= slack * e_ref + floor, (e_mine, e_ref)


def rel_err(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


CASES = ["gan_c4_t32_b4", "gan_c4_t32_b4_bigD", "gan_c128_t64_b4", "gan_c4_t20_b3", "gan_c4_t16_cond_lat"]


@pytest.mark.parametrize("name", CASES)
def test_steps_match_oracle_and_golden(engine_mod, name):
    g = load(name)
    S, eng, cfg, (real, numeric, latent, emot) = make(engine_mod, g)
    for it in range(int(g["n_steps"])):
        dm_d = [torch.from_numpy(g[f"s{it}.dm_d{j}"]).float() for j in range(2)]
        dm_g = [torch.from_numpy(g[f"s{it}.dm_g{j}"]).float() for j in range(2)]
        noise_d, alpha = torch.from_numpy(g[f"s{it}.noise_d"]), torch.from_numpy(g[f"s{it}.alpha"])
        noise_g = torch.from_numpy(g[f"s{it}.noise_g"])
        # ---- D-step ----
        eng.set_randoms(noise_d.cuda(), [m.cuda() for m in dm_d], alpha.cuda())
        eng.d_backward()
        rd64 = O.d_step(as_f64(S, cfg), real.double(), latent.double(), numeric.double(), noise_d.double(),
                        alpha.double(), [m.double() for m in dm_d])
        rd = O.d_step(S, real, latent, numeric, noise_d, alpha, dm_d)
        ld = eng.loss_d_out.cpu()
        # golden (reference) scalars: fp32 single-step tolerance rtol 1e-4
        assert abs(ld[0].item() - float(g[f"s{it}.loss_d"])) <= 1e-4 * abs(float(g[f"s{it}.loss_d"]))
        assert abs(eng.gp.item() - float(g[f"s{it}.gp"])) <= 1e-4
        B = eng.B
        np.testing.assert_allclose(eng.s[B:2 * B].cpu().numpy(), g[f"s{it}.d_real"], rtol=1e-4, atol=2e-6)
        # d_fake inherits the generator's BatchNorm conditioning (see the fake_d comment below)
        np.testing.assert_allclose(eng.s[2 * B:].cpu().numpy(), g[f"s{it}.d_fake"], rtol=5e-3, atol=2e-5)
        ok, errs = grad_ok(eng.s[2 * B:], rd["d_fake"], rd64["d_fake"], slack=8.0, floor=2e-6)
        assert ok, ("d_fake", errs)
        # (slices of the weight gradients: tolerance relative to the tensor's mean |g| -- individual small
        #  elements are differences of large cancelling terms)
        if it == 0:
            # train-mode BatchNorm over a handful of rows amplifies fp32 rounding (invstd up to 316 per
            # layer): against the reference's fp32 output allow 2e-3 / 2e-5, and require that we are no
            # further from the fp64 truth than a few times the reference's own fp32 arithmetic is.
            np.testing.assert_allclose(eng.fake_d.cpu().numpy(), g["s0.fake_d"], rtol=2e-3, atol=2e-5)
            ok, errs = grad_ok(eng.fake_d, rd["fake"], rd64["fake"], slack=8.0, floor=2e-6)
            assert ok, ("fake_d", errs)
            np.testing.assert_allclose(eng.D.g["conv.0.weight"][:4].cpu().numpy(), g["s0.dgrad_conv0_w"], rtol=1e-3,
                                       atol=5e-2 * float(g["s0.dgrad.conv.0.weight"][2]) / eng.D.p["conv.0.weight"].numel())
            np.testing.assert_allclose(eng.D.g["conv.4.weight"][:2].cpu().numpy(), g["s0.dgrad_conv4_w"], rtol=1e-3,
                                       atol=5e-2 * float(g["s0.dgrad.conv.4.weight"][2]) / eng.D.p["conv.4.weight"].numel())
            np.testing.assert_allclose(eng.D.g["fc.1.weight"][:4].cpu().numpy(), g["s0.dgrad_fc1_w"], rtol=1e-3,
                                       atol=5e-2 * float(g["s0.dgrad.fc.1.weight"][2]) / eng.D.p["fc.1.weight"].numel())
        for k, gr in rd["grads"].items():
            if k.endswith("bias") and k != "fc.1.bias" and not k.startswith("conv"):
                continue                     # real_fake.bias: exact cancellation noise
            got, g64 = eng.D.g[k], rd64["grads"][k]
            if k == "real_fake.weight":      # embedding half cancels to rounding noise
                got, gr, g64 = got[:, :256], gr[:, :256], g64[:, :256]
            ok, errs = grad_ok(got, gr, g64)
            assert ok, (it, k, errs)
        d_old = eng.D.data.clone()
        eng.d_update()
        # The embedding half of the critic head and its bias receive +1/B and -1/B contributions that
        # cancel to rounding noise, which Adam turns into +-lr steps (in the reference too).  Glue those
        # chaotic entries to the oracle's so that the G-step compares like with like.
        # Everything else must agree after the Adam update; then the critic is re-synchronised
        # ("teacher forcing") because the generator gradient is ill-conditioned in these tiny fixtures
        # (two train-mode BatchNorms over <= 64 rows: condition number ~1e5, the reference's own fp32
        # result is only good to ~2e-3 against fp64) -- each step is judged from identical inputs.
        # Adam normalises every element's step to ~lr, so elements whose gradient is rounding noise move
        # by +-lr either way: judge the UPDATE vectors (L2), not individual elements.
        for k, v in S.PD.items():
            o, n = eng.D.offsets[k]
            old = d_old[o:o + n].view(v.shape).cpu()
            upd, upd_ref = eng.D.p[k].cpu() - old, v - old
            if k in ("real_fake.bias", "fc.1.bias"):
                continue     # +1/B and -1/B contributions cancel (exactly, when all fc masks are 1): Adam-amplified noise
            if k == "real_fake.weight":
                upd, upd_ref = upd[:, :256], upd_ref[:, :256]
            if float(upd_ref.abs().max()) == 0.0:      # gradient cancels EXACTLY in the reference (e.g. fc.1.bias when
                assert float(upd.abs().max()) <= eng.lr_d    # every mask is 1): ours may carry a rounding residue
                continue
            # 0.1: cancellation-dominated gradients (e.g. conv biases: sum dz_fake - sum dz_real) carry
            # O(10 %) elementwise noise in BOTH fp32 implementations, and Adam maps every element to ~lr.
            assert rel_err(upd, upd_ref) < 0.1, (it, "D update", k, rel_err(upd, upd_ref))
        with torch.no_grad():
            for k, v in S.PD.items():
                eng.D.p[k].copy_(v)
            eng.params_changed()
        # ---- G-step ----
        eng.set_randoms(noise_g.cuda(), [m.cuda() for m in dm_g])
        eng.g_backward()
        rg64 = O.g_step(as_f64(S, cfg), latent.double(), numeric.double(), emot, noise_g.double(),
                        [m.double() for m in dm_g])
        rg = O.g_step(S, latent, numeric, emot, noise_g, dm_g)
        assert abs(eng.adv.item() - float(g[f"s{it}.adv"])) <= 1e-4 * max(1.0, abs(float(g[f"s{it}.adv"])))
        assert abs(eng.emo.item() - float(g[f"s{it}.emo"])) <= 1e-4
        if it == 0:
            np.testing.assert_allclose(eng.logits.cpu().numpy(), g["s0.logits"], rtol=1e-4, atol=2e-6)
        ok, errs = grad_ok(eng.notes, rg["fake"], rg64["fake"], slack=8.0, floor=2e-6)
        assert ok, ("fake_g", errs)
        for k, gr in rg["grads"].items():
            if k in NOISE_PARAMS_G:
                continue
            # floor 3e-3: in gan_c128_t64_b4 one ReLU/LeakyReLU mask decision sits within fp32 rounding of zero, and
            # every generator gradient moves by 2.0e-3 (relative L2) when it flips.  The reference's OWN fp32 result
            # flips with nothing but its CPU thread count: e_ref = 2.0e-3 with 128 threads, 2.9e-6 with 1 (measured;
            # collecting the CPU test modules sets 1 thread), while ours is 2.0e-3 -- so "slack x e_ref" alone made
            # this test depend on which other test files were collected.
            ok, errs = grad_ok(eng.GE.g[k], gr, rg64["grads"][k], slack=6.0, floor=3e-3)
            assert ok, (it, k, errs)
        ge_old = eng.GE.data.clone()
        eng.g_update()
        # post-update parameters (pre-BN conv biases excluded: pure Adam-amplified noise), then re-sync
        for k, v in S.PGE.items():
            if k in NOISE_PARAMS_G:
                continue
            o, n = eng.GE.offsets[k]
            old = ge_old[o:o + n].view(v.shape).cpu()
            e = rel_err(eng.GE.p[k].cpu() - old, v - old)
            assert e < 0.1, (it, "GE update", k, e)
        for k, v in S.BG.items():
            tol = 1e-5 if k.endswith("running_var") else 1e-3      # running_mean absorbs the noisy biases
            assert rel_err(eng.Gbuf[k], v) < tol, (it, k, rel_err(eng.Gbuf[k], v))
        with torch.no_grad():
            for k, v in S.PGE.items():
                eng.GE.p[k].copy_(v)
            for k, v in S.BG.items():
                eng.Gbuf[k].copy_(v)
            eng.params_changed()
    assert eng.num_batches_tracked == S.bn_batches
    # eval-mode generation (app.py contract)
    z = O.closed_form((eng.B, cfg["NOISE_DIM"]), 11.0, 1.0)
    with torch.no_grad():
        emb = O.feature_encoder_fwd(S.PE, numeric, None)
        gen, _ = O.generator_fwd(S.PG, S.BG, z, latent, emb, cfg["INTEGRATION_MODE"], cfg["MAX_NOTES"], train=False)
    out = eng.generate(z.cuda(), numeric.cuda(), latent.cuda())
    torch.testing.assert_close(out.cpu(), gen, rtol=1e-3, atol=1e-5)


def test_graph_replay_matches_eager(engine_mod):
    """The captured hipGraphs must reproduce the eager launch sequence bit for bit."""
    g = load("gan_c4_t32_b4_bigD")
    _, e1, cfg, _ = make(engine_mod, g)
    _, e2, _, _ = make(engine_mod, g)
    with torch.cuda.stream(e2.stream):
        for it in range(4):
            R = O.step_randoms(e1.B, cfg["NOISE_DIM"], seed=100 + it)
            for e, graph in ((e1, False), (e2, True)):
                e.set_randoms(R["noise_d"].cuda(), [m.cuda() for m in R["dm_d"]], R["alpha"].cuda())
                e.run("d_backward", graph)
                e.run("d_update", graph)
                e.set_randoms(R["noise_g"].cuda(), [m.cuda() for m in R["dm_g"]])
                e.run("g_backward", graph)
                e.run("g_update", graph)
        torch.cuda.synchronize()
    assert torch.equal(e1.D.data, e2.D.data)
    assert torch.equal(e1.GE.data, e2.GE.data)
    assert torch.equal(e1.loss_d_out, e2.loss_d_out)
    assert e1.num_batches_tracked == e2.num_batches_tracked


def test_production_flow_is_bitwise_reproducible_and_matches_the_dp_flow(engine_mod):
    """The production sub-steps (device Philox draws, tick-free updates, hipGraph replay) run twice from the same seed
    give identical bits (fixed-order reductions, no float atomics); and the data-parallel launch order with world = 1
    semantics emulated by hand (g_forward_rng before d_update, split backward graphs) ends in the same parameters up to
    the different Philox step the generator draw then sees -- so it is compared on its own second run instead."""
    from melo_gan_amd.gan.dp import DataParallel
    g = load("gan_c4_t32_b4")

    def run(flow):
        _, e, cfg, (real, numeric, latent, emot) = make(engine_mod, g)
        e.seed(77)
        dp = DataParallel(e, 1, None)
        with torch.cuda.stream(e.stream):
            for it in range(6):
                e.set_batch(real.cuda(), numeric.cuda(), latent.cuda(), emot.cuda())
                if flow == "step":
                    dp.step(True, g_step=(it % 2 == 1))
                else:                                   # the N > 1 ordering of DataParallel.step, without collectives
                    e.run("d_backward_rng")
                    if it % 2 == 1:
                        e.run("g_forward_rng")
                        e.run("d_update")
                        e.run("g_backward_a2")
                        e.run("g_backward_b")
                        e.run("g_update")
                    else:
                        e.run("d_update")
            torch.cuda.synchronize()
        assert torch.isfinite(e.D.data).all() and torch.isfinite(e.GE.data).all()
        # Philox draws: one per batch in the fused flow (dg_step_rng draws both halves at once), one per sub-step otherwise
        assert int(e.rng_step.item()) == (6 if flow == "step" else 6 + 3)
        assert float(e.D.state[0].item()) == 6.0 and float(e.GE.state[0].item()) == 3.0
        return e.D.data.clone(), e.GE.data.clone(), e.loss_d_out.clone()

    for flow in ("step", "dp_order"):
        a, b = run(flow), run(flow)
        for x, y in zip(a, b):
            assert torch.equal(x, y), flow


def test_smoke_entry(engine_mod):
    import __graft_entry__ as entry
    entry.smoke_check(verbose=False)


def test_workspaces_outlive_the_graphs_that_captured_them(engine_mod):
    """The trainer's default shape (B=32, T=512, C=4) with CRITIC_ITERS > 1: the critic step is captured several batches
    before the generator step first runs and asks for a larger 'wgrad_multi' scratch buffer.  The superseded buffer has
    its address baked into the critic graph: it must stay allocated (ops._ws_retired), and the replayed critic graph
    must keep matching an engine that never replays."""
    from melo_gan_amd import ops
    cfg, ed_cfg = O.default_gan_cfg(32, 512, 4), O.default_ed_cfg(4)
    engines = []
    for _ in range(2):
        e = engine_mod.GanEngine(cfg, ed_cfg, "cuda", 32)
        e.init_weights(3)
        e.seed(5)
        engines.append(e)
    real, numeric, latent, emot = O.synthetic_batch(32, 512, 4, cfg["LATENT_DIM"], 6, 1)
    e_graph, e_eager = engines
    def scratch():          # the engine stream's 'wgrad_multi' scratch buffer (ops.workspace keys by device, tag, stream)
        return [b for (d, tag, st), b in ops._ws_cache.items() if tag == "wgrad_multi" and st == e_graph.stream.cuda_stream][0]
    with torch.cuda.stream(e_graph.stream):
        for e in engines:
            e.set_batch(real.cuda(), numeric.cuda(), latent.cuda(), emot.cuda())
        ptrs = []
        for it in range(10):
            for e, graph in ((e_graph, True), (e_eager, False)):
                e.run("d_step_rng", graph)
                if it % 5 == 4:
                    e.run("g_step_rng", graph)
            if it == 2:                                   # critic graph exists, the generator step has not run yet
                assert not isinstance(e_graph._graphs["d_step_rng"], str) and "g_step_rng" not in e_graph._graphs
                ptrs.append(scratch().data_ptr())
        torch.cuda.synchronize()
    now = scratch().data_ptr()
    if now != ptrs[0]:                                    # the buffer grew: the captured one must still be alive
        assert any(b.data_ptr() == ptrs[0] for b in ops._ws_retired)
    assert torch.equal(e_graph.D.data, e_eager.D.data) and torch.equal(e_graph.GE.data, e_eager.GE.data)


def test_fused_step_matches_the_two_separate_steps(engine_mod):
    """dg_forward runs the critic step's and the generator step's E_num + generator forward as ONE 2B-row pass (same
    weights, per-half noise / dropout masks / BatchNorm statistics, running statistics updated twice).  With the same
    injected randoms it must reproduce the two separate passes: every gradient, both losses, both fake batches, the
    running statistics -- to fp32 summation-order accuracy (the 2B-row launches tile and split their reductions
    differently), and the replayed one-graph step must equal its eager self bit for bit."""
    g = load("gan_c128_t64_b4")
    S, e1, cfg, _ = make(engine_mod, g)
    _, e2, _, _ = make(engine_mod, g)
    R = O.step_randoms(e1.B, cfg["NOISE_DIM"], seed=31)
    for e in (e1, e2):
        e.set_randoms(R["noise_d"].cuda(), [m.cuda() for m in R["dm_d"]], R["alpha"].cuda())
        e.set_randoms(R["noise_g"].cuda(), [m.cuda() for m in R["dm_g"]])
    e1.d_backward(); gd1 = e1.D.grad.clone(); e1.d_update()
    e1.g_backward(); gg1 = e1.GE.grad.clone(); e1.g_update()
    e2.dg_forward()
    e2.d_backward(forward=False); gd2 = e2.D.grad.clone(); e2.d_update()
    e2.g_backward_a2(); e2.g_backward_b(); gg2 = e2.GE.grad.clone(); e2.g_update()
    torch.cuda.synchronize()
    assert e1.num_batches_tracked == e2.num_batches_tracked == 2
    for a, b, what in ((e1.fake_d, e2.fake_d, "fake_d"), (e1.notes, e2.notes, "fake_g"), (e1.emb_d, e2.emb_d, "emb_d"),
                       (e1.loss_d_out, e2.loss_d_out, "loss_d"), (e1.adv, e2.adv, "adv"), (e1.emo, e2.emo, "emo")):
        assert rel_err(b, a) < 2e-5, (what, rel_err(b, a))
    for k in e1.Gbuf:
        assert rel_err(e2.Gbuf[k], e1.Gbuf[k]) < 1e-5, k
    assert rel_err(gd2, gd1) < 1e-4, rel_err(gd2, gd1)
    for k, (o, n) in e1.GE.offsets.items():
        if k in NOISE_PARAMS_G:
            continue
        assert rel_err(gg2[o:o + n], gg1[o:o + n]) < 2e-3, (k, rel_err(gg2[o:o + n], gg1[o:o + n]))     # BatchNorm-conditioned
    # the production form: draw + both steps as one replayed graph == the same sequence launched eagerly
    _, e3, _, batch = make(engine_mod, g)
    _, e4, _, _ = make(engine_mod, g)
    with torch.cuda.stream(e4.stream):
        for e, graph in ((e3, False), (e4, True)):
            e.seed(9)
            for _ in range(4):
                e.run("dg_step_rng", graph)
        torch.cuda.synchronize()
    assert torch.equal(e3.D.data, e4.D.data) and torch.equal(e3.GE.data, e4.GE.data) and torch.equal(e3.notes, e4.notes)
    assert e3.num_batches_tracked == e4.num_batches_tracked == 8 and int(e4.rng_step.item()) == 4
    assert float(e4.D.state[0].item()) == 4.0 and float(e4.GE.state[0].item()) == 4.0


def test_split_flow_equals_the_single_graph_flow(engine_mod):
    """DataParallel.step on one GPU: back-to-back generator steps run the emotion branch as its own graph on a side stream
    (four graphs per batch: "split") or as a parallel branch of the one graph ("ingraph"); parameters, optimiser state and
    losses must equal the one-graph flow bit for bit, also when critic-only batches (which fall back to the one-graph
    flow) are mixed in."""
    from melo_gan_amd.gan.dp import DataParallel
    g = load("gan_c128_t64_b4")
    res = []
    for flow in ("none", "split", "ingraph"):
        os.environ["MELO_ED_FLOW"] = flow
        try:
            S, eng, cfg, batch = make(engine_mod, g, use_graph=True)
            dp = DataParallel(eng, 1, None)
            eng.seed(5)
            with torch.cuda.stream(eng.stream):
                for k, g_step in enumerate((True, True, True, False, True, True, False, False, True, True, True)):
                    eng.set_batch(*[t.cuda() for t in batch])
                    dp.step(True, g_step=g_step)
            torch.cuda.synchronize()
            res.append((eng.D.data.clone(), eng.GE.data.clone(), eng.GE.m.clone(), float(eng.loss_d_out[0]), float(eng.adv),
                        float(eng.emo), {k: v.clone() for k, v in eng.Gbuf.items()}, set(eng._graphs)))
        finally:
            os.environ.pop("MELO_ED_FLOW", None)
    a, b, c = res
    assert any(k.startswith("g_finish") for k in b[7]) and not any(k.startswith("g_finish") for k in a[7])   # the split flow ran
    assert "dg_fork_step_rng" in c[7]                                                                        # the forked graph ran
    for o in (b, c):
        assert torch.equal(a[0], o[0]) and torch.equal(a[1], o[1]) and torch.equal(a[2], o[2])
        assert a[3:6] == o[3:6]
        for k in a[6]:
            assert torch.equal(a[6][k], o[6][k]), k
