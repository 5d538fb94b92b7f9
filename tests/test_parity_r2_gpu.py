"""Round-2 parity additions (GPU, through the C-ABI):

  * FREE-RUNNING trajectories: the engine runs the fixtures' 2-3 (critic + generator) steps on its own state -- no
    re-synchronisation with the oracle between sub-steps -- and is compared with what the REFERENCE recorded for the same
    injected randoms (tests/golden/make_golden.py::gan_case): per-step losses and the end-of-run state checksums;
  * the production optimiser path (flat buffers, Adam state advanced by the Philox draw = mg_rng_fill_tick +
    mg_adam_flat_ticked, grad_scale = 1/world) against torch.optim.Adam / AdamW element by element;
  * batch-1 eval generation (BASELINE config 5, app.py:92-119) against reference fixtures, through the MIDI contract.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402  (the checker)
from test_oracle_golden import checksum, gen1_state  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["gan_c4_t32_b4", "gan_c4_t32_b4_bigD", "gan_c128_t64_b4", "gan_c4_t20_b3", "gan_c4_t16_cond_lat"]
# Stated trajectory tolerance (SURVEY hard part 3): losses of a free-running 2-3 step run within rtol 1e-3 of the
# reference's.  (gp and the emotion loss carry an absolute floor: they are O(1e-2..1) sums of cancelling terms.)
TRAJ_RTOL, TRAJ_ATOL = 1e-3, 2e-5
# Parameters whose gradient is mathematically zero (rounding noise that Adam turns into +-lr steps, in the reference as
# well): pre-BatchNorm conv biases (and the running means that absorb them), the critic head's bias and embedding half.
ZERO_GRAD = ("G.decoder.deconv.0.bias", "G.decoder.deconv.3.bias", "G.decoder.deconv.1.running_mean",
             "G.decoder.deconv.4.running_mean", "D.real_fake.bias", "D.real_fake.weight", "D.fc.1.bias")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def build(g):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.engine import GanEngine
    B, T, C = int(g["B"]), int(g["T"]), int(g["C"])
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    cfg["INTEGRATION_MODE"], ed_cfg["input_mode"] = str(g["mode"]), str(g["ed_mode"])
    S = O.build_gan_state(cfg, ed_cfg, "closed_form", d_scale=float(g["d_scale"]))
    eng = GanEngine(cfg, ed_cfg, "cuda", B)
    eng.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
    real, numeric, latent, emot = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, int(g["seed"]))
    if cfg["INTEGRATION_MODE"] == "conditioning":
        latent = O.closed_form((B, cfg["LATENT_DIM"]), 9.0, 0.5)
    eng.set_batch(real.cuda(), numeric.cuda(), latent.cuda(), emot.cuda())
    return S, eng, cfg, (real, numeric, latent, emot)


@pytest.mark.parametrize("name", CASES)
def test_free_running_trajectory_matches_reference(name):
    g = load(name)
    S, eng, cfg, (real, numeric, latent, emot) = build(g)
    n_steps = int(g["n_steps"])
    dev = {}
    for it in range(n_steps):
        cu = lambda k: torch.from_numpy(g[f"s{it}.{k}"]).cuda()  # noqa: E731
        B = eng.B
        eng.set_randoms(cu("noise_d"), [cu("dm_d0").float(), cu("dm_d1").float()], cu("alpha"))
        eng.d_backward()
        d_real = eng.s[B:2 * B].cpu().numpy().copy()          # critic rows: [x_hat | real | fake]
        # The critic's output OFFSET is a free, noise-driven degree of freedom of the model: the head's bias and the
        # embedding half of its weight receive +1/B and -1/B contributions that cancel exactly, so their gradient is
        # rounding noise and Adam moves them by +-lr_d per critic update (in the reference too).  loss_d is blind to it
        # (same offset on real and fake); the raw scores and adv = -mean(D(fake)) carry it:
        offset_noise = it * eng.lr_d * (1.0 + float(eng.emb_d.abs().sum(dim=1).max().item()))
        np.testing.assert_allclose(d_real, g[f"s{it}.d_real"], rtol=TRAJ_RTOL, atol=2e-5 + offset_noise)
        eng.d_update()
        offset_noise += eng.lr_d * (1.0 + float(eng.emb.abs().sum(dim=1).max().item()))
        eng.set_randoms(cu("noise_g"), [cu("dm_g0").float(), cu("dm_g1").float()])
        eng.g_backward()
        eng.g_update()
        got = dict(loss_d=eng.loss_d_out[0].item(), gp=eng.gp.item(), adv=eng.adv.item(), emo=eng.emo.item())
        for k, v in got.items():
            ref = float(g[f"s{it}.{k}"])
            atol = TRAJ_ATOL + (offset_noise if k == "adv" else 0.0)
            dev[f"s{it}.{k}"] = max(0.0, abs(v - ref) - atol) / max(abs(ref), 1e-30)
            assert abs(v - ref) <= TRAJ_RTOL * abs(ref) + atol, (name, it, k, v, ref, atol)
    # end-of-run state: the reference's checksums (sum, cos-weighted sum, L1 mass) of every tensor, relative to the L1 mass.
    # Adam moves every element by ~lr per step whatever its gradient, so elements whose gradient is rounding noise
    # differ by up to n_steps * lr each: the bound is rtol * L1 + (share of such elements, measured <= 2 %) * numel * n * lr.
    sd = eng.state_dicts()
    worst = {}
    for grp, lr in (("D", eng.lr_d), ("G", eng.lr_g), ("E_num", eng.lr_g)):
        tag = "E" if grp == "E_num" else grp
        for k, v in sd[grp].items():
            if f"{tag}.{k}" in ZERO_GRAD or k.endswith("num_batches_tracked"):
                continue
            ref = g[f"end.{tag}.{k}"]
            ck = checksum(v.float())
            err = np.abs(ck - ref).max() / max(abs(ref[2]), 1e-12)
            worst[f"{tag}.{k}"] = err
            bound = TRAJ_RTOL + 0.02 * v.numel() * n_steps * lr / max(abs(ref[2]), 1e-12)
            assert err <= bound, (name, tag, k, err, bound)
    # generation from the free-run state against the reference's: eval mode first (app.py contract).  It inherits the
    # noise of the pre-BatchNorm biases / running means (n_steps * lr_g of drift through two BatchNorms and
    # deconvolutions: measured <= 8e-5 absolute on outputs of ~5e-3).
    z = O.closed_form((eng.B, cfg["NOISE_DIM"]), 11.0, 1.0)
    out = eng.generate(z.cuda(), numeric.cuda(), latent.cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g["end.generated"], rtol=5e-3, atol=2e-4)
    # ... then on BATCH statistics (G.train(), E_num eval), which does not see those parameters and is ~60x larger
    # (1 / batch std instead of 1 / running std).  How tightly can a free-running end state be pinned?  The EXACT (fp64)
    # evaluation of the same 2-3 steps ends 2.1-2.7 % away from the reference's fp32 result in two of the five fixtures
    # (5e-6 in the others): the trajectory amplifies fp32 rounding that much in the reference itself.  So the bound is the
    # reference's own distance from exact arithmetic.  The three evaluations (reference fp32, exact fp64, ours) either
    # agree to ~1e-5 or sit on different BRANCHES (a ReLU / LeakyReLU decision or the sign of a near-zero gradient that
    # Adam's first step turns into a full +-lr move, BatchNorm over 12-64 rows amplifying it) 0.8-2.7 % apart; measured:
    #   fixture               ours vs ref   fp64 vs ref
    #   gan_c4_t32_b4         9.6e-6        5.6e-6        all on one branch
    #   gan_c4_t32_b4_bigD    3.1e-6        2.1e-2        ours with the reference, exact arithmetic elsewhere
    #   gan_c128_t64_b4       2.7e-2        2.7e-2        ours with exact arithmetic (1e-5 from it), the reference elsewhere
    #   gan_c4_t20_b3         1.8e-5        2.2e-5        one branch
    #   gan_c4_t16_cond_lat   7.8e-3        1.8e-5        ours on its own branch
    # Required: on the branch of the reference or of exact arithmetic to 1e-4 (+1.5x the exact run's own distance), or,
    # failing both, no further than the largest branch distance the reference itself shows against exact arithmetic (3 %).
    eng.noise.copy_(z.cuda())
    eng._e_fwd(train=False)
    eng._g_fwd(eng.notes, train=True)
    ref_tr = torch.from_numpy(g["end.generated_train"])
    S64 = O.GanState(S.cfg, S.ed_cfg, *[type(P)((k, v.double().clone()) for k, v in P.items())
                                         for P in (S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)])
    for it in range(n_steps):
        f = lambda k: torch.from_numpy(g[f"s{it}.{k}"]).double()  # noqa: E731
        O.d_step(S64, real.double(), latent.double(), numeric.double(), f("noise_d"), f("alpha"), [f("dm_d0"), f("dm_d1")])
        O.g_step(S64, latent.double(), numeric.double(), emot, f("noise_g"), [f("dm_g0"), f("dm_g1")])
    with torch.no_grad():
        emb64 = O.feature_encoder_fwd(S64.PE, numeric.double(), None)
        gen64, _ = O.generator_fwd(S64.PG, S64.BG, z.double(), latent.double(), emb64, cfg["INTEGRATION_MODE"],
                                   cfg["MAX_NOTES"], train=True)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
    e_mine, e_exact, e_mine64 = rel(eng.notes.cpu(), ref_tr), rel(gen64, ref_tr), rel(eng.notes.cpu(), gen64)
    same_branch = e_mine <= 1.5 * e_exact + 1e-4 or e_mine64 <= 1e-4
    assert same_branch or e_mine <= 3e-2, (name, "generated_train", e_mine, e_exact, e_mine64)
    print(name, "generated_train: ours vs reference", e_mine, " exact fp64 vs reference", e_exact, " ours vs fp64", e_mine64,
          "same branch" if same_branch else "OWN BRANCH")
    print(name, "max loss dev", max(dev.values()), "max checksum dev", max(worst.values()), max(worst, key=worst.get))


@pytest.mark.parametrize("world", [1, 8])
@pytest.mark.parametrize("decoupled_wd", [0.0, 0.01])
def test_production_optimiser_path_elementwise(world, decoupled_wd):
    """The flat, ticked update (mg_rng_fill_tick advances the Adam state, mg_adam_flat_ticked applies it) with the
    data-parallel 1/world factor against torch.optim.Adam / AdamW on the averaged gradient, element by element."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    n, lr, betas = 10007, 2e-4, (0.5, 0.9)
    gen = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=gen) * 0.02
    pr = p0.clone().requires_grad_(True)
    opt = (torch.optim.AdamW([pr], lr=lr, betas=betas, weight_decay=decoupled_wd) if decoupled_wd
           else torch.optim.Adam([pr], lr=lr, betas=betas))
    pad = (-n) % 4
    p = torch.cat([p0, torch.zeros(pad)]).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.zeros(4, dtype=torch.float64, device="cuda")
    rng_step = torch.zeros(1, dtype=torch.int64, device="cuda")
    noise = torch.empty(64, device="cuda")
    for it in range(4):
        shards = [torch.randn(n, generator=gen) * (10.0 ** torch.randint(-6, 1, (n,), generator=gen).float()) for _ in range(world)]
        gsum = torch.stack(shards).sum(0)                      # what the all-reduce(SUM) leaves in the flat gradient
        pr.grad = gsum / world
        opt.step()
        grad = torch.cat([gsum, torch.zeros(pad)]).cuda()
        if it % 2 == 0:      # production: the sub-step's Philox draw advances the Adam state, the update the Philox counter
            ops.rng_fill(noise, None, None, None, 0.2, 7, rng_step, tick_state=state, betas=betas)
            ops.adam_flat(p, grad, m, v, state, lr, *betas, weight_decay=decoupled_wd, grad_scale=1.0 / world,
                          ticked_rng_step=rng_step)
        else:                # split form (tests, data-parallel d_update)
            ops.adam_flat(p, grad, m, v, state, lr, *betas, weight_decay=decoupled_wd, grad_scale=1.0 / world)
        upd, upd_ref = (p[:n].cpu() - p0), (pr.detach() - p0)
        # elementwise: every step is ~lr in magnitude; 1e-4 of lr is ~50x tighter than a 1 % bias-correction error
        assert float((upd - upd_ref).abs().max()) <= 1e-4 * lr * (it + 1), (it, float((upd - upd_ref).abs().max()))
    assert float(state[0].item()) == 4.0 and int(rng_step.item()) == 2


@pytest.mark.parametrize("name", ["gen1_c4_t512", "gen1_c128_t256"])
def test_batch1_generation_matches_reference(name):
    """BASELINE config 5: E_num -> G in eval mode at B = 1 (app.py:92-119), eager and as a replayed hipGraph, and -- at
    the reference's note shape -- through the output contract (midi.notes_from_roll)."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import midi, ops
    from melo_gan_amd.gan.engine import GanEngine
    g = load(name)
    S, cfg, z, numeric = gen1_state(g)
    eng = GanEngine(cfg, O.default_ed_cfg(int(g["C"])), "cuda", 1)
    eng.load_state(S.PE, S.PG, S.BG, S.PD, S.PED, S.BED)
    out = eng.generate(z.cuda(), numeric.cuda()).cpu().numpy()
    np.testing.assert_allclose(eng.emb.cpu().numpy(), g["emb"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(eng.lat.cpu().numpy(), g["latent"], rtol=1e-4, atol=1e-6)
    # the fixture's last deconvolution is scaled x1600 (outputs spanning the MIDI writer's branches): every output is a sum of
    # ~160 products of magnitude ~1, i.e. fp32 summation-order noise of ~3e-5 absolute
    np.testing.assert_allclose(out, g["generated"], rtol=1e-3, atol=1e-4)
    with torch.cuda.stream(eng.stream):
        gr = ops.Graph()
        gr.begin()
        eng._e_fwd(train=False, gin=True)          # as generate() does: the chain launch also assembles the generator input
        eng._g_fwd(eng.notes, train=False)
        gr.end()
        eng.notes.zero_()
        gr.launch()
        torch.cuda.synchronize()
    assert np.array_equal(eng.notes.cpu().numpy(), out)
    if "notes" in g.files:
        notes, bpm = midi.notes_from_roll(out[0], bpm=100.0, scale="minor", root_key=2)
        ref = g["notes"]
        assert bpm == float(g["tempo"])
        # int() / threshold decisions sit on fp32 knife edges for a few of the 512 rows: allow 2 rows to fall the other way
        assert abs(len(notes) - len(ref)) <= 2
        if len(notes) == len(ref):
            got = np.array(notes, dtype=np.float64)
            assert int((got[:, :2] != ref[:, :2]).any(axis=1).sum()) <= 2
            np.testing.assert_allclose(got[:, 2:], ref[:, 2:], rtol=0, atol=1e-3)
