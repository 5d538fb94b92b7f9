"""The reference-shaped module surface (constructors, forward signatures, state_dict keys) and the trainer CLI
on the GPU, against the golden fixtures produced by the reference modules."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def build(B, T, C):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.models import Generator, Discriminator
    from melo_gan_amd.gan.feature_encoder import FeatureEncoder
    from melo_gan_amd.emotion_discriminator.ed_model import EmotionDiscriminator
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "closed_form")
    E = FeatureEncoder(6, [256, 128], 128)
    G = Generator(cfg["NOISE_DIM"], cfg["LATENT_DIM"], cfg["INTEGRATION_MODE"], max_notes=T, note_dim=C, numeric_embed_dim=128)
    D = Discriminator(T, C, numeric_embed_dim=128)
    ED = EmotionDiscriminator(ed_cfg)
    # state_dict surface == the reference's
    assert set(E.state_dict()) == set(S.PE)
    assert set(D.state_dict()) == set(S.PD)
    gkeys = set(S.PG) | set(S.BG) | {"decoder.deconv.1.num_batches_tracked", "decoder.deconv.4.num_batches_tracked"}
    assert set(G.state_dict()) == gkeys
    edkeys = set(S.PED) | set(S.BED) | {f"encoder.conv.{i}.net.1.num_batches_tracked" for i in range(4)}
    assert set(ED.state_dict()) == edkeys
    E.load_state_dict(S.PE)
    G.load_state_dict({**S.PG, **S.BG}, strict=False)
    D.load_state_dict(S.PD)
    ED.load_state_dict({**S.PED, **S.BED}, strict=False)
    return cfg, S, E.cuda(), G.cuda(), D.cuda(), ED.cuda().eval()


@pytest.mark.parametrize("name", ["layers_c4_t16_b2", "layers_c128_t32_b2"])
def test_module_forwards_match_reference(name):
    g = load(name)
    B, T, C = int(g["B"]), int(g["T"]), int(g["C"])
    cfg, S, E, G, D, ED = build(B, T, C)
    real, numeric, latent, emot = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, 3)
    E.eval()
    emb = E(numeric.cuda())
    np.testing.assert_allclose(emb.cpu().numpy(), g["E.emb_eval"], rtol=1e-4, atol=1e-6)
    noise = O.closed_form((B, cfg["NOISE_DIM"]), 5.0, 1.0).cuda()
    G.train()
    fake, lat = G(noise, latent.cuda(), emb)
    assert tuple(fake.shape) == (B, T, C)
    np.testing.assert_allclose(fake.cpu().numpy(), g["G.train.fake"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(lat.cpu().numpy(), g["G.train.latent"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(G.decoder.deconv[1].running_var.cpu().numpy(), g["G.train.running_var1"], rtol=1e-4)
    assert int(G.decoder.deconv[1].num_batches_tracked) == 1
    G.eval()
    fake_e, _ = G(noise, latent.cuda(), emb)
    np.testing.assert_allclose(fake_e.cpu().numpy(), g["G.eval.fake"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(D(real.cuda(), emb).cpu().numpy(), g["D.score"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ED(real.cuda()).cpu().numpy(), g["ED.logits"], rtol=1e-4, atol=2e-6)


def test_gradient_penalty_value_matches_reference():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.utils import compute_gradient_penalty, emotion_to_index, weights_init
    g = load("gan_c4_t32_b4_bigD")
    B, T, C = int(g["B"]), int(g["T"]), int(g["C"])
    cfg, S, E, G, D, ED = build(B, T, C)
    D.load_state_dict({k: v * 6.0 for k, v in S.PD.items()})
    real, numeric, latent, _ = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, int(g["seed"]))
    fake = torch.from_numpy(g["s0.fake_d"]).cuda()
    with torch.no_grad():
        emb = O.feature_encoder_fwd(S.PE, numeric, [torch.from_numpy(g[f"s0.dm_d{j}"]).float() for j in range(2)])
    gp = compute_gradient_penalty(D, real.cuda(), fake, emb.cuda(), "cuda", alpha=torch.from_numpy(g["s0.alpha"]).cuda())
    assert abs(gp.item() - float(g["s0.gp"])) < 1e-4
    assert emotion_to_index("Sad") == 1 and emotion_to_index([0, 0, 0, 1]) == 3 and emotion_to_index(None) == -1
    D.apply(weights_init)
    assert float(D.conv[0].bias.abs().max()) == 0.0 and 0.01 < float(D.conv[0].weight.std()) < 0.03


def test_trainer_cli_smoke(tmp_path):
    """CLI + YAML + checkpoint contract on synthetic data (2 epochs, tiny shapes)."""
    import yaml
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan import train_gan
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "config", "gan_config.yaml")))
    cfg.update(EPOCHS=2, BATCH_SIZE=4, MAX_NOTES=32, SAVE_FREQ=1, CRITIC_ITERS=2,
               CHECKPOINT_DIR=str(tmp_path / "ck"), LOG_DIR=str(tmp_path / "log"), SAMPLE_DIR=str(tmp_path / "s"))
    ed = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "config", "ed_config.yaml")))
    cp, ep = tmp_path / "gan.yaml", tmp_path / "ed.yaml"
    cp.write_text(yaml.safe_dump(cfg))
    ep.write_text(yaml.safe_dump(ed))
    train_gan.main(["--config", str(cp), "--ed_config", str(ep), "--ed_ckpt", str(tmp_path / "none.pth"), "--synthetic", "20"])
    full = torch.load(tmp_path / "ck" / "gan_epoch0002.pth", map_location="cpu")
    assert set(full) == {"epoch", "G", "D", "E_num", "opt_G", "opt_D"} and full["epoch"] == 2
    final = torch.load(tmp_path / "ck" / "gan_final.pth", map_location="cpu")
    assert set(final) == {"G", "E_num"}
    # the checkpoint loads into the reference-shaped modules (strict), as app.py:44-48 does
    from melo_gan_amd.gan.models import Generator
    from melo_gan_amd.gan.feature_encoder import FeatureEncoder
    G = Generator(128, 64, "warm_start", max_notes=32, note_dim=4, numeric_embed_dim=128)
    G.load_state_dict(final["G"], strict=True)
    FeatureEncoder(6, [256, 128], 128).load_state_dict(final["E_num"], strict=True)
    assert int(final["G"]["decoder.deconv.1.num_batches_tracked"]) > 0
    for v in final["G"].values():
        assert torch.isfinite(v.float()).all()


def test_resume_continues_from_a_full_checkpoint(tmp_path):
    """--resume: G (+ BatchNorm buffers), D, E_num and both optimisers come back from gan_epochNNNN.pth; the derived copies
    (WQ-layout weights, folded emotion discriminator) are refreshed; training continues at the next epoch."""
    import yaml
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan import train_gan
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "config", "gan_config.yaml")))
    cfg.update(EPOCHS=1, BATCH_SIZE=4, MAX_NOTES=32, SAVE_FREQ=1, CRITIC_ITERS=2,
               CHECKPOINT_DIR=str(tmp_path / "ck"), LOG_DIR=str(tmp_path / "log"), SAMPLE_DIR=str(tmp_path / "s"))
    ed = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "config", "ed_config.yaml")))
    e1 = train_gan.train(dict(cfg), dict(ed), str(tmp_path / "none.pth"), synthetic=20)
    ck = str(tmp_path / "ck" / "gan_epoch0001.pth")
    from melo_gan_amd.gan.engine import GanEngine
    from melo_gan_amd.gan import config as Cfg
    e2 = GanEngine(Cfg.with_gan_defaults(dict(cfg), require=False), dict(ed), "cuda", 4)
    e2.init_weights(7)
    assert train_gan.resume_checkpoint(e2, ck) == 1
    for a, b in ((e1.GE, e2.GE), (e1.D, e2.D)):
        assert torch.equal(a.data, b.data) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
        assert float(a.state[0]) == float(b.state[0]) > 0 and abs(float(a.state[1]) - float(b.state[1])) < 1e-12
    for k in e1.Gbuf:
        assert torch.equal(e1.Gbuf[k], e2.Gbuf[k])
    for k in e1.wq:
        assert torch.equal(e1.wq[k], e2.wq[k]), k
    assert e2.num_batches_tracked == e1.num_batches_tracked
    cfg2 = dict(cfg, EPOCHS=2)
    e3 = train_gan.train(cfg2, dict(ed), str(tmp_path / "none.pth"), synthetic=20, resume=ck)
    assert os.path.exists(tmp_path / "ck" / "gan_epoch0002.pth")
    assert float(e3.D.state[0]) > float(e1.D.state[0])
    with pytest.raises(KeyError):
        train_gan.resume_checkpoint(e2, str(tmp_path / "ck" / "gan_final.pth"))


def test_ed_checkpoint_shape_mismatch_raises_and_missing_keys_are_reported(tmp_path, capsys):
    """load_state_dict(strict=False) semantics (reference train_gan.py:121-128): missing keys are tolerated but named, a size
    mismatch raises; an incomplete spectral-norm triple counts as missing."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan import train_gan
    from melo_gan_amd.gan.engine import GanEngine
    eng = GanEngine(O.default_gan_cfg(4, 32, 4), O.default_ed_cfg(4), "cuda", 4)
    eng.init_weights(0)
    sd = {**{k: v.cpu().clone() for k, v in eng.ED.p.items()}, **{k: v.cpu().clone() for k, v in eng.EDbuf.items()}}
    w = sd.pop("classifier.head.weight")
    sd["classifier.head.weight_orig"], sd["classifier.head.weight_u"] = w, torch.ones(w.shape[0])       # no _v: incomplete triple
    sd["encoder.project.bias"] = sd["encoder.project.bias"] + 1.0
    p = str(tmp_path / "ed.pth")
    torch.save({"model": sd}, p)
    before = eng.ED.p["classifier.head.weight"].clone()
    assert train_gan.load_ed_checkpoint(eng, p)
    out = capsys.readouterr().out
    assert "classifier.head.weight" in out and "lacks 1 key" in out
    assert torch.equal(eng.ED.p["classifier.head.weight"], before)
    torch.testing.assert_close(eng.ED.p["encoder.project.bias"].cpu(), sd["encoder.project.bias"])
    sd["encoder.conv.1.net.0.weight"] = torch.zeros(3, 3, 3)
    torch.save({"model": sd}, p)
    with pytest.raises(RuntimeError, match="size mismatch"):
        train_gan.load_ed_checkpoint(eng, p)
