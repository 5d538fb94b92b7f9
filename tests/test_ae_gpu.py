"""VAE step (config 4 path, src/ae/train_ae.py:110-122) on the GPU against the reference-generated fixture and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# conv biases that feed a train-mode BatchNorm: gradient is mathematically zero (pure rounding noise)
PRE_BN_BIAS = {"encoder.conv.0.bias", "encoder.conv.3.bias", "encoder.conv.6.bias",
               "decoder.deconv.0.bias", "decoder.deconv.3.bias"}


def rel_err(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_vae_steps_match_reference():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.ae.engine import VaeEngine
    g = np.load(os.path.join(GOLD, "ae_t32_b4.npz"))
    B, T, L = int(g["B"]), int(g["T"]), int(g["latent_dim"])
    spec, bufs = O.vae_spec(T, L)
    P = O.fill_params(spec, 6.0, O.norm_affine_names(spec))
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    eng = VaeEngine(dict(MAX_NOTES=T, LATENT_DIM=L, BATCH_SIZE=B, LR=1e-4, WEIGHT_DECAY=1e-5), "cuda", B)
    assert list(eng.P.spec) == list(spec) and all(tuple(eng.P.spec[k]) == tuple(spec[k]) for k in spec)
    eng.load_state(P, Bf)
    opt = O.AdamState(P, 1e-4, (0.9, 0.999), 1e-8, weight_decay=1e-5, decoupled=True)
    x = torch.rand(B, T, 4, generator=torch.Generator().manual_seed(5)) * 2 - 1
    for it in range(int(g["n_steps"])):
        eps = torch.from_numpy(g[f"s{it}.eps"])
        eng.x.copy_(x.cuda())
        eng.eps.copy_(eps.cuda())
        eng.forward(True)
        eng.backward(10.0)
        P64 = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
        B64 = {k: v.double().clone() for k, v in Bf.items()}
        rec64, _, mu64, lv64 = O.vae_fwd(P64, B64, x.double(), eps.double(), T, True)
        l64, _, _ = O.vae_loss(rec64, x.double(), mu64, lv64, 10.0)
        g64 = dict(zip(P64, torch.autograd.grad(l64, list(P64.values()))))
        old = {k: v.clone() for k, v in P.items()}
        r = O.ae_step(P, Bf, opt, x, eps, 10.0, T)
        loss = eng.loss.cpu()
        assert abs(loss[0].item() - float(g[f"s{it}.loss"])) < 1e-5 and abs(loss[2].item() - float(g[f"s{it}.kld"])) < 1e-5
        if it == 0:
            np.testing.assert_allclose(eng.recon.cpu().numpy(), g["s0.recon"], rtol=1e-3, atol=1e-5)
            np.testing.assert_allclose(eng.mu.cpu().numpy(), g["s0.mu"], rtol=1e-3, atol=1e-6)
        # gradients: no further from fp64 than a few x the reference's own fp32 arithmetic
        P32 = {k: v.clone().requires_grad_(True) for k, v in old.items()}
        Bx = {k: v.clone() for k, v in B64.items()}
        rec32, _, mu32, lv32 = O.vae_fwd(P32, {k: v.float() for k, v in Bx.items()}, x, eps, T, True)
        l32, _, _ = O.vae_loss(rec32, x, mu32, lv32, 10.0)
        g32 = dict(zip(P32, torch.autograd.grad(l32, list(P32.values()))))
        for k in spec:
            if k in PRE_BN_BIAS:
                continue
            e_mine, e_ref = rel_err(eng.P.g[k], g64[k]), rel_err(g32[k], g64[k])
            assert e_mine <= 8 * e_ref + 5e-5, (it, k, e_mine, e_ref)
        eng.update()
        assert abs(eng.gnorm[0].item() - float(g[f"s{it}.grad_norm"])) < 2e-3 * float(g[f"s{it}.grad_norm"])
        # re-synchronise (teacher forcing) -- see tests/test_engine_gpu.py for why
        for k, v in P.items():
            upd, upd_ref = eng.P.p[k].cpu() - old[k], v - old[k]
            if upd_ref.norm() > 0 and k not in PRE_BN_BIAS:
                assert rel_err(upd, upd_ref) < 0.15, (it, k, rel_err(upd, upd_ref))
        eng.load_state(P, Bf)


def test_ae_trainer_cli_smoke(tmp_path):
    import yaml
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.ae import train_ae
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "config", "ae_config.yaml")))
    cfg.update(MAX_NOTES=32, BATCH_SIZE=8, EPOCHS=2, CHECKPOINT_DIR=str(tmp_path / "ck"), LOG_DIR=str(tmp_path / "log"))
    p = tmp_path / "ae.yaml"
    p.write_text(yaml.safe_dump(cfg))
    train_ae.main(["--config", str(p), "--synthetic", "32"])
    best = torch.load(tmp_path / "ck" / "ae_best.pth", map_location="cpu")
    final = torch.load(tmp_path / "ck" / "ae_final.pth", map_location="cpu")
    assert set(best) == {"epoch", "model_state"} and set(best["model_state"]) == set(final)
    spec, bufs = O.vae_spec(32, 8)
    assert set(spec) | set(bufs) <= set(final) and "encoder._linear.1.weight" in final
    assert all(torch.isfinite(v.float()).all() for v in final.values())


def test_latent_export_matches_the_oracle_encoder(tmp_path):
    """ae/encode.py path (SURVEY f-3): eval-mode encoder over an array whose length is not a batch multiple; mu against
    the oracle's eval forward; checkpoint round trip through the trainer's state_dict layout."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.ae import encode as E
    from melo_gan_amd.ae.engine import VaeEngine
    from melo_gan_amd.ae.train_ae import state_dict
    T, L, B = 32, 8, 4
    spec, bufs = O.vae_spec(T, L)
    P = O.fill_params(spec, 6.0, O.norm_affine_names(spec))
    Bf = {k: (torch.rand(s, generator=torch.Generator().manual_seed(1)) + 0.5 if k.endswith("running_var")
              else torch.randn(s, generator=torch.Generator().manual_seed(2)) * 0.1) for k, s in bufs.items()}
    src = VaeEngine(dict(MAX_NOTES=T, LATENT_DIM=L, BATCH_SIZE=B, LR=1e-4, WEIGHT_DECAY=1e-5), "cuda", B)
    src.load_state(P, Bf)
    ckpt = os.path.join(str(tmp_path), "ae_best.pth")
    torch.save({"epoch": 3, "model_state": state_dict(src)}, ckpt)
    notes = torch.rand(10, T, 4, generator=torch.Generator().manual_seed(7)) * 2 - 1
    np.save(os.path.join(str(tmp_path), "notes.npy"), notes.numpy())
    cfg_path = os.path.join(str(tmp_path), "ae.yaml")
    with open(cfg_path, "w") as f:
        f.write(f"MAX_NOTES: {T}\nLATENT_DIM: {L}\nBATCH_SIZE: {B}\nLR: 1.0e-4\nWEIGHT_DECAY: 1.0e-5\n")
    out = os.path.join(str(tmp_path), "feats", "encoder_feats.npy")
    E.main(["--model", ckpt, "--notes", os.path.join(str(tmp_path), "notes.npy"), "--out_file", out, "--config", cfg_path])
    got = np.load(out)
    _, _, mu, _ = O.vae_fwd(P, Bf, notes, torch.zeros(10, L), T, train=False)
    assert got.shape == (10, L)
    np.testing.assert_allclose(got, mu.numpy(), rtol=2e-3, atol=2e-5)
