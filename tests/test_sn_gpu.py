"""Spectral normalisation on the GPU (mg_spectral_norm_fwd / _bwd; `use_spectral_norm: true` in the emotion discriminator's
pre-training, src/emotion_discriminator/ed_model.py:29-32,79-82 inside train_ed.py:51-82; FeatureEncoder(use_sn=True),
src/gan/feature_encoder.py:24-31) against the oracle's restatement of torch.nn.utils.spectral_norm (pinned against PyTorch's
wrapper in tests/test_oracle_sn.py) and against PyTorch's wrapper itself."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402


def rel_err(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("shape", [(64, 4, 5), (256, 256, 3), (128, 64, 3), (16, 40), (7, 3)])
def test_kernels_match_the_restatement(shape):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(shape, generator=g) * 0.3
    rows, cols = shape[0], int(torch.tensor(shape[1:]).prod())
    u0 = F.normalize(torch.randn(rows, generator=g), dim=0)
    v0 = F.normalize(torch.randn(cols, generator=g), dim=0)
    dw = torch.randn(shape, generator=g)
    for train in (True, False):
        u, v = u0.clone(), v0.clone()
        wr = w.clone().requires_grad_(True)
        w_eff_ref = O.spectral_norm_weight(wr, u, v, train)
        (w_eff_ref * dw).sum().backward()
        ly = dict(w_orig=w.cuda(), w_eff=torch.empty(shape, device="cuda"), u=u0.cuda(), v=v0.cuda(), sigma=torch.zeros(1, device="cuda"),
                  dw=dw.cuda())
        ops.spectral_norm_fwd([ly], train)
        ops.spectral_norm_bwd([ly])
        torch.cuda.synchronize()
        # fp64 truth: sigma = u . (W v) cancels heavily for vectors that are not singular vectors yet, so the fp32 restatement
        # itself is only good to ~1e-5 there -- the kernel may not be worse than a small multiple of it
        u64, v64 = u0.double().clone(), v0.double().clone()
        w64 = w.double().clone().requires_grad_(True)
        w_eff64 = O.spectral_norm_weight(w64, u64, v64, train)
        (w_eff64 * dw.double()).sum().backward()
        assert rel_err(ly["w_eff"], w_eff64) <= 6.0 * rel_err(w_eff_ref, w_eff64) + 2e-6
        assert rel_err(ly["u"], u64) < 5e-6 and rel_err(ly["v"], v64) < 5e-6
        if not train:
            assert torch.equal(ly["u"].cpu(), u0) and torch.equal(ly["v"].cpu(), v0)
        assert rel_err(ly["dw"], w64.grad) <= 6.0 * rel_err(wr.grad, w64.grad) + 1e-5


def _sn_cfg(C):
    return dict(O.default_ed_cfg(C), dropout=0.2, use_spectral_norm=True)


def test_ed_pretraining_with_spectral_norm_matches_the_oracle():
    """Three training steps (train_ed.py:51-82) with every encoder convolution and both classifier layers spectrally
    normalised: loss, logits, the gradient that reaches weight_orig, the power-iteration buffers, AdamW -- teacher-forced from
    the oracle's state each step; then the eval-mode forward and the state_dict surface."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    from melo_gan_amd.emotion_discriminator.ed_model import EmotionDiscriminator
    B, T, C = 8, 32, 4
    ed_cfg = _sn_cfg(C)
    spec, bufs = O.emotion_disc_spec(ed_cfg)
    P = O.fill_params(spec, 9.0, O.norm_affine_names(spec))
    for v in P.values():
        if v.dim() >= 2:
            v.mul_(4.0)
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    g = torch.Generator().manual_seed(11)
    names = O.ed_sn_layers(ed_cfg)
    assert len(names) == 6
    for nm in names:
        shp = spec[nm + ".weight"]
        Bf[nm + ".weight_u"] = F.normalize(torch.randn(shp[0], generator=g), dim=0)
        Bf[nm + ".weight_v"] = F.normalize(torch.randn(int(torch.tensor(shp[1:]).prod()), generator=g), dim=0)
    cfg = dict(ed_cfg, batch_size=B, max_notes=T, optimizer=dict(name="AdamW", lr=2e-4, betas=[0.5, 0.999], weight_decay=0.01))
    eng = EdEngine(cfg, "cuda", B, T)
    assert eng.sn_names == names
    eng.load_state(P, Bf)
    opt = O.AdamState(P, 2e-4, (0.5, 0.999), 1e-8, weight_decay=0.01, decoupled=True)
    pre_bn_bias = tuple(f"encoder.conv.{i}.net.0.bias" for i in range(4))
    for it in range(3):
        x = torch.rand(B, T, C, generator=g) * 2 - 1
        y = torch.randint(0, 4, (B,), generator=g)
        dm = [(torch.rand(B, h, generator=g) >= 0.2).float() / 0.8 for h in (256, 128)]
        eng.set_batch(x.cuda(), y.cuda())
        eng.set_masks([m.cuda() for m in dm])
        eng.backward()
        P64 = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
        B64 = {k: v.double().clone() for k, v in Bf.items()}
        l64 = F.cross_entropy(O.emotion_disc_fwd(P64, B64, x.double(), ed_cfg, True, [m.double() for m in dm]), y)
        g64 = dict(zip(P64, torch.autograd.grad(l64, list(P64.values()))))
        old = {k: v.clone() for k, v in P.items()}
        r = O.ed_step(P, Bf, opt, x, y, ed_cfg, dm)
        assert abs(eng.loss.item() - float(r["loss"])) < 2e-5
        assert rel_err(eng.logits, r["logits"]) < 2e-3
        for k in spec:
            if k in pre_bn_bias:
                continue
            e_mine, e_ref = rel_err(eng.P.g[k], g64[k]), rel_err(r["grads"][k], g64[k])
            assert e_mine <= 6.0 * e_ref + 3e-4, (it, k, e_mine, e_ref)
        for nm in names:          # the power iteration moved the buffers exactly as the wrapper's would
            assert rel_err(eng.buf[nm + ".weight_u"], Bf[nm + ".weight_u"]) < 1e-5, (it, nm)
            assert rel_err(eng.buf[nm + ".weight_v"], Bf[nm + ".weight_v"]) < 1e-5, (it, nm)
        eng.update()
        for k in spec:
            if k in pre_bn_bias:
                continue
            upd, upd_ref = eng.P.p[k].cpu() - old[k], P[k] - old[k]
            assert rel_err(upd, upd_ref) < 0.1, (it, "AdamW update", k, rel_err(upd, upd_ref))
        eng.load_state(P, Bf)
    # eval-mode forward: no power iteration, running statistics
    x = torch.rand(B, T, C, generator=g) * 2 - 1
    eng.set_batch(x.cuda(), torch.zeros(B, dtype=torch.int64).cuda())
    u_before = {nm: eng.buf[nm + ".weight_u"].clone() for nm in names}
    eng.forward_eval()
    want = O.emotion_disc_fwd(P, Bf, x, ed_cfg, train=False)
    assert rel_err(eng.logits, want) < 2e-3
    assert all(torch.equal(eng.buf[nm + ".weight_u"], u_before[nm]) for nm in names)
    # the state_dict surface of the reference module with use_spectral_norm, and the mirror module's eval forward on it
    sd = eng.state_dict()
    mirror = EmotionDiscriminator(dict(ed_cfg)).cuda().eval()
    assert set(sd) == set(mirror.state_dict()), set(sd) ^ set(mirror.state_dict())
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in mirror.state_dict().items())
    mirror.load_state_dict(sd)
    assert rel_err(mirror(x.cuda()), want) < 2e-3
    # and PyTorch's own wrapper agrees with what was trained: a torch module with the reference's layer structure
    convs = [torch.nn.utils.spectral_norm(nn.Conv1d(ci, co, k, 1, k // 2)) for (ci, co, k) in eng.chans]
    for i, c in enumerate(convs):
        c.load_state_dict({"weight_orig": sd[f"encoder.conv.{i}.net.0.weight_orig"], "bias": sd[f"encoder.conv.{i}.net.0.bias"],
                           "weight_u": sd[f"encoder.conv.{i}.net.0.weight_u"], "weight_v": sd[f"encoder.conv.{i}.net.0.weight_v"]})
        c.eval()
        xin = torch.randn(2, eng.chans[i][0], 12, generator=g)
        w_eff = O.spectral_norm_weight(sd[f"encoder.conv.{i}.net.0.weight_orig"], sd[f"encoder.conv.{i}.net.0.weight_u"].clone(),
                                       sd[f"encoder.conv.{i}.net.0.weight_v"].clone(), train=False)
        torch.testing.assert_close(c(xin), F.conv1d(xin, w_eff, c.bias, 1, eng.chans[i][2] // 2), rtol=1e-5, atol=1e-5)


def test_graph_replay_of_the_spectral_norm_step_equals_eager():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    B, T, C = 4, 16, 4
    cfg = dict(_sn_cfg(C), batch_size=B, max_notes=T, optimizer=dict(name="AdamW", lr=2e-4, betas=[0.5, 0.999], weight_decay=0.01))
    engs = []
    for _ in range(2):
        e = EdEngine(cfg, "cuda", B, T)
        e.init_weights(3)
        engs.append(e)
    g = torch.Generator().manual_seed(1)
    x, y = (torch.rand(B, T, C, generator=g) * 2 - 1).cuda(), torch.randint(0, 4, (B,), generator=g).cuda()
    with torch.cuda.stream(engs[0].stream):
        for e in engs:
            e.set_batch(x, y)
        for it in range(5):
            engs[0].run("step_rng", True)
            engs[1].run("step_rng", False)
        torch.cuda.synchronize()
    assert torch.equal(engs[0].P.data, engs[1].P.data)
    for k in engs[0].buf:
        assert torch.equal(engs[0].buf[k], engs[1].buf[k]), k
    assert float(engs[0].buf["encoder.conv.0.net.0.weight_u"].norm()) == pytest.approx(1.0, abs=1e-5)


def test_feature_encoder_with_spectral_norm_matches_torchs_wrapper():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.feature_encoder import FeatureEncoder
    torch.manual_seed(0)
    sn = torch.nn.utils.spectral_norm
    ref = nn.Sequential(nn.LayerNorm(6), sn(nn.Linear(6, 256)), nn.GELU(), nn.Dropout(0.0), sn(nn.Linear(256, 128)), nn.GELU(),
                        nn.Dropout(0.0), nn.Linear(128, 128))
    mine = FeatureEncoder(6, (256, 128), 128, dropout=0.0, use_sn=True)
    assert set(mine.state_dict()) == {"net." + k for k in ref.state_dict()}
    mine.load_state_dict({"net." + k: v for k, v in ref.state_dict().items()})
    mine = mine.cuda()
    x = torch.randn(5, 6)
    for mode in ("train", "train", "eval"):          # two training forwards: the buffers move in step with the wrapper's
        getattr(ref, mode)()
        getattr(mine, mode)()
        want = ref(x)
        got = mine(x.cuda())
        assert rel_err(got, want) < 1e-4, mode
        assert rel_err(mine.net[1].weight_u, ref[1].weight_u) < 1e-5


def test_trainer_cli_with_spectral_norm_and_the_checkpoint_feeds_the_gan(tmp_path):
    """`use_spectral_norm: true` through the train_ed CLI path (synthetic, learnable labels): the loss falls, the checkpoint
    carries weight_orig / weight_u / weight_v, and the GAN trainer's frozen copy gets the NORMALISED weights (sigma folded at
    load time: train_gan.load_ed_checkpoint) -- its eval forward equals the training engine's."""
    import os
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator import train_ed
    from melo_gan_amd.gan import train_gan
    from melo_gan_amd.gan.engine import GanEngine
    cfg = dict(O.default_ed_cfg(4), dropout=0.2, batch_size=32, max_notes=32, num_epochs=6, seed=1, use_spectral_norm=True,
               optimizer=dict(name="AdamW", lr=2e-3, betas=[0.5, 0.999], weight_decay=0.0), metric_for_best="val_loss",
               early_stopping_patience=10, save_freq=3, checkpoint_dir=str(tmp_path), save_name="ed_best.pth")
    eng, best = train_ed.train(cfg, synthetic=512, use_graph=True)
    assert best < 1.2, best                                      # below ln 4 = 1.386
    ck = torch.load(os.path.join(str(tmp_path), "ed_best.pth"), map_location="cpu", weights_only=False)
    assert "encoder.conv.1.net.0.weight_orig" in ck["model"] and "classifier.net.3.weight_v" in ck["model"]
    assert "encoder.conv.1.net.0.weight" not in ck["model"]
    gan = GanEngine(O.default_gan_cfg(4, 32, 4), O.default_ed_cfg(4), "cuda", 4)
    gan.init_weights(0)
    assert train_gan.load_ed_checkpoint(gan, os.path.join(str(tmp_path), "ed_best.pth"))
    w_eff = O.spectral_norm_weight(ck["model"]["encoder.conv.1.net.0.weight_orig"], ck["model"]["encoder.conv.1.net.0.weight_u"].clone(),
                                   ck["model"]["encoder.conv.1.net.0.weight_v"].clone(), train=False)
    assert rel_err(gan.ED.p["encoder.conv.1.net.0.weight"], w_eff) < 1e-5
