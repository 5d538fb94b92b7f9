"""Emotion-discriminator pre-training step (SURVEY f-2, src/emotion_discriminator/train_ed.py:51-82) on the GPU against the
reference-generated fixtures and the oracle: train-mode BatchNorm + GELU, classifier dropout with injected masks,
cross-entropy, every parameter gradient, AdamW, running statistics, eval-mode forward afterwards."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# conv biases in front of a train-mode BatchNorm: mathematically zero gradient (pure rounding noise, Adam-amplified)
PRE_BN_BIAS = tuple(f"encoder.conv.{i}.net.0.bias" for i in range(4))


def rel_err(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def initial_state(C):
    ed_cfg = dict(O.default_ed_cfg(C), dropout=0.2)
    spec, bufs = O.emotion_disc_spec(ed_cfg)
    P = O.fill_params(spec, 9.0, O.norm_affine_names(spec))
    for v in P.values():
        if v.dim() >= 2:
            v.mul_(4.0)
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    return ed_cfg, spec, P, Bf


@pytest.mark.parametrize("name", ["ed_train_c4_t32_b8", "ed_train_c128_t16_b4"])
def test_ed_pretraining_steps_match_reference(name):
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, T, C, n_steps = int(g["B"]), int(g["T"]), int(g["C"]), int(g["n_steps"])
    ed_cfg, spec, P, Bf = initial_state(C)
    cfg = dict(ed_cfg, batch_size=B, max_notes=T, optimizer=dict(name="AdamW", lr=2e-4, betas=[0.5, 0.999], weight_decay=0.01))
    eng = EdEngine(cfg, "cuda", B, T)
    assert list(eng.P.spec) == list(spec) and all(tuple(eng.P.spec[k]) == tuple(spec[k]) for k in spec)
    eng.load_state(P, Bf)
    opt = O.AdamState(P, 2e-4, (0.5, 0.999), 1e-8, weight_decay=0.01, decoupled=True)
    for it in range(n_steps):
        x, y = torch.from_numpy(g[f"s{it}.x"]), torch.from_numpy(g[f"s{it}.y"])
        dm = [torch.from_numpy(g[f"s{it}.dm{j}"]).float() / 0.8 for j in range(2)]
        eng.set_batch(x.cuda(), y.cuda())
        eng.set_masks([m.cuda() for m in dm])
        eng.backward()
        # fp64 truth and the reference's own fp32 arithmetic from the same (re-synchronised) parameters
        P64 = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
        B64 = {k: v.double().clone() for k, v in Bf.items()}
        l64 = torch.nn.functional.cross_entropy(O.emotion_disc_fwd(P64, B64, x.double(), ed_cfg, True, [m.double() for m in dm]), y)
        g64 = dict(zip(P64, torch.autograd.grad(l64, list(P64.values()))))
        old = {k: v.clone() for k, v in P.items()}
        r = O.ed_step(P, Bf, opt, x, y, ed_cfg, dm)
        assert abs(eng.loss.item() - float(g[f"s{it}.loss"])) < 5e-6
        np.testing.assert_allclose(eng.logits.cpu().numpy(), g[f"s{it}.logits"], rtol=2e-3, atol=2e-5)
        if it == 0:
            np.testing.assert_allclose(eng.P.g["classifier.head.weight"].cpu().numpy(), g["s0.grad.head_w"], rtol=2e-3, atol=1e-6)
        for k in spec:
            if k in PRE_BN_BIAS:
                continue
            e_mine, e_ref = rel_err(eng.P.g[k], g64[k]), rel_err(r["grads"][k], g64[k])
            assert e_mine <= 6.0 * e_ref + 3e-4, (it, k, e_mine, e_ref)
        eng.update()
        for k in spec:
            if k in PRE_BN_BIAS:
                continue
            upd, upd_ref = eng.P.p[k].cpu() - old[k], P[k] - old[k]
            assert rel_err(upd, upd_ref) < 0.1, (it, "AdamW update", k, rel_err(upd, upd_ref))
        for k in Bf:
            if not k.endswith("num_batches_tracked"):
                tol = 2e-4 * (it + 1) if k.endswith("running_mean") else 0.0
                np.testing.assert_allclose(eng.buf[k].cpu().numpy(), Bf[k].numpy(), rtol=1e-4, atol=tol + 1e-6)
        # teacher forcing: each step is judged from identical inputs
        eng.load_state(P, Bf)
    np.testing.assert_allclose(eng.P.p["classifier.head.weight"].cpu().numpy(), g["end.head_w"], rtol=1e-4, atol=2e-6)
    eng.set_batch(torch.from_numpy(g["s0.x"]).cuda(), torch.from_numpy(g["s0.y"]).cuda())
    eng.forward_eval()
    np.testing.assert_allclose(eng.logits.cpu().numpy(), g["end.eval_logits"], rtol=2e-3, atol=1e-4)
    sd = eng.state_dict()
    assert int(sd["encoder.conv.0.net.1.num_batches_tracked"]) == n_steps
    assert set(f"end.{k}" for k in sd) == set(k for k in g.files if k.startswith("end.")) - {"end.head_w", "end.rm3", "end.eval_logits"}


def test_ed_graph_replay_trains():
    """backward_rng/update as replayed hipGraphs with device-drawn dropout masks: the loss of a fixed batch falls."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    cfg = dict(O.default_ed_cfg(4), dropout=0.2, optimizer=dict(name="AdamW", lr=1e-3, betas=[0.5, 0.999], weight_decay=0.0))
    eng = EdEngine(cfg, "cuda", 16, 32)
    _, spec, P, Bf = initial_state(4)
    eng.load_state(P, Bf)
    gen = torch.Generator().manual_seed(3)
    x = (torch.rand(16, 32, 4, generator=gen) * 2 - 1).cuda()
    y = torch.randint(0, 4, (16,), generator=gen).cuda()
    losses = []
    with torch.cuda.stream(eng.stream):
        eng.set_batch(x, y)
        for _ in range(60):
            eng.run("backward_rng")
            eng.run("update")
            losses.append(eng.loss.item())
    assert int(eng.rng_step.item()) == 60 and float(eng.P.state[0].item()) == 60.0
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])


def test_ed_trainer_cli_smoke_and_checkpoint_feeds_the_gan_trainer(tmp_path):
    """train_ed CLI path on synthetic, learnable labels: validation accuracy rises, the best checkpoint has the reference's
    layout ({'epoch','model','optimizer','cfg'}) and loads into the GAN engine's frozen emotion discriminator."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator import train_ed
    from melo_gan_amd.gan import train_gan
    from melo_gan_amd.gan.engine import GanEngine
    cfg = dict(O.default_ed_cfg(4), dropout=0.2, batch_size=32, max_notes=32, num_epochs=8, seed=1,
               optimizer=dict(name="AdamW", lr=2e-3, betas=[0.5, 0.999], weight_decay=0.0),
               scheduler=dict(name="ReduceLROnPlateau", mode="min", factor=0.5, patience=1, threshold=1e-4),
               metric_for_best="val_loss", early_stopping_patience=10, save_freq=4,
               checkpoint_dir=str(tmp_path), save_name="ed_best.pth")
    eng, best = train_ed.train(cfg, synthetic=512, use_graph=True)
    assert best < 1.0, best                                      # well below ln 4 = 1.386
    ck = torch.load(os.path.join(str(tmp_path), "ed_best.pth"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "cfg"}
    assert os.path.exists(os.path.join(str(tmp_path), "ed_epoch004.pth"))
    gcfg, ecfg = O.default_gan_cfg(4, 32, 4), O.default_ed_cfg(4)
    gan = GanEngine(gcfg, ecfg, "cuda", 4)
    gan.init_weights(0)
    assert train_gan.load_ed_checkpoint(gan, os.path.join(str(tmp_path), "ed_best.pth"))
    for k in gan.ED.spec:
        assert torch.equal(gan.ED.p[k].cpu(), ck["model"][k]), k
    assert torch.equal(gan.EDbuf["encoder.conv.0.net.1.running_var"].cpu(), ck["model"]["encoder.conv.0.net.1.running_var"])


def test_set_lr_reaches_the_one_graph_training_step():
    """ReduceLROnPlateau (train_ed.py:101-123) under graph replay: the learning rate is baked into the captured
    'step_rng' graph, so set_lr must drop it -- with lr = 0 a replayed step must leave every parameter untouched."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    cfg = dict(O.default_ed_cfg(4), dropout=0.2, optimizer=dict(name="AdamW", lr=1e-3, betas=[0.5, 0.999], weight_decay=0.0))
    eng = EdEngine(cfg, "cuda", 8, 32)
    _, spec, P, Bf = initial_state(4)
    eng.load_state(P, Bf)
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(8, 32, 4, generator=gen) * 2 - 1).cuda()
    y = torch.randint(0, 4, (8,), generator=gen).cuda()
    with torch.cuda.stream(eng.stream):
        eng.set_batch(x, y)
        for _ in range(3):                       # eager warm-up, capture, replay
            eng.run("step_rng")
        torch.cuda.synchronize()
        assert not isinstance(eng._graphs["step_rng"], str)
        before = eng.P.data.clone()
        eng.run("step_rng")
        torch.cuda.synchronize()
        assert not torch.equal(before, eng.P.data)           # lr 1e-3: the replayed step moves the parameters
        eng.set_lr(0.0)
        assert "step_rng" not in eng._graphs
        before = eng.P.data.clone()
        for _ in range(3):                       # eager, capture, replay -- all with lr = 0
            eng.run("step_rng")
        torch.cuda.synchronize()
        assert torch.equal(before, eng.P.data)
    assert float(eng.P.state[0].item()) == 7.0


def test_epoch_includes_the_trailing_partial_batch():
    """The reference's ED loaders have no drop_last and weight each batch by its size (ed_dataset.py:542-558,
    train_ed.py:75-82): an epoch over n = 2B + 3 samples evaluates all of them, n < B does not report 0.0, and a
    training epoch takes ceil(n / B) optimiser steps."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.emotion_discriminator import train_ed
    from melo_gan_amd.emotion_discriminator.engine import EdEngine
    B, T, C = 8, 32, 4
    cfg = dict(O.default_ed_cfg(C), dropout=0.2, optimizer=dict(name="AdamW", lr=1e-3, betas=[0.5, 0.999], weight_decay=0.0))
    ed_cfg, spec, P, Bf = initial_state(C)
    eng = EdEngine(cfg, "cuda", B, T)
    eng.load_state(P, Bf)
    gen = torch.Generator().manual_seed(11)
    for n in (2 * B + 3, 3):
        x = torch.rand(n, T, C, generator=gen) * 2 - 1
        y = torch.randint(0, 4, (n,), generator=gen)
        with torch.no_grad():
            logits = O.emotion_disc_fwd(P, Bf, x, ed_cfg, False, None)
            ref_loss = torch.nn.functional.cross_entropy(logits, y).item()
            ref_acc = (logits.argmax(1) == y).float().mean().item()
        with torch.cuda.stream(eng.stream):
            loss, acc = train_ed.run_epoch(eng, x.cuda(), y.cuda(), False, True)
            loss2, acc2 = train_ed.run_epoch(eng, x.cuda(), y.cuda(), False, True)     # second pass: graphs
        assert abs(loss - ref_loss) < 2e-5 and abs(acc - ref_acc) < 1e-6, (n, loss, ref_loss, acc, ref_acc)
        assert loss2 == loss and acc2 == acc
    steps0 = float(eng.P.state[0].item())
    with torch.cuda.stream(eng.stream):
        train_ed.run_epoch(eng, x.cuda(), y.cuda(), True, True, torch.Generator().manual_seed(0))
        x19 = (torch.rand(19, T, C, generator=gen) * 2 - 1).cuda()
        train_ed.run_epoch(eng, x19, torch.randint(0, 4, (19,), generator=gen).cuda(), True, True, torch.Generator().manual_seed(0))
    assert float(eng.P.state[0].item()) == steps0 + 1 + 3
    with pytest.raises(ValueError):
        train_ed.run_epoch(eng, x.cuda()[:0], y.cuda()[:0], False, True)


def test_out_of_range_label_poisons_the_loss_instead_of_reading_out_of_bounds():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    from melo_gan_amd.gan.utils import check_labels
    logits = torch.randn(6, 4, device="cuda")
    loss, dl = torch.zeros(1, device="cuda"), torch.zeros(6, 4, device="cuda")
    for bad in (-1, 4):
        y = torch.tensor([0, 1, 2, 3, bad, 1], device="cuda")
        ops.softmax_ce(logits, y, loss, dl, 1.0)
        assert torch.isnan(loss).all() and torch.isnan(dl[4]).all() and torch.isfinite(dl[:4]).all()
        with pytest.raises(ValueError, match="outside"):
            check_labels(y.cpu(), 4)
    ops.softmax_ce(logits, torch.tensor([0, 1, 2, 3, 3, 1], device="cuda"), loss, dl, 1.0)
    assert torch.isfinite(loss).all()
