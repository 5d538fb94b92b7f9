#!/usr/bin/env python3
"""
Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES
(imported from /root/reference, build container only -- the reference never
travels to the GPU box; only these small data fixtures do).

    python tests/golden/make_golden.py

What is recorded
  * inputs that are not closed-form (injected noise, alpha, dropout keep-masks,
    the synthetic batch is regenerated from its seed),
  * outputs of the reference modules: per-layer activations, logits, losses,
    gradient-penalty, gradient checksums, post-Adam parameter checksums,
    generated (T, C) tensors.
Weights are NOT stored: both sides fill them from the closed-form rule in
oracle/melo_oracle.py (fill_params), loaded here with load_state_dict.

The train *step* is not a reference function (src/gan/train_gan.py:63-285 is a
__main__ block), so it is restated here as straight-line calls over the imported
reference modules, following train_gan.py:183-251 line by line.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("MELO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
# src/gan/utils.py:11 imports pretty_midi at module level (absent here); only the
# MIDI writer needs it, so an empty stub module makes the training utils importable.
sys.modules.setdefault("pretty_midi", types.ModuleType("pretty_midi"))

from src.gan.models import Generator, Discriminator            # noqa: E402
from src.gan.feature_encoder import FeatureEncoder              # noqa: E402
from src.emotion_discriminator.ed_model import EmotionDiscriminator  # noqa: E402
from src.gan.utils import compute_gradient_penalty              # noqa: E402
from src.ae.model import VAE                                    # noqa: E402

from oracle import melo_oracle as O                             # noqa: E402

torch.set_num_threads(1)


def load_into(module: nn.Module, params, buffers=None):
    sd = module.state_dict()
    for k, v in params.items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), (k, sd.get(k, None) is None)
        sd[k] = v.clone()
    if buffers:
        for k, v in buffers.items():
            assert k in sd, k
            sd[k] = v.clone()
    module.load_state_dict(sd, strict=True)


def checksum(t: torch.Tensor):
    t = t.detach().double().flatten()
    w = torch.cos(0.11 * torch.arange(t.numel(), dtype=torch.float64))
    return np.array([t.sum().item(), (t * w).sum().item(), t.abs().sum().item()])


class DropCapture:
    """Records the keep-mask each nn.Dropout drew (train mode)."""

    def __init__(self, module):
        self.masks = []
        self.hooks = [m.register_forward_hook(self._hook) for m in module.modules() if isinstance(m, nn.Dropout)]

    def _hook(self, mod, inp, out):
        x = inp[0]
        if mod.training:
            keep = torch.where(x != 0, (out != 0).float(), torch.ones_like(x))
        else:
            keep = torch.ones_like(x)
        self.masks.append(keep.detach().clone())

    def pop(self):
        m, self.masks = self.masks, []
        return m


def build_reference(cfg, ed_cfg, S):
    E = FeatureEncoder(cfg["NUMERIC_INPUT_DIM"], cfg["ENCODER_HIDDEN"], cfg["ENCODER_OUT_DIM"])
    G = Generator(cfg["NOISE_DIM"], cfg["LATENT_DIM"], cfg["INTEGRATION_MODE"], max_notes=cfg["MAX_NOTES"],
                  note_dim=cfg["NOTE_DIM"], numeric_embed_dim=cfg["ENCODER_OUT_DIM"])
    D = Discriminator(cfg["MAX_NOTES"], cfg["NOTE_DIM"], numeric_embed_dim=cfg["ENCODER_OUT_DIM"])
    ED = EmotionDiscriminator(ed_cfg)
    load_into(E, S.PE)
    load_into(G, S.PG, S.BG)
    load_into(D, S.PD)
    load_into(ED, S.PED, S.BED)
    for p in ED.parameters():
        p.requires_grad = False
    ED.eval()
    return E, G, D, ED


def gan_case(name, B, T, C, n_steps=3, mode="warm_start", ed_mode="notes", seed=7, d_scale=1.0):
    cfg = O.default_gan_cfg(B, T, C)
    cfg["INTEGRATION_MODE"] = mode
    ed_cfg = O.default_ed_cfg(C)
    ed_cfg["input_mode"] = ed_mode
    S = O.build_gan_state(cfg, ed_cfg, "closed_form", d_scale=d_scale)
    E, G, D, ED = build_reference(cfg, ed_cfg, S)
    opt_G = torch.optim.Adam(list(G.parameters()) + list(E.parameters()), lr=cfg["LR_G"],
                             betas=(cfg["BETA1"], cfg["BETA2"]))
    opt_D = torch.optim.Adam(D.parameters(), lr=cfg["LR_D"], betas=(cfg["BETA1"], cfg["BETA2"]))
    real, numeric, latent, emot_idx = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, seed)
    if mode == "conditioning":
        latent = O.closed_form((B, cfg["LATENT_DIM"]), 9.0, 0.5)
    cap = DropCapture(E)
    crit = nn.CrossEntropyLoss()
    out = dict(B=B, T=T, C=C, n_steps=n_steps, seed=seed, mode=mode, ed_mode=ed_mode, d_scale=d_scale)
    G.train(); E.train(); D.train()
    dev = torch.device("cpu")
    for it in range(n_steps):
        # ---- D-step: train_gan.py:183-205 ----
        torch.manual_seed(1000 + it)
        opt_D.zero_grad()
        with torch.no_grad():
            emb_d = E(numeric)
            noise = torch.randn(B, cfg["NOISE_DIM"])
            fake_d, _ = G(noise, latent, emb_d)
        dm_d = cap.pop()
        d_real = D(real, emb_d)
        d_fake = D(fake_d.detach(), emb_d)
        rng_state = torch.get_rng_state()
        alpha = torch.rand(B, 1, 1)              # what compute_gradient_penalty will draw (utils.py:76)
        torch.set_rng_state(rng_state)
        gp = compute_gradient_penalty(D, real.data, fake_d.data, emb_d, dev)
        loss_d = torch.mean(d_fake) - torch.mean(d_real) + cfg["LAMBDA_GP"] * gp
        loss_d.backward()
        if it == 0:
            for k, p in D.named_parameters():
                out[f"s0.dgrad.{k}"] = checksum(p.grad)
            out["s0.dgrad_conv0_w"] = D.conv[0].weight.grad[:4].detach().numpy().copy()
            out["s0.dgrad_conv4_w"] = D.conv[4].weight.grad[:2].detach().numpy().copy()
            out["s0.dgrad_fc1_w"] = D.fc[1].weight.grad[:4].detach().numpy().copy()
        opt_D.step()
        # ---- G-step: train_gan.py:211-251 ----
        opt_G.zero_grad()
        emb_g = E(numeric)
        noise_g = torch.randn(B, cfg["NOISE_DIM"])
        fake_g, lat_g = G(noise_g, latent, emb_g)
        dm_g = cap.pop()
        d_fake_g = D(fake_g, emb_g)
        adv = -torch.mean(d_fake_g)
        ed_in = lat_g if ed_mode == "latent" else fake_g
        logits = ED(ed_in)
        emo = crit(logits, emot_idx)
        loss_g = adv + cfg["LAMBDA_EMOTION"] * emo
        loss_g.backward()
        if it == 0:
            for k, p in G.named_parameters():
                out[f"s0.ggrad.G.{k}"] = checksum(p.grad)
            for k, p in E.named_parameters():
                out[f"s0.ggrad.E.{k}"] = checksum(p.grad if p.grad is not None else torch.zeros_like(p))
            out["s0.logits"] = logits.detach().numpy().copy()
        opt_G.step()
        out[f"s{it}.noise_d"] = noise.numpy().copy()
        out[f"s{it}.alpha"] = alpha.numpy().copy()
        out[f"s{it}.noise_g"] = noise_g.numpy().copy()
        for j, m in enumerate(dm_d):
            out[f"s{it}.dm_d{j}"] = m.numpy().astype(np.uint8)
        for j, m in enumerate(dm_g):
            out[f"s{it}.dm_g{j}"] = m.numpy().astype(np.uint8)
        out[f"s{it}.loss_d"] = np.float64(loss_d.item())
        out[f"s{it}.gp"] = np.float64(gp.item())
        out[f"s{it}.d_real"] = d_real.detach().numpy().copy()
        out[f"s{it}.d_fake"] = d_fake.detach().numpy().copy()
        out[f"s{it}.adv"] = np.float64(adv.item())
        out[f"s{it}.emo"] = np.float64(emo.item())
        if it == 0:
            out["s0.fake_d"] = fake_d.numpy().copy()
    # post-training state
    for k, v in D.state_dict().items():
        out[f"end.D.{k}"] = checksum(v)
    for k, v in G.state_dict().items():
        out[f"end.G.{k}"] = checksum(v.float())
    for k, v in E.state_dict().items():
        out[f"end.E.{k}"] = checksum(v)
    # generated tensor in eval mode (app.py:92-119 contract: E_num -> G, eval)
    G.eval(); E.eval()
    with torch.no_grad():
        z = O.closed_form((B, cfg["NOISE_DIM"]), 11.0, 1.0)
        gen, _ = G(z, latent, E(numeric))
    out["end.generated"] = gen.numpy().copy()
    # the same generation with the generator's BatchNorm on BATCH statistics (G.train(), E_num still eval): unlike the
    # eval-mode output it does not depend on the pre-BatchNorm conv biases / running means, whose values are Adam-amplified
    # rounding noise in the reference itself -- the tight pin of a free-running end state
    G.train()
    with torch.no_grad():
        gen_tr, _ = G(z, latent, E(numeric))
    out["end.generated_train"] = gen_tr.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss_d", [out[f"s{i}.loss_d"] for i in range(n_steps)], "emo", out[f"s{n_steps-1}.emo"])


def layers_case(name, B, T, C):
    """Per-layer activations of D, G (train + eval), ED and E_num at a tiny shape."""
    cfg = O.default_gan_cfg(B, T, C)
    ed_cfg = O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "closed_form")
    E, G, D, ED = build_reference(cfg, ed_cfg, S)
    real, numeric, latent, emot_idx = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, 3)
    out = dict(B=B, T=T, C=C)
    acts = {}

    def hook(tag):
        def f(mod, inp, o):
            acts[tag] = o.detach().clone()
        return f
    hs = []
    for i, m in enumerate(D.conv):
        hs.append(m.register_forward_hook(hook(f"D.conv.{i}")))
    for i, m in enumerate(G.decoder.deconv):
        hs.append(m.register_forward_hook(hook(f"G.deconv.{i}")))
    for i, m in enumerate(G.decoder.pre):
        hs.append(m.register_forward_hook(hook(f"G.pre.{i}")))
    for i, blk in enumerate(ED.encoder.conv):
        hs.append(blk.register_forward_hook(hook(f"ED.block.{i}")))
    hs.append(ED.encoder.register_forward_hook(hook("ED.encoder")))
    E.eval(); D.train(); G.train()
    with torch.no_grad():
        emb = E(numeric)
        noise = O.closed_form((B, cfg["NOISE_DIM"]), 5.0, 1.0)
        fake, lat = G(noise, latent, emb)
        out["G.train.fake"] = fake.numpy().copy()
        out["G.train.latent"] = lat.numpy().copy()
        for k in list(acts):
            if k.startswith("G."):
                out["G.train." + k[2:]] = acts[k].numpy().copy()
        out["G.train.running_mean1"] = G.decoder.deconv[1].running_mean.numpy().copy()
        out["G.train.running_var1"] = G.decoder.deconv[1].running_var.numpy().copy()
        G.eval()
        fake_e, _ = G(noise, latent, emb)
        out["G.eval.fake"] = fake_e.numpy().copy()
        score = D(real, emb)
        out["D.score"] = score.numpy().copy()
        for k in list(acts):
            if k.startswith("D."):
                out[k] = acts[k].numpy().copy()
        logits = ED(real)
        out["ED.logits"] = logits.numpy().copy()
        for k in list(acts):
            if k.startswith("ED."):
                out[k] = acts[k].numpy().copy()
        out["E.emb_eval"] = emb.numpy().copy()
    # input gradients of D and ED (first-order), and the GP value for a fixed alpha
    x = real.clone().requires_grad_(True)
    s = D(x, emb).sum()
    out["D.input_grad"] = torch.autograd.grad(s, x)[0].numpy().copy()
    x = real.clone().requires_grad_(True)
    ce = nn.CrossEntropyLoss()(ED(x), emot_idx)
    out["ED.ce"] = np.float64(ce.item())
    out["ED.input_grad"] = torch.autograd.grad(ce, x)[0].numpy().copy()
    for h in hs:
        h.remove()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def ae_case(name, B, T, latent_dim=8, n_steps=2):
    cfg = dict(MAX_NOTES=T, LATENT_DIM=latent_dim)
    model = VAE(cfg)
    with torch.no_grad():
        model.encoder(torch.zeros(1, T, 4))         # materialise encoder._linear (train_ae.py:75-79)
    spec, bufs = O.vae_spec(T, latent_dim)
    P = O.fill_params(spec, 6.0, O.norm_affine_names(spec))
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    load_into(model, P, Bf)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, T, 4, generator=g) * 2 - 1
    out = dict(B=B, T=T, latent_dim=latent_dim, n_steps=n_steps)
    model.train()
    for it in range(n_steps):
        torch.manual_seed(2000 + it)
        st = torch.get_rng_state()
        eps = torch.randn(B, latent_dim)              # what reparameterize will draw (model.py:132)
        torch.set_rng_state(st)
        recon, z, mu, log_var = model(x)
        recon_loss = torch.nn.functional.mse_loss(recon, x)
        kld = -0.5 * torch.mean(1 + log_var - mu.pow(2) - log_var.exp())
        loss = recon_loss + 10.0 * kld
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        out[f"s{it}.eps"] = eps.numpy().copy()
        out[f"s{it}.loss"] = np.float64(loss.item())
        out[f"s{it}.recon_loss"] = np.float64(recon_loss.item())
        out[f"s{it}.kld"] = np.float64(kld.item())
        out[f"s{it}.grad_norm"] = np.float64(gn.item())
        if it == 0:
            out["s0.recon"] = recon.detach().numpy().copy()
            out["s0.mu"] = mu.detach().numpy().copy()
            out["s0.log_var"] = log_var.detach().numpy().copy()
    for k, v in model.state_dict().items():
        out[f"end.{k}"] = checksum(v.float())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", [out[f"s{i}.loss"] for i in range(n_steps)])


def ed_train_case(name, B, T, C, n_steps=3):
    """SURVEY f-2: the emotion discriminator's pre-training step (train_ed.py:51-82) on the reference module itself:
    train mode (BatchNorm batch statistics), CrossEntropyLoss, AdamW(lr 2e-4, betas (0.5, 0.999), wd 0.01).
    The classifier's dropout (p = 0.2) is live; the keep-masks nn.Dropout drew are captured and stored."""
    ed_cfg = dict(O.default_ed_cfg(C), dropout=0.2)
    ED = EmotionDiscriminator(ed_cfg)
    spec, bufs = O.emotion_disc_spec(ed_cfg)
    P = O.fill_params(spec, 9.0, O.norm_affine_names(spec))
    for k, v in P.items():            # x4 on the matrices: logits away from 0, per-class gradients of different sizes
        if v.dim() >= 2:
            v.mul_(4.0)
    Bf = {k: (torch.ones(s) if k.endswith("running_var") else torch.zeros(s)) for k, s in bufs.items()}
    load_into(ED, P, Bf)
    opt = torch.optim.AdamW(ED.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=0.01)
    crit = nn.CrossEntropyLoss()
    g = torch.Generator().manual_seed(11)
    out = dict(B=B, T=T, C=C, n_steps=n_steps)
    cap = DropCapture(ED)
    ED.train()
    for it in range(n_steps):
        x = torch.rand(B, T, C, generator=g) * 2 - 1
        y = torch.randint(0, 4, (B,), generator=g)
        opt.zero_grad()
        torch.manual_seed(3000 + it)
        logits = ED(x)
        for j, m in enumerate(cap.pop()):
            out[f"s{it}.dm{j}"] = m.numpy().copy()
        loss = crit(logits, y)
        loss.backward()
        opt.step()
        out[f"s{it}.x"], out[f"s{it}.y"] = x.numpy().copy(), y.numpy().copy()
        out[f"s{it}.loss"] = np.float64(loss.item())
        out[f"s{it}.logits"] = logits.detach().numpy().copy()
        if it == 0:
            out["s0.grad.head_w"] = ED.classifier.head.weight.grad.numpy().copy()
            out["s0.grad.conv0_w"] = ED.encoder.conv[0].net[0].weight.grad.numpy().copy()
    sd = ED.state_dict()
    for k, v in sd.items():
        out[f"end.{k}"] = checksum(v.float())
    out["end.head_w"] = sd["classifier.head.weight"].numpy().copy()
    out["end.rm3"] = sd["encoder.conv.3.net.1.running_mean"].numpy().copy()
    ED.eval()
    cap.pop()
    with torch.no_grad():
        out["end.eval_logits"] = ED(torch.from_numpy(out["s0.x"])).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", [out[f"s{i}.loss"] for i in range(n_steps)])


def midi_case(name):
    """Output contract (SURVEY f-1): the note events save_piano_roll_to_midi (src/gan/utils.py:95-161) derives from
    a generated (T, 4) tensor.  pretty_midi is absent, so a recording stand-in for the four names the function uses
    captures the Note objects it creates; the SMF byte serialisation itself (pretty_midi.write) is NOT pinned."""
    rec = {}

    class Note:
        def __init__(self, velocity, pitch, start, end):
            self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    class Instrument:
        def __init__(self, program):
            self.program, self.notes = program, []

    class PrettyMIDI:
        def __init__(self, initial_tempo=120.0):
            rec["tempo"] = initial_tempo
            self.instruments = []

        def write(self, path):
            rec["notes"] = [(n.velocity, n.pitch, n.start, n.end) for n in self.instruments[0].notes]
            rec["program"] = self.instruments[0].program

    pm = sys.modules["pretty_midi"]
    pm.Note, pm.Instrument, pm.PrettyMIDI = Note, Instrument, PrettyMIDI
    pm.instrument_name_to_program = lambda nm: {"Acoustic Grand Piano": 0, "Violin": 40}[nm]
    from src.gan.utils import save_piano_roll_to_midi
    g = torch.Generator().manual_seed(11)
    roll = (torch.rand(96, 4, generator=g) * 2.4 - 1.2).numpy().astype(np.float32)   # beyond [-1,1] to hit the clips
    out = dict(roll=roll)
    for tag, kw in (("a", dict(bpm=120.0, scale="major", root_key=0)),
                    ("b", dict(bpm=200.0, scale="minor_pentatonic", root_key=7, instrument_name="Violin")),
                    ("c", dict(bpm=45.0, scale="not_a_scale", root_key=3))):
        save_piano_roll_to_midi(roll, "/dev/null", **kw)
        out[f"{tag}.notes"] = np.array(rec["notes"], dtype=np.float64)
        out[f"{tag}.tempo"] = np.float64(rec["tempo"])
        out[f"{tag}.program"] = np.int64(rec["program"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k.endswith("notes")})


def install_midi_recorder(rec):
    """Recording stand-ins for the four pretty_midi names save_piano_roll_to_midi uses (pretty_midi is absent)."""
    class Note:
        def __init__(self, velocity, pitch, start, end):
            self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    class Instrument:
        def __init__(self, program):
            self.program, self.notes = program, []

    class PrettyMIDI:
        def __init__(self, initial_tempo=120.0):
            rec["tempo"] = initial_tempo
            self.instruments = []

        def write(self, path):
            rec["notes"] = [(n.velocity, n.pitch, n.start, n.end) for n in self.instruments[0].notes]
            rec["program"] = self.instruments[0].program

    pm = sys.modules["pretty_midi"]
    pm.Note, pm.Instrument, pm.PrettyMIDI = Note, Instrument, PrettyMIDI
    pm.instrument_name_to_program = lambda nm: {"Acoustic Grand Piano": 0, "Violin": 40}[nm]


def gen1_case(name, T, C, g_scale=4.0, last_scale=400.0):
    """BASELINE config 5 / app.py:92-119: batch-1 generation, E_num -> G in eval mode (dropout off, BatchNorm on
    NON-trivial running statistics), at the reference's own shape (T=512, C=4) and at the 128x256 roll.  Generator
    weights are the closed-form fill x g_scale so that the output spans the MIDI writer's branches; for C=4 the note
    events the reference's save_piano_roll_to_midi derives from the generated roll are recorded too."""
    cfg, ed_cfg = O.default_gan_cfg(1, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "closed_form")
    for k in S.PG:
        if k.endswith("weight") and S.PG[k].dim() > 1:
            S.PG[k].mul_(g_scale * (last_scale if k == "decoder.deconv.6.weight" else 1.0))
    S.BG.update(O.fill_buffers(O.generator_buffers(), 70.0))
    z = O.closed_form((1, cfg["NOISE_DIM"]), 13.0, 1.0)
    numeric = O.closed_form((1, 6), 17.0, 1.0)
    # centre every output channel (the last layer's bias is recorded in the fixture: not closed-form)
    S.PG["decoder.deconv.6.bias"].zero_()
    E, G, _, _ = build_reference(cfg, ed_cfg, S)
    G.eval(); E.eval()
    with torch.no_grad():
        S.PG["decoder.deconv.6.bias"].copy_(-G(z, None, E(numeric))[0][0].mean(0))
    E, G, _, _ = build_reference(cfg, ed_cfg, S)
    G.eval(); E.eval()
    with torch.no_grad():
        emb = E(numeric)
        gen, lat = G(z, None, emb)
    out = dict(bias6=S.PG["decoder.deconv.6.bias"].numpy().copy(), B=1, T=T, C=C, g_scale=g_scale, last_scale=last_scale, emb=emb.numpy().copy(), generated=gen.numpy().copy(), latent=lat.numpy().copy())
    if C == 4:
        rec = {}
        install_midi_recorder(rec)
        from src.gan.utils import save_piano_roll_to_midi
        save_piano_roll_to_midi(gen[0].numpy(), "/dev/null", bpm=100.0, scale="minor", root_key=2)
        out["notes"] = np.array(rec["notes"], dtype=np.float64)
        out["tempo"] = np.float64(rec["tempo"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "generated", tuple(gen.shape), "min/max", float(gen.min()), float(gen.max()),
          "notes", out.get("notes", np.zeros((0, 4))).shape)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gen1":           # only the cfg5 fixtures
    gen1_case("gen1_c4_t512", 512, 4)
    gen1_case("gen1_c128_t256", 256, 128)
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ed_train":     # only the f-2 fixtures
    ed_train_case("ed_train_c4_t32_b8", 8, 32, 4)
    ed_train_case("ed_train_c128_t16_b4", 4, 16, 128, n_steps=2)
elif __name__ == "__main__":
    gan_case("gan_c128_t64_b4", 4, 64, 128)                    # cfg1 shape at B=4
    gan_case("gan_c4_t32_b4", 4, 32, 4)                        # reference shape (C=4) scaled down
    gan_case("gan_c4_t32_b4_bigD", 4, 32, 4, d_scale=6.0)      # critic weights x6: GP far from 1, mixed masks
    gan_case("gan_c4_t20_b3", 3, 20, 4, n_steps=2)             # T % 8 != 0 -> zero-pad branch (models.py:76-81)
    gan_case("gan_c4_t16_cond_lat", 3, 16, 4, n_steps=2, mode="conditioning", ed_mode="latent")
    layers_case("layers_c4_t16_b2", 2, 16, 4)
    layers_case("layers_c128_t32_b2", 2, 32, 128)
    ae_case("ae_t32_b4", 4, 32)
    midi_case("midi_events")
    ed_train_case("ed_train_c4_t32_b8", 8, 32, 4)
    ed_train_case("ed_train_c128_t16_b4", 4, 16, 128, n_steps=2)
    gen1_case("gen1_c4_t512", 512, 4)
    gen1_case("gen1_c128_t256", 256, 128)
