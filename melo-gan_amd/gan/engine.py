"""GanEngine -- the WGAN-GP critic step and the generator step of Melo-GAN as explicit launch
sequences over libmelogan_hip (no autograd tape).

Restates /root/reference/src/gan/train_gan.py:183-205 (D-step) and :211-251 (G-step) with
hand-derived backward passes:

  * activations are channels-last (B, T, C) end to end -- the reference's permutes
    (src/gan/models.py:73,159; ed_model.py:65) disappear;
  * the three critic evaluations of the D-step (real, fake, x_hat) run as ONE batch of 3B, and
    so does their backward: the Wasserstein terms back-propagate ds = -1/B, +1/B and the
    gradient-penalty's autograd.grad(grad_outputs=ones) (src/gan/utils.py:80-87) back-propagates
    ds = 1 -- the same network, three upstream coefficients;
  * the penalty's second-order term is a hand-derived "tangent" forward pass through the critic
    with the LeakyReLU masks of the x_hat pass (D is piecewise linear: no BatchNorm,
    src/gan/models.py:144-146), after which every weight gradient is one two-segment wgrad:
    [real,fake | tangent] activations against [Wasserstein | penalty] output gradients;
  * parameters, gradients and Adam moments of each optimiser (D; G+E_num, as
    train_gan.py:136-145) live in ONE flat fp32 buffer each -> one fused Adam launch and one
    all-reduce bucket per step; state_dict tensors are views into it.

Randomness (noise, alpha, dropout keep-masks) is an INPUT of every step, so parity runs inject
the oracle's draws and production runs fill the same buffers from the device RNG.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import torch

from .. import ops
from ..ops import ACT_GELU, ACT_LRELU, ACT_RELU

Tensor = torch.Tensor
BN_EPS, BN_MOM, P_DROP = 1e-5, 0.1, 0.2


# ------------------------------------------------------------------------------------------
# parameter specs (reference state_dict names/shapes)
# ------------------------------------------------------------------------------------------
def feature_encoder_spec(in_dim=6, hidden_dims=(256, 128), out_dim=128):
    """src/gan/feature_encoder.py:16-42."""
    spec = OrderedDict([("net.0.weight", (in_dim,)), ("net.0.bias", (in_dim,))])
    prev, idx = in_dim, 1
    for h in hidden_dims:
        spec[f"net.{idx}.weight"] = (h, prev)
        spec[f"net.{idx}.bias"] = (h,)
        prev, idx = h, idx + 3
    spec[f"net.{idx}.weight"] = (out_dim, prev)
    spec[f"net.{idx}.bias"] = (out_dim,)
    return spec


def generator_spec(noise_dim, latent_dim, mode, hidden, max_notes, note_dim, numeric_embed_dim):
    """src/gan/models.py:86-106, :20-27, :33-64."""
    in_dim = noise_dim + numeric_embed_dim + (latent_dim if mode == "conditioning" else 0)
    red = max(1, max_notes // 8)
    return OrderedDict([
        ("noise_to_latent.net.0.weight", (hidden, in_dim)), ("noise_to_latent.net.0.bias", (hidden,)),
        ("noise_to_latent.net.2.weight", (latent_dim, hidden)), ("noise_to_latent.net.2.bias", (latent_dim,)),
        ("decoder.pre.0.weight", (512, latent_dim)), ("decoder.pre.0.bias", (512,)),
        ("decoder.pre.2.weight", (256 * red, 512)), ("decoder.pre.2.bias", (256 * red,)),
        ("decoder.deconv.0.weight", (256, 128, 5)), ("decoder.deconv.0.bias", (128,)),
        ("decoder.deconv.1.weight", (128,)), ("decoder.deconv.1.bias", (128,)),
        ("decoder.deconv.3.weight", (128, 64, 5)), ("decoder.deconv.3.bias", (64,)),
        ("decoder.deconv.4.weight", (64,)), ("decoder.deconv.4.bias", (64,)),
        ("decoder.deconv.6.weight", (64, note_dim, 5)), ("decoder.deconv.6.bias", (note_dim,)),
    ])


def discriminator_spec(note_dim, emb_dim=256, numeric_embed_dim=0):
    """src/gan/models.py:137-157."""
    return OrderedDict([
        ("conv.0.weight", (64, note_dim, 5)), ("conv.0.bias", (64,)),
        ("conv.2.weight", (128, 64, 5)), ("conv.2.bias", (128,)),
        ("conv.4.weight", (256, 128, 5)), ("conv.4.bias", (256,)),
        ("fc.1.weight", (emb_dim, 256)), ("fc.1.bias", (emb_dim,)),
        ("real_fake.weight", (1, emb_dim + numeric_embed_dim)), ("real_fake.bias", (1,)),
    ])


def emotion_disc_spec(cfg: dict):
    """src/emotion_discriminator/ed_model.py:52-58,115-145.  Returns (params, buffers, conv channel list)."""
    spec, bufs, chans = OrderedDict(), OrderedDict(), []
    prev = cfg.get("latent_dim", 128)
    if cfg.get("input_mode", "latent") == "notes":
        hid = cfg.get("notes_hidden", 256)
        in_ch, ch = cfg.get("note_dim", 4), 64
        for i in range(cfg.get("notes_blocks", 4)):
            k = 5 if i == 0 else 3
            chans.append((in_ch, ch, k))
            spec[f"encoder.conv.{i}.net.0.weight"] = (ch, in_ch, k)
            spec[f"encoder.conv.{i}.net.0.bias"] = (ch,)
            spec[f"encoder.conv.{i}.net.1.weight"] = (ch,)
            spec[f"encoder.conv.{i}.net.1.bias"] = (ch,)
            bufs[f"encoder.conv.{i}.net.1.running_mean"] = (ch,)
            bufs[f"encoder.conv.{i}.net.1.running_var"] = (ch,)
            in_ch, ch = ch, min(ch * 2, hid)
        spec["encoder.project.weight"] = (hid, in_ch)
        spec["encoder.project.bias"] = (hid,)
        prev = hid
    idx = 0
    for h in tuple(cfg.get("mlp_hidden", (256, 128))):
        spec[f"classifier.net.{idx}.weight"] = (h, prev)
        spec[f"classifier.net.{idx}.bias"] = (h,)
        prev, idx = h, idx + 3
    spec["classifier.head.weight"] = (cfg.get("n_classes", 4), prev)
    spec["classifier.head.bias"] = (cfg.get("n_classes", 4),)
    return spec, bufs, chans


class FlatParams:
    """All tensors of one optimiser in a single flat fp32 buffer (+ grads, Adam m/v, step state)."""

    def __init__(self, spec: "OrderedDict[str, tuple]", device, with_opt: bool = True, first=None):
        """`first`: tensor (or list of tensors) placed at offset 0 of the flat buffers (the dict order stays the spec's) -- the data-parallel
        wrapper all-reduces that tensor's gradient early and everything behind it as ONE contiguous range."""
        self.spec = spec
        self.n = sum(math.prod(s) for s in spec.values())
        pad = (-self.n) % 4
        self.data = torch.zeros(self.n + pad, device=device)
        self.p: Dict[str, Tensor] = OrderedDict()
        self.offsets = {}
        off = 0
        first = [first] if isinstance(first, str) else list(first or [])
        for k in first + [k for k in spec if k not in first]:
            n = math.prod(spec[k])
            self.offsets[k] = (off, n)
            off += n
        for k, s in spec.items():
            o, n = self.offsets[k]
            self.p[k] = self.data[o:o + n].view(s)
        self.g: Dict[str, Tensor] = OrderedDict()
        self.ticked = False      # Adam state already advanced by the sub-step's Philox draw (GanEngine.draw_randoms)
        if with_opt:
            self.grad = torch.zeros_like(self.data)
            self.m = torch.zeros_like(self.data)
            self.v = torch.zeros_like(self.data)
            self.state = torch.zeros(4, dtype=torch.float64, device=device)
            for k, s in spec.items():
                o, n = self.offsets[k]
                self.g[k] = self.grad[o:o + n].view(s)

    def load(self, params: Dict[str, Tensor], prefix: str = ""):
        for k in self.spec:
            src = params[prefix + k] if (prefix + k) in params else params[k]
            if tuple(src.shape) != tuple(self.spec[k]):
                raise ValueError(f"{k}: shape {tuple(src.shape)} != {self.spec[k]}")
            self.p[k].copy_(src.to(torch.float32))

    def state_dict(self) -> "OrderedDict[str, Tensor]":
        return OrderedDict((k, v.detach().cpu().clone()) for k, v in self.p.items())


class GanEngine:
    """One replica of the GAN training state on one GPU."""

    def __init__(self, cfg: dict, ed_cfg: dict, device="cuda", batch_size: Optional[int] = None, ed_dtype: str = "fp32"):
        """ed_dtype: "fp32" (the product default, and what every parity claim refers to) or "bf16": the SECONDARY
        configuration that stores the frozen emotion discriminator's activations and folded weights in bf16 and multiplies
        them on the bf16 matrix pipe with fp32 accumulation (csrc/conv_bf16.hip); everything trained stays fp32."""
        if ed_dtype not in ("fp32", "bf16"):
            raise ValueError(f"ed_dtype={ed_dtype!r}: expected 'fp32' or 'bf16'")
        self.ed_dtype = ed_dtype
        self.cfg, self.ed_cfg = dict(cfg), dict(ed_cfg)
        self.dev = torch.device(device)
        B = self.B = int(batch_size or cfg.get("BATCH_SIZE", 32))
        T = self.T = int(cfg["MAX_NOTES"])
        C = self.C = int(cfg["NOTE_DIM"])
        if T < 8:
            raise ValueError("MAX_NOTES < 8 (the reference's trim branch) is not supported")
        self.noise_dim, self.latent_dim = int(cfg["NOISE_DIM"]), int(cfg["LATENT_DIM"])
        self.mode = cfg.get("INTEGRATION_MODE", "conditioning")
        self.num_in = int(cfg.get("NUMERIC_INPUT_DIM", 6))
        self.E = int(cfg.get("ENCODER_OUT_DIM", 128))
        self.enc_hidden = tuple(cfg.get("ENCODER_HIDDEN", [256, 128]))
        if len(self.enc_hidden) != 2:
            raise ValueError("ENCODER_HIDDEN must have two entries (the reference default)")
        self.lambda_gp = float(cfg.get("LAMBDA_GP", 10.0))
        self.lambda_emo = float(cfg.get("LAMBDA_EMOTION", 1.0))
        self.lr_g, self.lr_d = float(cfg["LR_G"]), float(cfg["LR_D"])
        self.betas = (float(cfg.get("BETA1", 0.5)), float(cfg.get("BETA2", 0.9)))
        self.ed_mode = ed_cfg.get("input_mode", "notes")
        self.red = max(1, T // 8)
        self.in_dim = self.noise_dim + self.E + (self.latent_dim if self.mode == "conditioning" else 0)
        d = self.dev

        # ---- parameters ----
        gspec = generator_spec(self.noise_dim, self.latent_dim, self.mode, 512, T, C, self.E)
        espec = feature_encoder_spec(self.num_in, self.enc_hidden, self.E)
        ge = OrderedDict([("G." + k, s) for k, s in gspec.items()] + [("E." + k, s) for k, s in espec.items()])
        self.GE = FlatParams(ge, d, first=["G.decoder.pre.2.weight", "G.decoder.pre.2.bias"])
        self.D = FlatParams(discriminator_spec(C, 256, self.E), d)
        edspec, edbufs, self.ed_chans = emotion_disc_spec(self.ed_cfg)
        if self.ed_mode == "notes" and self.ed_chans and self.ed_chans[0][0] != C:
            raise ValueError(f"emotion discriminator config: note_dim={self.ed_chans[0][0]} but the GAN's NOTE_DIM={C} -- the "
                             "frozen classifier reads the generated notes, both must agree (ed_model.py:24, models.py:67)")
        self.ED = FlatParams(edspec, d, with_opt=False)
        self.EDbuf = {k: (torch.ones(s, device=d) if k.endswith("running_var") else torch.zeros(s, device=d))
                      for k, s in edbufs.items()}
        self.Gbuf = {"decoder.deconv.1.running_mean": torch.zeros(128, device=d),
                     "decoder.deconv.1.running_var": torch.ones(128, device=d),
                     "decoder.deconv.4.running_mean": torch.zeros(64, device=d),
                     "decoder.deconv.4.running_var": torch.ones(64, device=d)}
        self.num_batches_tracked = 0
        self.ed_scale = [torch.empty(co, device=d) for (_, co, _) in self.ed_chans]
        self.ed_shift = [torch.empty(co, device=d) for (_, co, _) in self.ed_chans]
        # the frozen ED's conv weights re-laid once as (Cin, Cout, K): its forward then takes the window GEMM's CNK
        # weight staging (coalesced dword loads + conflict-free 16-B LDS stores), 5-7 % faster than the NCK one
        self.ed_wt = [torch.empty(ci, co, k, device=d) for (ci, co, k) in self.ed_chans]
        # its three-tap layers by minimal filtering F(2,3) (csrc/conv_wino.hip: 2/3 of the direct form's matrix-pipe work;
        # filter transforms made once in fold_ed): forward where the layer has >= 128 output columns, data-gradient where it
        # has >= 128 input channels (= the gradient's columns; with 64 the launch is 128 workgroups and the direct kernel
        # wins: tools/wino_bench.py).  MELO_ED_WINO=0: the direct window GEMMs everywhere.
        wino = os.environ.get("MELO_ED_WINO", "1") == "1" and ed_dtype != "bf16"
        self.ed_wino_f = [bool(wino and k == 3 and co >= 128 and ops.wino3_supported(B, T, ci, co)) for (ci, co, k) in self.ed_chans]
        self.ed_wino_d = [bool(wino and k == 3 and ci >= 128 and ops.wino3_supported(B, T, co, ci)) for (ci, co, k) in self.ed_chans]
        wimg = lambda on, cin, n: torch.zeros(cin // 4, 4, n, 4, device=d) if on else None  # noqa: E731
        self.ed_wino_wf = [wimg(f, ci, co) for f, (ci, co, _) in zip(self.ed_wino_f, self.ed_chans)]     # refreshed by fold_ed
        self.ed_wino_wd = [wimg(f, co, ci) for f, (ci, co, _) in zip(self.ed_wino_d, self.ed_chans)]

        # ---- static inputs ----
        # Everything the E_num / generator forward touches has 2B rows: rows [0, B) belong to the critic step's pass
        # (no_grad, fake batch), rows [B, 2B) to the generator step's.  Both passes use the same weights (the generator is
        # updated only after the second one), so the production step runs them as ONE 2B-row pass with per-half dropout
        # masks, noise and BatchNorm statistics (dg_step_rng); the split forms run one half at a time.  Unsuffixed
        # attribute names are the generator-step half -- what the backward pass reads; `*_d` the critic-step half.
        z = lambda *s: torch.zeros(*s, device=d)  # noqa: E731
        B2 = 2 * B

        def halves(name, *shape):
            buf = z(B2, *shape)
            setattr(self, name + "_2", buf)
            setattr(self, name + "_d", buf[:B])
            setattr(self, name, buf[B:])

        halves("numeric", self.num_in)            # the batch's numeric features, staged into both halves
        self.latent = z(B, self.latent_dim)
        self.emot_idx = torch.zeros(B, dtype=torch.int64, device=d)
        halves("noise", self.noise_dim)
        self.alpha = z(B)
        self.dmask_2 = [z(B2, h) for h in self.enc_hidden]         # keep-mask * 1/(1-p)
        self.dmask_d, self.dmask = [m[:B] for m in self.dmask_2], [m[B:] for m in self.dmask_2]

        # ---- E_num / G activations ----
        h1, h2 = self.enc_hidden
        for nm, shape in (("e_x0", (self.num_in,)), ("e_xhat", (self.num_in,)), ("e_z1", (h1,)), ("e_h1", (h1,)),
                          ("e_z2", (h2,)), ("e_h2", (h2,)), ("emb", (self.E,)), ("gin", (self.in_dim,)), ("a_n0", (512,)),
                          ("lat", (self.latent_dim,)), ("a_p0", (512,)), ("a_p2", (256 * self.red,)),
                          ("y0", (self.red, 256)), ("z_d0", (2 * self.red, 128)), ("a_d0", (2 * self.red, 128)),
                          ("z_d3", (4 * self.red, 64)), ("a_d3", (4 * self.red, 64))):
            halves(nm, *shape)
        self.bn_mean_2, self.bn_invstd_2 = [z(2, 128), z(2, 64)], [z(2, 128), z(2, 64)]       # per half
        self.bn_mean, self.bn_invstd = [t[1] for t in self.bn_mean_2], [t[1] for t in self.bn_invstd_2]
        self.L3 = 8 * self.red

        # ---- critic activations for 3B rows: [x_hat | real | fake] ----
        # X0 holds 4B rolls [x_hat | real | fake (critic step) | fake (generator step)]: the critic step reads rows
        # [0, 3B) as one batch, the 2B-row generator pass writes rows [2B, 4B), the generator step reads [3B, 4B).
        c1 = lambda t: (t - 1) // 2 + 1  # noqa: E731
        self.T1, self.T2, self.T3 = c1(T), c1(c1(T)), c1(c1(c1(T)))
        Bd = 3 * B
        self.X0 = z(4 * B, T, C)
        self.A1, self.A2, self.A3 = z(Bd, self.T1, 64), z(Bd, self.T2, 128), z(Bd, self.T3, 256)
        self.H, self.Fh, self.s = z(Bd, 256), z(Bd, 256), z(Bd)
        self.dU, self.dH = z(Bd, 256), z(Bd, 256)
        self.dZ3, self.dZ2, self.dZ1 = z(Bd, self.T3, 256), z(Bd, self.T2, 128), z(Bd, self.T1, 64)
        self.gx = z(B, T, C)
        self.TAN0, self.TAN1, self.TAN2 = z(B, T, C), z(B, self.T1, 64), z(B, self.T2, 128)
        self.TZ3, self.ghb, self.gfb = z(B, self.T3, 256), z(B, 256), z(B, 256)
        self.norms, self.gp = z(B), z(1)
        self.ds_d = torch.cat([torch.ones(B), torch.full((B,), -1.0 / B), torch.full((B,), 1.0 / B)]).to(d)
        self.ds_g = torch.full((B,), -1.0 / B, device=d)
        self.loss_d_out, self.adv, self.emo = z(3), z(1), z(1)

        # ---- G-step extras ----
        self.real, self.fake_d = self.X0[B:2 * B], self.X0[2 * B:3 * B]
        self.notes = self.X0[3 * B:]              # generator output (G-step)
        self.dnotes = z(B, T, C)
        self.dn_dense = z(B, self.L3, C) if self.L3 != T else None
        self.demb = z(B, self.E)
        self.d_ad3, self.d_zd3 = z(B, 4 * self.red, 64), z(B, 4 * self.red, 64)
        self.d_ad0, self.d_zd0 = z(B, 2 * self.red, 128), z(B, 2 * self.red, 128)
        self.d_y0, self.d_p2 = z(B, self.red, 256), z(B, 256 * self.red)
        self.d_p0, self.d_lat, self.d_n0, self.d_gin = z(B, 512), z(B, self.latent_dim), z(B, 512), z(B, self.in_dim)
        self.d_ez2, self.d_ez1, self.d_ex0 = z(B, h2), z(B, h1), z(B, self.num_in)
        # emotion discriminator
        if self.ed_dtype == "bf16":
            if self.ed_mode != "notes":
                raise ValueError("ed_dtype='bf16' applies to the notes-mode emotion discriminator (its convolutions)")
            ci0 = C
            for (_, co, k) in self.ed_chans:
                if not ops.conv_s1_bf16_supported(B, T, ci0, co, k):
                    raise ValueError(f"ed_dtype='bf16': layer {ci0}->{co} k={k} at T={T} is outside the bf16 kernel's shapes "
                                     "(MAX_NOTES % 128, channels % 32 / % 64)")
                ci0 = co
            zb = lambda *shape: torch.zeros(*shape, device=d, dtype=torch.bfloat16)  # noqa: E731
            # forward image [k][co][ci] and data-gradient image [k][ci][co] (taps flipped) of every folded conv weight
            self.ed_wb_f = [zb(k, co, ci) for (ci, co, k) in self.ed_chans]
            self.ed_wb_d = [zb(k, ci, co) for (ci, co, k) in self.ed_chans]
        else:
            zb = z
        self.ed_z = [zb(B, T, co) for (_, co, _) in self.ed_chans]
        self.ed_a = [zb(B, T, co) for (_, co, _) in self.ed_chans]
        self.ed_dz = [zb(B, T, co) for (_, co, _) in self.ed_chans]
        hid = self.ed_cfg.get("notes_hidden", 256)
        mh = tuple(self.ed_cfg.get("mlp_hidden", (256, 128)))
        self.ed_feat_dim = hid if self.ed_mode == "notes" else self.ed_cfg.get("latent_dim", 128)
        if self.ed_mode != "notes" and self.ed_feat_dim != self.latent_dim:
            raise ValueError("ED latent_dim must equal the generator LATENT_DIM in 'latent' mode")
        self.ed_pool, self.ed_proj = z(B, self.ed_chans[-1][1] if self.ed_chans else 1), z(B, hid)
        self.ed_cz = [z(B, h) for h in mh]
        self.ed_ca = [z(B, h) for h in mh]
        self.ed_dcz = [z(B, h) for h in mh]
        self.n_classes = self.ed_cfg.get("n_classes", 4)
        self.logits, self.dlogits = z(B, self.n_classes), z(B, self.n_classes)
        self.ed_dproj, self.ed_dpool = z(B, hid), z(B, self.ed_pool.shape[1])
        self.ed_dfeat = z(B, self.ed_feat_dim)
        self.rng_seed = int(cfg.get("SEED", 42))
        self.rng_step = torch.zeros(1, dtype=torch.int64, device=d)
        self._graphs = {}
        # hipGraph capture is illegal on the null stream: every step runs on this side stream
        self.stream = torch.cuda.Stream(device=d)
        # side stream of the fused step's emotion branch (dg_step_rng); MELO_ED_SIDE=0: everything on one stream
        self.ed_side = torch.cuda.Stream(device=d) if os.environ.get("MELO_ED_SIDE", "1") == "1" else None
        self._tail_fork = False         # set by the forked step around g_backward_b
        self._d_wgrad_side, self._d_wgrad_ev = False, None
        self.ed_side_lds_pad = int(os.environ.get("MELO_ED_LDS_PAD", "42000"))      # g_ed_branch_side
        # its data-gradient convolutions: the same cap.  (In the forked-graph flow they mostly run after the main branch has
        # reached the join, yet lifting the cap for them measured SLOWER: 0.867 -> 0.882 ms per step; alone on the chip a
        # capped launch is only 8-10 % slower, tools/ed_pad_bench.py.)
        self.ed_side_lds_pad_bwd = int(os.environ.get("MELO_ED_LDS_PAD_BWD", str(self.ed_side_lds_pad)))
        # (Forked side streams for independent branches of a step were implemented and measured in round 1: every
        # fork/join cost more cross-queue latency than the overlap returned once the kernels filled the chip -- 1.70 ms
        # single-stream vs 1.73-1.82 -- and were removed; independent launches of one kernel share a launch instead.)
        self._init_wq()
        # x_hat rides in the launch that produces the critic step's fake batch where conv16 runs the last deconvolution
        # (for B and for 2B rows) and the generated roll fills the whole time axis (no zero-pad tail, models.py:78-81)
        self._mix_fused = (self.L3 == T and ("G.decoder.deconv.6.weight", "fwd") in self.wq and
                           all(ops.conv16_supported(nb, 4 * self.red, 64, C, True, 8 * self.red) for nb in (B, 2 * B)) and
                           os.environ.get("MELO_MIX_FUSED", "1") == "1")
        self._init_chains()
        self._bound = None          # bind_batches(): the step stages its own batch
        self._il = None             # the emotion branch's launch generator while a forked graph is captured interleaved
        # data parallelism with the collectives INSIDE the sub-steps (DataParallel, mode "ingraph"): an InGraphCollectives
        self.coll = None
        self._p2_pending = self._a_p0_gathered = False
        self.world_size = 1
        self.p2_world = 0          # > 0: decoder.pre.2's weight gradient comes from all-gathered factors (enable_p2_gather)
        self.capture_locked = False   # DataParallel.prepare(): every graph is captured before the first collective
        self._ed_folded = False

    # -------------------------------------------------------------------------------------
    # WQ-layout weight copies for conv16 (the stride-2 five-tap convolutions on 16x16 MFMA tiles)
    # -------------------------------------------------------------------------------------
    def _init_wq(self):
        """One WQ copy per (stride-2 convolution, direction it is used in) whose channel counts conv16 covers
        (reduction channels % 16, output columns % 32): written by the optimiser step itself (mg_adam_flat_wq) and by
        params_changed() after parameters were written from outside."""
        self.wq, self._wq_src = {}, []
        ent = {"D": [], "GE": []}

        def add(fp, key, name, direction, N, Cc, sn, sc):
            if Cc % 16 or N % 32:
                return
            t = torch.zeros(N * Cc * 5, device=self.dev)
            self.wq[(name, direction)] = t
            ent[key].append((fp.offsets[name][0], N, Cc, 5, sn, sc, t))
            self._wq_src.append((fp, name, t, N, Cc, sn, sc))
        for nm in ("conv.0.weight", "conv.2.weight", "conv.4.weight"):
            Cout, Cin, _ = self.D.spec[nm]
            add(self.D, "D", nm, "fwd", Cout, Cin, Cin * 5, 5)           # gather form: n = Cout, c = Cin
            add(self.D, "D", nm, "dgrad", Cin, Cout, 5, Cin * 5)         # transposed form: n = Cin, c = Cout
        for nm in ("G.decoder.deconv.0.weight", "G.decoder.deconv.3.weight", "G.decoder.deconv.6.weight"):
            Cin, Cout, _ = self.GE.spec[nm]
            add(self.GE, "GE", nm, "fwd", Cout, Cin, 5, Cout * 5)        # transposed form: n = Cout, c = Cin
            add(self.GE, "GE", nm, "dgrad", Cin, Cout, Cout * 5, 5)      # gather form: n = Cin, c = Cout
        self._wq_tab = {"D": ops.wq_table(ent["D"]) if ent["D"] else None,
                        "GE": ops.wq_table(ent["GE"]) if ent["GE"] else None}
        # the generator's update in two ranges (forked step, g_backward_b): [0, head) = decoder.pre.2 (89 % of the bytes; no
        # WQ copy inside), [head, n) = everything else with the table's offsets re-based
        o, cnt = self.GE.offsets["G.decoder.pre.2.bias"]
        self._ge_head = o + cnt if self.GE.offsets["G.decoder.pre.2.weight"][0] == 0 else 0
        if any(e[0] < self._ge_head for e in ent["GE"]):
            self._ge_head = 0
        self._wq_tab["GE_rest"] = ops.wq_table([(e[0] - self._ge_head,) + tuple(e[1:]) for e in ent["GE"]]) if ent["GE"] else None
        self._ge_head_done = False

    def _init_chains(self):
        """Which small per-sample layer stacks run as ONE row-chain launch (csrc/row_chain.hip) instead of a launch per
        layer: the numeric encoder (forward with the generator-input assembly; data-gradient), the critic's tail (fc +
        scoring head + their data-gradients) and the emotion classifier's tail (pooling, project, MLP, head, cross-entropy
        and every data-gradient).  Decided once from the shapes (a replayed graph runs no Python); MELO_CHAINS=0: the
        per-layer launches everywhere (what the chains are tested against)."""
        on = os.environ.get("MELO_CHAINS", "1") == "1"
        Ch = ops.Chain
        h1, h2 = self.enc_hidden
        PE, PD, PED = self._ep, self.D.p, self.ED.p
        self._chain_e = (on and self.num_in <= 64 and Ch.supported(self.num_in, h1, h2, self.E, self.noise_dim, max(self.latent_dim, 1))
                         and Ch.weights_ok(PE("net.1.weight"), PE("net.4.weight"), PE("net.7.weight")))
        PG = self._gp
        # ... extended over the generator's small front layers (noise_to_latent, decoder.pre.0: models.py:20-27,47-48) and,
        # backwards, their data-gradients: one launch from the numeric features to decoder.pre.2's input and back
        self._chain_gf = (self._chain_e and Ch.supported(self.in_dim, 512, max(self.latent_dim, 1)) and self.latent_dim % 4 == 0
                          and Ch.weights_ok(PG("noise_to_latent.net.0.weight"), PG("noise_to_latent.net.2.weight"),
                                            PG("decoder.pre.0.weight")) and os.environ.get("MELO_CHAIN_GF", "1") == "1")
        self._chain_d = on and Ch.supported(256, self.E) and Ch.weights_ok(PD["fc.1.weight"])
        mh = tuple(self.ed_cfg.get("mlp_hidden", (256, 128)))
        ws = [PED[f"classifier.net.{3 * j}.weight"] for j in range(len(mh))] + [PED["classifier.head.weight"]]
        dims = list(mh) + [self.ed_feat_dim]
        n_ops = 2 * (len(mh) + 1) + 3                     # load, MLP + head forward, CE, dlogits store, their data-gradients
        if self.ed_mode == "notes":
            ws.append(PED["encoder.project.weight"])
            dims.append(self.ed_pool.shape[1])
            n_ops += 2
        self._chain_ed = (on and self.n_classes <= 32 and Ch.supported(*dims) and Ch.weights_ok(*ws) and
                          n_ops <= ops.L.CHAIN_MAX_OPS)
        self.ed_loss_rows = torch.zeros(self.B, device=self.dev)      # per-sample cross-entropy terms (their mean is `emo`)
        self._gin_done = False

    def params_changed(self):
        """Call after writing parameters other than through the optimiser step (load_state and init_weights do):
        refreshes the derived copies -- the WQ-layout convolution weights and the folded emotion discriminator."""
        for fp, name, t, N, Cc, sn, sc in self._wq_src:
            ops.wq_relayout(fp.p[name], t, N, Cc, 5, sn, sc)
        # folded eagerly, never from inside a sub-step: a replayed hipGraph does not run the Python that would notice a
        # stale fold (and a fold launched during capture would be baked into every replay)
        self.fold_ed()

    class _Riders:
        """What a _conv5s2 launch did besides the convolution (the caller runs a separate kernel for what it did not)."""
        __slots__ = ("parts", "pooled", "perm", "mixed", "bnb")

        def __init__(self):
            self.parts, self.pooled, self.perm, self.mixed, self.bnb = None, False, False, False, None

    def _tick(self):
        """Interleaved capture of the forked step graph (dg_fork_step_rng): after a main-branch convolution launch, issue the
        emotion branch's next launch on the side stream."""
        il = self._il
        if il is not None:
            with torch.cuda.stream(self.ed_side):
                if next(il, "done") == "done":
                    self._il = None

    def _conv5s2(self, kind: str, x: Tensor, fp: "FlatParams", name: str, y: Tensor, stats=None, pool=None, perm=False,
                 mix=None, bnb=None, **epi):
        """One stride-2 five-tap convolution launch.  kind: conv_fwd / conv_dgrad (nn.Conv1d, weight (Cout,Cin,5)) or
        convT_fwd / convT_dgrad (nn.ConvTranspose1d, weight (Cin,Cout,5)).  conv16 (no split-K, no finish launch) wherever
        it covers the shape (faster on every cfg2 layer, tools/conv16_bench.py); the 64x64-tile kernel otherwise.
        Riders, taken by the same launch where conv16 runs it (returned _Riders says which):
          stats = rows per BatchNorm group: per-column partial statistics of what it stores (the BatchNorm that follows
                  then needs no reduction pass) -> .parts = (part, part_rows);
          pool = (B, N) tensor: the temporal mean of the output -> .pooled;
          perm: y is (B, N, Tout), the order of the Linear output the reference views as (B, C, L) -> .perm; when the
                launch cannot, NOTHING is launched and the caller takes its two-launch route;
          mix = (real, alpha, out, rows): the gradient penalty's interpolate of the first `rows` samples -> .mixed;
          bnb = (a, z, mean, invstd, act): y is the gradient reaching a train-mode BatchNorm layer: the two column sums of its
                backward -> .bnb = (part, part_rows) (ops.bn_train_bwd_parts is then ONE launch)."""
        w = fp.p[name]
        res = GanEngine._Riders()
        transposed = kind in ("conv_dgrad", "convT_fwd")
        direction = "fwd" if kind.endswith("fwd") else "dgrad"
        N = w.shape[0] if kind in ("conv_fwd", "convT_dgrad") else w.shape[1]
        wq = self.wq.get((name, direction))
        B, Tin, Cin = x.shape
        Ty = y.shape[2] if perm else y.shape[1]
        odd = transposed and Ty == 2 * Tin - 1 and kind == "conv_dgrad"
        if wq is not None and ops.conv16_supported(B, Tin, Cin, N, transposed, (2 * Tin - (1 if odd else 0)) if transposed else 0):
            kw = dict(epi)
            if pool is not None and kind == "conv_fwd" and ops.conv16_poolable(B, Tin, Cin, N):
                kw["pool"] = (pool, 1.0 / y.shape[1])
                res.pooled = True
            if stats is not None:
                tb, rows = ops.conv16_plan(B, Tin, N, transposed)
                if stats % tb == 0:                 # no tile straddles two groups
                    part = ops.workspace(12 * rows * N, x.device, "bn_part").view(torch.float32)
                    kw["stats"] = part
                    res.parts = (part, rows)
            if perm:
                kw["perm"] = res.perm = True
            if bnb is not None and y.shape[1] == (2 * Tin if transposed else (Tin + 4 - 5) // 2 + 1):
                rows = ops.conv16_plan(B, Tin, N, transposed)[1]
                bpart = ops.workspace(16 * rows * N, x.device, "bnb_part").view(torch.float64)
                kw["bnb"] = (bnb[0], bnb[1], bnb[2], bnb[3], bpart, bnb[4])
                res.bnb = (bpart, rows)
            if mix is not None:
                kw["mix"] = mix
                res.mixed = True
            ops.conv16(x, wq, y, N, transposed, odd=odd, **kw)
            self._tick()
            return res
        if perm:
            return res                              # not launched: the caller's transpose route
        if kind == "conv_fwd":
            ops.conv1d_fwd(x, w, y, 2, **epi)
        elif kind == "conv_dgrad":
            ops.conv1d_dgrad(x, w, y, 2, **epi)
        elif kind == "convT_fwd":
            ops.convT1d_fwd(x, w, y, **epi)
        else:
            ops.convT1d_dgrad(x, w, y, **epi)
        return res

    # -------------------------------------------------------------------------------------
    # state in / out
    # -------------------------------------------------------------------------------------
    def load_state(self, PE, PG, BG, PD, PED, BED):
        """Load CPU dicts keyed like the reference's state_dicts."""
        self.GE.load({**{"G." + k: v for k, v in PG.items()}, **{"E." + k: v for k, v in PE.items()}})
        self.D.load(PD)
        self.ED.load(PED)
        for k in self.Gbuf:
            self.Gbuf[k].copy_(BG[k])
        for k in self.EDbuf:
            self.EDbuf[k].copy_(BED[k])
        self.params_changed()

    def init_weights(self, seed: int = 42):
        """weights_init (src/gan/utils.py:37-45): N(0, 0.02) on every Conv*/Linear* weight of E_num, G, D,
        biases 0; BatchNorm/LayerNorm gamma=1, beta=0.  The frozen ED keeps a deterministic random fill
        (the reference tolerates a missing checkpoint, train_gan.py:127-128)."""
        g = torch.Generator().manual_seed(seed)
        for fp in (self.GE, self.D):
            for k, s in fp.spec.items():
                if k.endswith("bias"):
                    fp.p[k].zero_()
                elif len(s) == 1:
                    fp.p[k].fill_(1.0)
                else:
                    fp.p[k].copy_(torch.empty(s).normal_(0.0, 0.02, generator=g))
        for k, s in self.ED.spec.items():
            if k.endswith("bias"):
                self.ED.p[k].zero_()
            elif len(s) == 1:
                self.ED.p[k].fill_(1.0)
            else:
                fan_in = math.prod(s[1:])
                bound = 1.0 / math.sqrt(fan_in)
                self.ED.p[k].copy_((torch.rand(s, generator=g) * 2 - 1) * bound)
        self.params_changed()

    def state_dicts(self):
        """{'G','E_num','D','ED'} -> reference-format state_dicts (CPU)."""
        G, E = OrderedDict(), OrderedDict()
        for k, v in self.GE.p.items():
            (G if k.startswith("G.") else E)[k[2:]] = v.detach().cpu().clone()
        out_g = OrderedDict()
        for k, v in G.items():
            out_g[k] = v
            if k in ("decoder.deconv.1.bias", "decoder.deconv.4.bias"):     # BN buffers follow weight/bias
                base = k[:-len("bias")]
                out_g[base + "running_mean"] = self.Gbuf[base + "running_mean"].cpu().clone()
                out_g[base + "running_var"] = self.Gbuf[base + "running_var"].cpu().clone()
                out_g[base + "num_batches_tracked"] = torch.tensor(self.num_batches_tracked, dtype=torch.int64)
        ed = self.ED.state_dict()
        for k, v in self.EDbuf.items():
            ed[k] = v.cpu().clone()
        return {"G": out_g, "E_num": E, "D": self.D.state_dict(), "ED": ed}

    def set_batch(self, real: Tensor, numeric: Tensor, latent: Optional[Tensor], emot_idx: Tensor,
                  idx: Optional[Tensor] = None, real_idx="same"):
        """Stage one batch into the engine's input buffers.  Device sources go through ONE mg_stage_rows launch; with
        `idx` (int64 device tensor of B rows) the sources are whole resident arrays and the batch is gathered from
        them (real_idx=None: `real` alone is already a gathered batch).  Host sources take torch's copies."""
        srcs = [(real, self.real, idx if real_idx == "same" else real_idx), (numeric, self.numeric_d, idx),
                (numeric, self.numeric, idx)]
        if latent is not None and self.latent_dim > 0:
            srcs.append((latent, self.latent, idx))
        srcs.append((emot_idx, self.emot_idx, idx))
        if all(s.is_cuda and s.is_contiguous() and s.dtype == d.dtype and s.shape[1:] == d.shape[1:] and
               (i is not None or s.shape[0] == self.B) for s, d, i in srcs):
            ops.stage_rows(srcs, self.B)
            return
        if idx is not None:
            raise ValueError("set_batch: gathering by idx needs contiguous device sources of the engine's dtypes")
        self.real.copy_(real, non_blocking=True)
        self.numeric_d.copy_(numeric, non_blocking=True)
        self.numeric.copy_(numeric, non_blocking=True)
        if latent is not None:
            self.latent.copy_(latent, non_blocking=True)
        self.emot_idx.copy_(emot_idx, non_blocking=True)

    def bind_batches(self, real: Tensor, numeric: Tensor, latent: Optional[Tensor], emot_idx: Tensor,
                     order_len: Optional[int] = None):
        """Bind an HBM-resident split (or pool of batches): from now on every batch's first sub-step (the draw of
        d_backward_rng / dg_step_rng / ...) STAGES ITS OWN BATCH -- batch number (rng_step - batch_base) of the order written
        by set_order() (default: the rows in sequence) -- as the first launch of its graph.  No set_batch() call, i.e. no
        host-side launch and no extra graph boundary between steps (they cost ~20 us per step at cfg2).  The Philox step
        counter, advanced by the critic's Adam launch once per batch, is the batch counter."""
        srcs = [(real, self.real), (numeric, self.numeric_d), (numeric, self.numeric)]
        if latent is not None and self.latent_dim > 0:
            srcs.append((latent, self.latent))
        srcs.append((emot_idx, self.emot_idx))
        n = real.shape[0]
        for s_, d_ in srcs:
            if not (s_.is_cuda and s_.is_contiguous() and s_.dtype == d_.dtype and s_.shape[1:] == d_.shape[1:] and s_.shape[0] == n):
                raise ValueError("bind_batches: contiguous device arrays of the engine's dtypes and row shapes, equal length")
        self.batch_order_len = int(order_len or (n // self.B) * self.B)
        if self.batch_order_len < self.B:
            raise ValueError("bind_batches: fewer rows than one batch")
        self.batch_order = torch.arange(self.batch_order_len, dtype=torch.int64, device=self.dev)
        self.batch_base = self.rng_step.clone()
        self._bound = srcs
        if self._graphs and self.capture_locked:
            raise RuntimeError("bind_batches: after DataParallel.prepare() -- bind the split before the first step")
        self._graphs.clear()        # graphs captured before the binding do not stage

    def set_order(self, order: Tensor):
        """The epoch's order (int64 row indices, batch after batch; exactly batch_order_len of them): batch 0 of it is the
        next batch staged.  Stream-ordered device copies -- no synchronisation."""
        if self._bound is None or order.numel() != self.batch_order_len:
            raise ValueError("set_order: bind_batches() first; the order must have batch_order_len entries")
        self.batch_order.copy_(order.to(torch.int64), non_blocking=True)
        self.batch_base.copy_(self.rng_step)

    def unbind_batches(self):
        self._bound = None
        if not self.capture_locked:
            self._graphs.clear()

    def _stage_bound(self):
        if self._bound is not None:
            ops.stage_rows_cursor(self._bound, self.B, self.batch_order, self.batch_order_len, self.rng_step, self.batch_base)

    def set_randoms(self, noise: Tensor, drop_masks: Optional[Sequence[Tensor]], alpha: Optional[Tensor] = None,
                    half: Optional[str] = None):
        """Injected randomness of one sub-step.  drop_masks are {0,1} keep-masks (None => eval mode, no dropout).
        half: 'd' = the critic step's pass (default when alpha is given), 'g' = the generator step's."""
        half = half or ("d" if alpha is not None else "g")
        (self.noise_d if half == "d" else self.noise).copy_(noise, non_blocking=True)
        if alpha is not None:
            self.alpha.copy_(alpha.reshape(-1), non_blocking=True)
        if drop_masks is not None:
            for dst, m in zip(self.dmask_d if half == "d" else self.dmask, drop_masks):
                dst.copy_(m.to(torch.float32) * (1.0 / (1.0 - P_DROP)), non_blocking=True)

    def draw_randoms(self, with_alpha: bool):
        """Production path: one Philox launch fills noise, alpha and both dropout masks (draw order of the
        reference per sub-step: dropout masks, randn noise, rand alpha -- SURVEY section 3).  The same launch advances
        the Adam state of the optimiser this sub-step ends with (critic when alpha is drawn, generator otherwise), and
        that update advances the Philox step counter: no tick launches (see mg_rng_fill_tick)."""
        fp = self.D if with_alpha else self.GE
        if with_alpha:
            self._stage_bound()                   # a bound split: the batch's first sub-step stages it (bind_batches)
        # the two sub-steps draw from different Philox keys: under data parallelism the G-step's draw is issued before
        # the critic update has advanced the step counter (DataParallel.step)
        key = self.rng_seed if with_alpha else (self.rng_seed + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        noise, masks = (self.noise_d, self.dmask_d) if with_alpha else (self.noise, self.dmask)
        ops.rng_fill(noise, self.alpha if with_alpha else None, masks[0], masks[1], P_DROP,
                     key, self.rng_step, tick_state=fp.state, betas=self.betas)
        fp.ticked = True

    def _stamp(self, i: int):
        """MELO_STAMPS=1: a one-lane node writing the device clock into stamps[i] (tools/step_stamps.py): 0 step start, 1 fork
        (generator pass done), 2 / 3 emotion branch first / last node, 4 main branch at the join, 5 behind the join, 6 end."""
        if os.environ.get("MELO_STAMPS", "0") != "1":
            return
        if not hasattr(self, "stamps"):
            self.stamps = torch.zeros(8, dtype=torch.int64, device=self.dev)
        ops.stamp(self.stamps, i)

    def draw_randoms_both(self):
        """The fused step's draw: noise and dropout masks of BOTH halves plus alpha in one launch, which advances both
        optimisers' Adam states; the critic update advances the Philox counter."""
        self._stamp(0)
        # (Tried: the staging launch on the side stream beside the draw -- a fork and a join for two 5-10-us launches: 75.4 k
        # samples/s against 76.2 k without.)
        stage = None
        if self._bound is not None and os.environ.get("MELO_STAGE_RIDER", "1") == "1":
            # a bound split: the batch's staging rides in the draw's launch (both only read the step counter)
            stage = (self._bound, self.B, self.batch_order, self.batch_order_len, self.batch_base)
        else:
            self._stage_bound()
        ops.rng_fill(self.noise_2, self.alpha, self.dmask_2[0], self.dmask_2[1], P_DROP, self.rng_seed, self.rng_step,
                     tick_state=self.D.state, betas=self.betas, tick_state2=self.GE.state, stage=stage)
        self.D.ticked, self.GE.ticked = True, "nobump"

    def seed(self, seed: int):
        self.rng_seed = int(seed)
        self.rng_step.zero_()
        if self._bound is not None:
            self.batch_base.zero_()

    # -------------------------------------------------------------------------------------
    # forward pieces
    # -------------------------------------------------------------------------------------
    def _gp(self, k):
        return self.GE.p["G." + k]

    def _ep(self, k):
        return self.GE.p["E." + k]

    def _rows(self, which: str):
        """Row range of the 2B-row buffers: 'd' = critic-step half, 'g' = generator-step half, 'both'."""
        B = self.B
        return {"d": (0, B), "g": (B, 2 * B), "both": (0, 2 * B)}[which]

    def _e_fwd(self, train: bool, which: str = "g", gin: bool = False):
        """FeatureEncoder.forward (src/gan/feature_encoder.py:43-45) on the rows of `which`.  gin: the caller runs the
        generator next -- where the encoder is one row-chain launch it also assembles the generator's input
        [noise | embedding (| latent)] (models.py:116-122), which _g_fwd then does not."""
        P = self._ep
        r0, r1 = self._rows(which)
        v = lambda name: getattr(self, name + "_2")[r0:r1]  # noqa: E731
        if self._chain_e:
            m1, m2 = ((self.dmask_2[0][r0:r1], self.dmask_2[1][r0:r1]) if train else (None, None))
            ch = ops.Chain(r1 - r0)
            ch.load(0, v("numeric"))
            ch.layernorm(0, 1, P("net.0.weight"), P("net.0.bias"), xhat=v("e_xhat"), y=v("e_x0"))
            ch.linear_fwd(1, 2, P("net.1.weight"), P("net.1.bias"), ACT_GELU, m1, zout=v("e_z1"), out=v("e_h1"))
            ch.linear_fwd(2, 3, P("net.4.weight"), P("net.4.bias"), ACT_GELU, m2, zout=v("e_z2"), out=v("e_h2"))
            ch.linear_fwd(3, 4, P("net.7.weight"), P("net.7.bias"), out=v("emb"))
            if gin and self._chain_gf:
                # the generator's input assembled in LDS, then noise_to_latent and decoder.pre.0 in the same launch
                nd, E, G = self.noise_dim, self.E, self._gp
                ch.copy(4, 5, E, dst_at=nd)
                ch.load(5, v("noise"))
                if self.mode == "conditioning":
                    ch.load(5, self.latent, mod=self.B, at=nd + E)      # the batch's latent, the same rows for both halves
                ch.store(5, v("gin"))
                ch.linear_fwd(5, 0, G("noise_to_latent.net.0.weight"), G("noise_to_latent.net.0.bias"), ACT_RELU, out=v("a_n0"))
                ch.linear_fwd(0, 1, G("noise_to_latent.net.2.weight"), G("noise_to_latent.net.2.bias"), out=v("lat"))
                ch.linear_fwd(1, 2, G("decoder.pre.0.weight"), G("decoder.pre.0.bias"), ACT_RELU, out=v("a_p0"))
                ch.launch()
                self._gin_done = "front"
                return
            if gin:
                nd, E, g = self.noise_dim, self.E, v("gin")
                ch.store(4, g[:, nd:nd + E])
                ch.load(5, v("noise"))
                ch.store(5, g[:, :nd])
                if self.mode == "conditioning":
                    ch.load(5, self.latent, mod=self.B)          # the batch's latent, the same rows for both halves
                    ch.store(5, g[:, nd + E:nd + E + self.latent_dim])
            ch.launch()
            self._gin_done = gin
            return
        ops.layernorm_fwd(v("numeric"), v("e_x0"), v("e_xhat"), P("net.0.weight"), P("net.0.bias"))
        m1, m2 = ((self.dmask_2[0][r0:r1], self.dmask_2[1][r0:r1]) if train else (None, None))
        ops.linear_fwd(v("e_x0"), P("net.1.weight"), v("e_h1"), bias=P("net.1.bias"), zout=v("e_z1"), act=ACT_GELU, emul=m1)
        ops.linear_fwd(v("e_h1"), P("net.4.weight"), v("e_h2"), bias=P("net.4.bias"), zout=v("e_z2"), act=ACT_GELU, emul=m2)
        ops.linear_fwd(v("e_h2"), P("net.7.weight"), v("emb"), bias=P("net.7.bias"))

    def _g_fwd(self, out: Tensor, train: bool, which: str = "g"):
        """Generator.forward (src/gan/models.py:108-130, :66-83) on the rows of `which` into `out` (rows, T, C)."""
        P = self._gp
        r0, r1 = self._rows(which)
        n = r1 - r0
        v = lambda name: getattr(self, name + "_2")[r0:r1]  # noqa: E731
        # gin = [noise | embedding (| latent)]: the column blocks filled by one launch
        nd, E = self.noise_dim, self.E
        gin = v("gin")
        blocks = [(v("noise"), gin[:, :nd], None), (v("emb"), gin[:, nd:nd + E], None)]
        if self.mode == "conditioning":
            for h in range(n // self.B):          # the batch's latent, once per half
                blocks.append((self.latent, gin[h * self.B:(h + 1) * self.B, nd + E:nd + E + self.latent_dim], None))
        front, self._gin_done = self._gin_done, False
        if front:
            pass                                    # assembled by the encoder's chain launch (_e_fwd(gin=True))
        elif n == self.B:
            ops.stage_rows(blocks, n)
        else:       # one launch: the two-half blocks cover n rows, the per-half latent blocks B rows each
            ops.stage_rows(blocks[:2] + [(s_, d_, i_, self.B) for (s_, d_, i_) in blocks[2:]], n)
        if front != "front":                        # else noise_to_latent and pre.0 ran in that launch too
            ops.linear_fwd(gin, P("noise_to_latent.net.0.weight"), v("a_n0"), bias=P("noise_to_latent.net.0.bias"), act=ACT_RELU)
            ops.linear_fwd(v("a_n0"), P("noise_to_latent.net.2.weight"), v("lat"), bias=P("noise_to_latent.net.2.bias"))
            ops.linear_fwd(v("lat"), P("decoder.pre.0.weight"), v("a_p0"), bias=P("decoder.pre.0.bias"), act=ACT_RELU)
        if n <= ops.SKINNY_MAX_ROWS and self.red > 1:
            # pre.2's output straight into the channels-last (rows, L, 256) tensor the first deconvolution reads
            # (models.py:70-73: view(B, 256, L) + permute): the Linear launch walks its weight rows in that order
            ops.linear_fwd(v("a_p0"), P("decoder.pre.2.weight"), v("y0"), perm_L=self.red, bias=P("decoder.pre.2.bias"),
                           act=ACT_RELU)
        else:
            ops.linear_fwd(v("a_p0"), P("decoder.pre.2.weight"), v("a_p2"), bias=P("decoder.pre.2.bias"), act=ACT_RELU)
            ops.transpose_bcl_blc(v("a_p2").view(n, 256, self.red), v("y0"))
        st = self.B if train else None
        pr = self._conv5s2("convT_fwd", v("y0"), self.GE, "G.decoder.deconv.0.weight", v("z_d0"), stats=st, bias=P("decoder.deconv.0.bias"))
        self._bn(v("z_d0"), v("a_d0"), "decoder.deconv.1", 0, train, which, pr.parts)
        pr = self._conv5s2("convT_fwd", v("a_d0"), self.GE, "G.decoder.deconv.3.weight", v("z_d3"), stats=st, bias=P("decoder.deconv.3.bias"))
        self._bn(v("z_d3"), v("a_d3"), "decoder.deconv.4", 1, train, which, pr.parts)
        # the critic-step half's fake batch also leaves x_hat = alpha * real + (1 - alpha) * fake (utils.py:76-79) in X0[:B]
        mix = None
        if train and which in ("d", "both") and self._mix_fused and out.data_ptr() == self.fake_d.data_ptr():
            mix = (self.real, self.alpha, self.X0[:self.B], self.B)
        self._conv5s2("convT_fwd", v("a_d3"), self.GE, "G.decoder.deconv.6.weight", out, mix=mix, bias=P("decoder.deconv.6.bias"))
        if train:
            self.num_batches_tracked += n // self.B

    def _bn(self, z, a, name, i, train, which="g", parts=None):
        """parts: (partial statistics, rows) left by the producing conv16 launch: no reduction pass over z."""
        P = self._gp
        if train:
            groups = 2 if which == "both" else 1
            g0 = 0 if which in ("d", "both") else 1
            mean = self.bn_mean_2[i] if groups == 2 else self.bn_mean_2[i][g0]
            invstd = self.bn_invstd_2[i] if groups == 2 else self.bn_invstd_2[i][g0]
            if parts is not None:
                ops.bn_train_fwd_parts(parts[0], parts[1], groups, z, a, P(name + ".weight"), P(name + ".bias"),
                                       self.Gbuf[name + ".running_mean"], self.Gbuf[name + ".running_var"], mean, invstd,
                                       ACT_RELU, BN_MOM, BN_EPS)
                return
            ops.bn_train_fwd(z, a, P(name + ".weight"), P(name + ".bias"), self.Gbuf[name + ".running_mean"],
                             self.Gbuf[name + ".running_var"], mean, invstd, ACT_RELU, BN_MOM, BN_EPS, groups=groups)
        else:
            ops.bn_eval_fwd(z, a, P(name + ".weight"), P(name + ".bias"), self.Gbuf[name + ".running_mean"],
                            self.Gbuf[name + ".running_var"], ACT_RELU, BN_EPS)

    def _d_fwd(self, x: Tensor, nb: int, emb: Tensor, head: bool = True):
        """Discriminator.forward (src/gan/models.py:158-169) on the first nb rows of the critic buffers; `emb` (B rows)
        is the numeric embedding of row r % B.  head=False leaves the scoring head to _d_bwd_input(with_head=True),
        which runs it in the same launch as its gradient."""
        P = self.D.p
        self._conv5s2("conv_fwd", x, self.D, "conv.0.weight", self.A1[:nb], bias=P["conv.0.bias"], act=ACT_LRELU)
        self._conv5s2("conv_fwd", self.A1[:nb], self.D, "conv.2.weight", self.A2[:nb], bias=P["conv.2.bias"], act=ACT_LRELU)
        # AdaptiveAvgPool1d(1) rides in conv.4's launch where a sample's time axis is one wave tile (cfg2: 32 positions)
        if not self._conv5s2("conv_fwd", self.A2[:nb], self.D, "conv.4.weight", self.A3[:nb], pool=self.H[:nb],
                             bias=P["conv.4.bias"], act=ACT_LRELU).pooled:
            ops.meanT_fwd(self.A3[:nb], self.H[:nb])
        if self._chain_d and not head:
            return                  # fc + head + their data-gradients are ONE chain launch in _d_bwd_input(with_head=True)
        ops.linear_fwd(self.H[:nb], P["fc.1.weight"], self.Fh[:nb], bias=P["fc.1.bias"], act=ACT_LRELU)
        if head:
            ops.dhead_fwd(self.Fh[:nb], emb, P["real_fake.weight"].view(-1), P["real_fake.bias"], self.s[:nb])

    def _d_bwd_input(self, ds: Tensor, nb: int, emb: Tensor, demb: Optional[Tensor], with_head: bool = False, mean=None):
        """Back-propagate ds through the critic down to dZ1 (grad wrt conv.0's pre-activation).  mean = (src, out, scale): a
        scalar mean that rides in the pooling-backward launch (the generator's adversarial loss)."""
        P = self.D.p
        if with_head and self._chain_d and (demb is None or emb.shape[0] == nb):
            # the critic's tail as one launch: fc + LeakyReLU, the scoring head, its input gradient (ds is a constant of
            # the step) and fc's data-gradient (models.py:149-169)
            ch = ops.Chain(nb)
            ch.load(0, self.H[:nb])
            ch.linear_fwd(0, 1, P["fc.1.weight"], P["fc.1.bias"], ACT_LRELU, out=self.Fh[:nb])
            ch.dhead(1, 2, P["real_fake.weight"].view(-1), P["real_fake.bias"], emb, ds, self.s[:nb], demb=demb)
            ch.store(2, self.dU[:nb])
            ch.linear_dgrad(2, 3, P["fc.1.weight"], out=self.dH[:nb])
            ch.launch()
        else:
            if with_head and self._chain_d:         # _d_fwd left fc to the chain
                ops.linear_fwd(self.H[:nb], P["fc.1.weight"], self.Fh[:nb], bias=P["fc.1.bias"], act=ACT_LRELU)
            if with_head:
                ops.dhead_fwd_bwd(ds, self.Fh[:nb], emb, P["real_fake.weight"].view(-1), P["real_fake.bias"], self.s[:nb],
                                  self.dU[:nb], demb, nb_emb=nb if demb is not None else 0)
            else:
                ops.dhead_bwd(ds, self.Fh[:nb], P["real_fake.weight"].view(-1), self.dU[:nb], demb,
                              nb_emb=nb if demb is not None else 0)
            ops.linear_dgrad(self.dU[:nb], P["fc.1.weight"], self.dH[:nb])
        ops.meanT_bwd(self.dH[:nb], self.dZ3[:nb], gref=self.A3[:nb], gact=ACT_LRELU, mean=mean)
        self._conv5s2("conv_dgrad", self.dZ3[:nb], self.D, "conv.4.weight", self.dZ2[:nb], gref=self.A2[:nb], gact=ACT_LRELU)
        self._conv5s2("conv_dgrad", self.dZ2[:nb], self.D, "conv.2.weight", self.dZ1[:nb], gref=self.A1[:nb], gact=ACT_LRELU)

    def _require_fold(self):
        if not self._ed_folded:
            raise RuntimeError("GanEngine: the frozen emotion discriminator's parameters were written without "
                               "params_changed() -- its folded scale/shift and re-laid weights are stale")

    def fold_ed(self):
        """Eval-mode BatchNorm of the frozen ED folded with its conv bias into scale/shift; called by params_changed()
        (load_state, init_weights, load_ed_checkpoint, broadcast), outside any graph."""
        for i in range(len(self.ed_chans)):
            pre = f"encoder.conv.{i}.net"
            ops.bn_fold(self.ED.p[pre + ".1.weight"], self.ED.p[pre + ".1.bias"], self.EDbuf[pre + ".1.running_mean"],
                        self.EDbuf[pre + ".1.running_var"], self.ED.p[pre + ".0.bias"], self.ed_scale[i], self.ed_shift[i], BN_EPS)
            self.ed_wt[i].copy_(self.ED.p[pre + ".0.weight"].permute(1, 0, 2))
            ci, co, k = self.ed_chans[i]
            if self.ed_wino_f[i]:
                ops.wino3_weights(self.ED.p[pre + ".0.weight"], co, ci, ci * 3, 3, out=self.ed_wino_wf[i])
            if self.ed_wino_d[i]:
                ops.wino3_weights(self.ED.p[pre + ".0.weight"], ci, co, 3, ci * 3, flip=True, out=self.ed_wino_wd[i])
            if self.ed_dtype == "bf16":
                w = self.ED.p[pre + ".0.weight"]                       # (co, ci, k)
                ops.wb_relayout(w, self.ed_wb_f[i], co, ci, k, ci * k, k)
                ops.wb_relayout(w, self.ed_wb_d[i], ci, co, k, k, ci * k, flip=True)
        self._ed_folded = True

    def _ed_conv_fwd(self, i: int, x: Tensor):
        """Layer i of the frozen encoder, fp32: Conv1d -> folded BatchNorm (z kept) -> GELU (ed_model.py:24-46)."""
        ci, co, k = self.ed_chans[i]
        if self.ed_wino_f[i]:
            ops.conv_wino3(x, self.ed_wino_wf[i], self.ed_a[i], scale=self.ed_scale[i], shift=self.ed_shift[i],
                           zout=self.ed_z[i], act=ACT_GELU)
        else:
            ops.conv_gather(x, self.ed_wt[i], self.ed_a[i], co, k, 1, k, co * k, scale=self.ed_scale[i],
                            shift=self.ed_shift[i], zout=self.ed_z[i], act=ACT_GELU)

    def _ed_conv_dgrad(self, i: int, out: Tensor):
        """Layer i's input gradient from ed_dz[i], fp32; for i > 0 times GELU'(z) and the BatchNorm scale of layer i-1."""
        epi = dict(gref=self.ed_z[i - 1], gact=ACT_GELU, gscale=self.ed_scale[i - 1]) if i > 0 else {}
        if self.ed_wino_d[i]:
            ops.conv_wino3(self.ed_dz[i], self.ed_wino_wd[i], out, **epi)
        else:
            ops.conv1d_dgrad(self.ed_dz[i], self.ED.p[f"encoder.conv.{i}.net.0.weight"], out, 1, **epi)

    def _ed_fwd(self, notes: Tensor):
        """EmotionDiscriminator.forward in eval mode (ed_model.py:63-69,92-95,147-165)."""
        P = self.ED.p
        if self.ed_mode == "notes" and self.ed_dtype == "bf16":
            x = notes                                                  # fp32, converted on its way into LDS
            for i in range(len(self.ed_chans)):
                ops.conv_s1_bf16(x, self.ed_wb_f[i], self.ed_a[i], scale=self.ed_scale[i], shift=self.ed_shift[i],
                                 zout=self.ed_z[i], act=ACT_GELU)
                x = self.ed_a[i]
            ops.meanT_fwd_bf16(x, self.ed_pool)
            ops.linear_fwd(self.ed_pool, P["encoder.project.weight"], self.ed_proj, bias=P["encoder.project.bias"])
            feat = self.ed_proj
        elif self.ed_mode == "notes":
            x = notes
            for i in range(len(self.ed_chans)):
                self._ed_conv_fwd(i, x)
                x = self.ed_a[i]
            ops.meanT_fwd(x, self.ed_pool)
            ops.linear_fwd(self.ed_pool, P["encoder.project.weight"], self.ed_proj, bias=P["encoder.project.bias"])
            feat = self.ed_proj
        else:
            feat = self.lat
        for j in range(len(self.ed_cz)):
            ops.linear_fwd(feat, P[f"classifier.net.{3 * j}.weight"], self.ed_ca[j], bias=P[f"classifier.net.{3 * j}.bias"],
                           zout=self.ed_cz[j], act=ACT_GELU)
            feat = self.ed_ca[j]
        ops.linear_fwd(feat, P["classifier.head.weight"], self.logits, bias=P["classifier.head.bias"])

    def _ed_bwd(self, dnotes: Tensor):
        """Input gradient of the frozen ED: dlogits -> dnotes (notes mode) or -> ed_dfeat (latent mode)."""
        P = self.ED.p
        n = len(self.ed_cz)
        g = self.dlogits
        w = P["classifier.head.weight"]
        for j in reversed(range(n)):
            ops.linear_dgrad(g, w, self.ed_dcz[j], gref=self.ed_cz[j], gact=ACT_GELU)
            g, w = self.ed_dcz[j], P[f"classifier.net.{3 * j}.weight"]
        if self.ed_mode != "notes":
            ops.linear_dgrad(g, w, self.ed_dfeat)
            return
        ops.linear_dgrad(g, w, self.ed_dproj)
        ops.linear_dgrad(self.ed_dproj, P["encoder.project.weight"], self.ed_dpool)
        self._ed_bwd_convs(dnotes)

    def _ed_bwd_convs(self, dnotes: Tensor, mean=None):
        """From the pooled features' gradient back to the notes: pooling backward (times conv3's GELU' and BatchNorm scale)
        and the four convolutions' data-gradients.  mean: a scalar mean riding in the pooling-backward launch."""
        P = self.ED.p
        last = len(self.ed_chans) - 1
        if self.ed_dtype == "bf16":
            ops.meanT_bwd_bf16(self.ed_dpool, self.ed_dz[last], self.ed_z[last], ACT_GELU, self.ed_scale[last])
            for i in range(last, 0, -1):
                ops.conv_s1_bf16(self.ed_dz[i], self.ed_wb_d[i], self.ed_dz[i - 1], gref=self.ed_z[i - 1], gact=ACT_GELU,
                                 gscale=self.ed_scale[i - 1])
            ops.conv_s1_bf16(self.ed_dz[0], self.ed_wb_d[0], dnotes)   # fp32 out: the generator's gradient stays fp32
            return
        ops.meanT_bwd(self.ed_dpool, self.ed_dz[last], gref=self.ed_z[last], gact=ACT_GELU, gscale=self.ed_scale[last], mean=mean)
        for i in range(last, 0, -1):
            self._ed_conv_dgrad(i, self.ed_dz[i - 1])
        self._ed_conv_dgrad(0, dnotes)

    # -------------------------------------------------------------------------------------
    # D-step  (src/gan/train_gan.py:183-205)
    # -------------------------------------------------------------------------------------
    def d_backward(self, forward: bool = True):
        """forward=False: the fake batch (rows [2B, 3B) of X0) and the embedding of the critic-step half already exist
        (dg_forward ran the 2B-row generator pass)."""
        B = self.B
        P, G = self.D.p, self.D.g
        if forward:
            # no_grad: embedding (dropout ON) and fake batch (BN train mode: running stats move)
            self._e_fwd(True, "d", gin=True)
            self._g_fwd(self.fake_d, True, "d")
        if not self._mix_fused:                      # else deconv.6's launch of the critic-step half wrote x_hat (_g_fwd)
            ops.gp_interp(self.real, self.fake_d, self.alpha, self.X0[:B])
        self._d_fwd(self.X0[:3 * B], 3 * B, self.emb_d, head=False)
        # one backward for [x_hat | real | fake] with ds = [1 | -1/B | +1/B]
        self._d_bwd_input(self.ds_d, 3 * B, self.emb_d, None, with_head=True)
        self._conv5s2("conv_dgrad", self.dZ1[:B], self.D, "conv.0.weight", self.gx)
        ops.gp_penalty(self.gx, self.TAN0, self.norms, None, self.lambda_gp)      # mean((norm-1)^2): in wgan_d_loss
        # Weight gradients are [real,fake] activations x Wasserstein dZ + tangent activations x penalty dZ; each
        # becomes launchable as soon as its tangent exists (d(lambda*gp)/d(grad_xhat) pushed forward through the
        # masked linear critic).
        self._conv5s2("conv_fwd", self.TAN0, self.D, "conv.0.weight", self.TAN1, gref=self.A1[:B], gact=ACT_LRELU)
        self._conv5s2("conv_fwd", self.TAN1, self.D, "conv.2.weight", self.TAN2, gref=self.A2[:B], gact=ACT_LRELU)
        tz3_pooled = self._conv5s2("conv_fwd", self.TAN2, self.D, "conv.4.weight", self.TZ3, pool=self.ghb, gref=self.A3[:B],
                                   gact=ACT_LRELU).pooled
        # the three convolutions' weight gradients go out as ONE launch (+ one slab reduction) once their tangents
        # exist (ops.wgrad_multi): three launches of ~256 workgroups each plus three reductions before
        # (Tried: this launch on a THIRD stream beside the small dependent launches below.  With three streams in the forked
        # graph the emotion branch's first node ran 236 us after the fork instead of 10 -- mg_stamp -- and the step went from
        # 0.835 to 0.88 ms: a forked hipGraph keeps two branches concurrent, not three.)
        wjobs = [
            ops.conv1d_wgrad(self.X0[B:3 * B], self.dZ1[B:], G["conv.0.weight"], 2, self.TAN0, self.dZ1[:B],
                             db=G["conv.0.bias"], defer=True),
            ops.conv1d_wgrad(self.A1[B:], self.dZ2[B:], G["conv.2.weight"], 2, self.TAN1, self.dZ2[:B],
                             db=G["conv.2.bias"], defer=True),
            ops.conv1d_wgrad(self.A2[B:], self.dZ3[B:], G["conv.4.weight"], 2, self.TAN2, self.dZ3[:B],
                             db=G["conv.4.bias"], defer=True)]
        if self._d_wgrad_side:
            # forked step graph: this launch (+ its slab reduction), 70 us of the critic step, goes INTO the emotion branch's
            # stream -- that branch has the slack since its three-tap layers run on minimal filtering -- beside the three
            # small dependent launches below; d_update waits for it
            cur = torch.cuda.current_stream()
            self.ed_side.wait_stream(cur)
            with torch.cuda.stream(self.ed_side):
                ops.wgrad_multi(wjobs)
                self._d_wgrad_ev = torch.cuda.Event()
                self._d_wgrad_ev.record(self.ed_side)
        else:
            ops.wgrad_multi(wjobs)
        if not tz3_pooled:
            ops.meanT_fwd(self.TZ3, self.ghb)
        ops.linear_fwd(self.ghb, P["fc.1.weight"], self.gfb, gref=self.Fh[:B], gact=ACT_LRELU)
        ops.linear_wgrad(self.H[B:], self.dU[B:], G["fc.1.weight"], self.ghb, self.dU[:B], db=G["fc.1.bias"])
        # the critic's loss scalars (logging only) ride in the head's weight-gradient launch
        ops.dhead_wgrad(self.ds_d[B:], self.Fh[B:], self.emb_d, self.gfb, G["real_fake.weight"].view(-1), G["real_fake.bias"],
                        2 * B, B, loss=(self.s[B:], self.norms, self.lambda_gp, self.loss_d_out, self.gp, B))

    def d_backward_rng(self):
        """Production D-step front half: device RNG draw + d_backward as one capturable sequence."""
        self.draw_randoms(with_alpha=True)
        self.d_backward()

    def g_backward_rng(self):
        self.draw_randoms(with_alpha=False)
        self.g_backward()

    # whole sub-steps as ONE graph each (single-GPU production path: a graph boundary costs ~7 us more than a kernel
    # boundary inside a graph); the split forms above exist for data parallelism (collectives between them) and tests
    def d_step_rng(self):
        self.d_backward_rng()
        self.d_update()

    def g_step_rng(self):
        self.g_backward_rng()
        self.g_update()

    # ---- the fused step: one critic update + one generator update with ONE 2B-row generator pass ----
    def dg_forward(self):
        """E_num + generator forward of the critic step AND of the generator step as one pass over 2B rows: both use the
        same weights (train_gan.py:186-189 and :216-219 -- the generator is not updated in between), each half has its own
        noise, dropout masks and BatchNorm batch statistics, and the running statistics move twice, critic-step half first."""
        self._require_fold()
        self._p2_pending, self._a_p0_gathered = bool(self.coll is not None and self.p2_world), False
        self._e_fwd(True, "both", gin=True)
        self._g_fwd(self.X0[2 * self.B:], True, "both")
        self._stamp(1)

    def dg_step_rng(self):
        """Draw + critic step + generator step as ONE capturable sequence (single-GPU production path whenever a
        generator update follows the critic update on the same batch)."""
        self.draw_randoms_both()
        self.dg_forward()
        self.d_backward(forward=False)
        self.d_update()
        self.g_backward_a2()
        self.g_backward_b()
        self.g_update()

    def dg_fork_step_rng(self, draw: bool = True):
        """dg_step_rng with the frozen emotion discriminator's branch as a PARALLEL BRANCH of the same graph, on the side
        stream (MELO_ED_FLOW=ingraph; the default of the bf16-stored branch, which is too short for the split flow).  The
        branch needs only the generated batch and is needed only where the generator's backward starts; the main branch
        is captured FIRST: hipGraphLaunch feeds a graph's branches to their hardware queues in capture order and the
        branch fed second starts 100-300 us after the fork -- that must not be the critical path (emotion branch
        captured first: 0.947 ms/step; critic step first: 0.911; one stream: 0.958).  Launching the forked graph costs the
        host 0.87 ms per step against 0.64 for the plain one -- within 5 % of the fp32 step's GPU time, which is why the fp32
        engine takes the split flow (0.58 ms of host time for 0.926) instead."""
        if self.ed_side is None:
            return self.dg_step_rng()
        cur = torch.cuda.current_stream()
        if draw:                                      # draw=False: the randoms were injected (parity tests)
            self.draw_randoms_both()
        self.dg_forward()
        self.ed_side.wait_stream(cur)                 # fork
        self._fork_branches()
        cur.wait_stream(self.ed_side)                 # join: from here on dnotes needs the emotion branch's part
        self.g_critic_back()
        self._tail_fork = os.environ.get("MELO_TAIL_FORK", "1") == "1"
        try:
            self.g_backward_b()
        finally:
            self._tail_fork = False
        self.g_update()

    def _fork_branches(self):
        """The two parallel branches between fork and join: the critic step + the critic pass on the generated batch (this
        stream) beside the frozen emotion discriminator's branch (side stream).  Measured with mg_stamp inside the replayed
        graph (tools/step_stamps.py; the profiler's timeline of a forked graph is NOT the truth -- DESIGN section 6): the
        branch starts ~10 us after the fork; a forked hipGraph keeps TWO branches concurrent (a third stream delayed the
        branch's first node by 236 us); capture order does not matter.  Since the branch's three-tap layers run on minimal
        filtering it has ~70 us of slack against this stream, so the critic's convolution weight gradients (one launch + slab
        reduction, ~70 us) are enqueued INTO the branch's stream behind its classifier-tail chain (order "dwgrad_side", the
        default: 77.2 k samples/s against 76.3 k for "main_first", three alternations on one box; behind the pooling backward:
        the same; one launch later: no gain).  Other orders kept for experiments: main_first, ed_first, interleave."""
        order = os.environ.get("MELO_FORK_ORDER", "dwgrad_side")
        if order == "ed_first" or (order == "interleave" and not self._chain_ed):
            with torch.cuda.stream(self.ed_side):
                self.g_ed_branch_side()
            order = "done"
        if order == "interleave":
            self._require_fold()
            self._il = self._ed_steps(self.ed_side_lds_pad, self.ed_side_lds_pad_bwd)
        if order == "dwgrad_side" and self._chain_ed and self.ed_mode == "notes" and self.ed_dtype == "fp32":
            self._require_fold()
            il = self._ed_steps(self.ed_side_lds_pad, self.ed_side_lds_pad_bwd)
            n_first = len(self.ed_chans) + int(os.environ.get("MELO_DWGRAD_AFTER", "1"))   # forward convs + the tail chain
            with torch.cuda.stream(self.ed_side):
                self._stamp(2)
                for _ in range(n_first):
                    next(il)
            self._d_wgrad_side = True
            try:
                self.d_backward(forward=False)
            finally:
                self._d_wgrad_side = False
            with torch.cuda.stream(self.ed_side):
                for _ in il:
                    pass
                self._stamp(3)
            self.d_update()
            self.g_critic_front()
            return
        self.d_backward(forward=False)
        self.d_update()
        self.g_critic_front()
        if order == "interleave":
            while self._il is not None:          # whatever the main branch's ticks did not reach
                self._tick()
        elif order != "done":
            with torch.cuda.stream(self.ed_side):
                self.g_ed_branch_side()

    def dg_fork_rest(self):
        """Everything behind the 2B-row generator pass as ONE graph whose ROOT is the fork: the emotion branch on the side
        stream beside the critic step + critic pass, the join, the generator's backward and update (flow "fork2": two graphs
        per batch, dg_forward_rng + this)."""
        cur = torch.cuda.current_stream()
        self.ed_side.wait_stream(cur)
        self._fork_branches()
        cur.wait_stream(self.ed_side)
        self.g_critic_back()
        self.g_backward_b()
        self.g_update()

    # ---- the split flow (DataParallel.step on one GPU, the default MELO_ED_FLOW=split): the emotion branch as its own graph ----
    def dg_forward_rng(self):
        self.draw_randoms_both()
        self.dg_forward()

    def d_step_g_critic_front(self):
        self.d_backward(forward=False)
        self.d_update()
        self.g_critic_front()

    def g_finish(self):
        self.g_critic_back()
        self.g_backward_b()
        self.g_update()

    # the same pieces for the data-parallel step order (collectives in between: DataParallel._step)
    def d_backward_nofwd(self):
        self.d_backward(forward=False)

    def d_update_g_critic_front(self):
        self.d_update()
        self.g_critic_front()

    def dg_forward_d_backward_rng(self):
        """Data parallelism: everything of the fused step in front of the critic's gradient all-reduce."""
        self.draw_randoms_both()
        self.dg_forward()
        self.d_backward(forward=False)

    def g_backward_a_rng(self):
        self.draw_randoms(with_alpha=False)
        self.g_backward_a()

    def big_grad_slice(self):
        """(offset, numel) of decoder.pre.2.weight's gradient inside the flat G+E_num gradient buffer."""
        return self.GE.offsets["G.decoder.pre.2.weight"]

    def p2_grad_slice(self):
        """(offset, numel) of decoder.pre.2's weight AND bias gradients (adjacent, at the front of the flat buffer):
        what the factor-gather mode computes for the global batch and therefore keeps out of the all-reduce."""
        (ow, nw), (ob, nb) = self.GE.offsets["G.decoder.pre.2.weight"], self.GE.offsets["G.decoder.pre.2.bias"]
        assert ow == 0 and ob == nw
        return 0, nw + nb

    def _adam(self, fp, lr):
        """The optimiser step; after draw_randoms() the Adam state is already advanced (fp.ticked)."""
        lo, wq = 0, self._wq_tab["D" if fp is self.D else "GE"]
        if fp is self.GE and self._ge_head_done:          # [0, head) went out with g_backward_b's side branch
            lo, wq, self._ge_head_done = self._ge_head, self._wq_tab["GE_rest"], False
        ops.adam_flat(fp.data[lo:], fp.grad[lo:], fp.m[lo:], fp.v[lo:], fp.state, lr, *self.betas, grad_scale=1.0 / self.world_size,
                      ticked_rng_step=self.rng_step if fp.ticked is True else None, ticked=bool(fp.ticked), wq=wq)
        fp.ticked = False

    def _adam_ge_head(self):
        """decoder.pre.2's share of the generator update, on its own: an elementwise update, so the split changes no bit.
        Only with the state advanced by the step's draw (fp.ticked): this launch must not advance it."""
        h = self._ge_head
        fp = self.GE
        ops.adam_flat(fp.data[:h], fp.grad[:h], fp.m[:h], fp.v[:h], fp.state, self.lr_g, *self.betas,
                      grad_scale=1.0 / self.world_size, ticked=True)
        self._ge_head_done = True

    def d_update(self):
        if self._d_wgrad_ev is not None:
            torch.cuda.current_stream().wait_event(self._d_wgrad_ev)
            self._d_wgrad_ev = None
        if self.coll is not None:           # C1 (gan/dp.py): the critic's gradient; a pending generator step's a_p0 rides along
            self.coll.reduce_d(self, self._p2_pending and not self._a_p0_gathered)
            self._a_p0_gathered = self._p2_pending
        self._adam(self.D, self.lr_d)

    # -------------------------------------------------------------------------------------
    # G-step  (src/gan/train_gan.py:211-251)
    # -------------------------------------------------------------------------------------
    def g_backward(self):
        self.g_backward_a()
        self.g_backward_b()

    def g_forward(self):
        """E_num + generator forward of the G-step: independent of the critic, so under data parallelism it runs while
        the critic's gradient all-reduce is in flight (DataParallel.step)."""
        self._require_fold()
        self._p2_pending, self._a_p0_gathered = bool(self.coll is not None and self.p2_world), False
        self._e_fwd(True, "g", gin=True)
        self._g_fwd(self.notes, True, "g")

    def g_forward_rng(self):
        self.draw_randoms(with_alpha=False)
        self.g_forward()

    def g_backward_a(self):
        """G-step up to and including decoder.pre.2's weight gradient (89 % of the G+E_num gradient bytes): the
        data-parallel wrapper starts that slice's all-reduce here and overlaps it with g_backward_b."""
        self.g_forward()
        self.g_backward_a2()

    def g_backward_a2(self):
        """g_backward_a without the generator forward (see g_forward)."""
        self.g_ed_branch()
        self.g_critic_chain()

    def g_ed_branch_side(self):
        """g_ed_branch for the side stream (the split / in-graph fork flows): its convolutions keep ONE workgroup per CU
        resident instead of three, so the critical path's kernels on the main stream -- many of them small and dependent --
        find free registers and wave slots at once instead of waiting for 15-50-us workgroups to retire.  The branch has the
        slack: it is needed only where the generator's backward starts.  Measured (cfg2, same box, alternating):
        0.913 -> 0.895 ms per step."""
        self._require_fold()
        self._stamp(2)
        if self._chain_ed:
            for _ in self._ed_steps(self.ed_side_lds_pad, self.ed_side_lds_pad_bwd):
                pass
        else:
            with ops.conv_lds_pad(self.ed_side_lds_pad):
                self.g_ed_branch()
        self._stamp(3)

    def g_ed_branch(self):
        """The frozen emotion discriminator's forward, cross-entropy and input gradient on the generated batch: the only
        part of the generator step that does not touch the critic -- under data parallelism it runs while the critic's
        gradient all-reduce is in flight (DataParallel.step)."""
        self._require_fold()
        if self._chain_ed:
            return self._ed_branch_chain()
        self._ed_fwd(self.notes)
        ops.softmax_ce(self.logits, self.emot_idx, self.emo, self.dlogits, self.lambda_emo)
        self._ed_bwd(self.dnotes if self.ed_mode == "notes" else None)

    def _ed_branch_chain(self):
        for _ in self._ed_steps(0):
            pass

    def _ed_steps(self, lds_pad: int, lds_pad_bwd: Optional[int] = None):
        """g_ed_branch with the classifier's tail -- pooling, project, MLP, head, cross-entropy and every data-gradient back
        to the pooled features (ed_model.py:61,86-95,147-165) -- as ONE row-chain launch between the convolutions' forward
        and their data-gradients (was: 10 launches of ~5 us).  The loss scalar (the mean of the per-sample terms) rides in
        the pooling-backward launch.  A GENERATOR that yields after every launch, so that the forked step graph can capture
        the branch's launches INTERLEAVED with the main branch's (dg_fork_step_rng); lds_pad: the convolutions' occupancy
        cap when the branch runs beside the critical path (g_ed_branch_side)."""
        P = self.ED.p
        notes, bf16 = self.ed_mode == "notes", self.ed_dtype == "bf16"
        ch = ops.Chain(self.B)
        if notes:
            x = self.notes
            for i in range(len(self.ed_chans)):
                ci, co, k = self.ed_chans[i]
                if bf16:
                    ops.conv_s1_bf16(x, self.ed_wb_f[i], self.ed_a[i], scale=self.ed_scale[i], shift=self.ed_shift[i],
                                     zout=self.ed_z[i], act=ACT_GELU)
                else:
                    with ops.conv_lds_pad(lds_pad):
                        self._ed_conv_fwd(i, x)
                yield
                x = self.ed_a[i]
            if bf16:
                ops.meanT_fwd_bf16(x, self.ed_pool)
                yield
                ch.load(0, self.ed_pool)
            else:
                ch.mean_t(0, x, out=self.ed_pool)
            ch.linear_fwd(0, 1, P["encoder.project.weight"], P["encoder.project.bias"], out=self.ed_proj)
        else:
            ch.load(1, self.lat)
        cur = 1
        nxt = lambda c: (c + 1) % ops.L.CHAIN_SLOTS  # noqa: E731
        n = len(self.ed_cz)
        for j in range(n):
            ch.linear_fwd(cur, nxt(cur), P[f"classifier.net.{3 * j}.weight"], P[f"classifier.net.{3 * j}.bias"], ACT_GELU,
                          zout=self.ed_cz[j], out=self.ed_ca[j])
            cur = nxt(cur)
        ch.linear_fwd(cur, nxt(cur), P["classifier.head.weight"], P["classifier.head.bias"], out=self.logits)
        cur = nxt(cur)
        # F.cross_entropy (mean over the batch) forward + backward; the generator's loss weighs it by lambda_emo
        ch.softmax_ce(cur, nxt(cur), self.emot_idx, self.ed_loss_rows, self.lambda_emo / self.B, self.n_classes)
        cur = nxt(cur)
        ch.store(cur, self.dlogits)
        w = P["classifier.head.weight"]
        for j in reversed(range(n)):
            ch.linear_dgrad(cur, nxt(cur), w, gref=self.ed_cz[j], gact=ACT_GELU, out=self.ed_dcz[j])
            cur, w = nxt(cur), P[f"classifier.net.{3 * j}.weight"]
        if not notes:
            ch.linear_dgrad(cur, nxt(cur), w, out=self.ed_dfeat)
            ch.launch()
            yield
            ops.mean_scaled(self.ed_loss_rows, self.emo, 1.0)
            yield
            return
        ch.linear_dgrad(cur, nxt(cur), w, out=self.ed_dproj)
        cur = nxt(cur)
        ch.linear_dgrad(cur, nxt(cur), P["encoder.project.weight"], out=self.ed_dpool)
        ch.launch()
        yield
        if bf16:
            ops.mean_scaled(self.ed_loss_rows, self.emo, 1.0)
            yield
        yield from self._ed_bwd_convs_steps(self.dnotes, None if bf16 else (self.ed_loss_rows, self.emo, 1.0),
                                            lds_pad if lds_pad_bwd is None else lds_pad_bwd)

    def _ed_bwd_convs_steps(self, dnotes: Tensor, mean, lds_pad: int):
        """_ed_bwd_convs, one launch per step (see _ed_steps)."""
        P = self.ED.p
        last = len(self.ed_chans) - 1
        if self.ed_dtype == "bf16":
            ops.meanT_bwd_bf16(self.ed_dpool, self.ed_dz[last], self.ed_z[last], ACT_GELU, self.ed_scale[last])
            yield
            for i in range(last, 0, -1):
                ops.conv_s1_bf16(self.ed_dz[i], self.ed_wb_d[i], self.ed_dz[i - 1], gref=self.ed_z[i - 1], gact=ACT_GELU,
                                 gscale=self.ed_scale[i - 1])
                yield
            ops.conv_s1_bf16(self.ed_dz[0], self.ed_wb_d[0], dnotes)   # fp32 out: the generator's gradient stays fp32
            yield
            return
        ops.meanT_bwd(self.ed_dpool, self.ed_dz[last], gref=self.ed_z[last], gact=ACT_GELU, gscale=self.ed_scale[last], mean=mean)
        yield
        for i in range(last, 0, -1):
            with ops.conv_lds_pad(lds_pad):
                self._ed_conv_dgrad(i, self.ed_dz[i - 1])
            yield
        with ops.conv_lds_pad(lds_pad):
            self._ed_conv_dgrad(0, dnotes)
        yield

    def g_critic_chain(self):
        """Critic forward + input gradient on the generated batch (with the UPDATED critic), added to the emotion
        branch's gradient, then the generator's data-gradient chain down to decoder.pre.2."""
        self.g_critic_front()
        self.g_critic_back()

    def g_critic_front(self):
        """The part of g_critic_chain that does not need the emotion branch's result."""
        B = self.B
        self._d_fwd(self.notes, B, self.emb, head=False)
        # adv = -mean(D(fake)) (train_gan.py:224) rides in the pooling-backward launch: s is written two launches earlier
        self._d_bwd_input(self.ds_g, B, self.emb, self.demb, with_head=True, mean=(self.s[:B], self.adv, -1.0))
        self._stamp(4)

    def g_critic_back(self):
        B = self.B
        self._stamp(5)
        PG, GG = self._gp, (lambda k: self.GE.g["G." + k])
        self._conv5s2("conv_dgrad", self.dZ1[:B], self.D, "conv.0.weight", self.dnotes, accumulate=self.ed_mode == "notes")
        # ---- generator backward: the data-gradient chain down to decoder.pre.2, then pre.2's weight gradient -- 89 % of
        # the generator's gradient bytes, all-reduced while g_backward_b runs.  The deconvolutions' weight gradients
        # are not on that chain and wait in g_backward_b, where they widen the window the all-reduce hides in. ----
        dn = self.dnotes
        if self.dn_dense is not None:      # zero-padded tail rows carry no gradient (models.py:78-81)
            ops.copy_cols(self.dnotes.view(B, -1), 0, self.dn_dense.view(B, -1), 0, self.L3 * self.C)
            dn = self.dn_dense
        # BatchNorm backward: the two column sums ride in the data-gradient launch that produces the incoming gradient (conv16's
        # bnb rider), so the BatchNorm's own backward is ONE launch instead of a reduction pass plus an apply pass.  (Round 2
        # measured the rider as "no faster than the reduction pass" when that pass cost one of three launches; with two
        # launches of ~8 us at the dependent-launch floor on the tail's critical path it is -- MELO_BNB_RIDER=0: the old path.)
        rider = os.environ.get("MELO_BNB_RIDER", "1") == "1"
        for (dy_src, wname, da, a_, z_, dz_, bn, i) in (
                (dn, "G.decoder.deconv.6.weight", self.d_ad3, self.a_d3, self.z_d3, self.d_zd3, "decoder.deconv.4", 1),
                (self.d_zd3, "G.decoder.deconv.3.weight", self.d_ad0, self.a_d0, self.z_d0, self.d_zd0, "decoder.deconv.1", 0)):
            r = self._conv5s2("convT_dgrad", dy_src, self.GE, wname, da,
                              bnb=(a_, z_, self.bn_mean[i], self.bn_invstd[i], ACT_RELU) if rider else None)
            if r.bnb is not None:
                ops.bn_train_bwd_parts(r.bnb[0], r.bnb[1], da, a_, z_, dz_, PG(bn + ".weight"), self.bn_mean[i], self.bn_invstd[i],
                                       GG(bn + ".weight"), GG(bn + ".bias"), ACT_RELU)
            else:
                ops.bn_train_bwd(da, a_, z_, dz_, PG(bn + ".weight"), self.bn_mean[i], self.bn_invstd[i], GG(bn + ".weight"),
                                 GG(bn + ".bias"), ACT_RELU)
        # times pre.2's relu' (mask from y0 = its output as the deconvolution read it), stored in the reference's
        # (B, 256*red) order: the launch writes (b, c, l) itself where conv16 runs it, else a transpose follows
        if not self._conv5s2("convT_dgrad", self.d_zd0, self.GE, "G.decoder.deconv.0.weight", self.d_p2.view(B, 256, self.red),
                             perm=True, gref=self.y0, gact=ACT_RELU).perm:
            self._conv5s2("convT_dgrad", self.d_zd0, self.GE, "G.decoder.deconv.0.weight", self.d_y0, gref=self.y0, gact=ACT_RELU)
            ops.transpose_bcl_blc(self.d_y0, self.d_p2.view(B, 256, self.red))
        # pre.2's weight gradient: launched by g_backward_b with the other weight gradients, or -- data parallel, factor
        # gather -- by g_p2_wgrad from every rank's (d_p2, a_p0)

    def d_update_g_critic_chain(self):
        """Data parallelism: what follows the critic's gradient all-reduce, as one graph."""
        self.d_update()
        self.g_critic_chain()

    def g_backward_b(self, extra_jobs=()):
        B = self.B
        PG, GG = self._gp, (lambda k: self.GE.g["G." + k])
        PE, GEg = self._ep, (lambda k: self.GE.g["E." + k])
        # Weight / bias gradients are collected and go out at the end as one launch per kernel shape (ops.wgrad_multi):
        # the three deconvolutions (their inputs and output gradients were kept by g_backward_a2) and the six small
        # Linear layers, 12 launches at the launch floor before.
        dn = self.dn_dense if self.dn_dense is not None else self.dnotes
        jobs = []
        # `big`: everything whose operands g_critic_back has left behind, 95 % of this pass's gradient FLOPs, as launches of
        # their own (the same job list in every flow: a launch's slice plan -- hence the bits -- depends on its job list)
        big = list(extra_jobs)
        p2_job = None
        if self.coll is not None and self.p2_world:         # C2: pre.2's factors of every rank -> its GLOBAL weight gradient
            self.coll.gather_p2(self, not self._a_p0_gathered)
            p2_job = ops.linear_wgrad(self.a_p0_all, self.d_p2_all, GG("decoder.pre.2.weight"), db=GG("decoder.pre.2.bias"), defer=True)
        if not self.p2_world:
            p2_job = ops.linear_wgrad(self.a_p0, self.d_p2, GG("decoder.pre.2.weight"), db=GG("decoder.pre.2.bias"), defer=True)
        big.append(ops.convT1d_wgrad(self.a_d3, dn, GG("decoder.deconv.6.weight"), db=GG("decoder.deconv.6.bias"), defer=True))
        big.append(ops.convT1d_wgrad(self.a_d0, self.d_zd3, GG("decoder.deconv.3.weight"), db=GG("decoder.deconv.3.bias"),
                                     defer=True))
        big.append(ops.convT1d_wgrad(self.y0, self.d_zd0, GG("decoder.deconv.0.weight"), db=GG("decoder.deconv.0.bias"),
                                     defer=True))
        # (with the in-graph collectives of data parallelism too: they stay on THIS stream -- one communicator, one stream --
        # only launches go to the side; decoder.pre.2's gradient is global from the gathered factors, so its share of the
        # update does not wait for C3)
        tail_fork = self._tail_fork and self.ed_side is not None
        split = tail_fork and self._ge_head and self.GE.ticked and os.environ.get("MELO_ADAM_SPLIT", "0") == "1"
        cur = torch.cuda.current_stream()

        def side(fn):
            """The forked step graph's second fork: pre.2's and the deconvolutions' gradients (and pre.2's share of the update)
            on the side stream BESIDE the small dependent launches below (pre.2's data-gradient, the back chain, the LayerNorm
            parameters), which leave the chip nearly empty.  Other flows: the same launches on this stream -- a launch's
            slice plan depends on its job list, and every flow must produce the same bits."""
            if tail_fork:
                self.ed_side.wait_stream(cur)
                with torch.cuda.stream(self.ed_side):
                    fn()
            else:
                fn()
        if p2_job is not None:
            side(lambda: ops.wgrad_multi([p2_job], tag="_side"))
        side(lambda: ops.wgrad_multi(big, tag="_side2"))
        ops.linear_dgrad(self.d_p2, PG("decoder.pre.2.weight"), self.d_p0, gref=self.a_p0, gact=ACT_RELU)
        if split:
            # MELO_ADAM_SPLIT=1 (off by default): decoder.pre.2's share of the update (89 % of its bytes) as a launch of the side
            # branch, behind its gradient AND behind the launch above, the last reader of pre.2's weights in this step.  It was
            # worth 13 us when the tail's side branch was new; since the critic's weight gradients moved into the emotion
            # branch's stream the undivided update is faster (77.6 k vs 76.7 k samples/s, four alternations)
            side(self._adam_ge_head)
        jobs.append(ops.linear_wgrad(self.lat, self.d_p0, GG("decoder.pre.0.weight"), db=GG("decoder.pre.0.bias"), defer=True))
        jobs.append(ops.linear_wgrad(self.a_n0, self.d_lat, GG("noise_to_latent.net.2.weight"),
                                     db=GG("noise_to_latent.net.2.bias"), defer=True))
        jobs.append(ops.linear_wgrad(self.gin, self.d_n0, GG("noise_to_latent.net.0.weight"),
                                     db=GG("noise_to_latent.net.0.bias"), defer=True))
        # embedding gradient = generator-input slice + critic-head path, then E_num backward
        jobs.append(ops.linear_wgrad(self.e_h2, self.demb, GEg("net.7.weight"), db=GEg("net.7.bias"), defer=True))
        jobs.append(ops.linear_wgrad(self.e_h1, self.d_ez2, GEg("net.4.weight"), db=GEg("net.4.bias"), defer=True))
        jobs.append(ops.linear_wgrad(self.e_x0, self.d_ez1, GEg("net.1.weight"), db=GEg("net.1.bias"), defer=True))
        if self._chain_gf:
            # pre.0's, noise_to_latent's and the numeric encoder's data-gradients as ONE launch: from decoder.pre.2's input
            # gradient back to the LayerNorm output's (models.py:20-27,47-48 and feature_encoder.py:16-42 backwards)
            nd = self.noise_dim
            ch = ops.Chain(B)
            ch.load(0, self.d_p0)
            ch.linear_dgrad(0, 1, PG("decoder.pre.0.weight"), out=self.d_lat)
            if self.ed_mode != "notes":
                ch.load(1, self.ed_dfeat, accumulate=True)
                ch.store(1, self.d_lat)
            ch.linear_dgrad(1, 2, PG("noise_to_latent.net.2.weight"), gref=self.a_n0, gact=ACT_RELU, out=self.d_n0)
            ch.linear_dgrad(2, 3, PG("noise_to_latent.net.0.weight"), out=self.d_gin)
            ch.copy(3, 4, self.E, src_at=nd)
            ch.load(4, self.demb, accumulate=True)
            ch.store(4, self.demb)
            ch.linear_dgrad(4, 5, PE("net.7.weight"), gref=self.e_z2, gact=ACT_GELU, mask=self.dmask[1], out=self.d_ez2)
            ch.linear_dgrad(5, 0, PE("net.4.weight"), gref=self.e_z1, gact=ACT_GELU, mask=self.dmask[0], out=self.d_ez1)
            ch.linear_dgrad(0, 1, PE("net.1.weight"), out=self.d_ex0)
            ch.launch()
            ops.layernorm_bwd_params(self.d_ex0, self.e_xhat, GEg("net.0.weight"), GEg("net.0.bias"))
            ops.wgrad_multi(jobs)
            if tail_fork:
                torch.cuda.current_stream().wait_stream(self.ed_side)
            return
        ops.linear_dgrad(self.d_p0, PG("decoder.pre.0.weight"), self.d_lat)
        if self.ed_mode != "notes":
            ops.axpby(self.ed_dfeat, self.d_lat, 1.0, 1.0)
        ops.linear_dgrad(self.d_lat, PG("noise_to_latent.net.2.weight"), self.d_n0, gref=self.a_n0, gact=ACT_RELU)
        ops.linear_dgrad(self.d_n0, PG("noise_to_latent.net.0.weight"), self.d_gin)
        if self._chain_e:           # the encoder's data-gradient chain as one launch (feature_encoder.py:16-42 backwards)
            ch = ops.Chain(B)
            ch.load(0, self.d_gin[:, self.noise_dim:self.noise_dim + self.E])
            ch.load(0, self.demb, accumulate=True)
            ch.store(0, self.demb)
            ch.linear_dgrad(0, 1, PE("net.7.weight"), gref=self.e_z2, gact=ACT_GELU, mask=self.dmask[1], out=self.d_ez2)
            ch.linear_dgrad(1, 2, PE("net.4.weight"), gref=self.e_z1, gact=ACT_GELU, mask=self.dmask[0], out=self.d_ez1)
            ch.linear_dgrad(2, 3, PE("net.1.weight"), out=self.d_ex0)
            ch.launch()
        else:
            ops.copy_cols(self.d_gin, self.noise_dim, self.demb, 0, self.E, accumulate=True)
            ops.linear_dgrad(self.demb, PE("net.7.weight"), self.d_ez2, gref=self.e_z2, gact=ACT_GELU, emul=self.dmask[1])
            ops.linear_dgrad(self.d_ez2, PE("net.4.weight"), self.d_ez1, gref=self.e_z1, gact=ACT_GELU, emul=self.dmask[0])
            ops.linear_dgrad(self.d_ez1, PE("net.1.weight"), self.d_ex0)
        ops.layernorm_bwd_params(self.d_ex0, self.e_xhat, GEg("net.0.weight"), GEg("net.0.bias"))
        ops.wgrad_multi(jobs)
        if tail_fork:
            torch.cuda.current_stream().wait_stream(self.ed_side)

    def enable_p2_gather(self, world: int):
        """Data parallelism without all-reducing decoder.pre.2.weight's gradient (16.8 of the 18.8 MB at cfg2): that
        gradient is d_p2^T a_p0, a product of two per-sample factors of (B, 256 red) and (B, 512) floats -- 2.2 MB per
        rank.  The ranks all-gather the factors into d_p2_all / a_p0_all and every rank computes the GLOBAL batch's
        weight gradient itself (the same sum an all-reduce would deliver, `world` times the rows in one wgrad launch)."""
        self.p2_world = int(world)
        self.d_p2_all = torch.zeros(self.p2_world * self.B, 256 * self.red, device=self.dev)
        self.a_p0_all = torch.zeros(self.p2_world * self.B, 512, device=self.dev)

    def g_backward_p2b(self):
        """Second half of the G-step backward under enable_p2_gather: pre.2's global weight gradient from the gathered
        factors, then everything g_backward_b does."""
        self.g_backward_b([ops.linear_wgrad(self.a_p0_all, self.d_p2_all, self.GE.g["G.decoder.pre.2.weight"],
                                            db=self.GE.g["G.decoder.pre.2.bias"], defer=True)])

    def g_p2_wgrad(self):
        """pre.2's global weight (and bias) gradient from the gathered factors alone: g_backward_b (which skips it under
        enable_p2_gather) runs while the all-gather is in flight, this after it."""
        ops.linear_wgrad(self.a_p0_all, self.d_p2_all, self.GE.g["G.decoder.pre.2.weight"], db=self.GE.g["G.decoder.pre.2.bias"])

    def g_update(self):
        if self.coll is not None:
            self.coll.reduce_g(self)        # C3
            self._p2_pending = self._a_p0_gathered = False
        self._adam(self.GE, self.lr_g)
        self._stamp(6)

    # -------------------------------------------------------------------------------------
    # graph capture / replay
    # -------------------------------------------------------------------------------------
    def run(self, name: str, use_graph: bool = True):
        """Run one of the sub-step methods (d_backward, d_update, g_step_rng, dg_step_rng, ...), replaying its hipGraph
        when captured (the first call runs eagerly, which also warms every workspace)."""
        fn = getattr(self, name)
        if not use_graph:
            return fn()
        # An update graph exists in several forms (Adam state advanced by the preceding draw -- with or without the
        # Philox counter to advance -- or by itself); a replayed graph does not run the Python that tracks which one
        # applies, so it is tracked here by sub-step name.
        fp_upd = {"d_update": self.D, "g_update": self.GE, "d_update_g_critic_chain": self.D,
                  "d_step_g_critic_front": self.D, "g_finish": self.GE, "d_update_g_critic_front": self.D}.get(name)
        if name == "dg_fork_rest":          # both updates inside, both states advanced by dg_forward_rng's draw
            try:
                return self._run_graph(name, fn)
            finally:
                self.D.ticked = self.GE.ticked = False
        key = name + (f"#{fp_upd.ticked}" if fp_upd is not None and fp_upd.ticked else "")
        try:
            return self._run_graph(key, fn)
        finally:
            if name.endswith("_step_rng"):                      # draw and update(s) both inside: nothing left pending
                self.D.ticked = self.GE.ticked = False
            elif name in ("dg_forward_d_backward_rng", "dg_forward_rng"):   # the fused draw: both updates are still to come
                self.D.ticked, self.GE.ticked = True, "nobump"
            elif name.endswith("_rng"):
                (self.D if name.startswith("d_") else self.GE).ticked = True
            if fp_upd is not None:
                fp_upd.ticked = False

    def _run_graph(self, name: str, fn):
        st = self._graphs.get(name)
        if st is None:
            fn()                                  # eager warm-up (allocates workspaces, sets func attrs)
            self._graphs[name] = "warm"
            return
        if st == "warm":
            if getattr(self, "capture_locked", False):
                raise RuntimeError(f"GanEngine.run({name}): graph capture after DataParallel.prepare() -- every sub-step of "
                                   "a data-parallel run must be captured before the first collective is issued")
            st = self._capture(name, fn)
        st[0].launch()
        self.num_batches_tracked += st[1]

    def _capture(self, name: str, fn):
        """Capture fn's launches from the current stream into a hipGraph; returns (graph, BatchNorm forward passes it
        contains -- num_batches_tracked is host state a replay does not touch)."""
        if torch.cuda.current_stream() == torch.cuda.default_stream():
            raise RuntimeError("GanEngine.run: capture needs a non-default stream (use `with torch.cuda.stream(eng.stream)`)")
        nbt = self.num_batches_tracked
        torch.cuda.synchronize()
        g = ops.Graph()
        g.begin()
        try:
            fn()
        finally:
            g.end()
        delta, self.num_batches_tracked = self.num_batches_tracked - nbt, nbt
        self._graphs[name] = (g, delta)
        return self._graphs[name]

    # -------------------------------------------------------------------------------------
    # inference (app.py:92-119: E_num -> G in eval mode)
    # -------------------------------------------------------------------------------------
    def generate(self, noise: Tensor, numeric: Tensor, latent: Optional[Tensor] = None) -> Tensor:
        self.noise.copy_(noise)
        self.numeric.copy_(numeric)
        if latent is not None:
            self.latent.copy_(latent)
        self._e_fwd(False, "g", gin=True)
        self._g_fwd(self.notes, False, "g")
        return self.notes

