"""Pre-training of the emotion discriminator on MI355X: one step of
/root/reference/src/emotion_discriminator/train_ed.py:51-82 -- train-mode forward (BatchNorm batch statistics, classifier
dropout), mean cross-entropy, backward, AdamW -- over libmelogan_hip.  The frozen, eval-mode use of the same network on
the GAN hot path lives in melo_gan_amd.gan.engine.GanEngine.

Layout: activations (B, T, C) channels-last like everywhere else; parameters / gradients / Adam moments in one flat
fp32 buffer each (one fused AdamW launch); BatchNorm running statistics in `buf`.  Backward is hand-derived:
  logits -> classifier (Linear, GELU, Dropout)* -> project Linear -> mean over T -> [GELU, BatchNorm(train), Conv1d]*.

`use_spectral_norm: true` (ed_model.py:29-32,79-82: every encoder Conv1d and the classifier's hidden Linear layers wrapped in
torch.nn.utils.spectral_norm): the flat parameter buffer holds weight_orig, the buffers hold weight_u / weight_v; one launch
per forward runs the power iteration (training mode) and writes the effective weights w_orig / sigma the layers then use
(mg_spectral_norm_fwd), one launch after the weight gradients turns d w_eff into d w_orig (mg_spectral_norm_bwd).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import torch

from .. import ops
from ..ops import ACT_GELU, ACT_NONE
from ..gan.engine import FlatParams, emotion_disc_spec

Tensor = torch.Tensor


class EdEngine:
    """One replica of the emotion discriminator's training state on one GPU (input_mode == 'notes')."""

    def __init__(self, cfg: dict, device="cuda", batch_size: Optional[int] = None, max_notes: Optional[int] = None,
                 share: Optional["EdEngine"] = None):
        """share: another EdEngine whose parameters, optimiser state, BatchNorm buffers and Philox counter this one
        uses as its own -- the same model at a different batch size (the trailing partial batch of an epoch: the
        reference's loaders have no drop_last, ed_dataset.py:542-558)."""
        if cfg.get("input_mode", "latent") != "notes":
            raise ValueError("EdEngine: only input_mode='notes' (the convolutional encoder) is pre-trained here")
        self.cfg = cfg
        self.dev = d = torch.device(device)
        self.B = B = int(batch_size or cfg.get("batch_size", 64))
        self.T = T = int(max_notes or cfg.get("max_notes", 512))
        self.C = C = int(cfg.get("note_dim", 4))
        self.n_classes = int(cfg.get("n_classes", 4))
        self.p_drop = float(cfg.get("dropout", 0.2))
        opt = cfg.get("optimizer", {})
        self.lr = float(opt.get("lr", 2e-4))
        self.betas = tuple(float(b) for b in opt.get("betas", (0.9, 0.999)))
        self.weight_decay = float(opt.get("weight_decay", 0.0))
        self.decoupled = str(opt.get("name", "adamw")).lower() == "adamw"
        spec, bufs, chans = emotion_disc_spec(cfg)
        if share is not None:
            if share.P.spec != spec:
                raise ValueError("EdEngine(share=...): the two engines must have the same model configuration")
            self.P, self.buf = share.P, share.buf
        else:
            self.P = FlatParams(spec, d)
            self.buf: Dict[str, Tensor] = OrderedDict()
            for k, s in bufs.items():
                self.buf[k] = torch.ones(s, device=d) if k.endswith("running_var") else torch.zeros(s, device=d)
        self.chans = chans
        self.mlp = tuple(cfg.get("mlp_hidden", (256, 128)))
        # three-tap layers by minimal filtering (csrc/conv_wino.hip: 2/3 of the direct form's matrix-pipe work) where the
        # problem is big enough to be bound by it: forward with >= 128 output columns, data-gradient with >= 128 input channels
        # (the GAN engine's rule); the weights change every step, so both filter images of every such layer are transformed
        # by ONE launch at the top of the step.  MELO_ED_WINO=0: the direct window GEMMs.
        import os
        wino = os.environ.get("MELO_ED_WINO", "1") == "1" and B * T >= 4096
        self.wino_f = [bool(wino and k == 3 and co >= 128 and ops.wino3_supported(B, T, ci, co)) for (ci, co, k) in chans]
        self.wino_d = [bool(wino and k == 3 and ci >= 128 and i > 0 and ops.wino3_supported(B, T, co, ci)) for i, (ci, co, k) in enumerate(chans)]
        wimg = lambda on, cin, n: torch.zeros(cin // 4, 4, n, 4, device=d) if on else None  # noqa: E731
        self.wino_wf = [wimg(f_, ci, co) for f_, (ci, co, _) in zip(self.wino_f, chans)]
        self.wino_wd = [wimg(f_, co, ci) for f_, (ci, co, _) in zip(self.wino_d, chans)]
        # spectral normalisation: which layers, their u / v buffers (module state), effective weights and sigmas (derived)
        self.sn_names = []
        if cfg.get("use_spectral_norm", False):
            self.sn_names = [f"encoder.conv.{i}.net.0" for i in range(len(chans))] + [f"classifier.net.{3 * j}" for j in range(len(self.mlp))]
        if share is not None:
            self.w_eff, self.sn_sigma = share.w_eff, share.sn_sigma
        else:
            self.w_eff, self.sn_sigma = {}, {}
            for nm in self.sn_names:
                shp = spec[nm + ".weight"]
                self.buf[nm + ".weight_u"] = torch.zeros(shp[0], device=d)
                self.buf[nm + ".weight_v"] = torch.zeros(math.prod(shp[1:]), device=d)
                self.w_eff[nm] = torch.zeros(shp, device=d)
                self.sn_sigma[nm] = torch.ones(1, device=d)
        hid = cfg.get("notes_hidden", 256)
        f = lambda *s: torch.empty(*s, device=d)      # noqa: E731
        self.x = f(B, T, C)
        self.y = torch.zeros(B, dtype=torch.int64, device=d)
        self.z = [f(B, T, co) for (_, co, _) in chans]           # conv output (bias included), BatchNorm input
        self.a = [f(B, T, co) for (_, co, _) in chans]           # GELU(BatchNorm(z))
        self.dz = [f(B, T, co) for (_, co, _) in chans]
        self.da = [f(B, T, co) for (_, co, _) in chans]
        self.bn_mean = [f(co) for (_, co, _) in chans]
        self.bn_invstd = [f(co) for (_, co, _) in chans]
        self.pool, self.dpool = f(B, chans[-1][1]), f(B, chans[-1][1])
        self.proj, self.dproj = f(B, hid), f(B, hid)
        self.cz = [f(B, h) for h in self.mlp]                    # classifier pre-activations
        self.ca = [f(B, h) for h in self.mlp]                    # after GELU and dropout
        self.dcz = [f(B, h) for h in self.mlp]
        self.dmask = [torch.ones(B, h, device=d) for h in self.mlp]      # keep-mask / (1 - p)
        self.logits, self.dlogits = f(B, self.n_classes), f(B, self.n_classes)
        self.loss = torch.zeros(1, device=d)
        self.rng_step = share.rng_step if share is not None else torch.zeros(1, dtype=torch.int64, device=d)
        self.rng_seed = int(cfg.get("seed", 42))
        self.stream = share.stream if share is not None else torch.cuda.Stream(device=d)
        self._graphs = {}
        self._tails: Dict[int, "EdEngine"] = {}

    # ---- state in / out ---------------------------------------------------------------------------------
    def init_weights(self, seed: int = 42):
        """torch's default initialisation of the reference module (the trainer applies no weights_init to the ED):
        Conv1d / Linear weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm gamma = 1, beta = 0."""
        g = torch.Generator().manual_seed(seed)
        for k, s in self.P.spec.items():
            if ".net.1." in k:                                   # BatchNorm1d affine
                self.P.p[k].fill_(1.0 if k.endswith("weight") else 0.0)
                continue
            wshape = self.P.spec[k[:-len("bias")] + "weight"] if k.endswith("bias") else s
            bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
            self.P.p[k].copy_((torch.rand(s, generator=g) * 2 - 1) * bound)
        for k, v in self.buf.items():
            if k.endswith("weight_u") or k.endswith("weight_v"):      # torch: normalize(randn) (spectral_norm.py, SpectralNorm.apply)
                r = torch.randn(v.shape, generator=g)
                v.copy_(r / r.norm().clamp_min(1e-12))
            else:
                v.fill_(1.0 if k.endswith("running_var") else 0.0)

    def set_lr(self, lr: float):
        """ReduceLROnPlateau: the learning rate is a launch argument of the AdamW kernel, baked into every captured
        graph that contains the update ('update', 'update#ticked' and the one-graph training step 'step_rng'): those
        are dropped and re-captured with the new rate on their next run.  Engines sharing this one's parameters
        (tail()) follow."""
        self.lr = float(lr)
        for k in [k for k in self._graphs if k.startswith("update") or k == "step_rng"]:
            del self._graphs[k]
        for t in self._tails.values():
            t.set_lr(lr)

    def tail(self, rows: int) -> "EdEngine":
        """The same model (shared parameters / optimiser state / buffers) at a batch of `rows` < B samples."""
        if not 0 < rows < self.B:
            raise ValueError(f"tail: rows={rows} must be in (0, {self.B})")
        if rows not in self._tails:
            self._tails[rows] = EdEngine(self.cfg, self.dev, rows, self.T, share=self)
            self._tails[rows].lr = self.lr
        return self._tails[rows]

    def load_state(self, params: Dict[str, Tensor], buffers: Optional[Dict[str, Tensor]] = None):
        """params may be a torch state_dict of the reference module: a spectrally normalised layer's weight is `weight_orig`
        there, its `weight_u` / `weight_v` are buffers (taken from `params` too when `buffers` lacks them)."""
        params = dict(params)
        for nm in self.sn_names:
            if nm + ".weight_orig" in params:
                params[nm + ".weight"] = params[nm + ".weight_orig"]
        self.P.load(params)
        src = dict(params)
        src.update(buffers or {})
        for k, v in src.items():
            if k in self.buf:
                self.buf[k].copy_(v.to(torch.float32))

    def state_dict(self) -> "OrderedDict[str, Tensor]":
        """Keys and shapes of EmotionDiscriminator.state_dict() (ed_model.py), incl. num_batches_tracked."""
        sd = OrderedDict()
        P = self.P.state_dict()
        for i in range(len(self.chans)):
            pre = f"encoder.conv.{i}.net."
            sd[pre + "0.weight"], sd[pre + "0.bias"] = P[pre + "0.weight"], P[pre + "0.bias"]
            sd[pre + "1.weight"], sd[pre + "1.bias"] = P[pre + "1.weight"], P[pre + "1.bias"]
            sd[pre + "1.running_mean"] = self.buf[pre + "1.running_mean"].cpu().clone()
            sd[pre + "1.running_var"] = self.buf[pre + "1.running_var"].cpu().clone()
            sd[pre + "1.num_batches_tracked"] = torch.tensor(int(self.P.state[0].item()), dtype=torch.int64)
        for k, v in P.items():
            if k not in sd:
                sd[k] = v
        for nm in self.sn_names:          # torch.nn.utils.spectral_norm's keys: weight_orig (parameter), weight_u, weight_v (buffers)
            sd[nm + ".weight_orig"] = sd.pop(nm + ".weight")
            sd[nm + ".weight_u"] = self.buf[nm + ".weight_u"].cpu().clone()
            sd[nm + ".weight_v"] = self.buf[nm + ".weight_v"].cpu().clone()
        return sd

    def set_batch(self, x: Tensor, y: Tensor):
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def set_masks(self, masks: Sequence[Tensor]):
        """Injected classifier dropout masks, already scaled by 1/(1-p) (parity tests)."""
        for dst, m in zip(self.dmask, masks):
            dst.copy_(m.to(torch.float32), non_blocking=True)

    def draw_masks(self):
        """Production path: one Philox launch draws both keep-masks (scaled) and advances the AdamW state; the
        update advances the Philox counter (mg_rng_fill_tick / mg_adam_flat_ticked)."""
        ops.rng_fill(None, None, self.dmask[0], self.dmask[1] if len(self.dmask) > 1 else None, self.p_drop,
                     self.rng_seed, self.rng_step, tick_state=self.P.state, betas=self.betas)
        self.P.ticked = True

    # ---- the step ---------------------------------------------------------------------------------------
    def _sn_layers(self, with_grad: bool = False):
        out = []
        for nm in self.sn_names:
            ly = dict(w_orig=self.P.p[nm + ".weight"], w_eff=self.w_eff[nm], u=self.buf[nm + ".weight_u"],
                      v=self.buf[nm + ".weight_v"], sigma=self.sn_sigma[nm])
            if with_grad:
                ly["dw"] = self.P.g[nm + ".weight"]
            out.append(ly)
        return out

    def _w(self, name: str) -> Tensor:
        """The weight layer `name` computes with: w_orig / sigma when it is spectrally normalised."""
        return self.w_eff[name] if name in self.w_eff else self.P.p[name + ".weight"]

    def forward(self, train: bool = True):
        P, x = self.P.p, self.x
        if self.sn_names:          # power iteration (training mode) + effective weights of every normalised layer: one launch
            ops.spectral_norm_fwd(self._sn_layers(), train)
        jobs = []
        for i, (ci, co, _) in enumerate(self.chans):
            w = self._w(f"encoder.conv.{i}.net.0")
            if self.wino_f[i]:
                jobs.append((w, self.wino_wf[i], co, ci, ci * 3, 3, False))
            if self.wino_d[i] and train:
                jobs.append((w, self.wino_wd[i], ci, co, 3, ci * 3, True))
        if jobs:
            ops.wino3_weights_multi(jobs)
        for i in range(len(self.chans)):
            pre = f"encoder.conv.{i}.net."
            if self.wino_f[i]:
                ops.conv_wino3(x, self.wino_wf[i], self.z[i], bias=P[pre + "0.bias"])
            else:
                ops.conv1d_fwd(x, self._w(pre + "0"), self.z[i], 1, bias=P[pre + "0.bias"])
            if train:
                ops.bn_train_fwd(self.z[i], self.a[i], P[pre + "1.weight"], P[pre + "1.bias"], self.buf[pre + "1.running_mean"],
                                 self.buf[pre + "1.running_var"], self.bn_mean[i], self.bn_invstd[i], act=ACT_GELU)
            else:
                ops.bn_eval_fwd(self.z[i], self.a[i], P[pre + "1.weight"], P[pre + "1.bias"], self.buf[pre + "1.running_mean"],
                                self.buf[pre + "1.running_var"], act=ACT_GELU)
            x = self.a[i]
        ops.meanT_fwd(x, self.pool)
        ops.linear_fwd(self.pool, P["encoder.project.weight"], self.proj, bias=P["encoder.project.bias"])
        feat = self.proj
        for j in range(len(self.mlp)):
            ops.linear_fwd(feat, self._w(f"classifier.net.{3 * j}"), self.ca[j], bias=P[f"classifier.net.{3 * j}.bias"],
                           zout=self.cz[j], act=ACT_GELU, emul=self.dmask[j] if train else None)
            feat = self.ca[j]
        ops.linear_fwd(feat, P["classifier.head.weight"], self.logits, bias=P["classifier.head.bias"])

    def backward(self):
        """Train-mode forward + cross-entropy + gradients of every parameter into self.P.grad."""
        P, G = self.P.p, self.P.g
        self.forward(train=True)
        ops.softmax_ce(self.logits, self.y, self.loss, self.dlogits, 1.0)
        n = len(self.mlp)
        g, wname, inp = self.dlogits, "classifier.head", self.ca[n - 1]
        jobs = []        # weight gradients: collected, launched together at the end (ops.wgrad_multi)
        for j in reversed(range(n)):
            jobs.append(ops.linear_wgrad(inp, g, G[wname + ".weight"], db=G[wname + ".bias"], defer=True))
            # d/d(pre-activation) = dropout mask * GELU'(cz)
            ops.linear_dgrad(g, self._w(wname), self.dcz[j], gref=self.cz[j], gact=ACT_GELU, emul=self.dmask[j])
            g, wname = self.dcz[j], f"classifier.net.{3 * j}"
            inp = self.ca[j - 1] if j > 0 else self.proj
        jobs.append(ops.linear_wgrad(self.proj, g, G[wname + ".weight"], db=G[wname + ".bias"], defer=True))
        ops.linear_dgrad(g, self._w(wname), self.dproj)
        jobs.append(ops.linear_wgrad(self.pool, self.dproj, G["encoder.project.weight"], db=G["encoder.project.bias"], defer=True))
        ops.linear_dgrad(self.dproj, P["encoder.project.weight"], self.dpool)
        last = len(self.chans) - 1
        ops.meanT_bwd(self.dpool, self.da[last])
        for i in range(last, -1, -1):
            pre = f"encoder.conv.{i}.net."
            ops.bn_train_bwd(self.da[i], self.a[i], self.z[i], self.dz[i], P[pre + "1.weight"], self.bn_mean[i],
                             self.bn_invstd[i], G[pre + "1.weight"], G[pre + "1.bias"], act=ACT_GELU, beta=P[pre + "1.bias"])
            xin = self.a[i - 1] if i > 0 else self.x
            jobs.append(ops.conv1d_wgrad(xin, self.dz[i], G[pre + "0.weight"], 1, db=G[pre + "0.bias"], defer=True))
            if i > 0 and self.wino_d[i]:
                ops.conv_wino3(self.dz[i], self.wino_wd[i], self.da[i - 1])
            elif i > 0:
                ops.conv1d_dgrad(self.dz[i], self._w(pre + "0"), self.da[i - 1], 1)
        ops.wgrad_multi(jobs)
        if self.sn_names:          # the launches above left d w_eff: through w_eff = w_orig / sigma to d w_orig
            ops.spectral_norm_bwd(self._sn_layers(with_grad=True))

    def backward_rng(self):
        self.draw_masks()
        self.backward()

    def step_rng(self):
        """Mask draw + forward/backward + AdamW as ONE capturable sequence (one graph launch per training step)."""
        self.backward_rng()
        self.update()

    def update(self):
        fp = self.P
        ops.adam_flat(fp.data, fp.grad, fp.m, fp.v, fp.state, self.lr, *self.betas,
                      weight_decay=self.weight_decay if self.decoupled else 0.0,
                      ticked_rng_step=self.rng_step if fp.ticked else None)
        fp.ticked = False

    def run(self, name: str, use_graph: bool = True):
        """hipGraph replay of backward_rng / update / forward_eval (first call eager, second call captures)."""
        fn = getattr(self, name)
        if not use_graph:
            return fn()
        key = name + ("#ticked" if name == "update" and self.P.ticked else "")
        try:
            st = self._graphs.get(key)
            if st is None:
                fn()
                self._graphs[key] = "warm"
                return
            if st == "warm":
                if torch.cuda.current_stream() == torch.cuda.default_stream():
                    raise RuntimeError("EdEngine.run: capture needs a non-default stream")
                torch.cuda.synchronize()
                g = ops.Graph()
                g.begin()
                try:
                    fn()
                finally:
                    g.end()
                self._graphs[key] = st = g
            st.launch()
        finally:
            if name.endswith("_rng"):
                self.P.ticked = True
            if name in ("update", "step_rng"):
                self.P.ticked = False

    def forward_eval(self):
        self.forward(train=False)
