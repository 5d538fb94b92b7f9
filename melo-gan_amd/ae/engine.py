"""VaeEngine -- one training step of the reference's VAE (src/ae/model.py:4-148, src/ae/train_ae.py:110-122) as an
explicit launch sequence over libmelogan_hip: ConvEncoder (3x Conv1d s2 + BN + ReLU -> Linear) -> fc_mu /
fc_log_var -> reparameterise -> ConvDecoder (2x Linear -> 3x ConvTranspose1d, BN + ReLU, Tanh) -> MSE + beta*KLD ->
backward -> clip_grad_norm_(1.0) -> AdamW(lr 1e-4, wd 1e-5).  Channels-last throughout; the (B, 128, L) flatten
order of the reference (model.py:46) is kept by one transpose so that encoder._linear and decoder.pre.2 keep the
reference's weight layout.  The VAE is hard-wired to 4 note features (model.py:29,111,121).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

from .. import ops
from ..gan.engine import FlatParams, BN_EPS, BN_MOM
from ..ops import ACT_RELU, ACT_TANH

ENC = ((0, 4, 32), (3, 32, 64), (6, 64, 128))       # (index in encoder.conv, Cin, Cout)
DEC = ((0, 128, 64), (3, 64, 32), (6, 32, 4))       # (index in decoder.deconv, Cin, Cout)


def vae_spec(max_notes: int, latent_dim: int, hidden_dim: int = 512):
    L = max_notes
    for _ in range(3):
        L = (L - 1) // 2 + 1
    red = max(1, max_notes // 8)
    spec, bufs = OrderedDict(), OrderedDict()
    for i, ci, co in ENC:
        spec[f"encoder.conv.{i}.weight"], spec[f"encoder.conv.{i}.bias"] = (co, ci, 5), (co,)
        spec[f"encoder.conv.{i + 1}.weight"], spec[f"encoder.conv.{i + 1}.bias"] = (co,), (co,)
        bufs[f"encoder.conv.{i + 1}.running_mean"], bufs[f"encoder.conv.{i + 1}.running_var"] = (co,), (co,)
    spec["encoder._linear.1.weight"], spec["encoder._linear.1.bias"] = (hidden_dim, 128 * L), (hidden_dim,)
    spec["fc_mu.weight"], spec["fc_mu.bias"] = (latent_dim, hidden_dim), (latent_dim,)
    spec["fc_log_var.weight"], spec["fc_log_var.bias"] = (latent_dim, hidden_dim), (latent_dim,)
    spec["decoder.pre.0.weight"], spec["decoder.pre.0.bias"] = (hidden_dim, latent_dim), (hidden_dim,)
    spec["decoder.pre.2.weight"], spec["decoder.pre.2.bias"] = (128 * red, hidden_dim), (128 * red,)
    for i, ci, co in DEC:
        spec[f"decoder.deconv.{i}.weight"], spec[f"decoder.deconv.{i}.bias"] = (ci, co, 5), (co,)
        if i != 6:
            spec[f"decoder.deconv.{i + 1}.weight"], spec[f"decoder.deconv.{i + 1}.bias"] = (co,), (co,)
            bufs[f"decoder.deconv.{i + 1}.running_mean"], bufs[f"decoder.deconv.{i + 1}.running_var"] = (co,), (co,)
    return spec, bufs, L, red


class VaeEngine:
    def __init__(self, cfg: dict, device="cuda", batch_size: Optional[int] = None, share: Optional["VaeEngine"] = None):
        """share: another VaeEngine whose parameters, optimiser state and BatchNorm buffers this one uses as its own --
        the same model at a different batch size (the validation loader's trailing partial batch, train_ae.py:67, and
        the batch-1 reconstructions of train_ae.py:173-188)."""
        self.cfg = dict(cfg)
        self.dev = torch.device(device)
        B = self.B = int(batch_size or cfg.get("BATCH_SIZE", 32))
        T = self.T = int(cfg["MAX_NOTES"])
        if T < 8:
            raise ValueError("MAX_NOTES < 8 is not supported")
        self.latent = int(cfg["LATENT_DIM"])
        self.lr, self.wd = float(cfg.get("LR", 1e-4)), float(cfg.get("WEIGHT_DECAY", 1e-5))
        spec, bufs, self.Lenc, self.red = vae_spec(T, self.latent)
        d = self.dev
        if share is not None:
            if share.P.spec != spec:
                raise ValueError("VaeEngine(share=...): the two engines must have the same model configuration")
            self.P, self.buf = share.P, share.buf
        else:
            self.P = FlatParams(spec, d)
            self.buf = {k: (torch.ones(s, device=d) if k.endswith("running_var") else torch.zeros(s, device=d)) for k, s in bufs.items()}
        self._tails = {}
        z = lambda *s: torch.zeros(*s, device=d)  # noqa: E731
        self.x, self.eps = z(B, T, 4), z(B, self.latent)
        Ts = [T]
        for _ in range(3):
            Ts.append((Ts[-1] - 1) // 2 + 1)
        self.Ts = Ts
        self.ez = [z(B, Ts[i + 1], co) for i, (_, _, co) in enumerate(ENC)]
        self.ea = [z(B, Ts[i + 1], co) for i, (_, _, co) in enumerate(ENC)]
        self.e_mean = [z(co) for _, _, co in ENC]
        self.e_istd = [z(co) for _, _, co in ENC]
        self.flat = z(B, 128 * self.Lenc)                 # (B, 128, L) reference flatten order
        self.h, self.mu, self.lv, self.zl = z(B, 512), z(B, self.latent), z(B, self.latent), z(B, self.latent)
        self.p0, self.p2 = z(B, 512), z(B, 128 * self.red)
        self.y0 = z(B, self.red, 128)
        Ld = [self.red, 2 * self.red, 4 * self.red, 8 * self.red]
        self.Ld = Ld
        self.dz_ = [z(B, Ld[i + 1], co) for i, (_, _, co) in enumerate(DEC)]      # pre-BN / pre-tanh outputs
        self.da_ = [z(B, Ld[i + 1], co) for i, (_, _, co) in enumerate(DEC[:2])]  # post BN+ReLU
        self.d_mean = [z(co) for _, _, co in DEC[:2]]
        self.d_istd = [z(co) for _, _, co in DEC[:2]]
        self.recon = z(B, T, 4)
        self.loss = z(3)
        self.gnorm = z(2)
        # gradients of activations
        self.g_recon, self.g_rec_dense = z(B, T, 4), (z(B, Ld[3], 4) if Ld[3] != T else None)
        self.g_da = [torch.zeros_like(t) for t in self.da_]
        self.g_dz = [torch.zeros_like(t) for t in self.dz_]
        self.g_y0, self.g_p2, self.g_p0 = z(B, self.red, 128), z(B, 128 * self.red), z(B, 512)
        self.g_z, self.g_mu, self.g_lv = z(B, self.latent), z(B, self.latent), z(B, self.latent)
        self.k_mu, self.k_lv = z(B, self.latent), z(B, self.latent)
        self.g_h, self.g_flat = z(B, 512), z(B, 128 * self.Lenc)
        self.g_ea = [torch.zeros_like(t) for t in self.ea]
        self.g_ez = [torch.zeros_like(t) for t in self.ez]
        self.num_batches_tracked = 0
        self.stream = share.stream if share is not None else torch.cuda.Stream(device=d)
        self.world_size = 1

    def tail(self, rows: int) -> "VaeEngine":
        """The same model (shared parameters and buffers) at a batch of `rows` samples."""
        if rows == self.B:
            return self
        if rows <= 0:
            raise ValueError("tail: rows must be positive")
        if rows not in self._tails:
            self._tails[rows] = VaeEngine(self.cfg, self.dev, rows, share=self)
        return self._tails[rows]

    def load_state(self, P, Bf):
        self.P.load(P)
        for k in self.buf:
            self.buf[k].copy_(Bf[k])

    def init_weights(self, seed: int = 42):
        """torch default initialisers are what the reference uses for the VAE (no weights_init in train_ae.py)."""
        g = torch.Generator().manual_seed(seed)
        for k, s in self.P.spec.items():
            if len(s) == 1:
                self.P.p[k].fill_(1.0 if k.endswith("weight") and (".conv." in k or ".deconv." in k) and int(k.split(".")[2]) in (1, 4, 7) else 0.0)
            else:
                fan_in = s[1] * (s[2] if len(s) == 3 else 1)
                bound = 1.0 / fan_in ** 0.5
                self.P.p[k].copy_((torch.rand(s, generator=g) * 2 - 1) * bound)

    # ---------------------------------------------------------------------------------------------
    def forward(self, train: bool = True):
        p = self.P.p
        a = self.x
        for j, (i, ci, co) in enumerate(ENC):
            ops.conv1d_fwd(a, p[f"encoder.conv.{i}.weight"], self.ez[j], 2, bias=p[f"encoder.conv.{i}.bias"])
            nm = f"encoder.conv.{i + 1}"
            if train:
                ops.bn_train_fwd(self.ez[j], self.ea[j], p[nm + ".weight"], p[nm + ".bias"], self.buf[nm + ".running_mean"],
                                 self.buf[nm + ".running_var"], self.e_mean[j], self.e_istd[j], ACT_RELU, BN_MOM, BN_EPS)
            else:
                ops.bn_eval_fwd(self.ez[j], self.ea[j], p[nm + ".weight"], p[nm + ".bias"], self.buf[nm + ".running_mean"],
                                self.buf[nm + ".running_var"], ACT_RELU, BN_EPS)
            a = self.ea[j]
        # (B, L, 128) -> (B, 128, L): the reference's flatten order (model.py:46)
        ops.transpose_bcl_blc(a, self.flat.view(self.B, 128, self.Lenc))
        ops.linear_fwd(self.flat, p["encoder._linear.1.weight"], self.h, bias=p["encoder._linear.1.bias"], act=ACT_RELU)
        ops.linear_fwd(self.h, p["fc_mu.weight"], self.mu, bias=p["fc_mu.bias"])
        ops.linear_fwd(self.h, p["fc_log_var.weight"], self.lv, bias=p["fc_log_var.bias"])
        ops.reparam_fwd(self.mu, self.lv, self.eps, self.zl)
        ops.linear_fwd(self.zl, p["decoder.pre.0.weight"], self.p0, bias=p["decoder.pre.0.bias"], act=ACT_RELU)
        ops.linear_fwd(self.p0, p["decoder.pre.2.weight"], self.p2, bias=p["decoder.pre.2.bias"], act=ACT_RELU)
        ops.transpose_bcl_blc(self.p2.view(self.B, 128, self.red), self.y0)
        a = self.y0
        for j, (i, ci, co) in enumerate(DEC):
            if i == 6:
                ops.convT1d_fwd(a, p["decoder.deconv.6.weight"], self.recon, bias=p["decoder.deconv.6.bias"], act=ACT_TANH)
                break
            ops.convT1d_fwd(a, p[f"decoder.deconv.{i}.weight"], self.dz_[j], bias=p[f"decoder.deconv.{i}.bias"])
            nm = f"decoder.deconv.{i + 1}"
            if train:
                ops.bn_train_fwd(self.dz_[j], self.da_[j], p[nm + ".weight"], p[nm + ".bias"], self.buf[nm + ".running_mean"],
                                 self.buf[nm + ".running_var"], self.d_mean[j], self.d_istd[j], ACT_RELU, BN_MOM, BN_EPS)
            else:
                ops.bn_eval_fwd(self.dz_[j], self.da_[j], p[nm + ".weight"], p[nm + ".bias"], self.buf[nm + ".running_mean"],
                                self.buf[nm + ".running_var"], ACT_RELU, BN_EPS)
            a = self.da_[j]
        if train:
            self.num_batches_tracked += 1

    def backward(self, beta: float):
        p, g, B = self.P.p, self.P.g, self.B
        ops.vae_loss(self.recon, self.x, self.mu, self.lv, beta, self.loss, self.g_recon, self.k_mu, self.k_lv)
        # tanh' and (if T % 8 != 0) drop the zero-padded tail rows
        ops.act_bwd(self.g_recon, self.g_recon, gref=self.recon, gact=ACT_TANH)
        dn = self.g_recon
        jobs = []        # weight gradients: collected, launched together at the end (ops.wgrad_multi)
        if self.g_rec_dense is not None:
            ops.copy_cols(self.g_recon.view(B, -1), 0, self.g_rec_dense.view(B, -1), 0, self.Ld[3] * 4)
            dn = self.g_rec_dense
        ins = [self.y0, self.da_[0], self.da_[1]]
        for j in (2, 1, 0):
            i = DEC[j][0]
            jobs.append(ops.convT1d_wgrad(ins[j], dn, g[f"decoder.deconv.{i}.weight"], db=g[f"decoder.deconv.{i}.bias"], defer=True))
            if j == 0:
                ops.convT1d_dgrad(dn, p[f"decoder.deconv.{i}.weight"], self.g_y0)
                break
            ops.convT1d_dgrad(dn, p[f"decoder.deconv.{i}.weight"], self.g_da[j - 1])
            nm = f"decoder.deconv.{DEC[j - 1][0] + 1}"
            ops.bn_train_bwd(self.g_da[j - 1], self.da_[j - 1], self.dz_[j - 1], self.g_dz[j - 1], p[nm + ".weight"],
                             self.d_mean[j - 1], self.d_istd[j - 1], g[nm + ".weight"], g[nm + ".bias"], ACT_RELU)
            dn = self.g_dz[j - 1]
        ops.transpose_bcl_blc(self.g_y0, self.g_p2.view(B, 128, self.red), gref=self.p2, gact=ACT_RELU)
        jobs.append(ops.linear_wgrad(self.p0, self.g_p2, g["decoder.pre.2.weight"], db=g["decoder.pre.2.bias"], defer=True))
        ops.linear_dgrad(self.g_p2, p["decoder.pre.2.weight"], self.g_p0, gref=self.p0, gact=ACT_RELU)
        jobs.append(ops.linear_wgrad(self.zl, self.g_p0, g["decoder.pre.0.weight"], db=g["decoder.pre.0.bias"], defer=True))
        ops.linear_dgrad(self.g_p0, p["decoder.pre.0.weight"], self.g_z)
        ops.reparam_bwd(self.g_z, self.lv, self.eps, self.k_mu, self.k_lv, self.g_mu, self.g_lv)
        jobs.append(ops.linear_wgrad(self.h, self.g_mu, g["fc_mu.weight"], db=g["fc_mu.bias"], defer=True))
        jobs.append(ops.linear_wgrad(self.h, self.g_lv, g["fc_log_var.weight"], db=g["fc_log_var.bias"], defer=True))
        ops.linear_dgrad(self.g_mu, p["fc_mu.weight"], self.g_h)
        ops.linear_dgrad(self.g_lv, p["fc_log_var.weight"], self.g_h, accumulate=True)
        ops.act_bwd(self.g_h, self.g_h, gref=self.h, gact=ACT_RELU)
        jobs.append(ops.linear_wgrad(self.flat, self.g_h, g["encoder._linear.1.weight"], db=g["encoder._linear.1.bias"], defer=True))
        ops.linear_dgrad(self.g_h, p["encoder._linear.1.weight"], self.g_flat)
        ops.transpose_bcl_blc(self.g_flat.view(B, 128, self.Lenc), self.g_ea[2])      # back to (B, L, 128)
        ins = [self.x, self.ea[0], self.ea[1]]
        for j in (2, 1, 0):
            i = ENC[j][0]
            nm = f"encoder.conv.{i + 1}"
            ops.bn_train_bwd(self.g_ea[j], self.ea[j], self.ez[j], self.g_ez[j], p[nm + ".weight"], self.e_mean[j],
                             self.e_istd[j], g[nm + ".weight"], g[nm + ".bias"], ACT_RELU)
            jobs.append(ops.conv1d_wgrad(ins[j], self.g_ez[j], g[f"encoder.conv.{i}.weight"], 2, db=g[f"encoder.conv.{i}.bias"], defer=True))
            if j > 0:
                ops.conv1d_dgrad(self.g_ez[j], p[f"encoder.conv.{i}.weight"], self.g_ea[j - 1], 2)
        ops.wgrad_multi(jobs)

    def update(self):
        """clip_grad_norm_(1.0) (train_ae.py:121) folded into the fused AdamW through a device scalar."""
        ops.grad_norm_clip(self.P.grad[:self.P.n], 1.0, self.gnorm)
        ops.adam_flat(self.P.data, self.P.grad, self.P.m, self.P.v, self.P.state, self.lr, 0.9, 0.999, 1e-8, self.wd,
                      grad_scale=1.0 / self.world_size, gs_dev=self.gnorm[1:2])

    def step(self, x: torch.Tensor, eps: torch.Tensor, beta: float):
        self.x.copy_(x)
        self.eps.copy_(eps)
        self.forward(train=True)
        self.backward(beta)
        self.update()
