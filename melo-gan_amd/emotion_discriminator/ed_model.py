"""EmotionDiscriminator -- mirror of /root/reference/src/emotion_discriminator/ed_model.py:25-165.

Same cfg keys (input_mode, latent_dim, note_dim, notes_hidden, notes_blocks, mlp_hidden, n_classes, dropout,
use_spectral_norm), same state_dict keys (encoder.conv.{i}.net.{0,1}.*, encoder.project.*,
classifier.net.{0,3}.*, classifier.head.*; with use_spectral_norm the wrapped layers' weight_orig / weight_u / weight_v).  forward() is the eval-mode forward used on the GAN hot path
(train_gan.py:131-133): BatchNorm with running statistics folded into the conv epilogue, dropout identity.
"""
from typing import Dict

import torch
import torch.nn as nn

from .. import ops


def _sn(m: nn.Module, use_sn: bool) -> nn.Module:
    """ed_model.py:29-32,79-82: the layer wrapped in torch.nn.utils.spectral_norm -- here for its state_dict surface
    (weight_orig, weight_u, weight_v); forward() below computes the normalised weight itself (effective_weight)."""
    return torch.nn.utils.spectral_norm(m) if use_sn else m


def effective_weight(m: nn.Module) -> torch.Tensor:
    """m.weight, or for a spectrally normalised layer weight_orig / sigma(u, v) as the wrapper computes it in eval mode
    (no power iteration): one mg_spectral_norm_fwd launch."""
    if not hasattr(m, "weight_orig"):
        return m.weight
    w = m.weight_orig.detach().contiguous()
    w_eff, sigma = torch.empty_like(w), torch.empty(1, device=w.device)
    ops.spectral_norm_fwd([dict(w_orig=w, w_eff=w_eff, u=m.weight_u.detach().clone(), v=m.weight_v.detach().clone(), sigma=sigma)],
                          train=False)
    return w_eff


class ConvBlock1D(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, use_sn=False):
        super().__init__()
        self.net = nn.Sequential(_sn(nn.Conv1d(in_ch, out_ch, kernel_size, stride, padding), use_sn), nn.BatchNorm1d(out_ch),
                                 nn.GELU())


class NotesEncoder(nn.Module):
    def __init__(self, note_dim=4, hidden_dim=256, num_blocks=4, use_sn=False):
        super().__init__()
        layers, in_ch, ch = [], note_dim, 64
        for i in range(num_blocks):
            layers.append(ConvBlock1D(in_ch, ch, 5 if i == 0 else 3, padding=2 if i == 0 else 1, use_sn=use_sn))
            in_ch, ch = ch, min(ch * 2, hidden_dim)
        self.conv = nn.Sequential(*layers)
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.project = nn.Linear(in_ch, hidden_dim)


class MLPClassifier(nn.Module):
    def __init__(self, in_dim, hidden_dims=(256, 128), n_classes=4, dropout=0.2, use_sn=False):
        super().__init__()
        layers, prev = [], in_dim
        for h in hidden_dims:
            layers += [_sn(nn.Linear(prev, h), use_sn), nn.GELU(), nn.Dropout(dropout)]
            prev = h
        self.net = nn.Sequential(*layers)
        self.head = nn.Linear(prev, n_classes)


class EmotionDiscriminator(nn.Module):
    def __init__(self, cfg: Dict):
        super().__init__()
        self.cfg = cfg.copy()
        self.input_mode = cfg.get("input_mode", "latent")
        self.n_classes = cfg.get("n_classes", 4)
        use_sn = bool(cfg.get("use_spectral_norm", False))      # training with it: emotion_discriminator/engine.py::EdEngine
        mlp = tuple(cfg.get("mlp_hidden", (256, 128)))
        if self.input_mode == "latent":
            self.encoder = None
            self.classifier = MLPClassifier(cfg.get("latent_dim", 128), mlp, self.n_classes, cfg.get("dropout", 0.2), use_sn)
        elif self.input_mode == "notes":
            hid = cfg.get("notes_hidden", 256)
            self.encoder = NotesEncoder(cfg.get("note_dim", 4), hid, cfg.get("notes_blocks", 4), use_sn)
            self.classifier = MLPClassifier(hid, mlp, self.n_classes, cfg.get("dropout", 0.2), use_sn)
        else:
            raise ValueError("input_mode must be 'latent' or 'notes'")

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("only the eval-mode forward of the frozen ED is implemented (call .eval())")
        x = x.contiguous().float()
        dev, B = x.device, x.shape[0]
        if self.input_mode == "notes":
            if x.dim() != 3:
                raise ValueError(f"Expected notes input shape (B, T, note_dim), got {x.shape}")
            for blk in self.encoder.conv:
                conv, bn = blk.net[0], blk.net[1]
                C = conv.out_channels
                sc, sh = torch.empty(C, device=dev), torch.empty(C, device=dev)
                ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.bias, sc, sh, bn.eps)
                y = torch.empty(B, x.shape[1], C, device=dev)
                ops.conv1d_fwd(x, effective_weight(conv), y, 1, scale=sc, shift=sh, act=ops.ACT_GELU)
                x = y
            h = torch.empty(B, x.shape[2], device=dev)
            ops.meanT_fwd(x, h)
            feats = torch.empty(B, self.encoder.project.out_features, device=dev)
            ops.linear_fwd(h, self.encoder.project.weight, feats, bias=self.encoder.project.bias)
        else:
            if x.dim() != 2:
                raise ValueError(f"Expected latent input shape (B, latent_dim), got {x.shape}")
            feats = x
        for m in self.classifier.net:
            if isinstance(m, nn.Linear):
                y = torch.empty(B, m.out_features, device=dev)
                ops.linear_fwd(feats, effective_weight(m), y, bias=m.bias, act=ops.ACT_GELU)
                feats = y
        logits = torch.empty(B, self.n_classes, device=dev)
        ops.linear_fwd(feats, self.classifier.head.weight, logits, bias=self.classifier.head.bias)
        return logits

    def predict_proba(self, x):
        return torch.softmax(self.forward(x), dim=-1)

    def predict(self, x):
        return self.forward(x).argmax(dim=-1)
