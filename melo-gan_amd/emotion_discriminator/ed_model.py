"""EmotionDiscriminator -- mirror of /root/reference/src/emotion_discriminator/ed_model.py:25-165.

Same cfg keys (input_mode, latent_dim, note_dim, notes_hidden, notes_blocks, mlp_hidden, n_classes, dropout,
use_spectral_norm), same state_dict keys (encoder.conv.{i}.net.{0,1}.*, encoder.project.*,
classifier.net.{0,3}.*, classifier.head.*).  forward() is the eval-mode forward used on the GAN hot path
(train_gan.py:131-133): BatchNorm with running statistics folded into the conv epilogue, dropout identity.
"""
from typing import Dict

import torch
import torch.nn as nn

from .. import ops


class ConvBlock1D(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, use_sn=False):
        super().__init__()
        self.net = nn.Sequential(nn.Conv1d(in_ch, out_ch, kernel_size, stride, padding), nn.BatchNorm1d(out_ch), nn.GELU())


class NotesEncoder(nn.Module):
    def __init__(self, note_dim=4, hidden_dim=256, num_blocks=4, use_sn=False):
        super().__init__()
        layers, in_ch, ch = [], note_dim, 64
        for i in range(num_blocks):
            layers.append(ConvBlock1D(in_ch, ch, 5 if i == 0 else 3, padding=2 if i == 0 else 1))
            in_ch, ch = ch, min(ch * 2, hidden_dim)
        self.conv = nn.Sequential(*layers)
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.project = nn.Linear(in_ch, hidden_dim)


class MLPClassifier(nn.Module):
    def __init__(self, in_dim, hidden_dims=(256, 128), n_classes=4, dropout=0.2, use_sn=False):
        super().__init__()
        layers, prev = [], in_dim
        for h in hidden_dims:
            layers += [nn.Linear(prev, h), nn.GELU(), nn.Dropout(dropout)]
            prev = h
        self.net = nn.Sequential(*layers)
        self.head = nn.Linear(prev, n_classes)


class EmotionDiscriminator(nn.Module):
    def __init__(self, cfg: Dict):
        super().__init__()
        self.cfg = cfg.copy()
        self.input_mode = cfg.get("input_mode", "latent")
        self.n_classes = cfg.get("n_classes", 4)
        if cfg.get("use_spectral_norm", False):
            raise NotImplementedError("use_spectral_norm is false in the reference's ed_config.yaml")
        mlp = tuple(cfg.get("mlp_hidden", (256, 128)))
        if self.input_mode == "latent":
            self.encoder = None
            self.classifier = MLPClassifier(cfg.get("latent_dim", 128), mlp, self.n_classes, cfg.get("dropout", 0.2))
        elif self.input_mode == "notes":
            hid = cfg.get("notes_hidden", 256)
            self.encoder = NotesEncoder(cfg.get("note_dim", 4), hid, cfg.get("notes_blocks", 4))
            self.classifier = MLPClassifier(hid, mlp, self.n_classes, cfg.get("dropout", 0.2))
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
                ops.conv1d_fwd(x, conv.weight, y, 1, scale=sc, shift=sh, act=ops.ACT_GELU)
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
                ops.linear_fwd(feats, m.weight, y, bias=m.bias, act=ops.ACT_GELU)
                feats = y
        logits = torch.empty(B, self.n_classes, device=dev)
        ops.linear_fwd(feats, self.classifier.head.weight, logits, bias=self.classifier.head.bias)
        return logits

    def predict_proba(self, x):
        return torch.softmax(self.forward(x), dim=-1)

    def predict(self, x):
        return self.forward(x).argmax(dim=-1)
