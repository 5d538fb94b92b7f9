"""FeatureEncoder -- mirror of /root/reference/src/gan/feature_encoder.py:5-45 over the HIP library.

Same constructor, same state_dict keys (net.0 LayerNorm, net.{1,4,7} Linear).  forward() is an
inference/eval-or-train *forward only* (no autograd tape): training runs through
melo_gan_amd.gan.engine.GanEngine, which owns the hand-derived backward.
"""
import torch
import torch.nn as nn

from .. import ops


class FeatureEncoder(nn.Module):
    def __init__(self, in_dim: int, hidden_dims=(256, 128), out_dim: int = 128, dropout: float = 0.2,
                 use_sn: bool = False):
        super().__init__()
        if use_sn:
            raise NotImplementedError("spectral norm (ENCODER_USE_SN) is never enabled by the reference configs")
        layers = [nn.LayerNorm(in_dim)]
        prev = in_dim
        for h in hidden_dims:
            layers += [nn.Linear(prev, h), nn.GELU(), nn.Dropout(dropout)]
            prev = h
        layers.append(nn.Linear(prev, out_dim))
        self.net = nn.Sequential(*layers)      # parameter container only: forward never calls it
        self.p_drop = dropout

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous().float()
        B = x.shape[0]
        h = torch.empty_like(x)
        ops.layernorm_fwd(x, h, None, self.net[0].weight, self.net[0].bias)
        lin = [m for m in self.net if isinstance(m, nn.Linear)]
        for i, m in enumerate(lin):
            y = torch.empty(B, m.out_features, device=x.device)
            last = i == len(lin) - 1
            mask = None
            if not last and self.training and self.p_drop > 0:
                mask = (torch.rand(B, m.out_features, device=x.device) >= self.p_drop).float() / (1.0 - self.p_drop)
            ops.linear_fwd(h, m.weight, y, bias=m.bias, act=ops.ACT_NONE if last else ops.ACT_GELU, emul=mask)
            h = y
        return h
