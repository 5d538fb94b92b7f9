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
        # use_sn (feature_encoder.py:24-31): the hidden Linear layers wrapped in torch.nn.utils.spectral_norm -- state_dict
        # surface weight_orig / weight_u / weight_v; forward() normalises with mg_spectral_norm_fwd.  The reference's trainer
        # never passes it (train_gan.py builds FeatureEncoder without use_sn; ENCODER_USE_SN is read by nothing), so the GAN
        # engine trains the un-normalised encoder only.
        layers = [nn.LayerNorm(in_dim)]
        prev = in_dim
        for h in hidden_dims:
            lin = nn.Linear(prev, h)
            layers += [torch.nn.utils.spectral_norm(lin) if use_sn else lin, nn.GELU(), nn.Dropout(dropout)]
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
            w = m.weight
            if hasattr(m, "weight_orig"):      # power iteration in training mode, as the wrapper's pre-forward hook does
                w0 = m.weight_orig.detach().contiguous()
                w, sigma = torch.empty_like(w0), torch.empty(1, device=x.device)
                u, v = m.weight_u.detach().clone(), m.weight_v.detach().clone()
                ops.spectral_norm_fwd([dict(w_orig=w0, w_eff=w, u=u, v=v, sigma=sigma)], train=self.training)
                if self.training:
                    m.weight_u.copy_(u)
                    m.weight_v.copy_(v)
            ops.linear_fwd(h, w, y, bias=m.bias, act=ops.ACT_NONE if last else ops.ACT_GELU, emul=mask)
            h = y
        return h
