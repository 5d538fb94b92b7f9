"""Generator and Discriminator (WGAN-GP critic) -- mirror of /root/reference/src/gan/models.py.

Constructor signatures (models.py:86-87, :137), forward signatures (:108, :158) and state_dict keys
(noise_to_latent.net.*, decoder.pre.*, decoder.deconv.*; conv.*, fc.1.*, real_fake.*) are the
reference's; the arithmetic runs on libmelogan_hip with channels-last activations, so the reference's
permutes (models.py:73,159) do not exist here.  forward() is forward-only (no autograd tape); training
runs through melo_gan_amd.gan.engine.GanEngine.
"""
import torch
import torch.nn as nn

from .. import ops


class NoiseToLatent(nn.Module):
    """models.py:20-29 (parameter container; Generator.forward runs the kernels)."""

    def __init__(self, noise_dim, out_dim, hidden=512):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(noise_dim, hidden), nn.ReLU(True), nn.Linear(hidden, out_dim))


class GeneratorDecoder(nn.Module):
    """models.py:32-83."""

    def __init__(self, latent_dim=128, max_notes=512, out_channels=4):
        super().__init__()
        self.latent_dim, self.max_notes = latent_dim, max_notes
        self.reduced_len = max(1, max_notes // 8)
        self.pre = nn.Sequential(nn.Linear(latent_dim, 512), nn.ReLU(True),
                                 nn.Linear(512, 256 * self.reduced_len), nn.ReLU(True))
        self.deconv = nn.Sequential(
            nn.ConvTranspose1d(256, 128, 5, 2, 2, 1), nn.BatchNorm1d(128), nn.ReLU(True),
            nn.ConvTranspose1d(128, 64, 5, 2, 2, 1), nn.BatchNorm1d(64), nn.ReLU(True),
            nn.ConvTranspose1d(64, out_channels, 5, 2, 2, 1))


def _bn_relu(z, bn: nn.BatchNorm1d, train: bool):
    a = torch.empty_like(z)
    if train:
        C = z.shape[-1]
        sm, si = torch.empty(C, device=z.device), torch.empty(C, device=z.device)
        ops.bn_train_fwd(z, a, bn.weight, bn.bias, bn.running_mean, bn.running_var, sm, si, ops.ACT_RELU,
                         bn.momentum, bn.eps)
        bn.num_batches_tracked += 1
    else:
        ops.bn_eval_fwd(z, a, bn.weight, bn.bias, bn.running_mean, bn.running_var, ops.ACT_RELU, bn.eps)
    return a


class Generator(nn.Module):
    def __init__(self, noise_dim=128, latent_dim=128, mode="conditioning", hidden=512, max_notes=512, note_dim=4,
                 numeric_embed_dim=0):
        super().__init__()
        assert mode in ("conditioning", "warm_start")
        if max_notes < 8:
            raise ValueError("max_notes < 8 (the reference's trim branch) is not supported")
        self.mode, self.noise_dim, self.latent_dim = mode, noise_dim, latent_dim
        self.max_notes, self.note_dim, self.numeric_embed_dim = max_notes, note_dim, numeric_embed_dim
        self.input_dim = noise_dim + numeric_embed_dim + (latent_dim if mode == "conditioning" else 0)
        self.noise_to_latent = NoiseToLatent(self.input_dim, latent_dim, hidden=hidden)
        self.decoder = GeneratorDecoder(latent_dim=latent_dim, max_notes=max_notes, out_channels=note_dim)

    @torch.no_grad()
    def forward(self, noise, encoder_latent=None, numeric_embedding=None):
        """noise (B, noise_dim), encoder_latent (B, latent_dim) [conditioning mode], numeric_embedding
        (B, numeric_embed_dim) -> (notes (B, max_notes, note_dim), latent (B, latent_dim))."""
        inputs = [noise]
        if self.numeric_embed_dim > 0:
            assert numeric_embedding is not None, "numeric_embedding is required"
            inputs.append(numeric_embedding)
        if self.mode == "conditioning":
            assert encoder_latent is not None, "conditioning mode requires encoder latent input"
            inputs.append(encoder_latent)
        x = torch.cat([t.float() for t in inputs], dim=1).contiguous()
        B, dev = x.shape[0], x.device
        n2l, dec = self.noise_to_latent.net, self.decoder
        h = torch.empty(B, n2l[0].out_features, device=dev)
        ops.linear_fwd(x, n2l[0].weight, h, bias=n2l[0].bias, act=ops.ACT_RELU)
        latent = torch.empty(B, self.latent_dim, device=dev)
        ops.linear_fwd(h, n2l[2].weight, latent, bias=n2l[2].bias)
        p0 = torch.empty(B, 512, device=dev)
        ops.linear_fwd(latent, dec.pre[0].weight, p0, bias=dec.pre[0].bias, act=ops.ACT_RELU)
        red = dec.reduced_len
        p2 = torch.empty(B, 256 * red, device=dev)
        ops.linear_fwd(p0, dec.pre[2].weight, p2, bias=dec.pre[2].bias, act=ops.ACT_RELU)
        y0 = torch.empty(B, red, 256, device=dev)
        ops.transpose_bcl_blc(p2.view(B, 256, red), y0)
        z = torch.empty(B, 2 * red, 128, device=dev)
        ops.convT1d_fwd(y0, dec.deconv[0].weight, z, bias=dec.deconv[0].bias)
        a = _bn_relu(z, dec.deconv[1], self.training)
        z = torch.empty(B, 4 * red, 64, device=dev)
        ops.convT1d_fwd(a, dec.deconv[3].weight, z, bias=dec.deconv[3].bias)
        a = _bn_relu(z, dec.deconv[4], self.training)
        out = torch.zeros(B, self.max_notes, self.note_dim, device=dev)      # zero-pad rows (models.py:78-81)
        ops.convT1d_fwd(a, dec.deconv[6].weight, out, bias=dec.deconv[6].bias)
        return out, latent


class Discriminator(nn.Module):
    """WGAN-GP critic; returns one score per sample (models.py:132-169)."""

    def __init__(self, max_notes=512, note_dim=4, emb_dim=256, numeric_embed_dim=0):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv1d(note_dim, 64, 5, 2, 2), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv1d(64, 128, 5, 2, 2), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv1d(128, 256, 5, 2, 2), nn.LeakyReLU(0.2, inplace=True))
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(nn.Flatten(), nn.Linear(256, emb_dim), nn.LeakyReLU(0.2, inplace=True))
        self.combined_dim = emb_dim + numeric_embed_dim
        self.real_fake = nn.Linear(self.combined_dim, 1)

    @torch.no_grad()
    def forward(self, notes, numeric_embedding=None):
        x = notes.contiguous().float()
        B, dev = x.shape[0], x.device
        for i in (0, 2, 4):
            conv = self.conv[i]
            T = (x.shape[1] - 1) // 2 + 1
            y = torch.empty(B, T, conv.out_channels, device=dev)
            ops.conv1d_fwd(x, conv.weight, y, 2, bias=conv.bias, act=ops.ACT_LRELU)
            x = y
        h = torch.empty(B, x.shape[2], device=dev)
        ops.meanT_fwd(x, h)
        f = torch.empty(B, self.fc[1].out_features, device=dev)
        ops.linear_fwd(h, self.fc[1].weight, f, bias=self.fc[1].bias, act=ops.ACT_LRELU)
        s = torch.empty(B, device=dev)
        emb = numeric_embedding.contiguous().float() if numeric_embedding is not None else None
        ops.dhead_fwd(f, emb, self.real_fake.weight.view(-1), self.real_fake.bias, s)
        return s
