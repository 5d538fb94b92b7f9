"""The oracle's restatement of torch.nn.utils.spectral_norm (oracle/melo_oracle.py::spectral_norm_weight), pinned against
PyTorch's own wrapper -- the dependency the reference calls at src/emotion_discriminator/ed_model.py:29-32,79-82 and
src/gan/feature_encoder.py:24-31: effective weight, the in-place power iteration of the buffers, and the gradient that
reaches weight_orig, over several training steps and in eval mode."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import melo_oracle as O


@pytest.mark.parametrize("kind", ["conv", "linear"])
def test_spectral_norm_restatement_equals_torchs_wrapper(kind):
    torch.manual_seed(3)
    if kind == "conv":
        m = nn.Conv1d(6, 10, 3, padding=1)
        x = torch.randn(4, 6, 12)
    else:
        m = nn.Linear(9, 7)
        x = torch.randn(5, 9)
    m = torch.nn.utils.spectral_norm(m)
    w = m.weight_orig.detach().clone().requires_grad_(True)
    u, v = m.weight_u.detach().clone(), m.weight_v.detach().clone()
    m.train()
    for step in range(3):
        y = m(x)
        (y * y).sum().backward()
        w_eff = O.spectral_norm_weight(w, u, v, train=True)
        y2 = F.conv1d(x, w_eff, m.bias.detach(), 1, 1) if kind == "conv" else F.linear(x, w_eff, m.bias.detach())
        (y2 * y2).sum().backward()
        torch.testing.assert_close(y2, y.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(u, m.weight_u, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(v, m.weight_v, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(w.grad, m.weight_orig.grad, rtol=1e-5, atol=1e-6)
        with torch.no_grad():                      # one plain SGD step on both, gradients cleared
            m.weight_orig -= 0.05 * m.weight_orig.grad
            w -= 0.05 * w.grad
        m.weight_orig.grad = None
        w.grad = None
    m.eval()
    u0, v0 = u.clone(), v.clone()
    y = m(x)
    w_eff = O.spectral_norm_weight(w.detach(), u, v, train=False)
    y2 = F.conv1d(x, w_eff, m.bias.detach(), 1, 1) if kind == "conv" else F.linear(x, w_eff, m.bias.detach())
    torch.testing.assert_close(y2, y.detach(), rtol=1e-6, atol=1e-6)
    assert torch.equal(u, u0) and torch.equal(v, v0)          # eval: no power iteration


def test_ed_forward_with_spectral_norm_matches_a_torch_module_built_like_the_reference():
    """emotion_disc_fwd(use_spectral_norm) against an nn.Module assembled the way ed_model.py:25-95 does (Conv1d/Linear wrapped
    in torch's spectral_norm, BatchNorm, GELU, mean over time, project, MLP) in training mode, dropout off."""
    torch.manual_seed(0)
    cfg = dict(input_mode="notes", note_dim=4, notes_hidden=32, notes_blocks=2, mlp_hidden=[16, 8], n_classes=4, dropout=0.0,
               use_spectral_norm=True)
    sn = torch.nn.utils.spectral_norm
    convs = nn.ModuleList([sn(nn.Conv1d(4, 64, 5, 1, 2)), sn(nn.Conv1d(64, 32, 3, 1, 1))])
    bns = nn.ModuleList([nn.BatchNorm1d(64), nn.BatchNorm1d(32)])
    project = nn.Linear(32, 32)
    lins = nn.ModuleList([sn(nn.Linear(32, 16)), sn(nn.Linear(16, 8))])
    head = nn.Linear(8, 4)
    P, Bf = {}, {}
    for i in range(2):
        P[f"encoder.conv.{i}.net.0.weight"] = convs[i].weight_orig.detach().clone()
        P[f"encoder.conv.{i}.net.0.bias"] = convs[i].bias.detach().clone()
        Bf[f"encoder.conv.{i}.net.0.weight_u"] = convs[i].weight_u.detach().clone()
        Bf[f"encoder.conv.{i}.net.0.weight_v"] = convs[i].weight_v.detach().clone()
        P[f"encoder.conv.{i}.net.1.weight"] = bns[i].weight.detach().clone()
        P[f"encoder.conv.{i}.net.1.bias"] = bns[i].bias.detach().clone()
        Bf[f"encoder.conv.{i}.net.1.running_mean"] = bns[i].running_mean.clone()
        Bf[f"encoder.conv.{i}.net.1.running_var"] = bns[i].running_var.clone()
    P["encoder.project.weight"], P["encoder.project.bias"] = project.weight.detach().clone(), project.bias.detach().clone()
    for j in range(2):
        P[f"classifier.net.{3 * j}.weight"] = lins[j].weight_orig.detach().clone()
        P[f"classifier.net.{3 * j}.bias"] = lins[j].bias.detach().clone()
        Bf[f"classifier.net.{3 * j}.weight_u"] = lins[j].weight_u.detach().clone()
        Bf[f"classifier.net.{3 * j}.weight_v"] = lins[j].weight_v.detach().clone()
    P["classifier.head.weight"], P["classifier.head.bias"] = head.weight.detach().clone(), head.bias.detach().clone()
    x = torch.randn(3, 16, 4)
    h = x.permute(0, 2, 1)
    for i in range(2):
        h = F.gelu(bns[i](convs[i](h)))
    h = project(h.mean(dim=2))
    for j in range(2):
        h = F.gelu(lins[j](h))
    want = head(h)
    got = O.emotion_disc_fwd(P, Bf, x, cfg, train=True)
    torch.testing.assert_close(got, want.detach(), rtol=1e-5, atol=1e-6)
    assert O.ed_sn_layers(cfg) == ["encoder.conv.0.net.0", "encoder.conv.1.net.0", "classifier.net.0", "classifier.net.3"]
