"""Training utilities -- mirror of /root/reference/src/gan/utils.py:30-90 (seed_everything, weights_init,
emotion_to_index, compute_gradient_penalty).  The MIDI half of that file lives in melo_gan_amd.midi."""
import random

import numpy as np
import torch

from .. import ops


def seed_everything(seed=42):
    """utils.py:30-35."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def weights_init(m):
    """utils.py:37-45: class name contains 'Conv' or 'Linear' -> weight ~ N(0, 0.02), bias = 0."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 or classname.find("Linear") != -1:
        w = getattr(m, "weight", None)
        if isinstance(w, torch.Tensor):
            torch.nn.init.normal_(w.data, 0.0, 0.02)
        if getattr(m, "bias", None) is not None:
            torch.nn.init.constant_(m.bias.data, 0.0)


def check_labels(idx, n_classes: int = 4, what: str = "emotion labels"):
    """Class indices must lie in [0, n_classes): the reference's CrossEntropyLoss raises on anything else (and
    emotion_to_index maps unknown / missing labels to -1).  Raises ValueError naming the first offenders."""
    t = torch.as_tensor(idx)
    bad = ((t < 0) | (t >= n_classes)).nonzero().flatten()
    if bad.numel():
        rows = bad[:8].tolist()
        raise ValueError(f"{what}: {bad.numel()} of {t.numel()} outside [0, {n_classes}) -- first rows {rows}, "
                         f"values {t.flatten()[bad[:8]].tolist()} (unknown emotion names map to -1)")
    return t


def emotion_to_index(emotion):
    """utils.py:63-73."""
    if emotion is None:
        return -1
    if isinstance(emotion, (list, tuple, np.ndarray)):
        arr = np.array(emotion)
        if arr.ndim == 1 and arr.size == 4:
            return int(np.argmax(arr))
        return int(arr)
    if isinstance(emotion, str):
        return {"happy": 0, "sad": 1, "angry": 2, "calm": 3}.get(emotion.lower(), -1)
    try:
        return int(emotion)
    except Exception:
        return -1


@torch.no_grad()
def compute_gradient_penalty(D, real_samples, fake_samples, numeric_embedding, device=None, alpha=None):
    """VALUE of the WGAN-GP penalty (utils.py:75-90) for a melo_gan_amd Discriminator: interpolate, critic
    forward, hand-derived input gradient, mean((||grad||_2 - 1)^2).  The second-order term needed to TRAIN
    the critic is GanEngine.d_backward's tangent pass; this function is for monitoring/evaluation."""
    real = real_samples.contiguous().float()
    fake = fake_samples.contiguous().float()
    B, T, C = real.shape
    dev = real.device
    if alpha is None:
        alpha = torch.rand(B, device=dev)
    xh = torch.empty_like(real)
    ops.gp_interp(real, fake, alpha.reshape(-1).contiguous().float(), xh)
    acts, x = [], xh
    for i in (0, 2, 4):
        conv = D.conv[i]
        y = torch.empty(B, (x.shape[1] - 1) // 2 + 1, conv.out_channels, device=dev)
        ops.conv1d_fwd(x, conv.weight, y, 2, bias=conv.bias, act=ops.ACT_LRELU)
        acts.append(y)
        x = y
    h = torch.empty(B, 256, device=dev)
    ops.meanT_fwd(x, h)
    f = torch.empty(B, D.fc[1].out_features, device=dev)
    ops.linear_fwd(h, D.fc[1].weight, f, bias=D.fc[1].bias, act=ops.ACT_LRELU)
    ones = torch.ones(B, device=dev)
    dU, dH = torch.empty_like(f), torch.empty_like(h)
    ops.dhead_bwd(ones, f, D.real_fake.weight.view(-1), dU)
    ops.linear_dgrad(dU, D.fc[1].weight, dH)
    dz = torch.empty_like(acts[2])
    ops.meanT_bwd(dH, dz, gref=acts[2], gact=ops.ACT_LRELU)
    for i, conv_i in ((1, 4), (0, 2)):
        dprev = torch.empty_like(acts[i])
        ops.conv1d_dgrad(dz, D.conv[conv_i].weight, dprev, 2, gref=acts[i], gact=ops.ACT_LRELU)
        dz = dprev
    gx = torch.empty_like(real)
    ops.conv1d_dgrad(dz, D.conv[0].weight, gx, 2)
    norms, gp = torch.empty(B, device=dev), torch.empty(1, device=dev)
    ops.gp_penalty(gx, None, norms, gp, 1.0)
    return gp[0]
