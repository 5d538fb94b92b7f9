"""CPU: host-side logic that needs no GPU -- label validation, the plateau scheduler, DataParallel step orders live in
test_dp_cpu.py."""
import numpy as np
import pytest
import torch

import melo_gan_amd  # noqa: F401
from melo_gan_amd.gan.utils import check_labels, emotion_to_index


def test_check_labels_rejects_what_cross_entropy_would():
    ok = check_labels(torch.tensor([0, 3, 1, 2]), 4)
    assert ok.tolist() == [0, 3, 1, 2]
    for bad in ([0, -1, 2], [4], [0, 1, 100]):
        with pytest.raises(ValueError, match="outside"):
            check_labels(torch.tensor(bad), 4, "labels")
    # emotion_to_index (src/gan/utils.py:63-73) maps unknown names / None to -1: caught when the dataset is built
    idx = [emotion_to_index(e) for e in ("happy", "SAD", None, "bored", np.array([0, 0, 1, 0]), 3)]
    assert idx == [0, 1, -1, -1, 2, 3]
    with pytest.raises(ValueError, match="2 of 6"):
        check_labels(torch.tensor(idx), 4)


def test_plateau_matches_torch_reduce_lr_on_plateau():
    from melo_gan_amd.emotion_discriminator.train_ed import Plateau
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=1e-4)
    mine, lr = Plateau("min", 0.5, 2, 1e-4), 1.0
    g = np.random.default_rng(0)
    for m in np.concatenate([np.linspace(2, 1, 5), 1 + 0.01 * g.random(12), np.linspace(0.9, 0.5, 4), 0.5 + 0.01 * g.random(9)]):
        sch.step(float(m))
        lr = mine.step(float(m), lr)
        assert lr == opt.param_groups[0]["lr"]


def test_optimizer_checkpoint_is_a_torch_adam_state_dict():
    """opt_G / opt_D of gan_epochNNNN.pth (reference train_gan.py:267-276) are `torch.optim.Adam.state_dict()`s of the
    reference's optimisers: a torch Adam over parameters of the same shapes, in the same order, loads them as they are --
    and the flat buffers take them back (resume)."""
    import torch
    from collections import OrderedDict
    from melo_gan_amd.gan.engine import FlatParams
    from melo_gan_amd.gan.train_gan import adam_state_dict, load_adam_state_dict
    spec = OrderedDict([("a.weight", (3, 4)), ("a.bias", (3,)), ("b.weight", (2, 3, 5))])
    fp = FlatParams(spec, "cpu", first="b.weight")           # flat order differs from the spec's order on purpose
    g = torch.Generator().manual_seed(0)
    fp.m.copy_(torch.randn(fp.m.shape, generator=g))
    fp.v.copy_(torch.rand(fp.v.shape, generator=g))
    fp.state[0] = 7.0
    sd = adam_state_dict(fp, 2e-4, (0.5, 0.9))
    params = [torch.nn.Parameter(torch.zeros(s)) for s in spec.values()]
    opt = torch.optim.Adam(params, lr=1.0, betas=(0.1, 0.2))
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 2e-4 and tuple(opt.param_groups[0]["betas"]) == (0.5, 0.9)
    for i, k in enumerate(spec):
        off, n = fp.offsets[k]
        assert torch.equal(opt.state[params[i]]["exp_avg"], fp.m[off:off + n].view(spec[k]))
        assert torch.equal(opt.state[params[i]]["exp_avg_sq"], fp.v[off:off + n].view(spec[k]))
        assert float(opt.state[params[i]]["step"]) == 7.0
    fp2 = FlatParams(spec, "cpu", first="b.weight")
    load_adam_state_dict(fp2, opt.state_dict())
    assert torch.equal(fp2.m[:fp.n], fp.m[:fp.n]) and torch.equal(fp2.v[:fp.n], fp.v[:fp.n])
    assert float(fp2.state[0]) == 7.0 and abs(float(fp2.state[1]) - 0.5 ** 7) < 1e-15 and abs(float(fp2.state[2]) - 0.9 ** 7) < 1e-15


def test_optimizer_state_dict_is_validated_on_load():
    """A partial or mismatched optimiser checkpoint must fail with a message, not restore a wrong step silently."""
    import pytest
    import torch
    from collections import OrderedDict
    from melo_gan_amd.gan.engine import FlatParams
    from melo_gan_amd.gan.train_gan import adam_state_dict, load_adam_state_dict
    spec = OrderedDict([("a.weight", (3, 4)), ("a.bias", (3,))])
    fp = FlatParams(spec, "cpu")
    fp.state[0] = 3.0
    good = adam_state_dict(fp, 1e-4, (0.5, 0.9))
    sd = {"state": {0: good["state"][0]}, "param_groups": good["param_groups"]}
    with pytest.raises(ValueError, match="parameter entries"):
        load_adam_state_dict(fp, sd)
    sd = adam_state_dict(fp, 1e-4, (0.5, 0.9))
    sd["state"][1]["exp_avg"] = torch.zeros(4)
    with pytest.raises(ValueError, match="shape"):
        load_adam_state_dict(fp, sd)
    sd = adam_state_dict(fp, 1e-4, (0.5, 0.9))
    sd["state"][1]["step"] = torch.tensor(9.0)
    with pytest.raises(ValueError, match="step counts differ"):
        load_adam_state_dict(fp, sd)
    fp.m.fill_(1.0)
    load_adam_state_dict(fp, {"state": {}, "param_groups": good["param_groups"]})       # never stepped: fresh moments
    assert float(fp.m.abs().sum()) == 0.0 and float(fp.state[0]) == 0.0


def test_spectral_norm_checkpoint_weights_fold_at_load():
    """A frozen emotion discriminator trained with use_spectral_norm (ed_model.py:29-32): the weight its eval-mode forward
    uses is weight_orig / (u^T W v); train_gan.spectral_norm_weight reproduces what torch's wrapper computes."""
    import torch
    from melo_gan_amd.gan.train_gan import spectral_norm_weight
    torch.manual_seed(0)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv1d(4, 8, 5, padding=2))
    lin = torch.nn.utils.spectral_norm(torch.nn.Linear(6, 3))
    for m in (conv, lin):                     # a few training-mode forwards move u / v, as training would
        m.train()
        for _ in range(3):
            m(torch.randn(2, 4, 16) if m is conv else torch.randn(2, 6))
        m.eval()
        m(torch.randn(2, 4, 16) if m is conv else torch.randn(2, 6))        # eval forward: sets .weight from the stored u, v
        sd = {"x." + k: v for k, v in m.state_dict().items()}
        got = spectral_norm_weight(sd, "x.weight")
        torch.testing.assert_close(got, m.weight.detach(), rtol=1e-6, atol=1e-7)
    assert spectral_norm_weight({"x.weight": torch.ones(2)}, "x.weight").sum() == 2 and spectral_norm_weight({}, "x.weight") is None
