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
