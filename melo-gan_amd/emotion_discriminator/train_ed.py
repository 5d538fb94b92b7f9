#!/usr/bin/env python3
"""Pre-training of the emotion discriminator -- the MI355X-native counterpart of
/root/reference/src/emotion_discriminator/train_ed.py (same YAML keys, same loop: train epoch, validation epoch,
ReduceLROnPlateau, best / periodic checkpoints {"epoch","model","optimizer","cfg"}, early stopping):

    python -m melo_gan_amd.emotion_discriminator.train_ed --config config/ed_config.yaml

Data: the row-aligned arrays the GAN trainer already uses, `<splits_dir>/<stem of {split}_split_csv>/{notes,emotion}.npy`
(kept resident in HBM); the reference's per-file manifest/.npz loader (ed_dataset.py) is out of scope.  Like the
reference's loaders (no drop_last, ed_dataset.py:542-558) an epoch ends with the trailing partial batch, run by a second
engine of that batch size over the same parameters (EdEngine.tail); metrics are sample-weighted (train_ed.py:75-82).
--synthetic N trains on N random rolls (smoke runs without the git-ignored dataset).  input_mode must be 'notes'.
"""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from .. import ops
from ..gan import config as C
from ..gan.utils import check_labels, emotion_to_index, seed_everything
from .engine import EdEngine


class Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau (rel threshold mode, cooldown 0, min_lr 0), train_ed.py:101-123."""

    def __init__(self, mode="min", factor=0.5, patience=5, threshold=1e-4):
        self.mode, self.factor, self.patience, self.threshold = mode, factor, patience, threshold
        self.best = float("inf") if mode == "min" else -float("inf")
        self.bad = 0

    def step(self, metric: float, lr: float) -> float:
        better = metric < self.best * (1 - self.threshold) if self.mode == "min" else metric > self.best * (1 + self.threshold)
        if better:
            self.best, self.bad = metric, 0
            return lr
        self.bad += 1
        if self.bad > self.patience:
            self.bad = 0
            return lr * self.factor
        return lr


def load_split(cfg: dict, split: str, device):
    key = f"{split}_split_csv"
    if not cfg.get(key):
        raise ValueError(f"Missing split csv for '{split}' in config; expected key '{key}'.")
    d = os.path.join(cfg.get("splits_dir", os.path.dirname(cfg[key]) or "data/splits"), Path(cfg[key]).stem)
    paths = [os.path.join(d, n) for n in ("notes.npy", "emotion.npy")]
    if not all(os.path.exists(p) for p in paths):
        raise FileNotFoundError(f"{paths}: export the split to notes.npy / emotion.npy (the per-file .npz loader of the "
                                "reference is not implemented)")
    notes = torch.from_numpy(np.ascontiguousarray(np.load(paths[0]), dtype=np.float32)).to(device)
    labels = check_labels(torch.tensor([emotion_to_index(e) for e in np.load(paths[1], allow_pickle=True)], dtype=torch.int64),
                          int(cfg.get("n_classes", 4)), f"{split} split labels").to(device)
    return notes, labels


def synthetic_split(n, T, Cn, seed, device):
    """Learnable toy labels: the class is the quadrant of (mean pitch, mean velocity) of a roll."""
    g = np.random.default_rng(seed)
    x = g.uniform(-1, 1, (n, T, Cn)).astype(np.float32)
    bias = g.uniform(-0.5, 0.5, (n, 1, 2)).astype(np.float32)
    x[:, :, :2] = np.clip(x[:, :, :2] * 0.5 + bias, -1, 1)
    y = (x[:, :, 0].mean(1) > 0).astype(np.int64) * 2 + (x[:, :, 1].mean(1) > 0).astype(np.int64)
    return torch.from_numpy(x).to(device), torch.from_numpy(y).to(device)


def run_epoch(eng: EdEngine, x, y, train: bool, use_graph: bool, gen=None):
    """train_ed.py:51-82: sample-weighted mean loss and accuracy of one pass; one device->host read per epoch."""
    n, B = x.shape[0], eng.B
    if n == 0:
        raise ValueError("run_epoch: the split is empty")
    perm = torch.randperm(n, generator=gen).to(x.device) if train else torch.arange(n, device=x.device)
    acc = torch.zeros(2, device=x.device)
    for lo in range(0, n, B):
        rows = min(B, n - lo)
        e = eng if rows == B else eng.tail(rows)          # trailing partial batch: same parameters, smaller batch
        idx = perm[lo:lo + rows]
        e.set_batch(x.index_select(0, idx), y.index_select(0, idx))
        if train:
            e.run("step_rng", use_graph)
        else:
            e.run("forward_eval", use_graph)
            ops.softmax_ce(e.logits, e.y, e.loss, None, 1.0)
        acc[0:1] += e.loss * rows
        acc[1:2] += (e.logits.argmax(dim=1) == e.y).float().sum()
    loss, a = (acc / n).tolist()
    return loss, a


def save_checkpoint(eng: EdEngine, cfg: dict, epoch: int, is_best: bool):
    """train_ed.py:30-48: <checkpoint_dir>/<save_name> for the best model, ed_epochNNN.pth otherwise."""
    os.makedirs(cfg.get("checkpoint_dir", "data/models/ed"), exist_ok=True)
    name = cfg.get("save_name", "ed_best.pth") if is_best else f"ed_epoch{epoch:03d}.pth"
    path = os.path.join(cfg.get("checkpoint_dir", "data/models/ed"), name)
    fp = eng.P
    opt = {"state": {"step": float(fp.state[0].item()), "exp_avg": fp.m.cpu(), "exp_avg_sq": fp.v.cpu()},
           "param_groups": [{"lr": eng.lr, "betas": eng.betas, "eps": 1e-8, "weight_decay": eng.weight_decay}],
           "layout": {k: list(v) for k, v in fp.offsets.items()}}
    torch.save({"epoch": epoch, "model": eng.state_dict(), "optimizer": opt, "cfg": cfg}, path)
    return path


def train(cfg: dict, synthetic: int = 0, use_graph: bool = True):
    if cfg.get("input_mode", "latent") != "notes":
        raise ValueError("melo_gan_amd.emotion_discriminator.train_ed: input_mode must be 'notes'")
    if not torch.cuda.is_available():
        raise RuntimeError("melo_gan_amd has no CPU path: a MI355X (ROCm) device is required")
    seed_everything(cfg.get("seed", 42))
    device = torch.device("cuda", torch.cuda.current_device())
    T, Cn = int(cfg.get("max_notes", 512)), int(cfg.get("note_dim", 4))
    if synthetic:
        xt, yt = synthetic_split(synthetic, T, Cn, cfg.get("seed", 42), device)
        xv, yv = synthetic_split(max(synthetic // 4, int(cfg.get("batch_size", 64))), T, Cn, cfg.get("seed", 42) + 1, device)
    else:
        (xt, yt), (xv, yv) = load_split(cfg, "train", device), load_split(cfg, "val", device)
    eng = EdEngine(cfg, device, int(cfg.get("batch_size", 64)), T)
    eng.init_weights(cfg.get("seed", 42))
    sch = cfg.get("scheduler") or {}
    plateau = None
    if str(sch.get("name", "")).lower() == "reducelronplateau":
        plateau = Plateau(sch.get("mode", "min"), sch.get("factor", 0.5), sch.get("patience", 5), sch.get("threshold", 1e-4))
    epochs, patience = cfg.get("num_epochs", 50), cfg.get("early_stopping_patience", 10)
    by_loss = cfg.get("metric_for_best", "val_loss") == "val_loss"
    best, best_epoch = (float("inf") if by_loss else 0.0), 0
    gen = torch.Generator().manual_seed(cfg.get("seed", 42))
    print("Starting Emotion Discriminator Training")
    print("Input mode:", cfg["input_mode"], "| Device:", device, "| Epochs:", epochs, "| Best metric target:",
          cfg.get("metric_for_best", "val_loss"))
    with torch.cuda.stream(eng.stream):
        for epoch in range(1, epochs + 1):
            tl, ta = run_epoch(eng, xt, yt, True, use_graph, gen)
            vl, va = run_epoch(eng, xv, yv, False, use_graph)
            metric = vl if by_loss else va
            if plateau is not None:
                lr = plateau.step(metric, eng.lr)
                if lr != eng.lr:
                    eng.set_lr(lr)
            better = metric < best if by_loss else metric > best
            status = ""
            if better:
                best, best_epoch = metric, epoch
                save_checkpoint(eng, cfg, epoch, True)
                status = "New Best"
            print(f"[Epoch {epoch:03d}] Train-Loss={tl:.4f}, Train-Acc={ta:.3f} | Val-Loss={vl:.4f}, Val-Acc={va:.3f} {status}")
            if epoch - best_epoch >= patience:
                print(f"Early stopping triggered at epoch {epoch}. Best epoch was {best_epoch}.")
                break
            if epoch % cfg.get("save_freq", 5) == 0:
                save_checkpoint(eng, cfg, epoch, False)
    print("Training Completed. Best epoch:", best_epoch, "Best metric:", best)
    return eng, best


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="config/ed_config.yaml")
    ap.add_argument("--synthetic", type=int, default=0, help="train on N synthetic rolls instead of the split arrays")
    ap.add_argument("--epochs", type=int, default=None, help="override num_epochs")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args(argv)
    cfg = C.load_config(args.config)
    if args.epochs is not None:
        cfg["num_epochs"] = args.epochs
    train(cfg, args.synthetic, not args.no_graph)


if __name__ == "__main__":
    main()
