#!/usr/bin/env python3
"""VAE trainer -- counterpart of /root/reference/src/ae/train_ae.py (same --config flag and ae_config.yaml keys:
MAX_NOTES, LATENT_DIM, BATCH_SIZE, LR, WEIGHT_DECAY, EPOCHS, KLD_WARMUP_EPOCHS, BETA, EARLY_STOP_PATIENCE,
CHECKPOINT_DIR, LOG_DIR; same checkpoints: ae_best.pth {'epoch','model_state'}, ae_final.pth = bare state_dict).

    python -m melo_gan_amd.ae.train_ae --config config/ae_config.yaml [--synthetic N]

Per batch (train_ae.py:110-122): forward, MSE + beta*KLD, backward, clip_grad_norm_(1.0), AdamW -- all inside
VaeEngine.  Per epoch (train_ae.py:101-107,126-199): KL warm-up beta, validation with beta=1, ReduceLROnPlateau
(factor 0.5, patience 5, min_lr 1e-6), best-checkpoint, early stopping.  Data: row-aligned
<SPLITS_DIR>/{train,val}/notes.npy arrays kept in HBM (the reference's per-file .npz loader with in-loader
augmentation -- all augmentations are disabled by its config -- is host I/O and out of scope).  Per epoch the first
(up to) 6 validation rolls are reconstructed in eval mode and written as <RECON_DIR>/ep<E>_val<NNN>_{in,out}.mid
(train_ae.py:94,173-188; RECON_FREQ, RECON_DIR) through melo_gan_amd.midi.
"""
import argparse
import os

import numpy as np
import torch

from .. import ops
from ..midi import save_recon_midi
from ..gan.config import load_config
from ..gan.train_gan import _scalar_writer
from .engine import VaeEngine


def _load_split(cfg, name, device):
    p = os.path.join(cfg.get("SPLITS_DIR", "data/splits"), name, "notes.npy")
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found (export the split's normalised notes to one array, or use --synthetic)")
    return torch.from_numpy(np.load(p).astype(np.float32)).to(device)


def state_dict(eng: VaeEngine):
    sd = eng.P.state_dict()
    for k, v in eng.buf.items():
        sd[k] = v.cpu().clone()
        if k.endswith("running_var"):
            sd[k.replace("running_var", "num_batches_tracked")] = torch.tensor(eng.num_batches_tracked, dtype=torch.int64)
    return sd


def train(cfg, synthetic: int = 0):
    if not torch.cuda.is_available():
        raise RuntimeError("melo_gan_amd has no CPU path: a MI355X (ROCm) device is required")
    dev = torch.device("cuda", torch.cuda.current_device())
    model_dir, log_dir = cfg.get("CHECKPOINT_DIR", "models/ae"), cfg.get("LOG_DIR", "experiments/ae")
    os.makedirs(model_dir, exist_ok=True)
    B, T = cfg["BATCH_SIZE"], cfg["MAX_NOTES"]
    if synthetic:
        g = torch.Generator().manual_seed(0)
        train_x = (torch.rand(synthetic, T, 4, generator=g) * 2 - 1).to(dev)
        val_x = (torch.rand(max(B, synthetic // 4), T, 4, generator=g) * 2 - 1).to(dev)
    else:
        train_x, val_x = _load_split(cfg, "train", dev), _load_split(cfg, "val", dev)
    print(f"Train files: {len(train_x)}   Val files: {len(val_x)}")
    eng = VaeEngine(cfg, dev, B)
    eng.init_weights(0)
    writer = _scalar_writer(log_dir)
    best_val, no_improve, patience = float("inf"), 0, cfg.get("EARLY_STOP_PATIENCE", 10)
    warm, final_beta = cfg.get("KLD_WARMUP_EPOCHS", 25), float(cfg.get("BETA", 1.0))
    sched_bad, sched_best = 0, float("inf")
    gen = torch.Generator().manual_seed(0)
    eps = torch.empty(B, eng.latent, device=dev)
    acc = torch.zeros(3, device=dev)
    with torch.cuda.stream(eng.stream):
        for epoch in range(1, cfg["EPOCHS"] + 1):
            beta = final_beta if epoch >= warm else min(final_beta, (epoch / warm) * final_beta)
            perm = torch.randperm(len(train_x), generator=gen).to(dev)
            acc.zero_()
            nb = len(train_x) // B
            for i in range(nb):
                eng.step(train_x.index_select(0, perm[i * B:(i + 1) * B]), eps.normal_(), beta)
                acc += eng.loss
            tr = (acc / max(1, nb)).tolist()
            acc.zero_()
            vb = 0
            # validation: eval-mode BN, beta = 1; drop_last=False (train_ae.py:67), every batch counts once (:153-155)
            for i in range(0, len(val_x), B):
                e = eng.tail(min(B, len(val_x) - i))
                e.x.copy_(val_x[i:i + e.B])
                e.eps.copy_(eps[:e.B].normal_())
                e.forward(train=False)
                ops.vae_loss(e.recon, e.x, e.mu, e.lv, 1.0, e.loss)
                acc += e.loss
                vb += 1
            if vb == 0:
                raise ValueError("the validation split is empty")
            va = (acc / vb).tolist()
            # ReduceLROnPlateau(factor 0.5, patience 5, min_lr 1e-6), train_ae.py:80
            if va[0] < sched_best * (1 - 1e-4):
                sched_best, sched_bad = va[0], 0
            else:
                sched_bad += 1
                if sched_bad > 5:
                    eng.lr, sched_bad = max(eng.lr * 0.5, 1e-6), 0
            print(f"[Epoch {epoch}] Train: {tr[0]:.6f} (Recon: {tr[1]:.6f}, KLD: {tr[2]:.6f}) | "
                  f"Val: {va[0]:.6f} (Recon: {va[1]:.6f}, KLD: {va[2]:.6f})")
            for tag, v in (("loss/train_total", tr[0]), ("loss/train_recon", tr[1]), ("loss/train_kld", tr[2]),
                           ("loss/val_total", va[0]), ("loss/val_recon", va[1]), ("loss/val_kld", va[2]),
                           ("lr", eng.lr), ("beta", beta)):
                writer.add_scalar(tag, v, epoch)
            # reconstructions of the first (up to) 6 validation rolls, eval mode, batch 1 (train_ae.py:94,173-188)
            if epoch % cfg.get("RECON_FREQ", 1) == 0:
                recon_dir = cfg.get("RECON_DIR", os.path.join(log_dir, "reconstructions"))
                e1 = eng.tail(1)
                for j in range(min(6, len(val_x))):
                    e1.x.copy_(val_x[j:j + 1])
                    e1.eps.copy_(eps[:1].normal_())
                    e1.forward(train=False)
                    save_recon_midi(val_x[j].cpu().numpy(), e1.recon[0].cpu().numpy(), recon_dir, f"ep{epoch}_val{j:03d}")
            if va[0] < best_val:
                best_val, no_improve = va[0], 0
                torch.save({"epoch": epoch, "model_state": state_dict(eng)}, os.path.join(model_dir, "ae_best.pth"))
                print("Saved new best model ->", os.path.join(model_dir, "ae_best.pth"))
            else:
                no_improve += 1
            if no_improve >= patience:
                print("No improvement for", patience, "epochs. Early stopping.")
                break
    writer.close()
    print("Training complete. Best val:", best_val)
    torch.save(state_dict(eng), os.path.join(model_dir, "ae_final.pth"))
    return eng


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="config/ae_config.yaml", help="Path to config yaml")
    parser.add_argument("--synthetic", type=int, default=0)
    parser.add_argument("--epochs", type=int, default=None)
    args = parser.parse_args(argv)
    cfg = load_config(args.config)
    if args.epochs is not None:
        cfg["EPOCHS"] = args.epochs
    train(cfg, args.synthetic)


if __name__ == "__main__":
    main()
