#!/usr/bin/env python3
"""Latent export -- counterpart of /root/reference/src/ae/encode.py:124-139: run the trained VAE encoder in eval mode over
a split and save `mu` (N, LATENT_DIM) as encoder_feats.npy, the array the GAN / ED datasets look for.

    python -m melo_gan_amd.ae.encode --model models/ae/ae_best.pth --manifest data/splits/train_split.csv \
        --out_file data/splits/train/encoder_feats.npy --config config/ae_config.yaml

Input rows come from `<dir of manifest>/<split>/notes.npy` (or --notes PATH): the row-aligned fast-NPY export of the
split, in manifest order; the reference's per-file .npz loader is out of scope.
"""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from ..gan.config import load_config
from .engine import VaeEngine


def load_model(eng: VaeEngine, path: str):
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model_state"] if "model_state" in ckpt else ckpt          # encode.py:118-119
    eng.load_state({k: sd[k] for k in eng.P.spec}, {k: sd[k] for k in eng.buf})


@torch.no_grad()
def encode(eng: VaeEngine, notes: torch.Tensor) -> np.ndarray:
    """mu of every row of `notes` (N, T, 4), eval-mode BatchNorm, batches of eng.B (last one padded)."""
    n, B = notes.shape[0], eng.B
    out = torch.empty(n, eng.latent, device=notes.device)
    eng.eps.zero_()
    with torch.cuda.stream(eng.stream):
        for i in range(0, n, B):
            chunk = notes[i:i + B]
            eng.x[:chunk.shape[0]].copy_(chunk)
            if chunk.shape[0] < B:
                eng.x[chunk.shape[0]:].zero_()
            eng.forward(train=False)
            out[i:i + chunk.shape[0]].copy_(eng.mu[:chunk.shape[0]])
    torch.cuda.synchronize()
    return out.cpu().numpy()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Encode MIDI .npz files into latent vectors using VAE")
    ap.add_argument("--model", type=str, required=True, help="Path to trained ae_best.pth")
    ap.add_argument("--manifest", type=str, default=None, help="split CSV; its rows' export is <dir>/<split>/notes.npy")
    ap.add_argument("--notes", type=str, default=None, help="notes.npy to encode (overrides --manifest)")
    ap.add_argument("--out_file", type=str, required=True, help="Output .npy file for latents (mu)")
    ap.add_argument("--config", type=str, default="config/ae_config.yaml", help="Config with LATENT_DIM, MAX_NOTES")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("melo_gan_amd has no CPU path: a MI355X (ROCm) device is required")
    cfg = load_config(args.config)
    path = args.notes
    if path is None:
        if args.manifest is None:
            raise SystemExit("give --manifest or --notes")
        split = Path(args.manifest).stem.replace("_split", "")
        path = os.path.join(os.path.dirname(args.manifest), split, "notes.npy")
    notes = torch.from_numpy(np.ascontiguousarray(np.load(path), dtype=np.float32)).cuda()
    print(f"Found {len(notes)} files to encode")
    eng = VaeEngine(cfg, "cuda", min(int(cfg.get("BATCH_SIZE", 32)), 256))
    load_model(eng, args.model)
    latents = encode(eng, notes)
    os.makedirs(os.path.dirname(os.path.abspath(args.out_file)), exist_ok=True)
    np.save(args.out_file, latents)
    print(f"Saved latents ({latents.shape}) -> {args.out_file}")


if __name__ == "__main__":
    main()
