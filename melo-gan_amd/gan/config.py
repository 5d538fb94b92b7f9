"""YAML configuration with the reference's key names and code defaults.

gan_config.yaml keys and their `cfg.get` defaults: /root/reference/src/gan/train_gan.py:41-60,
74-155,267; ed_config.yaml keys: src/emotion_discriminator/ed_model.py:118-143.
"""
from __future__ import annotations

import yaml

# code defaults of the reference trainer (the YAML file may override any of them)
GAN_DEFAULTS = dict(SEED=42, DEVICE="cuda", BATCH_SIZE=32, NUMERIC_INPUT_DIM=6, ENCODER_OUT_DIM=128,
                    ENCODER_HIDDEN=[256, 128], INTEGRATION_MODE="conditioning", BETA1=0.5, BETA2=0.9,
                    LAMBDA_GP=10.0, LAMBDA_EMOTION=1.0, CRITIC_ITERS=5, SAVE_FREQ=5, SPLITS_DIR="data/splits",
                    PROCESSED_DIR="data/processed")
GAN_REQUIRED = ("TRAIN_SPLIT", "NOISE_DIM", "LATENT_DIM", "MAX_NOTES", "NOTE_DIM", "LR_G", "LR_D", "LOG_DIR",
                "CHECKPOINT_DIR", "SAMPLE_DIR", "EPOCHS")


def load_config(path: str) -> dict:
    """train_gan.py:35-37."""
    with open(path) as f:
        return yaml.safe_load(f)


def with_gan_defaults(cfg: dict, require: bool = True) -> dict:
    out = dict(GAN_DEFAULTS)
    out.update(cfg)
    if require:
        missing = [k for k in GAN_REQUIRED if k not in out]
        if missing:
            raise KeyError(f"gan config is missing required keys {missing} (the reference indexes cfg[...] for them)")
    return out


def default_gan_cfg(B: int, T: int, C: int) -> dict:
    """config/gan_config.yaml values that reach the hot path, at a chosen (B, T, C)."""
    return dict(BATCH_SIZE=B, NUMERIC_INPUT_DIM=6, ENCODER_OUT_DIM=128, ENCODER_HIDDEN=[256, 128], NOISE_DIM=128,
                LATENT_DIM=64, INTEGRATION_MODE="warm_start", MAX_NOTES=T, NOTE_DIM=C, LR_G=2e-4, LR_D=1e-4,
                BETA1=0.5, BETA2=0.9, LAMBDA_GP=10.0, LAMBDA_EMOTION=5.0, CRITIC_ITERS=5, SEED=42)


def default_ed_cfg(note_dim: int) -> dict:
    """config/ed_config.yaml keys read by EmotionDiscriminator."""
    return dict(input_mode="notes", note_dim=note_dim, notes_hidden=256, notes_blocks=4, mlp_hidden=[256, 128],
                n_classes=4, dropout=0.2, use_spectral_norm=False, latent_dim=64)
