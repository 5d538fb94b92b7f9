#!/usr/bin/env python3
"""WGAN-GP trainer with numeric-feature conditioning and a frozen emotion discriminator -- the
MI355X-native counterpart of /root/reference/src/gan/train_gan.py (same CLI, same YAML keys and
defaults, same checkpoint layout):

    python -m melo_gan_amd.gan.train_gan --config config/gan_config.yaml \
        --ed_config config/ed_config.yaml --ed_ckpt data/models/ed/ed_best.pth

Differences that do not change results: the epoch's batches come from an HBM-resident copy of the split
(no DataLoader workers), every D-step / G-step is a replayed hipGraph over libmelogan_hip, and the three
per-epoch scalars are accumulated on the device and read back once per epoch (the reference calls .item()
three times per batch, train_gan.py:205,250-251).  Extra flags (--synthetic, --epochs, --no-graph) exist
for smoke runs without the (git-ignored) dataset.

Data parallel (not in the reference, which is single-GPU): launched under torch.distributed.run
(`--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`) every rank holds a replica, takes every N-th batch of the
epoch's (identically shuffled) order and averages gradients through melo_gan_amd.gan.dp.DataParallel (RCCL over
xGMI); rank 0 logs and checkpoints.
"""
import argparse
import json
import os
import time

import torch

from . import config as C
from .dataset import GANDataset
from .dp import DataParallel
from .engine import GanEngine
from .utils import seed_everything


def _scalar_writer(log_dir):
    os.makedirs(log_dir, exist_ok=True)
    try:                                    # same tags as the reference when tensorboard is installed
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        class _Jsonl:
            def __init__(self, d):
                self.f = open(os.path.join(d, "scalars.jsonl"), "a")

            def add_scalar(self, tag, value, step):
                self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "wall_time": time.time()}) + "\n")
                self.f.flush()

            def close(self):
                self.f.close()
        return _Jsonl(log_dir)


def load_ed_checkpoint(eng: GanEngine, path: str):
    """train_gan.py:121-128: load_state_dict(strict=False); missing file => random ED with a warning."""
    if not os.path.exists(path):
        print(f"[WARN] ED checkpoint not found at {path}. ED will be random!")
        return False
    print(f"[INFO] Loading pre-trained Emotion Discriminator from {path}")
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["model"] if "model" in ckpt else ckpt
    # strict=False semantics: a key the checkpoint lacks keeps its initial value (reported, not silent -- a spectral-norm
    # triple with one part missing counts as missing); a key of the wrong shape raises, as load_state_dict does
    missing, bad = [], []
    for k in eng.ED.spec:
        w = spectral_norm_weight(sd, k)
        if w is None:
            missing.append(k)
        elif tuple(w.shape) != tuple(eng.ED.spec[k]):
            bad.append(f"{k}: checkpoint {tuple(w.shape)} vs model {tuple(eng.ED.spec[k])}")
        else:
            eng.ED.p[k].copy_(w.float())
    for k in eng.EDbuf:
        if k not in sd:
            missing.append(k)
        elif tuple(sd[k].shape) != tuple(eng.EDbuf[k].shape):
            bad.append(f"{k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(eng.EDbuf[k].shape)}")
        else:
            eng.EDbuf[k].copy_(sd[k].float())
    if bad:
        raise RuntimeError("Error(s) in loading state_dict for EmotionDiscriminator: size mismatch for " + "; ".join(bad))
    if missing:
        print(f"[WARN] ED checkpoint {path} lacks {len(missing)} key(s), left at their initial values: {', '.join(missing)}")
    eng.fold_ed()          # derived scale / shift / re-laid weights: eagerly, outside any graph
    return True


def spectral_norm_weight(sd: dict, key: str):
    """sd[key], or -- for a layer the reference wrapped in nn.utils.spectral_norm (`use_spectral_norm`, ed_model.py:29-32,
    79-82: state_dict keys `<key>_orig`, `<key>_u`, `<key>_v`) -- the weight that wrapper uses in eval mode:
    weight_orig / sigma with sigma = u^T W v over W = weight_orig flattened to (out, -1), no power iteration.  The GAN step
    only ever runs the emotion discriminator frozen and in eval mode, so folding sigma in at load time is exact."""
    if key in sd:
        return sd[key]
    if key + "_orig" in sd and key + "_u" in sd and key + "_v" in sd:
        w = sd[key + "_orig"].double()
        sigma = torch.dot(sd[key + "_u"].double(), w.flatten(1).mv(sd[key + "_v"].double()))
        return (w / sigma).float()
    return None


def save_checkpoint(eng: GanEngine, path: str, epoch=None, full=True):
    """train_gan.py:267-282: {'epoch','G','D','E_num','opt_G','opt_D'} every SAVE_FREQ epochs; final {'G','E_num'}."""
    sd = eng.state_dicts()
    if not full:
        torch.save({"G": sd["G"], "E_num": sd["E_num"]}, path)
        return
    betas = tuple(eng.betas)
    torch.save({"epoch": epoch, "G": sd["G"], "D": sd["D"], "E_num": sd["E_num"],
                "opt_G": adam_state_dict(eng.GE, eng.lr_g, betas), "opt_D": adam_state_dict(eng.D, eng.lr_d, betas)}, path)


def adam_state_dict(fp, lr: float, betas, eps: float = 1e-8, weight_decay: float = 0.0) -> dict:
    """The flat optimiser state as `torch.optim.Adam.state_dict()` of the reference's optimisers (train_gan.py:136-145:
    opt_G over list(G.parameters()) + list(E_num.parameters()), opt_D over D.parameters()): parameter i of the optimiser is
    entry i of the flat buffer's spec (module definition order, the generator's tensors before the numeric encoder's), with
    its own `step`, `exp_avg`, `exp_avg_sq`.  `optimizer.load_state_dict()` of a reference trainer accepts it as is."""
    step = float(fp.state[0].item())
    state = {}
    for i, k in enumerate(fp.spec):
        off, n = fp.offsets[k]
        state[i] = {"step": torch.tensor(step, dtype=torch.float32),
                    "exp_avg": fp.m[off:off + n].view(fp.spec[k]).detach().cpu().clone(),
                    "exp_avg_sq": fp.v[off:off + n].view(fp.spec[k]).detach().cpu().clone()}
    group = {"lr": float(lr), "betas": tuple(float(b) for b in betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(fp.spec)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(fp, sd: dict):
    """Inverse of adam_state_dict (resume): fills the flat moment buffers, the step counter and the running beta powers
    the update kernel keeps beside it (state = [step, beta1^step, beta2^step, -])."""
    state = sd["state"]
    if not state:                       # an optimiser that never stepped: fresh moments
        fp.m.zero_(); fp.v.zero_(); fp.state.zero_()
        return
    if len(state) != len(fp.spec):
        raise ValueError(f"optimizer state_dict has {len(state)} parameter entries, this optimiser has {len(fp.spec)}")
    steps = set()
    for i, k in enumerate(fp.spec):
        if i not in state:
            raise ValueError(f"optimizer state_dict lacks entry {i} ({k})")
        st = state[i]
        for part in ("exp_avg", "exp_avg_sq"):
            if tuple(st[part].shape) != tuple(fp.spec[k]):
                raise ValueError(f"optimizer state_dict entry {i} ({k}).{part}: shape {tuple(st[part].shape)} != {tuple(fp.spec[k])}")
        steps.add(float(st["step"]))
    if len(steps) != 1:
        raise ValueError(f"optimizer state_dict: per-parameter step counts differ ({sorted(steps)}); the flat optimiser keeps one")
    for i, k in enumerate(fp.spec):
        off, n = fp.offsets[k]
        fp.m[off:off + n].copy_(state[i]["exp_avg"].reshape(-1).to(fp.m.device))
        fp.v[off:off + n].copy_(state[i]["exp_avg_sq"].reshape(-1).to(fp.v.device))
    step = steps.pop()
    b1, b2 = sd["param_groups"][0]["betas"]
    fp.state[0], fp.state[1], fp.state[2] = step, float(b1) ** step, float(b2) ** step


def resume_checkpoint(eng: GanEngine, path: str) -> int:
    """Continue from a gan_epochNNNN.pth written by save_checkpoint (or by the reference trainer, train_gan.py:267-276):
    G (with its BatchNorm buffers), D, E_num, both optimisers.  Returns the epoch the checkpoint was written after."""
    ck = torch.load(path, map_location="cpu")
    for key in ("G", "D", "E_num", "opt_G", "opt_D"):
        if key not in ck:
            raise KeyError(f"{path}: not a full training checkpoint (no '{key}')")
    eng.GE.load({**{"G." + k: v for k, v in ck["G"].items()}, **{"E." + k: v for k, v in ck["E_num"].items()}})
    eng.D.load(ck["D"])
    for k in eng.Gbuf:
        eng.Gbuf[k].copy_(ck["G"][k].float())
    nbt = [v for k, v in ck["G"].items() if k.endswith("num_batches_tracked")]
    if nbt:
        eng.num_batches_tracked = int(nbt[0])
    load_adam_state_dict(eng.GE, ck["opt_G"])
    load_adam_state_dict(eng.D, ck["opt_D"])
    eng.params_changed()
    return int(ck.get("epoch") or 0)


def train(cfg: dict, ed_cfg: dict, ed_ckpt: str, synthetic: int = 0, use_graph: bool = True, resume: str = None):
    cfg = C.with_gan_defaults(cfg, require=not synthetic)
    seed_everything(cfg.get("SEED", 42))
    if not torch.cuda.is_available():
        raise RuntimeError("melo_gan_amd has no CPU path: a MI355X (ROCm) device is required")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("MELO_SHARE_GPU") == "1":     # rehearsal on a single-GPU box: ranks share the device
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("MELO_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    device = torch.device("cuda", torch.cuda.current_device())
    log = print if rank == 0 else (lambda *a, **k: None)
    log(f"Using main device: {device}" + (f" ({world} data-parallel ranks)" if world > 1 else ""))
    B = cfg.get("BATCH_SIZE", 32)
    if synthetic:
        ds = GANDataset.synthetic(synthetic, cfg["MAX_NOTES"], cfg["NOTE_DIM"], cfg["LATENT_DIM"], cfg.get("SEED", 42), device)
    else:
        ds = GANDataset.from_split(cfg, cfg["TRAIN_SPLIT"], cfg.get("ENCODER_FEATS_TRAIN"), device)
    log(f"Train set size: {len(ds)}")
    # ED_DTYPE: bf16 = the secondary configuration (frozen emotion discriminator stored in bf16); the
    # reference has no such key and the default is its fp32 arithmetic
    eng = GanEngine(cfg, ed_cfg, device, B, ed_dtype=str(cfg.get("ED_DTYPE", "fp32")))
    eng.init_weights(cfg.get("SEED", 42))
    load_ed_checkpoint(eng, ed_ckpt)
    start_epoch = resume_checkpoint(eng, resume) if resume else 0
    if resume:
        log(f"[INFO] Resumed from {resume} (epoch {start_epoch})")
    dp = DataParallel(eng, world, dist)
    dp.broadcast_params()
    eng.seed(cfg.get("SEED", 42) + rank)            # rank-offset Philox key: every shard draws its own noise
    writer = _scalar_writer(cfg.get("LOG_DIR", "experiments/gan/logs")) if rank == 0 else None
    os.makedirs(cfg.get("CHECKPOINT_DIR", "experiments/gan/checkpoints"), exist_ok=True)
    os.makedirs(cfg.get("SAMPLE_DIR", "experiments/gan/samples"), exist_ok=True)
    critic_iters = cfg.get("CRITIC_ITERS", 5)
    sums = torch.zeros(3, device=device)        # sum loss_d, sum g_adv, sum g_emo (device-side accumulation)
    shuffle_gen = torch.Generator().manual_seed(cfg.get("SEED", 42))
    log("Starting WGAN-GP Training with Emotion Guidance...")

    def epoch_batches():
        """Per batch of this rank's share of the epoch: stage it (or, for a bound resident split, nothing: the step's first
        launch gathers batch k of the epoch's order on the device) and yield its index."""
        if ds.resident:
            for k in range(ds.start_epoch(eng, shuffle_gen)):
                yield k
            return
        # every rank walks the same shuffled order and takes the batches rank, rank + world, ...; a trailing
        # incomplete round is dropped so that all ranks issue the same collectives
        usable = len(ds) // B - (len(ds) // B) % world
        for gi, batch in enumerate(ds.batches(B, shuffle_gen)):
            if gi >= usable:
                break
            if gi % world != rank:
                continue
            batch.stage(eng)           # one gather launch from the streamed rolls into the engine's buffers
            yield gi // world

    with torch.cuda.stream(eng.stream):
        if ds.resident:
            ds.bind(eng, B, rank, world)
        for epoch in range(start_epoch + 1, cfg["EPOCHS"] + 1):
            sums.zero_()
            steps = 0
            for batch_idx in epoch_batches():
                g_step = (batch_idx + 1) % critic_iters == 0
                dp.step(use_graph, g_step)
                sums[0:1] += eng.loss_d_out[0:1]
                if g_step:
                    sums[1:2] += eng.adv
                    sums[2:3] += eng.emo
                steps += 1
            s = sums.tolist()                                    # the epoch's only device->host sync
            g_steps = max(1, steps // critic_iters)
            steps = max(1, steps)
            log(f"Epoch {epoch}/{cfg['EPOCHS']} | D_loss: {s[0] / steps:.4f} | G_adv: {s[1] / g_steps:.4f} | "
                  f"G_emo: {s[2] / g_steps:.4f}")
            if rank != 0:
                continue
            writer.add_scalar("Loss/Critic", s[0] / steps, epoch)
            writer.add_scalar("Loss/Generator_Adv", s[1] / g_steps, epoch)
            writer.add_scalar("Loss/Generator_Emo", s[2] / g_steps, epoch)
            if epoch % cfg.get("SAVE_FREQ", 5) == 0:
                save_checkpoint(eng, os.path.join(cfg["CHECKPOINT_DIR"], f"gan_epoch{epoch:04d}.pth"), epoch, full=True)
    if rank == 0:
        save_checkpoint(eng, os.path.join(cfg["CHECKPOINT_DIR"], "gan_final.pth"), full=False)
        writer.close()
    if world > 1:
        dist.barrier(async_op=True).wait()
        dist.destroy_process_group()
    log("Training Complete.")
    return eng


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="config/gan_config.yaml", help="Path to the main GAN config")
    parser.add_argument("--ed_config", type=str, default="config/ed_config.yaml", help="Path to the ED config")
    parser.add_argument("--ed_ckpt", type=str, default="data/models/ed/ed_best.pth")
    parser.add_argument("--synthetic", type=int, default=0, help="train on N synthetic rolls instead of TRAIN_SPLIT")
    parser.add_argument("--epochs", type=int, default=None, help="override EPOCHS")
    parser.add_argument("--no-graph", action="store_true")
    parser.add_argument("--resume", type=str, default=None, help="gan_epochNNNN.pth to continue from (G, D, E_num, opt_G, opt_D)")
    args = parser.parse_args(argv)
    cfg = C.load_config(args.config)
    ed_cfg = C.load_config(args.ed_config)
    if args.epochs is not None:
        cfg["EPOCHS"] = args.epochs
    train(cfg, ed_cfg, args.ed_ckpt, args.synthetic, not args.no_graph, args.resume)


if __name__ == "__main__":
    main()
