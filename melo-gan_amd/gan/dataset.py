"""Data plane of the GAN trainer: the reference's fast-NPY contract, resident in HBM.

/root/reference/src/gan/dataset.py:30-56,165-174 -- three row-aligned arrays
<SPLITS_DIR>/<split>/{notes,emotion,numeric_features}.npy (+ optional encoder_feats.npy); an item is
(notes float32 (T,C), emotion str|int|one-hot, latent float32 (LATENT_DIM) or zeros, numeric float32 (6)).
Instead of 4 DataLoader worker processes restarted every epoch (train_gan.py:80) the whole split is
uploaded once (MI355X has 288 GB; the reference split is 897 x 512 x 4 floats = 7 MB) and batches are
device-side index_selects of a per-epoch permutation: shuffle=True, drop_last=True semantics.
The slow per-file .npz path of the reference is out of scope.

Splits that do not fit (or should not live) in HBM: `resident=False` keeps the note rolls in PINNED host memory and
streams them: a batch's rows are gathered on the host into one of two pinned staging buffers and
copied host->device on a side stream while the previous batch trains; the consumer stream waits on the copy's event.
The small per-sample arrays (numeric features, latents, labels) stay on the device either way.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .utils import check_labels, emotion_to_index


class Batch:
    """One batch of the epoch's order.  Nothing is gathered until it is used: `stage(engine)` gathers the rows straight
    into the engine's input buffers in one launch; iterating / unpacking it yields the four gathered device tensors
    (notes, numeric, latent, emot_idx) like a DataLoader batch.  A rank that skips the batch pays nothing."""

    def __init__(self, ds, idx, notes=None):
        self.ds, self.idx, self.notes = ds, idx, notes      # notes: already gathered (streamed mode)

    def stage(self, eng):
        ds = self.ds
        if self.notes is None:
            eng.set_batch(ds.notes, ds.numeric, ds.latent, ds.emot_idx, idx=self.idx)
        else:
            eng.set_batch(self.notes, ds.numeric, ds.latent, ds.emot_idx, idx=self.idx, real_idx=None)

    def tensors(self):
        ds, idx = self.ds, self.idx
        notes = self.notes if self.notes is not None else ds.notes.index_select(0, idx)
        return (notes, ds.numeric.index_select(0, idx), ds.latent.index_select(0, idx), ds.emot_idx.index_select(0, idx))

    def __iter__(self):
        return iter(self.tensors())


class GANDataset:
    def __init__(self, notes: np.ndarray, emotions, numeric: np.ndarray, latent: Optional[np.ndarray],
                 latent_dim: int, device="cuda", resident: bool = True):
        n = notes.shape[0]
        if not (len(emotions) == n and numeric.shape[0] == n):
            raise ValueError("NPY file length mismatch (notes, emotions, numeric_features)")
        if latent is not None and latent.shape[0] != n:
            print(f"[WARN] latent_feats length mismatch ({latent.shape[0]}) vs notes ({n}). Ignoring latent_feats.")
            latent = None
        self.n = n
        self.device = torch.device(device)
        self.resident = bool(resident)
        host_notes = torch.from_numpy(np.ascontiguousarray(notes, dtype=np.float32))
        self.notes = host_notes.to(device) if self.resident else host_notes.pin_memory()
        self.numeric = torch.from_numpy(np.ascontiguousarray(numeric, dtype=np.float32)).to(device)
        self.latent = (torch.from_numpy(np.ascontiguousarray(latent, dtype=np.float32)) if latent is not None
                       else torch.zeros(n, latent_dim)).to(device)
        self.emot_idx = check_labels(torch.tensor([emotion_to_index(e) for e in emotions], dtype=torch.int64),
                                     4, "GANDataset emotions").to(device)
        self._copy_stream = None

    @classmethod
    def from_split(cls, cfg: dict, split_csv: str, latent_feats_path: Optional[str] = None, device="cuda",
                   resident: Optional[bool] = None):
        """prepare_dataset of the reference trainer (train_gan.py:39-60), fast-NPY path only."""
        splits_dir = cfg.get("SPLITS_DIR", "data/splits")
        name = Path(split_csv).stem
        d = os.path.join(splits_dir, name)
        paths = {k: os.path.join(d, k + ".npy") for k in ("notes", "emotion", "numeric_features")}
        missing = [p for p in paths.values() if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(
                f"fast-NPY arrays not found ({missing}); the per-file .npz path of the reference is not "
                "implemented -- export the split to notes.npy / emotion.npy / numeric_features.npy")
        latent = np.load(latent_feats_path) if latent_feats_path and os.path.exists(latent_feats_path) else None
        notes = np.load(paths["notes"], mmap_mode="r")
        if resident is None:        # DATA_RESIDENT: true / false / absent = resident while the rolls take < 64 GiB
            resident = cfg.get("DATA_RESIDENT", notes.nbytes < (64 << 30))
        return cls(notes, np.load(paths["emotion"], allow_pickle=True),
                   np.load(paths["numeric_features"]), latent, cfg["LATENT_DIM"], device, resident)

    @classmethod
    def synthetic(cls, n: int, T: int, C: int, latent_dim: int, seed: int = 42, device="cuda", resident: bool = True):
        """SURVEY section 8(d) recipe: real ~ U(-1,1), numeric ~ N(0,1), latent = 0, labels uniform over 4."""
        g = np.random.default_rng(seed)
        return cls(g.uniform(-1, 1, (n, T, C)).astype(np.float32), g.integers(0, 4, n),
                   g.standard_normal((n, 6)).astype(np.float32), None, latent_dim, device, resident)

    def __len__(self):
        return self.n

    # ---- the resident split bound to an engine: every training step stages its own batch on the device ----
    def bind(self, eng, batch_size: int, rank: int = 0, world: int = 1) -> int:
        """Bind the resident arrays to `eng` (GanEngine.bind_batches): the step's first launch gathers batch k of the
        epoch's order itself.  Returns the batches this rank takes per epoch (drop_last; a trailing incomplete round of
        ranks is dropped so that all ranks issue the same collectives)."""
        if not self.resident:
            raise ValueError("GANDataset.bind: only an HBM-resident split can be bound (streamed splits use batches())")
        nb = self.n // batch_size
        self._bind = (int(batch_size), int(rank), int(world), (nb - nb % world) // world)
        if self._bind[3] < 1:
            raise ValueError("GANDataset.bind: the split holds less than one batch per rank")
        eng.bind_batches(self.notes, self.numeric, self.latent, self.emot_idx, order_len=self._bind[3] * batch_size)
        return self._bind[3]

    def start_epoch(self, eng, generator: Optional[torch.Generator] = None) -> int:
        """A new shuffled order (the same permutation stream batches() draws from): this rank's batches rank, rank + world,
        ... of it, written to the engine's order buffer with stream-ordered copies.  Returns the number of batches."""
        B, rank, world, nb = self._bind
        perm = torch.randperm(self.n, generator=generator)
        idx = torch.cat([perm[g * B:(g + 1) * B] for g in range(rank, nb * world, world)])
        eng.set_order(idx.to(self.device, non_blocking=True))
        return nb

    def batches(self, batch_size: int, generator: Optional[torch.Generator] = None):
        """One epoch: shuffled, drop_last (train_gan.py:80).  Yields device tensors."""
        perm_host = torch.randperm(self.n, generator=generator)
        perm = perm_host.to(self.device)
        nb = self.n // batch_size
        if self.resident:
            for i in range(nb):
                yield Batch(self, perm[i * batch_size:(i + 1) * batch_size])
            return
        # streamed: batch i+1 is gathered on the host and copied on the side stream while batch i trains
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        shape = (batch_size,) + tuple(self.notes.shape[1:])
        stage = [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        dev = [torch.empty(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
        ready = [torch.cuda.Event(), torch.cuda.Event()]          # copy i landed in dev[i % 2]
        freed = [torch.cuda.Event(), torch.cuda.Event()]          # the consumer is done with dev[i % 2]
        cur = torch.cuda.current_stream(self.device)

        def issue(i):
            s = i % 2
            ready[s].synchronize()                               # stage[s] is free once its previous copy completed
            torch.index_select(self.notes, 0, perm_host[i * batch_size:(i + 1) * batch_size], out=stage[s])
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(freed[s])
                dev[s].copy_(stage[s], non_blocking=True)
                ready[s].record(self._copy_stream)

        for s in range(2):
            ready[s].record(cur)
            freed[s].record(cur)
        if nb > 0:
            issue(0)
        for i in range(nb):
            if i + 1 < nb:
                issue(i + 1)
            s = i % 2
            cur.wait_event(ready[s])
            idx = perm[i * batch_size:(i + 1) * batch_size]
            yield Batch(self, idx, notes=dev[s])
            freed[s].record(torch.cuda.current_stream(self.device))
