#!/bin/bash
# 2 ranks sharing one GPU over gloo: one epoch of the trainer CLI on synthetic rolls in MELO_DP_MODE=overlap, =gather
# and =allreduce from the same seed; prints the largest relative difference of the final generator tensors.
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/dp_equiv}
mkdir -p $OUT
for m in overlap gather allreduce; do
python3 - <<PY
import yaml
c = yaml.safe_load(open("config/gan_config.yaml"))
c.update(EPOCHS=1, BATCH_SIZE=8, MAX_NOTES=64, SAVE_FREQ=1, CRITIC_ITERS=2, CHECKPOINT_DIR="$OUT/$m/ck", LOG_DIR="$OUT/$m/log",
         SAMPLE_DIR="$OUT/$m/s")
yaml.safe_dump(c, open("$OUT/$m.yaml", "w"))
PY
MELO_DP_MODE=$m MELO_SHARE_GPU=1 MELO_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 \
  --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2954$((RANDOM % 10)) -m melo_gan_amd.gan.train_gan \
  --config $OUT/$m.yaml --ed_config config/ed_config.yaml --ed_ckpt $OUT/none.pth --synthetic 64 > $OUT/$m.log 2>&1
tail -2 $OUT/$m.log
done
python3 - <<PY
import torch
b = torch.load("$OUT/allreduce/ck/gan_final.pth", map_location="cpu")
worst = 0.0
for mode in ("overlap", "gather"):
    a = torch.load(f"$OUT/{mode}/ck/gan_final.pth", map_location="cpu")
    for part in ("G", "E_num"):
        for k in a[part]:
            x, y = a[part][k].double(), b[part][k].double()
            if k in ("decoder.deconv.0.bias", "decoder.deconv.3.bias"):
                continue        # biases in front of a train-mode BatchNorm: zero gradient, Adam-amplified rounding noise
            if x.numel() and y.abs().max() > 0:
                worst = max(worst, float((x - y).norm() / y.norm()))
print("largest relative difference gather vs allreduce:", worst)
assert worst < 1e-5
PY
