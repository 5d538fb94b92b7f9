"""Data-parallel "gather" mode on the GPU engine (melo-gan_amd/gan/dp.py): decoder.pre.2.weight's gradient computed from
the all-gathered per-sample factors equals the sum of the shards' own weight gradients -- what an all-reduce delivers.
Two engines in one process stand in for two ranks; the collectives are done by hand."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import melo_oracle as O  # noqa: E402


def rel_err(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_factor_gather_equals_summed_shard_gradients():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.engine import GanEngine
    B, T, C, WORLD = 4, 32, 4, 2
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    engs = [GanEngine(cfg, ed_cfg, "cuda", B) for _ in range(WORLD)]
    inputs = []
    for r, e in enumerate(engs):
        e.init_weights(seed=3)                       # identical replicas
        g = torch.Generator().manual_seed(50 + r)    # each rank its own shard and its own draws
        real, numeric, latent, emot = O.synthetic_batch(B, T, C, cfg["LATENT_DIM"], 6, 70 + r)
        noise = torch.randn(tuple(e.noise.shape), generator=g)
        masks = [(torch.rand(tuple(m.shape), generator=g) < 0.8).float() for m in e.dmask]
        inputs.append((real.cuda(), numeric.cuda(), latent.cuda(), emot.cuda(), noise.cuda(), [m.cuda() for m in masks]))
    assert torch.equal(engs[0].GE.data, engs[1].GE.data)

    def feed(e, x):
        e.set_batch(*x[:4])
        e.set_randoms(x[4], x[5])

    # the shards' own gradients (single-GPU path), summed = what all-reduce(SUM) hands every rank
    ref = []
    for e, x in zip(engs, inputs):
        feed(e, x)
        e.g_backward()
        ref.append(e.GE.grad.clone())
    want = ref[0] + ref[1]
    off, n = engs[0].p2_grad_slice()                    # pre.2's weight and bias: adjacent, first in the flat buffer
    assert off == 0 and n == 256 * engs[0].red * 512 + 256 * engs[0].red

    # gather mode
    for e, x in zip(engs, inputs):
        e.enable_p2_gather(WORLD)
        e.GE.grad.fill_(float("nan"))
        feed(e, x)
        e.g_backward_a()
    d_all = torch.cat([e.d_p2 for e in engs])        # all_gather_into_tensor: rank order
    a_all = torch.cat([e.a_p0 for e in engs])
    for e in engs:
        e.d_p2_all.copy_(d_all)
        e.a_p0_all.copy_(a_all)
        e.g_backward_p2b()
    assert not torch.isnan(engs[0].GE.grad).any()
    assert torch.equal(engs[0].GE.grad[:n], engs[1].GE.grad[:n])            # every rank holds the same global gradient
    assert rel_err(engs[0].GE.grad[:n], want[:n]) < 2e-6
    rest = engs[0].GE.grad[n:] + engs[1].GE.grad[n:]                         # the remaining all-reduce
    assert rel_err(rest, want[n:]) < 2e-6
    o, m = engs[0].GE.offsets["G.decoder.pre.2.bias"]
    assert o + m == n and rel_err(engs[0].GE.grad[o:o + m], want[o:o + m]) < 2e-6   # bias: column sums of the gathered d_p2


@pytest.mark.timeout(600)
def test_two_rank_trainer_gather_equals_allreduce(tmp_path):
    """The trainer CLI under torch.distributed.run with 2 ranks (sharing this GPU, gloo): one epoch in MELO_DP_MODE=gather
    and in =allreduce from the same seed ends in the same generator / encoder parameters (tools/dp_mode_equivalence.sh)."""
    import os
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run(["bash", os.path.join(root, "tools", "dp_mode_equivalence.sh"), str(tmp_path)], cwd=root,
                       capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "largest relative difference gather vs allreduce" in r.stdout


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["gather", "allreduce", "overlap"])
def test_step_orders_run_on_a_one_rank_rccl_group(mode):
    """bench.py with MELO_FORCE_DP=1: a 1-rank RCCL ("nccl") process group and the N > 1 step order on this GPU, through
    warm-up, graph capture and replay.  Guards the interplay of on-stream collectives, the process group's watchdog
    thread and hipGraph capture (an abort in the first version of this path, ops._quiesce_process_group)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, MELO_FORCE_DP="1", MELO_DP_MODE=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "4", "--no-cpu-baseline",
                        "--profile-steps", "0"], cwd=root, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    for v in line["losses"].values():
        assert v == v and abs(v) < 1e6
