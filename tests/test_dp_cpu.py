"""N>1 path on CPU: two gloo ranks, each computing the critic/generator gradients of ITS shard with the oracle,
averaged through melo_gan_amd's DataParallel wrapper, against a single-process emulation of the same
semantics (shard-local BatchNorm statistics, mean of shard gradients).  No GPU involved."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import melo_oracle as O

B_SHARD, T, C, WORLD = 2, 16, 4, 2


def shard_grads(rank):
    """Flat (D, GE) gradients of one shard's D-step and G-step from identical initial parameters."""
    cfg, ed_cfg = O.default_gan_cfg(B_SHARD, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "closed_form", d_scale=6.0)
    real, numeric, latent, emot = O.synthetic_batch(B_SHARD, T, C, cfg["LATENT_DIM"], 6, 100 + rank)
    R = O.step_randoms(B_SHARD, cfg["NOISE_DIM"], seed=200 + rank)
    dcopy = {k: v.clone() for k, v in S.PD.items()}
    rd = O.d_step(S, real, latent, numeric, R["noise_d"], R["alpha"], R["dm_d"])
    for k in S.PD:                                   # undo the local Adam step: replicas update after the all-reduce
        S.PD[k].copy_(dcopy[k])
    rg = O.g_step(S, latent, numeric, emot, R["noise_g"], R["dm_g"])
    fd = torch.cat([rd["grads"][k].flatten() for k in S.PD])
    fg = torch.cat([rg["grads"][k].flatten() for k in rg["grads"]])
    return fd, fg


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.dp import DataParallel
    torch.set_num_threads(1)
    fd, fg = shard_grads(rank)
    params = torch.full((8,), float(rank))
    NBIG = 5000                             # stand-in for decoder.pre.2's weight + bias slice of the flat gradient
    eng = types.SimpleNamespace(D=types.SimpleNamespace(grad=fd.clone(), data=params.clone()),
                                GE=types.SimpleNamespace(grad=fg.clone(), data=params.clone() + 1), world_size=1,
                                p2_grad_slice=lambda: (0, NBIG))
    eng.d_p2, eng.a_p0 = torch.full((3, 4), float(rank + 1)), torch.full((3, 2), float(10 * (rank + 1)))

    def enable_p2_gather(world):
        eng.d_p2_all, eng.a_p0_all = torch.zeros(world * 3, 4), torch.zeros(world * 3, 2)
    eng.enable_p2_gather = enable_p2_gather
    os.environ["MELO_DP_MODE"] = "allreduce"
    dp = DataParallel(eng, WORLD, dist)
    dp.broadcast_params()
    assert eng.world_size == WORLD
    assert torch.equal(eng.D.data, torch.zeros(8)) and torch.equal(eng.GE.data, torch.ones(8))   # rank 0's values
    dp.allreduce_d(async_op=True)           # asynchronous form: complete after _wait()
    dp._wait()
    dp.allreduce_g()
    if rank == 0:
        torch.save({"d": eng.D.grad / WORLD, "g": eng.GE.grad / WORLD}, out)

    # DataParallel.step(): the full step's launch order in each mode.  The fake engine writes a sub-step's gradients
    # when that sub-step "runs" and checks what must already be reduced / gathered at each point.
    d_sum, g_sum = fd.clone(), fg.clone()
    dist.all_reduce(d_sum)
    dist.all_reduce(g_sum)
    log = []

    def run(name, use_graph):
        log.append(name)
        if name in ("d_backward_rng", "dg_forward_d_backward_rng"):
            eng.D.grad.copy_(fd)
        elif name == "g_ed_branch":                  # may run while the critic's all-reduce is in flight: touches nothing of it
            pass
        elif name in ("d_update", "d_update_g_critic_chain"):     # the critic's all-reduce must have completed
            assert torch.equal(eng.D.grad, d_sum), "critic update before its all-reduce finished"
        elif name == "g_backward_b":
            eng.GE.grad.copy_(fg)
        elif name in ("g_p2_wgrad", "g_backward_p2b"):   # every rank's factors, in rank order
            want = torch.cat([torch.full((3, 4), float(r + 1)) for r in range(WORLD)])
            assert torch.equal(eng.d_p2_all, want) and torch.equal(eng.a_p0_all[:, 0], 10 * want[:, 0])
            if name == "g_backward_p2b":
                eng.GE.grad.copy_(fg)
            eng.GE.grad[:NBIG].copy_(g_sum[:NBIG])   # what the engine computes from them: the global-batch gradient
        elif name == "g_update":
            assert torch.equal(eng.GE.grad, g_sum), "generator update with a partially reduced gradient"
    eng.run = run
    head = ["dg_forward_d_backward_rng", "g_ed_branch", "d_update_g_critic_chain"]
    orders = {"overlap": head + ["g_backward_b", "g_p2_wgrad", "g_update"], "gather": head + ["g_backward_p2b", "g_update"],
              "allreduce": ["dg_forward_d_backward_rng", "g_ed_branch", "d_update_g_critic_chain", "g_backward_b", "g_update"]}
    for mode, order in orders.items():
        os.environ["MELO_DP_MODE"] = mode
        os.environ["MELO_DP_MODE"] = "auto"
        assert DataParallel(eng, 8, dist).mode == "overlap" and DataParallel(eng, WORLD, dist).mode == "gather"
        os.environ["MELO_DP_MODE"] = mode
        dp = DataParallel(eng, WORLD, dist)
        assert dp.mode == mode
        for _ in range(2):
            log.clear()
            dp.step(True, g_step=True)
            assert log == order, (mode, log)
            log.clear()
            dp.step(True, g_step=False)
            assert log == ["d_backward_rng", "d_update"], (mode, log)
    os.environ.pop("MELO_DP_MODE")

    # ---- the ingraph order (round 3): the collectives are issued from INSIDE the engine's sub-steps through engine.coll, so
    # the step is the single-GPU flow (one sub-step per batch).  Here over gloo with the torch.distributed adapter; on GPUs
    # the same calls go to a private RCCL communicator and are captured into the step's graphs (tests/test_dp_gpu.py). ----
    from melo_gan_amd.gan.rccl import TorchDistComm
    e2 = types.SimpleNamespace(D=types.SimpleNamespace(grad=fd.clone(), data=params.clone()),
                               GE=types.SimpleNamespace(grad=fg.clone(), data=params.clone() + 1), world_size=1, coll=None,
                               p2_world=0, p2_grad_slice=lambda: (0, NBIG))
    e2.d_p2, e2.a_p0 = torch.full((3, 4), float(rank + 1)), torch.full((3, 2), float(10 * (rank + 1)))

    def enable2(world):
        e2.p2_world = world
        e2.d_p2_all, e2.a_p0_all = torch.zeros(world * 3, 4), torch.zeros(world * 3, 2)
    e2.enable_p2_gather = enable2
    want = torch.cat([torch.full((3, 4), float(r + 1)) for r in range(WORLD)])
    seen = []

    def run2(name, use_graph):
        seen.append(name)
        e2.D.grad.copy_(fd)
        if name == "d_step_rng":
            e2.coll.reduce_d(e2, False)
            assert torch.equal(e2.D.grad, d_sum)
            return
        assert name == "dg_step_rng"
        e2.a_p0_all.zero_(); e2.d_p2_all.zero_()
        e2.coll.reduce_d(e2, True)                       # C1: the critic's gradient + pre.2's input factor
        assert torch.equal(e2.D.grad, d_sum) and torch.equal(e2.a_p0_all[:, 0], 10 * want[:, 0])
        e2.coll.gather_p2(e2, False)                     # C2
        assert torch.equal(e2.d_p2_all, want)
        e2.GE.grad.copy_(fg)
        e2.GE.grad[:NBIG].copy_(g_sum[:NBIG])            # pre.2's global gradient, computed from the gathered factors
        e2.coll.reduce_g(e2)                             # C3: the rest
        assert torch.equal(e2.GE.grad, g_sum)
    e2.run = run2
    dp2 = DataParallel(e2, WORLD, dist, comm=TorchDistComm(dist))
    assert dp2.mode == "ingraph" and dp2.active and e2.coll is not None and e2.p2_world == WORLD and e2.world_size == WORLD
    dp2.prepare(True)                                    # nothing to prepare: no capture-ahead constraint
    dp2.step(True, g_step=True)
    dp2.step(True, g_step=False)
    assert seen == ["dg_step_rng", "d_step_rng"], seen
    os.environ["MELO_DP_MODE"] = "ingraph"               # asked for explicitly but no communicator can exist over gloo
    try:
        DataParallel(e2, WORLD, dist)
        raise AssertionError("ingraph without a communicator must be refused")
    except ValueError:
        pass
    os.environ.pop("MELO_DP_MODE")
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_average_matches_emulation(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "avg.pt")
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    got = torch.load(out)
    torch.set_num_threads(1)
    ds, gs = zip(*[shard_grads(r) for r in range(WORLD)])
    torch.testing.assert_close(got["d"], sum(ds) / WORLD, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(got["g"], sum(gs) / WORLD, rtol=1e-6, atol=1e-9)


def test_critic_gradient_is_shardable():
    """Every critic loss term is a mean of per-sample terms: with the SAME fake batch, the mean
    of two half-batch gradients equals the full-batch gradient (the critic has no BatchNorm)."""
    cfg, ed_cfg = O.default_gan_cfg(4, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "closed_form", d_scale=6.0)
    real, numeric, _, _ = O.synthetic_batch(4, T, C, cfg["LATENT_DIM"], 6, 5)
    fake = O.closed_form((4, T, C), 3.3, 0.5)
    alpha = torch.rand(4, 1, 1, generator=torch.Generator().manual_seed(1))
    emb = O.feature_encoder_fwd(S.PE, numeric, None).detach()

    def grads(sl):
        P = {k: v.clone().requires_grad_(True) for k, v in S.PD.items()}
        loss = (O.discriminator_fwd(P, fake[sl], emb[sl]).mean() - O.discriminator_fwd(P, real[sl], emb[sl]).mean()
                + 10.0 * O.gradient_penalty(P, real[sl], fake[sl], emb[sl], alpha[sl]))
        return torch.autograd.grad(loss, list(P.values()))
    full, a, b = grads(slice(0, 4)), grads(slice(0, 2)), grads(slice(2, 4))
    for f, x, y in zip(full, a, b):
        torch.testing.assert_close((x + y) / 2, f, rtol=1e-4, atol=1e-7)
