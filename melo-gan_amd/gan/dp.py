"""Data-parallel wrapper: one process per GPU, identical replicas, per-rank batch shard.

Every loss of the hot path is a mean over per-sample terms, so averaging the
ranks' flat gradient buffers equals the global-batch gradient: all-reduce(SUM) of the flat fp32
gradient buffers (critic 1.25 MB every batch; generator + numeric encoder 18.8 MB on generator
steps), the 1/world factor folded into the fused Adam launch (grad_scale).  On ROCm the "nccl"
backend is RCCL over xGMI.

Three step orders (MELO_DP_MODE, default "gather"):
  gather     decoder.pre.2.weight -- 16.8 of the generator's 18.8 MB -- is never all-reduced: its gradient is
             d_p2^T a_p0, so the ranks all-gather those two per-sample factors (2.2 MB per rank) and each computes
             the global batch's weight gradient itself (GanEngine.enable_p2_gather).  Per generator step: all-reduce
             critic (1.25 MB), all-gather 2 MB + 128 KB per rank, all-reduce the remaining 2 MB.  Bytes received
             per rank at N = 2 / 4 / 8: 4.4 / 9 / 18 MB instead of 20 / 30 / 35 MB.  No overlap with compute.
  allreduce  one all-reduce per optimiser, no overlap.
  overlap    the critic's all-reduce beside the generator forward and pre.2's slice beside the rest of backward.
             Measured on one MI355X with a 1-rank RCCL group (bench.py, MELO_FORCE_DP=1): a collective left in
             flight across a hipGraph launch costs ~85 us each -- 1.53 ms/step against 1.35 for "allreduce" and
             1.32 without collectives -- more than the transfer it can hide, hence not the default.

BatchNorm semantics under sharding (decision, SURVEY hard part 4): the generator's train-mode
BatchNorm uses the LOCAL shard's statistics (DDP-style), i.e. each rank runs exactly the
reference's B=64 model; running statistics are rank-local and rank 0's are checkpointed.
The wrapper is backend-agnostic (works on CPU tensors with gloo) so that the N>1 logic is
testable without GPUs.
"""
from __future__ import annotations

import os

class DataParallel:
    def __init__(self, engine, world_size: int, dist=None, group=None, force_collectives: bool = False):
        """force_collectives: issue the collectives (and take the overlapped step order) even at world_size 1 -- a
        rehearsal of the N > 1 control path on a single-GPU box."""
        self.engine, self.world, self.dist, self.group = engine, int(world_size), dist, group
        self.active = dist is not None and (self.world > 1 or force_collectives)
        self._dry = os.environ.get("MELO_DP_DRY") == "1"      # rehearsal: the N > 1 step order without the collectives
        self.mode = os.environ.get("MELO_DP_MODE", "gather")
        if self.mode not in ("gather", "allreduce", "overlap"):
            raise ValueError(f"MELO_DP_MODE={self.mode}: expected gather | allreduce | overlap")
        if self.mode == "gather" and not hasattr(engine, "enable_p2_gather"):
            self.mode = "allreduce"
        if self.active and self.mode == "gather":
            engine.enable_p2_gather(self.world)
        if self.active and self.mode == "overlap":
            engine.p2_in_a2 = True       # pre.2's weight gradient must exist when g_backward_a2 ends
        engine.world_size = self.world
        self._pending = []

    def _flat_state(self):
        e = self.engine
        bufs = [e.D.data, e.GE.data]
        if getattr(e, "ED", None) is not None:
            bufs.append(e.ED.data)
        for d in (getattr(e, "Gbuf", {}), getattr(e, "EDbuf", {})):
            bufs.extend(d.values())
        return bufs

    def broadcast_params(self, src: int = 0):
        """Once at start: every replica gets rank `src`'s parameters and buffers."""
        if not self.active:
            return
        bufs = self._flat_state()
        for t in bufs:
            self.dist.broadcast(t, src=src, group=self.group, async_op=True).wait()
        if bufs and bufs[0].is_cuda:
            # the broadcasts are ordered on the launching stream only; the step's graphs replay on the engine's stream
            import torch
            torch.cuda.synchronize()

    def _allreduce(self, flat):
        if not self.active or self._dry:
            return
        # A synchronous collective runs ON the calling (engine) stream: no cross-stream hop (an async_op + wait pair
        # cost ~45 us more per collective on one MI355X with a 1-rank RCCL group).  Its completion event is recorded
        # on that stream and polled by the process group's watchdog thread, which must not coincide with a graph
        # capture on the same stream: ops.Graph.begin() drains the device and the watchdog first.
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)

    def allreduce_d(self):
        self._allreduce(self.engine.D.grad)

    def allreduce_g(self):
        self._allreduce(self.engine.GE.grad)

    # ---- overlapped variant for the generator step --------------------------------------------------------
    # decoder.pre.2.weight is 89 % of the G+E_num gradient bytes and is final ~100 us before the end of backward
    # (SURVEY hard part 5).  start_g_big() launches its all-reduce asynchronously (RCCL runs on its own stream,
    # ordered after the launching stream) while the caller enqueues the rest of backward; finish_g() all-reduces
    # the two small remaining slices and waits for the big one.
    def start_g_big(self):
        if not self.active or self._dry:
            return
        off, n = self.engine.big_grad_slice()
        self._pending.append(self.dist.all_reduce(self.engine.GE.grad[off:off + n], op=self.dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True))

    def finish_g(self):
        """All-reduces what start_g_big() left (one contiguous range: the engine places the big tensor first in the
        flat buffer) and waits for the big slice."""
        if not self.active:
            return
        off, n = self.engine.big_grad_slice()
        g = self.engine.GE.grad
        if off > 0:
            self._allreduce(g[:off])
        if off + n < g.numel():
            self._allreduce(g[off + n:])
        self._wait()

    def _wait(self):
        for w in self._pending:
            w.wait()
        self._pending.clear()

    # ---- the critic's gradient: latency-bound 1.25 MB, hidden behind the G-step's generator forward -------------
    def start_d(self):
        if not self.active or self._dry:
            return
        self._pending.append(self.dist.all_reduce(self.engine.D.grad, op=self.dist.ReduceOp.SUM, group=self.group,
                                                  async_op=True))

    def step(self, use_graph: bool = True, g_step: bool = True):
        """One training step (1 critic update, optionally 1 generator update) on the batch already set with
        engine.set_batch().  world == 1: two graphs per sub-step.  world > 1: the step order of MELO_DP_MODE (module
        docstring).  "gather" / "allreduce": synchronous collectives on the engine's stream; "overlap": asynchronous
        ones on RCCL's stream beside the engine's."""
        e = self.engine
        if not self.active:
            # one graph per batch: the critic step alone, or critic + generator step with ONE 2B-row generator pass
            e.run("dg_step_rng" if g_step and hasattr(e, "dg_step_rng") else "d_step_rng", use_graph)
            if g_step and not hasattr(e, "dg_step_rng"):
                e.run("g_step_rng", use_graph)
            return
        e.run("d_backward_rng", use_graph)          # Philox draw (noise, alpha, dropout masks) + D fwd/bwd
        if not g_step:
            self.allreduce_d()
            e.run("d_update", use_graph)
            return
        if self.mode == "overlap":
            self.start_d()
            e.run("g_forward_rng", use_graph)
            self._wait()
            e.run("d_update", use_graph)
            e.run("g_backward_a2", use_graph)
            self.start_g_big()
            e.run("g_backward_b", use_graph)
            self.finish_g()
            e.run("g_update", use_graph)
            return
        self.allreduce_d()
        e.run("d_update", use_graph)
        if self.mode == "allreduce":
            e.run("g_backward_rng", use_graph)
            self.allreduce_g()
        else:
            e.run("g_backward_a_rng", use_graph)    # ... down to pre.2's output gradient; no pre.2 weight gradient yet
            self.gather_p2()
            e.run("g_backward_p2b", use_graph)      # pre.2's global weight gradient + the rest of backward
            self.allreduce_g_rest()
        e.run("g_update", use_graph)

    # ---- factor gather for decoder.pre.2.weight (mode "gather") ----------------------------------------------
    def gather_p2(self):
        e = self.engine
        for src, dst in ((e.d_p2, e.d_p2_all), (e.a_p0, e.a_p0_all)):
            if self._dry:
                dst[:src.shape[0]].copy_(src)
            elif src.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal of several ranks on one GPU: gloo has no all_gather for device tensors
                host = dst.cpu()
                self.dist.all_gather_into_tensor(host, src.cpu(), group=self.group)
                dst.copy_(host)
            else:
                self.dist.all_gather_into_tensor(dst, src, group=self.group)

    def allreduce_g_rest(self):
        """Everything of the generator / numeric-encoder gradient except pre.2's weight and bias (first in the flat
        buffer; both come out of the gathered factors as global-batch sums already)."""
        e = self.engine
        off, n = e.p2_grad_slice() if hasattr(e, "p2_grad_slice") else e.big_grad_slice()
        if off != 0:
            raise RuntimeError("gather mode expects decoder.pre.2 at offset 0 of the flat gradient")
        self._allreduce(e.GE.grad[n:])
