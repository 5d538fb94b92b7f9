"""Data-parallel wrapper: one process per GPU, identical replicas, per-rank batch shard.

Every loss of the hot path is a mean over per-sample terms, so averaging the
ranks' flat gradient buffers equals the global-batch gradient.  One collective per optimiser per
step: all-reduce(SUM) of the flat fp32 gradient buffer (critic 1.25 MB every batch; generator +
numeric encoder 18.8 MB on generator steps); the 1/world factor is folded into the fused Adam
launch (grad_scale).  On ROCm the "nccl" backend is RCCL over xGMI.

BatchNorm semantics under sharding (decision, SURVEY hard part 4): the generator's train-mode
BatchNorm uses the LOCAL shard's statistics (DDP-style), i.e. each rank runs exactly the
reference's B=64 model; running statistics are rank-local and rank 0's are checkpointed.
The wrapper is backend-agnostic (works on CPU tensors with gloo) so that the N>1 logic is
testable without GPUs.
"""
from __future__ import annotations

class DataParallel:
    def __init__(self, engine, world_size: int, dist=None, group=None):
        self.engine, self.world, self.dist, self.group = engine, int(world_size), dist, group
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
        if self.world == 1 or self.dist is None:
            return
        for t in self._flat_state():
            self.dist.broadcast(t, src=src, group=self.group)

    def _allreduce(self, flat):
        if self.world == 1 or self.dist is None:
            return
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
        if self.world == 1 or self.dist is None:
            return
        off, n = self.engine.big_grad_slice()
        self._pending.append(self.dist.all_reduce(self.engine.GE.grad[off:off + n], op=self.dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True))

    def finish_g(self):
        """All-reduces what start_g_big() left (one contiguous range: the engine places the big tensor first in the
        flat buffer) and waits for the big slice."""
        if self.world == 1 or self.dist is None:
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
        if self.world == 1 or self.dist is None:
            return
        self._pending.append(self.dist.all_reduce(self.engine.D.grad, op=self.dist.ReduceOp.SUM, group=self.group,
                                                  async_op=True))

    def step(self, use_graph: bool = True, g_step: bool = True):
        """One training step (1 critic update, optionally 1 generator update) on the batch already set with
        engine.set_batch().  world == 1: two graphs per sub-step.  world > 1: the critic's all-reduce overlaps the
        generator forward of the G-step (which does not read the critic), decoder.pre.2.weight's all-reduce overlaps
        the tail of backward, the rest of the generator gradient is one more all-reduce."""
        e = self.engine
        e.run("d_backward_rng", use_graph)          # Philox draw (noise, alpha, dropout masks) + D fwd/bwd
        if self.world == 1 or self.dist is None:
            e.run("d_update", use_graph)
            if g_step:
                e.run("g_backward_rng", use_graph)
                e.run("g_update", use_graph)
            return
        if not g_step:
            self.allreduce_d()
            e.run("d_update", use_graph)
            return
        self.start_d()
        e.run("g_forward_rng", use_graph)
        self._wait()
        e.run("d_update", use_graph)
        e.run("g_backward_a2", use_graph)
        self.start_g_big()
        e.run("g_backward_b", use_graph)
        self.finish_g()
        e.run("g_update", use_graph)
