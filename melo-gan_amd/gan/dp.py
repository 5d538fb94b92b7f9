"""Data-parallel wrapper: one process per GPU, identical replicas, per-rank batch shard.

Every loss of the hot path is a mean over per-sample terms, so averaging the
ranks' flat gradient buffers equals the global-batch gradient: all-reduce(SUM) of the flat fp32
gradient buffers (critic 1.25 MB every batch; generator + numeric encoder 18.8 MB on generator
steps), the 1/world factor folded into the fused Adam launch (grad_scale).  On ROCm the "nccl"
backend is RCCL over xGMI.

Step orders (MELO_DP_MODE).  The default on real GPUs is **ingraph** (round 3): a PRIVATE RCCL communicator (gan/rccl.py,
ctypes over librccl: `ncclAllReduce` / `ncclAllGather` take the engine's stream and are captured like kernel launches), so
the N > 1 step is exactly the N = 1 step -- one graph per batch, or the split flow's four with the emotion branch on the side
stream -- with three collective nodes inside it:

    C1  group{ all-reduce(critic gradient, 1.25 MB), all-gather(a_p0: pre.2's INPUT factor, 128 KB per rank) }   in front of the
        critic's Adam launch (a_p0 exists since the generator pass: it rides in the collective that is issued anyway)
    C2  all-gather(d_p2: pre.2's OUTPUT-gradient factor, 2 MB per rank)     in front of the weight-gradient launches, which
        compute pre.2's GLOBAL weight gradient from the gathered factors (never all-reduced: 16.8 of the 18.8 MB)
    C3  all-reduce(everything else of the generator / encoder gradient, 2 MB)    in front of the generator's Adam launch

No host involvement between the pieces, no process-group watchdog (torch.distributed only broadcasts the parameters at
start-up), nothing to prepare() before the first collective.  The older orders below stay for backends without such a
communicator (gloo: the CPU tests, several ranks rehearsing on one GPU) and as the reference the new one is tested against.

The older orders: a batch's step is a fixed sequence of hipGraphs with the collectives between them.  With a generator update:

    G1  dg_forward_d_backward_rng   one Philox draw, the 2B-row E_num + generator pass, the critic step's forward/backward
    C1  all-reduce(critic gradient, 1.25 MB)
    G2  g_ed_branch                 frozen emotion discriminator forward + input gradient: does not touch the critic
    G3  d_update_g_critic_chain     critic Adam, critic forward/backward on the generated batch, generator data
                                    gradients down to decoder.pre.2
    C2  all-gather of pre.2's two gradient factors (modes gather / overlap) -- see below
    G4  g_backward_b                deconvolution + Linear weight gradients, the rest of the backward chain
    G5  g_p2_wgrad                  pre.2's global weight gradient from the gathered factors (inside G4 when C2 is synchronous)
    C3  all-reduce(everything else of the generator / encoder gradient, 2 MB)
    G6  g_update

Optional (MELO_DP_SIDE=1, off by default -- see __init__): the emotion branch (G2) launched on the engine's SIDE stream right
after the generator pass -- G1 split into G1a (draw + generator pass) and G1b (critic step's backward), G3 into G3a (critic
Adam + critic pass over the generated batch) and G3b (from the first use of the emotion branch's gradient on) -- beside G1b,
C1 and G3a; the main stream joins it in front of G3b.  Measured on one MI355X with a 1-rank RCCL group: gather 1.023 -> 0.984
ms/step, allreduce 0.995 -> 0.957, overlap 1.105 -> 1.049; results bit-identical.

Step orders (MELO_DP_MODE; default "auto" = "overlap" from 8 ranks up, "gather" below):
  gather     every collective synchronous, on the engine's stream, in program order.
  overlap    C2 -- the only transfer large enough to be worth it, 18 MB received per rank at N = 8 -- is issued
             asynchronously (RCCL's own stream, ordered behind the engine's by an event) and runs beside G4 (~150 us of
             weight gradients that do not depend on it); C1 and C3 stay synchronous.
  allreduce  one synchronous all-reduce per optimiser, no factor gather.
Why not everything asynchronous: measured on one MI355X with a 1-rank RCCL group (the transfers take no time there, so
what shows is the price of the mechanism): no collectives 1.176 ms/step, allreduce 1.214, gather 1.245, C1 and C2 both
asynchronous 1.345 -- and 1.352 with the collectives synchronous but G2 / G4 forked onto a side stream instead.  A
cross-stream fork + join around a hipGraph launch costs ~50 us on this platform whichever side carries the collective,
so overlap pays only for a transfer that takes longer than that: C2 at N = 8 (~75-150 us expected), not C1 (~30 us).

Factor gather: decoder.pre.2.weight is 16.8 of the generator's 18.8 MB and its gradient is d_p2^T a_p0, so the ranks
all-gather those two per-sample factors (2.2 MB per rank) and each computes the global batch's weight gradient itself
(GanEngine.enable_p2_gather).  Bytes received per rank at N = 2 / 4 / 8: 4.4 / 9 / 18 MB instead of 20 / 30 / 35 MB.

Graph capture and the process group's watchdog: torch's watchdog thread polls the completion event of every collective,
and HIP refuses an event query while the stream the event was recorded on is capturing.  A synchronous collective
records its event on the ENGINE's stream, so no capture may follow one.  prepare() therefore runs -- once, before the
first collective -- two dry steps of each kind (collectives replaced by local copies) that warm up and capture every
graph the step order uses, restores the training state, and locks the engine against later captures.

BatchNorm semantics under sharding (decision, SURVEY hard part 4): the generator's train-mode
BatchNorm uses the LOCAL shard's statistics (DDP-style), i.e. each rank runs exactly the
reference's B=64 model; running statistics are rank-local and rank 0's are checkpointed.
The wrapper is backend-agnostic (works on CPU tensors with gloo) so that the N>1 logic is
testable without GPUs.
"""
from __future__ import annotations

import os
import sys

MODES = ("auto", "ingraph", "overlap", "gather", "allreduce")


class InGraphCollectives:
    """The three collectives of a step, issued from INSIDE the engine's sub-steps (GanEngine.d_update / g_backward_b /
    g_update call these through `engine.coll`) on the engine's stream -- part of whatever graph the sub-step is captured
    into.  comm: gan/rccl.py::RcclComm (or TorchDistComm: eager only)."""

    def __init__(self, comm):
        self.comm = comm

    def reduce_d(self, e, with_a_p0: bool):
        """C1: the critic's gradient summed over the ranks; on a generator step pre.2's input factor rides along."""
        c = self.comm
        c.group_start()
        c.all_reduce(e.D.grad)
        if with_a_p0:
            c.all_gather(e.a_p0, e.a_p0_all)
        c.group_end()

    def gather_p2(self, e, with_a_p0: bool):
        """C2: pre.2's output-gradient factor of every rank (and the input factor if C1 did not carry it)."""
        c = self.comm
        c.group_start()
        c.all_gather(e.d_p2, e.d_p2_all)
        if with_a_p0:
            c.all_gather(e.a_p0, e.a_p0_all)
        c.group_end()

    def reduce_g(self, e):
        """C3: the generator / encoder gradient except what the gathered factors already gave globally."""
        if getattr(e, "p2_world", 0):
            off, n = e.p2_grad_slice()
            if off != 0:
                raise RuntimeError("the factor gather expects decoder.pre.2 at offset 0 of the flat gradient")
            self.comm.all_reduce(e.GE.grad[n:])
        else:
            self.comm.all_reduce(e.GE.grad)


class DataParallel:
    def __init__(self, engine, world_size: int, dist=None, group=None, force_collectives: bool = False, comm=None):
        """force_collectives: issue the collectives (and take the N > 1 step order) even at world_size 1 -- a
        rehearsal of the N > 1 control path on a single-GPU box.  comm: a ready communicator for the ingraph order (tests);
        by default a private RCCL communicator is created over `dist`'s ranks when its backend is nccl."""
        self.engine, self.world, self.dist, self.group = engine, int(world_size), dist, group
        self.active = (dist is not None or comm is not None) and (self.world > 1 or force_collectives)
        self._dry = os.environ.get("MELO_DP_DRY") == "1"      # rehearsal: the N > 1 step order without the collectives
        self.mode = os.environ.get("MELO_DP_MODE", "auto")
        if self.mode not in MODES:
            raise ValueError(f"MELO_DP_MODE={self.mode}: expected one of {MODES}")
        self.comm = None
        can_ingraph = hasattr(engine, "coll") and hasattr(engine, "enable_p2_gather")
        if self.active and can_ingraph and self.mode in ("auto", "ingraph"):
            if comm is not None:
                self.comm = comm
            elif dist is not None and self._backend() == "nccl":
                # "auto": a private communicator that cannot be created (librccl missing from torch's lib directory, an
                # id exchange that fails) must not take the job down -- every rank then agrees on the round-2 order over
                # torch.distributed.  MELO_DP_MODE=ingraph asks for it explicitly and gets the error.
                from .rccl import RcclComm
                err = None
                try:
                    self.comm = RcclComm.from_process_group(dist, group)
                except Exception as ex:      # noqa: BLE001
                    err = ex
                ok = torch_ok = 1 if err is None else 0
                try:
                    import torch
                    t = torch.tensor([ok], device="cuda", dtype=torch.int32)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                    torch_ok = int(t.item())
                except Exception:            # noqa: BLE001
                    pass
                if err is not None and self.mode == "ingraph":
                    raise err
                if not torch_ok:             # some rank failed: nobody uses the private communicator
                    if self.comm is not None:
                        self.comm.destroy()
                        self.comm = None
                    if err is not None:
                        print(f"[melo_gan_amd.dp] in-graph collectives unavailable ({err}); falling back to the torch.distributed "
                              "step order", file=sys.stderr)
            elif self.mode == "ingraph":
                raise ValueError("MELO_DP_MODE=ingraph needs the nccl (RCCL) backend or an explicit communicator")
        if self.comm is not None:
            self.mode = "ingraph"
        elif self.mode in ("auto", "ingraph"):
            self.mode = "overlap" if self.world >= 8 else "gather"
        if self.mode != "allreduce" and not hasattr(engine, "enable_p2_gather"):
            self.mode = "allreduce"
        if self.active and self.mode != "allreduce":
            engine.enable_p2_gather(self.world)
        if self.mode == "ingraph":
            engine.coll = InGraphCollectives(self.comm)
        engine.world_size = self.world
        self._pending = []
        self._prepared = False
        self._prev_g = False                                        # was the previous batch a generator step (split flow)
        # N > 1: MELO_DP_SIDE=1 puts the emotion branch on the side stream as well.  Off by default: it could only be measured
        # on a 1-rank RCCL group (-40..-55 us per step in every mode), not with real inter-GPU collectives, and two
        # processes SHARING one GPU over gloo (the rehearsal setup) fall to 440 ms per step with it (process time slicing
        # between two queues each) -- the proven one-stream order stays the default for runs nobody could rehearse.
        self._dp_side = os.environ.get("MELO_DP_SIDE", "0") == "1" and getattr(engine, "ed_dtype", "fp32") == "fp32" \
            and hasattr(engine, "d_update_g_critic_front")
        # split | ingraph | none.  Default: the split flow for the fp32 engine; the bf16-stored emotion branch is a third as
        # long and the three extra graph launches cost more than hiding it returns (0.831 -> 0.872 ms): it forks inside
        # the one graph instead (0.770)
        self._ed_flow = os.environ.get("MELO_ED_FLOW") or "ingraph"
        # (round 2 took "split" for the fp32 engine: 96 launches per step made the forked graph's launch cost the host 0.87 ms,
        #  more than the GPU step; at 60 launches the host is back ahead and one graph with the fork inside beats the four
        #  graphs of the split flow by their three extra boundaries: 0.886 -> 0.868 ms per step)

    def _backend(self):
        try:
            return self.dist.get_backend(self.group)
        except Exception:  # noqa: BLE001  (stand-ins without a backend notion)
            return None

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
        if not self.active or self.dist is None:
            return
        bufs = self._flat_state()
        for t in bufs:
            self.dist.broadcast(t, src=src, group=self.group, async_op=True).wait()
        if bufs and bufs[0].is_cuda:
            # the broadcasts are ordered on the launching stream only; the step's graphs replay on the engine's stream
            import torch
            torch.cuda.synchronize()
        if hasattr(self.engine, "params_changed"):
            self.engine.params_changed()         # derived copies (WQ-layout conv weights, folded emotion discriminator)

    # ---- graph capture before the first collective ---------------------------------------------------------
    def prepare(self, use_graph: bool = True):
        """Warm up and capture every hipGraph of this mode's step order (with and without a generator update) by two
        dry steps each -- the engine's first run of a sub-step is eager, the second captures -- then restore the training
        state and forbid later captures.  Must run on the engine's stream, after set_batch() has staged finite inputs,
        before the first step().  A no-op for engines without graphs (the CPU stand-ins of the tests) or use_graph=False."""
        e = self.engine
        self._prepared = True
        if not self.active or not use_graph or not hasattr(e, "_capture") or self.mode == "ingraph":
            return                          # ingraph: the collectives are graph nodes, nothing to capture ahead of them
        import torch
        keep = [e.D.data, e.D.grad, e.D.m, e.D.v, e.D.state, e.GE.data, e.GE.grad, e.GE.m, e.GE.v, e.GE.state, e.rng_step]
        keep += list(e.Gbuf.values())
        saved = [t.clone() for t in keep]
        nbt, ticked = e.num_batches_tracked, (e.D.ticked, e.GE.ticked)
        dry, self._dry = self._dry, True
        try:
            for g_step in (True, True, False, False):
                self._step(True, g_step)
        finally:
            self._dry = dry
        torch.cuda.synchronize()
        for t, v in zip(keep, saved):
            t.copy_(v)
        e.params_changed()
        e.num_batches_tracked, (e.D.ticked, e.GE.ticked) = nbt, ticked
        torch.cuda.synchronize()
        e.capture_locked = True

    # ---- collectives -----------------------------------------------------------------------------------------
    def _allreduce(self, flat, async_op: bool = False):
        if not self.active or self._dry:
            return
        # A synchronous collective runs ON the calling (engine) stream: no cross-stream hop.  An asynchronous one runs on
        # RCCL's stream behind an event of the calling stream; wait() makes the calling stream wait for it (no host block).
        w = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            self._pending.append(w)

    def _wait(self):
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def allreduce_d(self, async_op: bool = False):
        self._allreduce(self.engine.D.grad, async_op)

    def allreduce_g(self):
        self._allreduce(self.engine.GE.grad)

    def gather_p2(self, async_op: bool = False):
        """All-gather of decoder.pre.2's gradient factors (d_p2: (B, 256 red), a_p0: (B, 512)) in rank order."""
        e = self.engine
        for src, dst in ((e.d_p2, e.d_p2_all), (e.a_p0, e.a_p0_all)):
            if not self.active or self._dry:
                dst[:src.shape[0]].copy_(src)
            elif src.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal of several ranks on one GPU: gloo has no all_gather for device tensors
                host = dst.cpu()
                self.dist.all_gather_into_tensor(host, src.cpu(), group=self.group)
                dst.copy_(host)
            else:
                w = self.dist.all_gather_into_tensor(dst, src, group=self.group, async_op=async_op)
                if async_op:
                    self._pending.append(w)

    def allreduce_g_rest(self):
        """Everything of the generator / numeric-encoder gradient except pre.2's weight and bias (first in the flat
        buffer; both come out of the gathered factors as global-batch sums already)."""
        e = self.engine
        off, n = e.p2_grad_slice() if hasattr(e, "p2_grad_slice") else e.big_grad_slice()
        if off != 0:
            raise RuntimeError("gather mode expects decoder.pre.2 at offset 0 of the flat gradient")
        self._allreduce(e.GE.grad[n:])

    # ---- the step --------------------------------------------------------------------------------------------
    def step(self, use_graph: bool = True, g_step: bool = True):
        """One training step (1 critic update, optionally 1 generator update) on the batch already set with
        engine.set_batch().  world == 1: one graph per batch, or the split flow below when generator steps follow each
        other.  world > 1: the step order of MELO_DP_MODE (module
        docstring)."""
        e = self.engine
        if not self.active or self.mode == "ingraph":
            # one GPU -- or N GPUs with the collectives inside the graphs (engine.coll): the same flows
            side = getattr(e, "ed_side", None)
            back_to_back, self._prev_g = self._prev_g and g_step, g_step
            if back_to_back and use_graph and side is not None and self._ed_flow == "split":
                # The split flow: the frozen emotion discriminator's branch (a third of the step's MFMA work; needs only the
                # generated batch, is needed only where the generator's backward starts) runs as its OWN graph on a side
                # stream beside the critic step, whose ~50 launches are mostly small dependent kernels that leave the matrix
                # pipes idle.  Four graphs per batch: [draw + 2B-row generator pass] -> fork -> side: [emotion branch] beside
                # main: [critic step + critic pass over the generated batch] -> join -> [generator backward + update].
                # Only when generator steps follow each other (the host is then a whole step ahead of the GPU): measured
                # 0.959 -> 0.929 ms/batch at 1 critic : 1 generator update, but 0.474 -> 0.503 at the reference's 5 : 1
                # (Streams).
                import torch
                cur = torch.cuda.current_stream()
                e.run("dg_forward_rng", True)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    e.run("g_ed_branch_side", True)
                e.run("d_step_g_critic_front", True)
                cur.wait_stream(side)
                e.run("g_finish", True)
                return
            if back_to_back and use_graph and side is not None and self._ed_flow == "fork2":
                e.run("dg_forward_rng", True)
                e.run("dg_fork_rest", True)
                return
            if back_to_back and use_graph and side is not None and self._ed_flow == "ingraph":
                e.run("dg_fork_step_rng", True)       # the branch inside the one graph (GanEngine.dg_fork_step_rng)
                return
            # one graph per batch: the critic step alone, or critic + generator step with ONE 2B-row generator pass
            e.run("dg_step_rng" if g_step else "d_step_rng", use_graph)
            return
        if not self._prepared:
            self.prepare(use_graph)
        self._step(use_graph, g_step)

    def _step(self, use_graph: bool, g_step: bool):
        e = self.engine
        if not g_step:
            e.run("d_backward_rng", use_graph)        # Philox draw (noise, alpha, dropout masks) + D fwd/bwd
            self.allreduce_d()
            e.run("d_update", use_graph)
            return
        side = getattr(e, "ed_side", None) if self._dp_side else None
        if side is not None:
            # the frozen emotion discriminator's branch on a side stream, beside the critic step's backward, C1 and the
            # critic's pass over the generated batch (as on one GPU: DataParallel.step, "the split flow")
            import torch
            cur = torch.cuda.current_stream()
            e.run("dg_forward_rng", use_graph)            # G1a
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                e.run("g_ed_branch_side", use_graph)      # G2, side stream
            e.run("d_backward_nofwd", use_graph)          # G1b
            self.allreduce_d()                            # C1
            e.run("d_update_g_critic_front", use_graph)   # G3a
            cur.wait_stream(side)
            e.run("g_critic_back", use_graph)             # G3b
        else:
            e.run("dg_forward_d_backward_rng", use_graph)
            self.allreduce_d()                            # C1
            e.run("g_ed_branch", use_graph)
            e.run("d_update_g_critic_chain", use_graph)
        if self.mode == "allreduce":
            e.run("g_backward_b", use_graph)
            self.allreduce_g()
        elif self.mode == "gather":
            self.gather_p2()                          # C2, synchronous
            e.run("g_backward_p2b", use_graph)        # pre.2's global weight gradient shares the weight-gradient launch
            self.allreduce_g_rest()
        else:
            self.gather_p2(async_op=True)             # C2 ...
            e.run("g_backward_b", use_graph)          # ... beside the weight gradients and the rest of the backward chain
            self._wait()
            e.run("g_p2_wgrad", use_graph)
            self.allreduce_g_rest()                   # C3
        e.run("g_update", use_graph)
