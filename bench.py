#!/usr/bin/env python3
"""Headline benchmark: piano-roll samples/sec of the full (1 critic + 1 generator) step of
Melo-GAN's GAN training hot path (G + D + frozen emotion-D), B=64 per GPU, 128x256 rolls.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one D-step (src/gan/train_gan.py:183-205) + one G-step (:211-251) on one batch of
synthetic (B, T=256, C=128) rolls resident in HBM, including the per-step device RNG draws,
the gradient all-reduce (N>1) and both Adam updates.  value = N*B*K / max-over-ranks wall time of
exactly K steps (barrier + synchronize on both sides).  Beside it, "event_timing": every step of a
second window of max(K, 64) steps between its own pair of HIP events on the engine's stream --
median / p10 / p90 per step (median of >= 50 hipEvent-timed iterations).
Behind the W warm-up steps (the first of them eager / capturing, the GPU mostly idle) --settle-steps (default 32) MORE untimed
steps of the same workload are replayed before the timed window, so that a short window (the driver's K = 20) does not start
on clocks that are still ramping (0.875 -> 0.825 ms per step over the first ~15 replays, tools/first_steps.py); the line
reports them as `warmup_settle_steps`.

The same JSON line carries
  roofline     : the dominant kernel SYMBOL (wino3_kernel: the emotion discriminator's three-tap layers by minimal
                 filtering F(2,3), 5 launches = 20.1 of the step's 56.9 algorithmic GFLOP).  The launches are recorded
                 during one step and replayed (same tensors, HIP events on the launch stream): ALGORITHMIC (direct-form)
                 FLOPs / that time, against the dense fp32-MFMA peak (157.3 TFLOP/s) -- the kernel EXECUTES 2/3 of those
                 FLOPs on the matrix pipe, `mfma_executed_frac` is the pipe's own utilisation;
  cpu_baseline : the oracle (PyTorch-CPU fp32 restatement of the reference step) timed on this
                 host on a bounded number of the same steps (rank 0, N=1 only): once with every physical
                 core this process may use, once with 1 thread; CPU model and counts stated.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

B_PER_GPU, T, C = 64, 256, 128
# the dominant kernel: the emotion discriminator's three-tap layers by minimal filtering F(2,3) (csrc/conv_wino.hip: conv1-3
# forward, conv2-3 data-gradient = 5 launches, 20.1 of the step's algorithmic GFLOP); MELO_ED_WINO=0 puts the direct
# stride-1 K=3 window GEMM back (both weight-layout instantiations)
DOMINANT = ("wino3_kernel",)
DOMINANT_NAME = "wino3_kernel"
DOMINANT_DIRECT = ("conv_wgemm_kernel<1,3,false,true,1,1>", "conv_wgemm_kernel<1,3,false,false,1,1>")
DOMINANT_DIRECT_NAME = "conv_wgemm_kernel<1,3,false,{true|false},1,1>"
WINO_EXECUTED = 2.0 / 3.0        # matrix-pipe FLOPs executed per algorithmic (direct-form) FLOP: 4 channel GEMMs per output pair, not 6
CONV16 = tuple("conv16_kernel<%s,%d,%s>" % (a, b, c) for a in ("false", "true") for b in (1, 2) for c in ("false", "true"))
CONV16_NAME = "conv16_kernel<{false|true},{1|2},{false|true}>"
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: dense fp32 matrix peak
# conv+linear FLOPs per sample of one (1D+1G) step at cfg2 as the reference executes it
MFLOP_PER_SAMPLE = 889.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-steps", type=int, default=32,
                    help="further untimed steps behind the warm-up, until the GPU's clocks have settled (reported in the line)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the all-cores CPU leg (the 1-thread "
                    "leg gets the same budget and at least 2 steps)")
    ap.add_argument("--profile-steps", type=int, default=20,
                    help="replays of the dominant kernel's launches (one step's worth each) for the roofline leg; 0 = skip")
    ap.add_argument("--launch-flops", default=None, metavar="FILE",
                    help="write {kernel symbol: launches and GFLOP per step} of one eager step to FILE (tools/pmc_mfma_busy.py)")
    ap.add_argument("--ed-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: the SECONDARY configuration -- the frozen emotion discriminator's activations and folded "
                         "weights stored in bf16, fp32 accumulate (reported with its own dtype string, never the headline)")
    ap.add_argument("--workload", default="gan", choices=["gan", "ae", "gen1", "ed"],
                    help="gan: the headline cfg2 step (default); ae: BASELINE config 4 (VAE step, B=256, T=256, C=4); "
                         "gen1: BASELINE config 5 (batch-1 E_num->G generation latency); ed: emotion-discriminator "
                         "pre-training step (SURVEY f-2) at the cfg2 shape")
    return ap.parse_args()


class RecordHook:
    """Remembers every launch of the selected kernel symbols (symbol, FLOPs, a closure that re-issues it)."""

    def __init__(self, symbols):
        self.symbols, self.records = set(symbols), []

    def __call__(self, symbol, flops, launch=None):
        hook = self

        class Ctx:
            def __enter__(self_c):
                return self_c

            def __exit__(self_c, *a):
                if symbol in hook.symbols and launch is not None:
                    hook.records.append((symbol, flops, launch))
                return False
        return Ctx()


def time_dominant(ops, records, reps):
    """Summed duration (ms) of `reps` replays of the recorded launches (same tensors, same order as in the step), each
    launch between its OWN pair of HIP events on the launch stream: one dispatch at a time, which is what rocprofv3's
    kernel trace reports too.  (Back-to-back replay from one hipGraph under a single event pair lets a launch's ramp
    overlap its predecessor's drain and read ~5 % shorter than the kernel takes inside the step.)"""
    for _, _, launch in records:                 # warm
        if launch() != 0:
            raise RuntimeError("replayed launch failed")
    pairs = [(ops.Event(), ops.Event()) for _ in range(reps * len(records))]
    i = 0
    for _ in range(reps):
        for _, _, launch in records:
            a, b = pairs[i]
            i += 1
            a.record()
            rc = launch()
            b.record()
            if rc != 0:
                raise RuntimeError(f"replayed launch failed: rc={rc}")
    torch.cuda.synchronize()
    return sum(a.elapsed_ms(b) for a, b in pairs)


class EventHook:
    """Brackets every launch of the selected kernel symbols with a pair of HIP events on the launch stream (the in-step leg)."""

    def __init__(self, ops, symbols):
        self.ops, self.symbols, self.pairs = ops, set(symbols), []

    def __call__(self, symbol, flops, launch=None):
        hook = self

        class Ctx:
            def __enter__(self_c):
                if symbol in hook.symbols:
                    self_c.a, self_c.b = hook.ops.Event(), hook.ops.Event()
                    self_c.a.record()
                return self_c

            def __exit__(self_c, *a):
                if symbol in hook.symbols:
                    self_c.b.record()
                    hook.pairs.append((symbol, flops, self_c.a, self_c.b))
                return False
        return Ctx()


def time_in_step(ops, eng, symbols, reps):
    """The selected kernels' durations INSIDE the production step: `reps` eager runs of the forked step (emotion branch on
    the side stream beside the critic step, as the replayed graph has it), every selected launch between its own event pair on
    the stream it is launched on.  Returns (total ms, total flops, launches).  Eager launches leave the GPU a little less
    contended than the graph replay does (host launch latency between kernels); rocprofv3's per-kernel average over the
    replayed graphs is in profiles/ (tools/make_profiles.sh)."""
    hook = EventHook(ops, symbols)
    run = eng.dg_fork_step_rng if eng.ed_side is not None else eng.dg_step_rng
    run()                                        # warm
    torch.cuda.synchronize()
    ops.set_launch_hook(hook)
    try:
        for _ in range(reps):
            run()
            eng.D.ticked = eng.GE.ticked = False
        torch.cuda.synchronize()
    finally:
        ops.set_launch_hook(None)
    ms = sum(a.elapsed_ms(b) for _, _, a, b in hook.pairs)
    return ms, sum(f for _, f, _, _ in hook.pairs), len(hook.pairs)


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed PMC measurement (profiles/rNN_traffic_dominant_
    kernel.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied -- tools/pmc_traffic.py).
    The counters cannot be read from inside this process, so the line carries the committed figure and says where it is from."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_dominant_kernel.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return float(d["traffic_bytes_per_launch"]), (f"profiles/{os.path.basename(files[-1])}: {d.get('launches', '?')} launches, "
                                                       "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, "
                                                       "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024")
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable {files[-1]}: {e}"


def host_cpu():
    """(threads to use, physical cores of the host, logical CPUs this process may run on, model name).  Threads = the
    physical cores this process can actually use: logical CPUs of the affinity mask, limited by a cgroup CPU quota if
    there is one, and never more than one thread per physical core (ATen's CPU kernels lose with SMT siblings)."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    phys = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    physical = len(cores) or logical
    n = min(logical, physical)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if os.environ.get("MELO_CPU_THREADS"):
        n = int(os.environ["MELO_CPU_THREADS"])
    return max(1, n), physical, logical, model


def cpu_baseline(seconds: float):
    """Oracle (the CPU restatement, pinned to the reference by tests/golden) on the host cores: all usable physical
    cores, then 1 thread.  Bounded: `seconds` per leg (at least 2 timed steps)."""
    from oracle import melo_oracle as O
    threads, physical, logical, model = host_cpu()
    cfg, ed_cfg = O.default_gan_cfg(B_PER_GPU, T, C), O.default_ed_cfg(C)
    S = O.build_gan_state(cfg, ed_cfg, "weights_init", seed=42)
    real, numeric, latent, emot = O.synthetic_batch(B_PER_GPU, T, C, cfg["LATENT_DIM"], 6, 42)

    def one(i):
        R = O.step_randoms(B_PER_GPU, cfg["NOISE_DIM"], seed=i)
        O.d_step(S, real, latent, numeric, R["noise_d"], R["alpha"], R["dm_d"])
        O.g_step(S, latent, numeric, emot, R["noise_g"], R["dm_g"])

    def leg(nthreads, warm):
        torch.set_num_threads(nthreads)
        for i in range(warm):
            one(i)
        ts = []
        t0 = time.perf_counter()
        while True:
            t1 = time.perf_counter()
            one(warm + len(ts))
            ts.append(time.perf_counter() - t1)
            if (time.perf_counter() - t0 >= seconds and len(ts) >= 2) or len(ts) >= 5000:
                break
        ts.sort()
        return B_PER_GPU / ts[len(ts) // 2], len(ts), time.perf_counter() - t0

    v_all, n_all, el_all = leg(threads, 2)
    v_one, n_one, el_one = leg(1, 1)
    torch.set_num_threads(threads)
    return dict(value=round(v_all, 2), unit="samples/s", cores=threads, kind="port",
                sample=f"median of {n_all} full (1D+1G) steps of the same B=64, T=256, C=128 workload, fp32, "
                       f"torch {torch.__version__} CPU, {threads} threads, {el_all:.1f} s",
                value_1thread=round(v_one, 2),
                sample_1thread=f"median of {n_one} steps, 1 thread, {el_one:.1f} s",
                cpu_model=model, host_physical_cores=physical, usable_logical_cpus=logical)


def side_workload(args):
    """Secondary BASELINE configs (single GPU, not the headline metric): one JSON line each."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    torch.cuda.set_device(0)
    if args.workload == "ed":
        from melo_gan_amd.emotion_discriminator.engine import EdEngine
        from melo_gan_amd.gan.config import default_ed_cfg
        Bv = B_PER_GPU
        cfg = dict(default_ed_cfg(C), dropout=0.2, optimizer=dict(name="AdamW", lr=2e-4, betas=[0.5, 0.999], weight_decay=0.0))
        eng = EdEngine(cfg, "cuda", Bv, T)
        eng.init_weights(0)
        x = torch.rand(Bv, T, C, device="cuda") * 2 - 1
        y = torch.randint(0, 4, (Bv,), device="cuda")
        with torch.cuda.stream(eng.stream):
            eng.set_batch(x, y)
            for _ in range(max(args.warmup, 3)):
                eng.run("step_rng")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.run("step_rng")
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        # conv FLOPs: forward + data-gradient (not into the input) + weight-gradient of ED conv0-3
        gf = 2.0 * Bv * T * sum(ci * co * k for ci, co, k in eng.chans) * 3 / 1e9 - 2.0 * Bv * T * eng.chans[0][0] * eng.chans[0][1] * eng.chans[0][2] / 1e9
        print(json.dumps({"metric": "emotion-discriminator pre-training samples/sec, batch=64 128x256 roll", "value": round(Bv * args.steps / el, 1),
                          "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "f-2: ED train step B=64, T=256, C=128 (train-mode BN+GELU, dropout, CE, AdamW)",
                                                          "step_gflop": round(gf, 2)},
                          "losses": {"ce": round(eng.loss.item(), 5)}}), flush=True)
        return
    if args.workload == "ae":
        from melo_gan_amd.ae.engine import VaeEngine
        Bv, Tv = 256, 256
        eng = VaeEngine(dict(MAX_NOTES=Tv, LATENT_DIM=8, BATCH_SIZE=Bv, LR=1e-4, WEIGHT_DECAY=1e-5), "cuda", Bv)
        eng.init_weights(0)
        x = torch.rand(Bv, Tv, 4, device="cuda") * 2 - 1
        eps = torch.empty(Bv, 8, device="cuda")
        with torch.cuda.stream(eng.stream):
            def one():
                eng.step(x, eps.normal_(), 10.0)
            for _ in range(max(args.warmup, 3)):
                one()
            torch.cuda.synchronize()
            g = ops.Graph()
            g.begin()
            eng.forward(True); eng.backward(10.0); eng.update()
            g.end()
            g.launch(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eps.normal_()
                eng.eps.copy_(eps)
                g.launch()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print(json.dumps({"metric": "VAE samples/sec (ae_config step), batch=256 T=256 C=4", "value": round(Bv * args.steps / el, 1),
                          "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "cfg4: VAE step B=256, T=256, C=4, latent 8, beta 10"},
                          "losses": {"total": round(eng.loss[0].item(), 5)}}), flush=True)
        return
    from melo_gan_amd.gan.engine import GanEngine
    from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg
    for (Tg, Cg) in ((512, 4), (256, 128)):
        eng = GanEngine(default_gan_cfg(1, Tg, Cg), default_ed_cfg(Cg), "cuda", 1)
        eng.init_weights(42)
        z, num = torch.randn(1, 128, device="cuda"), torch.randn(1, 6, device="cuda")
        with torch.cuda.stream(eng.stream):
            for _ in range(5):
                eng.generate(z, num)
            torch.cuda.synchronize()
            g = ops.Graph()
            g.begin()
            eng._e_fwd(train=False, gin=True); eng._g_fwd(eng.notes, train=False)
            g.end()
            lat = []
            for _ in range(1000):
                eng.noise.copy_(z)
                t0 = time.perf_counter()
                g.launch()
                out = eng.notes[0, 0, 0].item()          # D2H of the result = end of the request (app.py:105)
                lat.append(time.perf_counter() - t0)
        lat.sort()
        print(json.dumps({"metric": "batch-1 generation latency (E_num -> G, eval)", "value": round(1e6 * lat[500], 1), "unit": "us p50",
                          "p99_us": round(1e6 * lat[990], 1), "n_gpus": 1, "steps": 1000, "higher_is_better": False, "dtype": "f32",
                          "config": {"workload": f"cfg5: B=1, T={Tg}, C={Cg}, hipGraph replay + 4-byte D2H sync"}}), flush=True)


def main():
    args = parse()
    if args.workload != "gan":
        return side_workload(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # Rehearsal knobs (never set by the driver): MELO_DIST_BACKEND=gloo + MELO_SHARE_GPU=1 let N ranks share one
    # GPU on a single-GPU box so the multi-rank control path can be exercised without an 8-GPU node.
    backend = os.environ.get("MELO_DIST_BACKEND", "nccl")
    if os.environ.get("MELO_SHARE_GPU") == "1":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    # MELO_FORCE_DP=1 (rehearsal, never set by the driver): a 1-rank RCCL group and the N > 1 step order on one GPU
    force_dp = os.environ.get("MELO_FORCE_DP") == "1" and world == 1
    if force_dp:
        os.environ.setdefault("MASTER_PORT", "29577")
    dist_on = world > 1 or force_dp
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    from melo_gan_amd.gan.engine import GanEngine
    from melo_gan_amd.gan.dp import DataParallel
    from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg

    cfg, ed_cfg = default_gan_cfg(B_PER_GPU, T, C), default_ed_cfg(C)
    eng = GanEngine(cfg, ed_cfg, f"cuda:{local_rank}", B_PER_GPU, ed_dtype=args.ed_dtype)
    eng.init_weights(seed=42)                       # identical on every rank
    dp = DataParallel(eng, world, dist if dist_on else None, force_collectives=force_dp)
    dp.broadcast_params()
    # synthetic data resident in HBM: a small pool of per-rank batches (SURVEY 8d recipe)
    g = torch.Generator().manual_seed(42 + 1000 * rank)
    pool = []
    for _ in range(4):
        real = (torch.rand(B_PER_GPU, T, C, generator=g) * 2 - 1).cuda()
        numeric = torch.randn(B_PER_GPU, 6, generator=g).cuda()
        latent = torch.zeros(B_PER_GPU, cfg["LATENT_DIM"]).cuda()
        emot = torch.randint(0, 4, (B_PER_GPU,), generator=g).cuda()
        pool.append((real, numeric, latent, emot))
    eng.seed(1234 + rank)                            # rank-offset Philox key: every shard draws its own noise
    use_graph = not args.no_graph
    # the pool bound as a resident split: the step's first launch gathers its batch on the device (batch k = rows of pool[k % 4]),
    # as the trainer does with the epoch's shuffled order (GANDataset.bind) -- no host-side staging launch between steps
    bound = [torch.cat([b[j] for b in pool]) for j in range(4)]
    eng.bind_batches(*bound)

    def step(i):
        dp.step(use_graph)        # melo-gan_amd/gan/dp.py: the graphs of one step and, for N > 1, the in-graph collectives

    def local_step():
        """One eager step of the production launch sequence on THIS rank alone: no collectives of any kind."""
        keep = (eng.coll, eng.p2_world, eng.world_size)
        eng.coll, eng.p2_world = None, 0
        try:
            DataParallel(eng, 1, None).step(False)
        finally:
            eng.coll, eng.p2_world, eng.world_size = keep

    def barrier():
        if dist_on:
            # this script's own collectives stay off the engine's stream (async_op + wait on the process group's stream)
            dist.barrier(async_op=True).wait()

    settle_steps = 0
    with torch.cuda.stream(eng.stream):
        for i in range(max(args.warmup, 3 if use_graph else 1)):
            step(i)
        # The first warm-up steps are eager / capturing and leave the GPU mostly idle; a timed window that starts right
        # behind them sees the clocks still ramping (per-step events: 0.875 ms falling to 0.825 over the first ~15 replays
        # -- tools/first_steps.py).  The metric is steady-state throughput, so --settle-steps MORE untimed steps of the same
        # workload are replayed first (the same count on every rank); the line reports them (`warmup_settle_steps`).
        if use_graph and args.settle_steps > 0:
            for i in range(args.settle_steps):
                step(i)
            settle_steps = args.settle_steps
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        host_el = time.perf_counter() - t0           # the host's share: enqueueing K steps (it runs ahead of the GPU)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True).wait()
            el = float(t.item())

        # ---- per-step HIP events: a second window of >= 64 steps, each between its own event pair on the
        # engine's stream; the host runs ahead, so the events see device time only ----
        n_ev = max(args.steps, 64)
        evs = [ops.Event() for _ in range(n_ev + 1)]
        evs[0].record()
        for i in range(n_ev):
            step(i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        barrier()
        per = sorted(evs[i].elapsed_ms(evs[i + 1]) for i in range(n_ev))
        ev_stats = [per[n_ev // 2], per[n_ev // 10], per[(9 * n_ev) // 10]]
        if dist_on:      # the slowest rank's median (the ranks are in lock-step through the collectives anyway)
            t = torch.tensor(ev_stats, device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True).wait()
            ev_stats = t.tolist()
        event_timing = {"method": "hipEvent pair per step on the engine stream", "steps": n_ev,
                        "median_ms": round(ev_stats[0], 4), "p10_ms": round(ev_stats[1], 4), "p90_ms": round(ev_stats[2], 4),
                        "value_at_median": round(world * B_PER_GPU / (ev_stats[0] * 1e-3), 1)}

        # ---- secondary: the reference's schedule, CRITIC_ITERS = 5 critic updates per generator update ----
        sched = None
        if args.profile_steps > 0:
            iters = max(4, min(20, args.steps // 5))
            for k in range(2 * 5):
                dp.step(use_graph, g_step=(k % 5 == 4))
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(iters * 5):
                dp.step(use_graph, g_step=(k % 5 == 4))
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            el5 = time.perf_counter() - t0
            if dist_on:
                t = torch.tensor([el5], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True).wait()
                el5 = float(t.item())
            sched = {"metric": "samples/s at the reference schedule (5 critic updates : 1 generator update)",
                     "value": round(world * B_PER_GPU * iters * 5 / el5, 1), "batches": iters * 5,
                     "ms_per_batch": round(1e3 * el5 / (iters * 5), 4)}

        # ---- roofline leg: the dominant kernel's launches of one step, recorded and replayed under HIP events ----
        roof = roof2 = None
        if rank == 0 and args.profile_steps > 0:
            hook = RecordHook(DOMINANT + DOMINANT_DIRECT + CONV16)
            ops.set_launch_hook(hook)
            # the production launch sequence of one step, eagerly (the hook sees every launch) -- through a LOCAL wrapper and
            # with the engine's in-graph collectives switched off: only rank 0 runs this leg, so it must not issue
            # collectives (it did in an earlier version of this file and would have left rank 0 waiting for the others)
            local_step()
            torch.cuda.synchronize()
            ops.set_launch_hook(None)

            def leg(symbols, name):
                recs = [r for r in hook.records if r[0] in symbols]
                if not recs:
                    return None
                reps = args.profile_steps
                ms = time_dominant(ops, recs, reps)
                launches, flops = reps * len(recs), reps * sum(r[1] for r in recs)
                achieved = flops / (ms * 1e-3) / 1e12
                # HBM bytes per launch need the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
                # tools/pmc_traffic.py): they cannot be read from inside this process, so `traffic` is null here and the
                # measured figure lives under profiles/ with the command that produced it
                return dict(bound="mfma", achieved=round(achieved, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=None, kernel=name,
                            how="one dispatch at a time (recorded launches replayed, HIP event pair each)",
                            launches=launches, launches_per_step=len(recs), avg_us=round(1e3 * ms / launches, 2),
                            avg_gflop_per_launch=round(flops / launches / 1e9, 3))
            dom, dom_name = DOMINANT, DOMINANT_NAME
            roof = leg(dom, dom_name)
            if roof is None:                      # MELO_ED_WINO=0: the direct window GEMM is the dominant symbol again
                dom, dom_name = DOMINANT_DIRECT, DOMINANT_DIRECT_NAME
                roof = leg(dom, dom_name)
            elif any(r[0] in DOMINANT_DIRECT for r in hook.records):
                roof["note"] = ("conv1's data-gradient (64 output columns) stays on the direct window GEMM "
                                "(conv_wgemm_kernel<1,3,...>): not in this symbol")
            if roof is not None and dom is DOMINANT:
                roof["mfma_executed_frac"] = round(roof["frac"] * WINO_EXECUTED, 4)
                roof["flops"] = "algorithmic = the direct convolution's 2*B*T*N*Cin*3 per launch; executed on the matrix pipe: 2/3 of it"
            roof2 = leg(CONV16, CONV16_NAME)
            # the same kernels INSIDE the step (two streams, as replayed): frac_in_step is the figure to hold against the
            # rocprofv3 per-kernel average of profiles/; `frac` / `achieved` stay the one-dispatch-at-a-time measurement
            keep = (eng.coll, eng.p2_world, eng.world_size)
            eng.coll, eng.p2_world = None, 0
            try:
                for r_, syms in ((roof, dom), (roof2, CONV16)):
                    if r_ is None:
                        continue
                    ms_i, fl_i, n_i = time_in_step(ops, eng, syms, max(4, args.profile_steps // 2))
                    r_["in_step"] = {"achieved": round(fl_i / (ms_i * 1e-3) / 1e12, 2), "frac": round(fl_i / (ms_i * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                     "avg_us": round(1e3 * ms_i / n_i, 2), "launches": n_i,
                                     "how": "eager forked step (emotion branch on the side stream), HIP event pair per launch"}
            finally:
                eng.coll, eng.p2_world, eng.world_size = keep
            if roof is not None:
                tb, prov = measured_traffic()
                roof["traffic"] = tb
                roof["traffic_provenance"] = prov
                # algorithmic bytes per launch, averaged over the six launches of a step (ED conv1-3: forward reads x and w,
                # writes the activation AND the pre-activation; data-gradient reads dy, the saved pre-activation and w, writes dx)
                algo, n_l = 0.0, 0
                for ci, co in ((64, 128), (128, 256), (256, 256)):
                    xb, yb, wb = 4.0 * B_PER_GPU * T * ci, 4.0 * B_PER_GPU * T * co, 4.0 * ci * co * 3
                    algo += xb + 2 * yb + wb
                    n_l += 1
                    if dom is DOMINANT_DIRECT or ci >= 128:      # conv1's data-gradient is not a wino3 launch
                        algo += yb + 2 * xb + wb
                        n_l += 1
                roof["traffic_algorithmic"] = round(algo / n_l, 0)
                if tb:
                    roof["traffic_over_algorithmic"] = round(tb / (algo / n_l), 3)
                if "in_step" in roof and dom is DOMINANT:
                    roof["in_step"]["mfma_executed_frac"] = round(roof["in_step"]["frac"] * WINO_EXECUTED, 4)
        if rank == 0 and args.launch_flops:
            tally = {}

            def count(symbol, flops, launch=None):
                t = tally.setdefault(symbol, {"launches_per_step": 0, "gflop_per_step": 0.0})
                t["launches_per_step"] += 1
                t["gflop_per_step"] += flops / 1e9
                return ops._NullCtx()
            ops.set_launch_hook(count)
            local_step()
            torch.cuda.synchronize()
            ops.set_launch_hook(None)
            with open(args.launch_flops, "w") as f:
                json.dump(tally, f, indent=1)
        # launches per step = kernel nodes of the step's graph(s) as captured
        step_graphs = [v for k, v in eng._graphs.items() if k.split("#")[0] in ("dg_fork_step_rng", "dg_step_rng") and not isinstance(v, str)]
        launches_per_step = step_graphs[0][0].kernel_nodes if step_graphs else None
        loss_d, adv, emo = eng.loss_d_out[0].item(), eng.adv.item(), eng.emo.item()

    # ---- secondary configuration beside the fp32 headline (BASELINE.json configs[1] names bf16): the same step with the
    # frozen emotion discriminator's activations / folded weights STORED in bf16 (fp32 accumulate); all trained state fp32 ----
    sec = None
    if rank == 0 and world == 1 and args.ed_dtype == "fp32" and args.profile_steps > 0 and use_graph:
        try:
            eng2 = GanEngine(cfg, ed_cfg, f"cuda:{local_rank}", B_PER_GPU, ed_dtype="bf16")
            eng2.init_weights(seed=42)
            eng2.seed(1234)
            eng2.bind_batches(*bound)
            dp2 = DataParallel(eng2, 1, None)
            n2 = max(args.steps, 64)
            with torch.cuda.stream(eng2.stream):
                for i in range(5):
                    dp2.step(True)
                torch.cuda.synchronize()
                evs = [ops.Event() for _ in range(n2 + 1)]
                evs[0].record()
                for i in range(n2):
                    dp2.step(True)
                    evs[i + 1].record()
                torch.cuda.synchronize()
            per = sorted(evs[i].elapsed_ms(evs[i + 1]) for i in range(n2))
            med = per[n2 // 2]
            sec = {"metric": "piano-roll samples/sec (G+D step), batch=64 128x256 roll -- SECONDARY configuration",
                   "dtype": "f32; frozen emotion discriminator stored in bf16 (fp32 accumulate)",
                   "value": round(B_PER_GPU / (med * 1e-3), 1), "unit": "samples/s", "ms_per_step": round(med, 4),
                   "steps": n2, "timing": "median of per-step hipEvent pairs",
                   "losses": {"loss_d": round(eng2.loss_d_out[0].item(), 5), "adv": round(eng2.adv.item(), 5),
                              "emo": round(eng2.emo.item(), 5)}}
            del eng2, dp2
        except Exception as e:  # the headline line must not depend on the secondary configuration
            sec = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)
    if dist_on:
        dist.barrier(async_op=True).wait()
        dist.destroy_process_group()
    if rank == 0:
        value = world * B_PER_GPU * args.steps / el
        line = {
            "metric": "piano-roll samples/sec (G+D step), batch=64 128x256 roll",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "warmup_settle_steps": settle_steps, "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.ed_dtype == "fp32" else "f32; frozen emotion discriminator stored in bf16 (fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "cfg2: 128x256 piano-roll, batch=64 per GPU, full G+D+emotion-D step "
                                   "(1 critic update incl. gradient penalty + 1 generator update)",
                       "global_batch": world * B_PER_GPU, "T": T, "C": C, "parallelism": f"dp{world}",
                       "graph": use_graph, "step_gflop": round(MFLOP_PER_SAMPLE * B_PER_GPU / 1e3, 2),
                       # the reference's step computes the critic's weight gradients in the generator step as well and
                       # throws them away (train_gan.py:225): 1.69 GF per step that this build does not execute
                       "step_gflop_executed": round(MFLOP_PER_SAMPLE * B_PER_GPU / 1e3 - 1.69, 2),
                       "step_tflops_executed": round((MFLOP_PER_SAMPLE * B_PER_GPU / 1e3 - 1.69) / (1e3 * el / args.steps), 2),
                       "launches_per_step": launches_per_step,
                       "dp_mode": dp.mode if dp.active else None},
            "host_ms_per_step": round(1e3 * host_el / args.steps, 4),
            "event_timing": event_timing, "roofline": roof, "roofline_stride2_family": roof2, "cpu_baseline": cpu,
            "secondary_bf16_ed": sec,
            "secondary": sched,
            "losses": {"loss_d": round(loss_d, 5), "g_adv": round(adv, 5), "g_emo": round(emo, 5)},
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
