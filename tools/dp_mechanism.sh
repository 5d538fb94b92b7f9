#!/bin/bash
# What the N > 1 step order costs before any byte moves: bench.py on ONE GPU with a 1-rank RCCL group (MELO_FORCE_DP=1: the
# data-parallel control path, collectives included, transfers take no time) against the plain single-GPU line, alternating on
# the same box.   gpurun -- 'bash tools/dp_mechanism.sh' -> gpurun_out/dp_mechanism.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/dp_mechanism.txt
line() { python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]
print('%-44s ms/step %.4f  event median %.4f  dp_mode %s  kernel nodes/step %s' % ('$1', d['ms_per_step'], d['event_timing']['median_ms'], d['config']['dp_mode'], d['config']['launches_per_step']))"; }
: > $O
for i in 1 2 3; do
  python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | line "plain single GPU" | tee -a $O
  MELO_FORCE_DP=1 python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | line "1-rank RCCL group, ingraph (default)" | tee -a $O
done
for m in gather allreduce; do
  MELO_FORCE_DP=1 MELO_DP_MODE=$m python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | line "1-rank RCCL group, $m (round-2 order)" | tee -a $O
done
