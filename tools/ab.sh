#!/bin/bash
# Alternating A/B of bench.py settings on ONE box in ONE call (the pool's boxes differ by ~2 %, runs drift by ~0.5 %: only
# same-call alternations of 300 steps resolve a 5-us change of the 0.82-ms step -- DESIGN section 7 has the casualties of
# anything less).   gpurun -- 'bash tools/ab.sh 4 "A=1" "MELO_TAIL_FORK=0" ...'   -> samples/s per setting and round, then means
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
declare -A SUM
for i in $(seq 1 $N); do
  for cfg in "$@"; do
    v=$(env $cfg python3 $R/bench.py --steps 300 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "
import sys, json
print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
    echo "round $i  $cfg  $v"
    SUM[$cfg]=$(python3 -c "print(${SUM[$cfg]:-0} + $v)")
  done
done
for cfg in "$@"; do python3 -c "v = ${SUM[$cfg]} / $N; print('mean  %-40s %.1f samples/s = %.4f ms/step' % ('$cfg', v, 64e3 / v))"; done
