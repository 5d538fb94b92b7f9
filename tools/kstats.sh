#!/bin/bash
# Per-kernel time tables of bench.py under several environment settings, one rocprofv3 --kernel-trace --stats run each:
#   gpurun -- 'bash tools/kstats.sh TAG "A=1 B=0" "A=0" ...'   -> gpurun_out/kstats_TAG/<n>.csv + <n>.json (bench line) + summary.txt
# The program follows `--` directly (no env / bash -c wrapper under the profiler): settings are exported in this shell.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/kstats_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
n=0
for cfg in "$@"; do
  (
    for kv in $cfg; do export "$kv"; done
    python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $O/$n.json 2> $O/$n.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$n -o s -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
    cp $O/st$n/s_kernel_stats.csv $O/$n.csv; rm -rf $O/st$n
  )
  echo "[$n] $cfg: $(python3 -c "import json,sys; print(json.load(open('$O/$n.json'))['ms_per_step'])" 2>/dev/null)" | tee -a $O/summary.txt
  n=$((n+1))
done
python3 $R/tools/kstats_table.py $O >> $O/summary.txt 2>&1
tail -60 $O/summary.txt
