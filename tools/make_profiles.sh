#!/bin/bash
# Regenerates the measurement artefacts quoted in DESIGN.md on a GPU box, into gpurun_out/profiles_$1/ (copy the ones to be
# judged into profiles/ with the round prefix).  One gpurun call:  gpurun --timeout 1100 -- 'bash tools/make_profiles.sh r03'
# PMC passes run alone (--pmc with --kernel-trace only), the program directly after `--` (no wrapper).
set -e -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1/7] bench line"; python3 $R/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "[2/7] kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
cp $OUT/stats/s_kernel_stats.csv $OUT/bench_kernel_stats.csv; rm -rf $OUT/stats
echo "[3/7] MFMA busy"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- python3 $R/bench.py --no-graph --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --launch-flops $OUT/launch_flops.json > /dev/null 2>&1
python3 $R/tools/pmc_mfma_busy.py $OUT/pmc_mfma/m_counter_collection.csv $OUT/pmc_mfma/m_kernel_trace.csv $OUT/launch_flops.json > $OUT/mfma_busy_pmc.json; rm -rf $OUT/pmc_mfma
echo "[4/7] FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --no-graph --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
echo "[5/7] WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $R/bench.py --no-graph --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv > $OUT/traffic_dominant_kernel.json; rm -rf $OUT/pmc_fetch $OUT/pmc_write
echo "[6/7] layer benches"; python3 $R/tools/layer_bench.py > $OUT/layer_bench.txt 2>/dev/null; python3 $R/tools/conv16_bench.py > $OUT/conv16_bench.txt 2>/dev/null
python3 $R/tools/wino_bench.py > $OUT/wino_bench.txt 2>/dev/null
echo "[7/7] other workloads"; python3 $R/tools/ref_shape_bench.py > $OUT/other_workloads.txt 2>/dev/null
for w in ae gen1 ed; do python3 $R/bench.py --workload $w --no-cpu-baseline >> $OUT/other_workloads.txt 2>/dev/null; done
ls -la $OUT
echo "[8/8] chains + timeline"; (cd $R/tools && python3 chain_bench.py > $OUT/chain_bench.txt 2>/dev/null; python3 ed_pad_bench.py > $OUT/ed_pad_bench.txt 2>/dev/null)
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o s -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
python3 $R/tools/timeline.py $OUT/tl/s_kernel_trace.csv 30 > $OUT/step_timeline.txt; rm -rf $OUT/tl
python3 $R/tools/step_stamps.py 200 > $OUT/step_stamps.txt 2>/dev/null
ls -la $OUT
