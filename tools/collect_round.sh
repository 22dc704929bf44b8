#!/bin/bash
# Round evidence (run on the GPU box from the repo root:  bash tools/collect_round.sh r2 ; results land in gpurun_out/<tag>/):
#   bench.json                 the default bench line (incl. CPU baseline + parity block)
#   bench_dcunet.json          python bench.py --workload dcunet      (bench_convtasnet.json / bench_demucs.json: --workload convtasnet / demucs)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline`
#   kernel_stats_serial.csv    the same with SEHIP_NO_SIDE_STREAM=1 (every kernel alone on the GPU) and no roofline pass
#   pmc_fetch/, pmc_write/     PMC passes over the REAL step (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one;
#                              no trace domain besides --kernel-trace) -> tools/traffic_summary.py -> profiles/<tag>_traffic.json
set +e
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py 2>$OUT/bench.err | grep '^{' | tail -1 > $OUT/bench.json
timeout 600 python bench.py --workload dcunet --steps 20 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_dcunet.json
timeout 600 python bench.py --workload convtasnet 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_convtasnet.json
timeout 600 python bench.py --workload demucs --steps 30 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_demucs.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_under_rocprof.json
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dcunet -o run -- python3 $ROOT/bench.py --workload dcunet --steps 5 --warmup 2 --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_convtasnet -o run -- python3 $ROOT/bench.py --workload convtasnet --steps 10 --warmup 3 --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_demucs -o run -- python3 $ROOT/bench.py --workload demucs --steps 5 --warmup 2 > /dev/null 2>&1
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_demucs_serial -o run -- python3 $ROOT/bench.py --workload demucs --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for d in prof prof_serial prof_dcunet prof_convtasnet prof_demucs prof_demucs_serial; do
  f=$(ls $OUT/$d/*/run_kernel_stats.csv $OUT/$d/run_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats${d#prof}.csv
  rm -f $OUT/$d/*/run_kernel_trace.csv $OUT/$d/run_kernel_trace.csv
done
cd $ROOT
python tools/traffic_summary.py $OUT 3 > $OUT/traffic.json 2>$OUT/traffic.err
cut -c1-600 $OUT/bench.json
ls -la $OUT
