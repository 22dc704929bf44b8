#!/bin/bash
# Round-end evidence: the default bench line, rocprofv3 kernel stats of the same command, and the serial (no side stream)
# kernel stats.  Run on the GPU box from the repo root:  bash tools/collect_final.sh ; results land in gpurun_out/final/.
set +e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | tail -1 > $OUT/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_under_rocprof.json
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cp $OUT/prof/run_kernel_stats.csv $OUT/kernel_stats.csv
cp $OUT/prof_serial/run_kernel_stats.csv $OUT/kernel_stats_serial.csv
rm -rf $OUT/prof/run_kernel_trace.csv $OUT/prof_serial/run_kernel_trace.csv
cat $OUT/bench.json | cut -c1-400
