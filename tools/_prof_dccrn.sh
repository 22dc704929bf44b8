# kernel stats of the DCCRN step, overlapped and serial: gpurun -- bash tools/_prof_dccrn.sh <tag>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3}; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>$O/prof.err | grep '^{' | tail -1 > $O/bench_under_rocprof.json
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof_serial.log 2>&1
for d in prof prof_serial; do
  f=$(ls $O/$d/*/run_kernel_stats.csv $O/$d/run_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats${d#prof}.csv
done
cd $R; python tools/trace_gaps.py $(ls $O/prof/*/run_kernel_trace.csv $O/prof/run_kernel_trace.csv 2>/dev/null | head -1) > $O/gaps.txt 2>&1
rm -rf $O/prof $O/prof_serial
cut -c1-300 $O/bench_under_rocprof.json; head -30 $O/gaps.txt
