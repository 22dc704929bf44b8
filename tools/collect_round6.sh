#!/bin/bash
# Round-6 evidence (on the GPU box, from the repo root:  bash tools/collect_round6.sh <commit> ; results land in gpurun_out/r6c/,
# the summaries are then copied into profiles/ by hand):
#   bench.json / bench_{dcunet,convtasnet,demucs}.json   bench lines incl. roofline (traffic measured IN the run), cpu_baseline, parity
#   kernel_stats.csv / kernel_stats_serial.csv           rocprofv3 --kernel-trace --stats of the DCCRN bench (overlapped / every kernel alone)
#   kernel_stats_{dcunet,convtasnet,demucs}.csv          the same for the other workloads (overlapped)
#   traffic.json / traffic_{dcunet,convtasnet,demucs}.json   FETCH_SIZE / WRITE_SIZE PMC passes over the real step (tools/traffic_summary.py)
#   mfma_util.json                                       MFMA busy / instruction counters per kernel class (tools/mfma_util_summary.py)
#   gaps.txt / gaps_{dcunet,convtasnet,demucs}.txt       two-queue timeline of one overlapped step (tools/trace_gaps.py)
set +e
COMMIT=${1:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6c
mkdir -p $OUT
cd $ROOT
timeout 1200 python bench.py 2>$OUT/bench.err | grep '^{' | tail -1 > $OUT/bench.json
for w in dcunet convtasnet demucs; do
  timeout 900 python bench.py --workload $w --steps 20 --warmup 3 2>$OUT/bench_$w.err | grep '^{' | tail -1 > $OUT/bench_$w.json
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_under_rocprof.json
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1
for w in dcunet convtasnet demucs; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o run -- python3 $ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-traffic > /dev/null 2>&1
done
for w in dccrn dcunet convtasnet demucs; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pm_$w/pmc_fetch -o run -- python3 $ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pm_$w/pmc_write -o run -- python3 $ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1
done
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_mfmaA -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_mfmaB -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1
for d in prof prof_serial prof_dcunet prof_convtasnet prof_demucs; do
  f=$(ls $OUT/$d/*/run_kernel_stats.csv $OUT/$d/run_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats${d#prof}.csv
done
t=$(ls $OUT/prof/*/run_kernel_trace.csv $OUT/prof/run_kernel_trace.csv 2>/dev/null | head -1)
cd $ROOT
[ -n "$t" ] && SEHIP_TRACE_TIMELINE=1 python tools/trace_gaps.py $t > $OUT/gaps.txt 2>&1
# the same two-queue timeline for the other workloads (their step starts at a different kernel)
for ws in "convtasnet:void ctn_encoder_fwd" "dcunet:dcunet_pack_input_kernel" "demucs:dmx_prep_up_kernel"; do
  w=${ws%%:*}; first=${ws#*:}
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$w -o run -- python3 $ROOT/bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1)
  tt=$(ls $OUT/tl_$w/*/run_kernel_trace.csv $OUT/tl_$w/run_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$tt" ] && SEHIP_TRACE_START="$first" SEHIP_TRACE_TIMELINE=1 python tools/trace_gaps.py $tt > $OUT/gaps_$w.txt 2>&1
  rm -rf $OUT/tl_$w
done
# where a conv_gemm_v3 launch spends its cycles (s_memtime stamps; needs tools/_var_stamps.so: python tools/build_variant.py stamps conv3.hip -DC3_STAMPS)
[ -f tools/_var_stamps.so ] && python tools/c3_stamps.py enc3.fwd enc4.fwd enc5.fwd dec2.dg dec1.dg dec0.dg dec0.fwd0 dec2.fwd0 enc3.dg0 enc4.dg0 2>/dev/null | grep -v amdgpu.ids > $OUT/c3_phase_budget.txt
python tools/traffic_summary.py $OUT/pm_dccrn 3 > $OUT/traffic.json 2>$OUT/traffic.err
for w in dcunet convtasnet demucs; do python tools/traffic_summary.py $OUT/pm_$w 3 > $OUT/traffic_$w.json 2>>$OUT/traffic.err; done
python tools/mfma_util_summary.py $OUT $COMMIT > $OUT/mfma_util.json 2>$OUT/mfma.err
python - <<PY
import json
for suffix in ("", "_dcunet", "_convtasnet", "_demucs"):
    p = "$OUT/traffic%s.json" % suffix
    try:
        j = json.load(open(p)); j["_whole_step"]["commit"] = "$COMMIT"
        if suffix: j["_whole_step"].pop("algorithmic_bytes_per_step", None)      # (that constant is DCCRN's)
        json.dump(j, open(p, "w"), indent=1)
    except Exception as e:
        print("traffic stamp failed", p, e)
PY
rm -rf $OUT/prof $OUT/prof_serial $OUT/prof_dcunet $OUT/prof_convtasnet $OUT/prof_demucs $OUT/pm_dccrn $OUT/pm_dcunet $OUT/pm_convtasnet $OUT/pm_demucs $OUT/pmc_mfmaA $OUT/pmc_mfmaB
cut -c1-300 $OUT/bench.json; ls -la $OUT
