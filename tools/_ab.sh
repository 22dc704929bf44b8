timeout 900 python -m pytest tests/test_gpu_convtasnet.py tests/test_pit.py tests/test_gpu_evaluate.py -m gpu -x -q 2>&1 | tail -3
for v in "A=1" "SEHIP_CTN_NO_MFMA_DECODER=1" "SEHIP_CTN_DEC_WGS=256" "SEHIP_CTN_DEC_WGS=1024"; do
  echo "== $v"
  env $v python bench.py --workload convtasnet --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tl/ctn2 -o run -- python3 $R/bench.py --workload convtasnet --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
f=$(ls $R/gpurun_out/tl/ctn2/*/run_kernel_stats.csv $R/gpurun_out/tl/ctn2/run_kernel_stats.csv 2>/dev/null | head -1)
grep -i "decoder_fwd" $f | cut -c1-40,150-260
rm -rf $R/gpurun_out/tl/ctn2
