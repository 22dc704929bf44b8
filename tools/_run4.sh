set -x
python -m pytest tests/test_gpu_dense_group.py -x -q 2>&1 | tail -15 > gpurun_out/r5_t4.txt
python -m pytest tests/test_gpu_c1_fullsize.py tests/test_gpu_solver.py tests/test_gpu_deterministic.py tests/test_gpu_lstm_fused.py -x -q 2>&1 | tail -5 >> gpurun_out/r5_t4.txt
for v in base SEHIP_NO_DENSE_GROUP=1 base SEHIP_NO_DENSE_GROUP=1; do
  if [ $v = base ]; then python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b4.txt
  else env $v python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b4.txt; fi
done
bash tools/_tl.sh > gpurun_out/r5_tl.log 2>&1; cp gpurun_out/tl/gaps.txt gpurun_out/r5_gaps_b.txt
