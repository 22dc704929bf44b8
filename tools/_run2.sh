set -x
PAIRS=dec0.fwd0+dec0.fwd1,dec1.fwd0+dec1.fwd1,dec2.fwd0+dec2.fwd1,enc5.dg0+enc5.dg1,enc4.dg0+enc4.dg1,enc3.dg0+enc3.dg1
python -m pytest tests/test_gpu_c1_fullsize.py tests/test_gpu_ops_local.py tests/test_gpu_paper_widths.py -x -q 2>&1 | tail -4 > gpurun_out/r5_t2.txt
SEHIP_NAMES=$PAIRS python tools/gemm_variants.py SEHIP_NO_C3_PAIR=1 base SEHIP_C3_PAIR_TM=8 SEHIP_C3_PAIR_TM=6 SEHIP_NO_C3_PAIR=1 base > gpurun_out/r5_ab2.txt 2>&1
for v in base SEHIP_NO_C3_PAIR=1 SEHIP_C3_PAIR_TM=6 SEHIP_C3_PAIR_TM=8 base SEHIP_NO_C3_PAIR=1; do
  if [ $v = base ]; then python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b2.txt
  else env $v python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b2.txt; fi
done
