set -x
NAMES=enc3.fwd,enc4.fwd,enc5.fwd,dec2.dg,dec1.dg,dec0.dg,dec0.fwd0+dec0.fwd1,dec1.fwd0+dec1.fwd1,dec2.fwd0+dec2.fwd1,enc5.dg0+enc5.dg1,enc4.dg0+enc4.dg1,enc3.dg0+enc3.dg1
python -m pytest tests/test_gpu_c1_fullsize.py tests/test_gpu_ops_local.py tests/test_gpu_paper_widths.py tests/test_gpu_stream_edges.py -x -q 2>&1 | tail -4 > gpurun_out/r5_t3.txt
SEHIP_NAMES=$NAMES python tools/gemm_variants.py lib:tools/_var_r4conv3.so base lib:tools/_var_r4conv3.so base > gpurun_out/r5_ab3.txt 2>&1
python tools/c3_stamps.py enc3.fwd enc4.fwd dec2.dg dec0.dg dec0.fwd0 enc3.dg0 > gpurun_out/r5_c3_stamps_3.txt 2>&1
for v in base lib base lib; do
  if [ $v = base ]; then python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b3.txt
  else SEHIP_LIB=tools/_var_r4conv3.so python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b3.txt; fi
done
