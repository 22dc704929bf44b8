set -x
NAMES=enc3.fwd,enc4.fwd,enc5.fwd,dec0.fwd0,dec0.fwd1,dec1.fwd0,dec1.fwd1,dec2.fwd0,dec2.fwd1,dec2.dg,dec1.dg,dec0.dg,enc5.dg0,enc5.dg1,enc4.dg0,enc4.dg1,enc3.dg0,enc3.dg1
python -m pytest tests/test_gpu_c1_fullsize.py tests/test_gpu_ops_local.py -x -q 2>&1 | tail -4 > gpurun_out/r5_t1.txt
SEHIP_VARIANT_SUMS=1 SEHIP_NAMES=$NAMES python tools/gemm_variants.py lib:tools/_var_r4conv3.so base lib:tools/_var_r4conv3.so base > gpurun_out/r5_ab1.txt 2>&1
python tools/c3_stamps.py enc3.fwd enc4.fwd dec2.dg dec0.dg dec0.fwd0 enc3.dg0 > gpurun_out/r5_c3_stamps_1.txt 2>&1
python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | cut -c1-200 > gpurun_out/r5_b1.txt
SEHIP_LIB=tools/_var_r4conv3.so python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r5_b1.txt
