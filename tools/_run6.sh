set -x
python -m pytest tests/test_gpu_c1_fullsize.py tests/test_gpu_ops_local.py tests/test_gpu_paper_widths.py tests/test_gpu_stream_edges.py -x -q 2>&1 | tail -4 > gpurun_out/r5_t6.txt
SEHIP_NAMES=enc3.fwd,enc4.fwd,enc5.fwd,dec2.dg,dec1.dg,dec0.dg python tools/gemm_variants.py SEHIP_NO_C3_TAIL=1 base SEHIP_NO_C3_TAIL=1 base > gpurun_out/r5_ab6.txt 2>&1
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$*', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b6.txt; }
run A=0
run SEHIP_NO_C3_TAIL=1
run A=0
run SEHIP_NO_C3_TAIL=1
