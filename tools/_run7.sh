set -x
python -m pytest tests/test_gpu_fused_tail.py tests/test_gpu_solver.py tests/test_gpu_frontend.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu" | tail -8 > gpurun_out/r5_t7.txt
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$*', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b7.txt; }
run A=0
run SEHIP_NO_FUSED_TAIL=1
run A=0
run SEHIP_NO_FUSED_TAIL=1
