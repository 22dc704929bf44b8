python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_c1_fullsize.py tests/test_gpu_dcunet.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu" | tail -6
bash tools/_ab_step.sh base SEHIP_NO_PARALLEL_HEAD=1 base SEHIP_NO_PARALLEL_HEAD=1 base SEHIP_NO_PARALLEL_HEAD=1
python bench.py --steps 50 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']); print(json.dumps(j['roofline'])[:600]); [print(k) for k in j['kernel_classes']]"
