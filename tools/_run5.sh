set -x
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$*', json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r5_b5.txt; }
run A=0
run SEHIP_W3_CLASSES=7
run SEHIP_W3_CLASSES=7 SEHIP_W3_WGS=128
run SEHIP_W3_CLASSES=7 SEHIP_W3_WGS=192
run SEHIP_W3_CLASSES=3
run SEHIP_W3_CLASSES=5
run SEHIP_W3_CLASSES=7 SEHIP_W3_DEC_J=8
run SEHIP_W3_CLASSES=7 SEHIP_W3_DEC_J=12
run SEHIP_W3_WGS=128
run SEHIP_W3_WGS=160
run SEHIP_CW_SPLITS=16
run SEHIP_C3_PAIR_TM=8
run SEHIP_C3_PAIR_TM=6
run A=0
