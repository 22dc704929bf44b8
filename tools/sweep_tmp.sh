run() { env "$@" python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for g in 256 512 768 1024; do echo SW=$g; SEHIP_SW_WGS=$g python tools/wg_table.py wg 2>&1 | grep "small_wgrad" | awk '{print $1, $2}' | tr '\n' ' '; echo; done
for i in 1 2; do
run SEHIP_SW_WGS=256
run SEHIP_SW_WGS=512
run SEHIP_SW_WGS=768
done
