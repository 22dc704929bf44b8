# usage: bash tools/_ab_step.sh "VAR=1" "lib:tools/_var_x.so" base ...   -> ms_per_step of bench.py for every variant, in order
for v in "$@"; do
  if [ "$v" = base ]; then e="A_=0"; elif [ "${v#lib:}" != "$v" ]; then e="SEHIP_LIB=${v#lib:}"; else e="$v"; fi
  env $e python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; print('$v', round(json.loads(sys.stdin.read())['ms_per_step'], 4))"
done
