# usage: ab_env.sh VAR "v1 v2 ..." "cfg1 cfg2" [reps]
var=$1; vals=$2; cfgs=$3; reps=${4:-1}
for rep in $(seq $reps); do
for v in $vals; do
  for cfg in $cfgs; do
    env $var=$v timeout -k 10 200 python bench.py --config configs/$cfg.yaml --no-cpu-baseline --no-kernel-timing --no-secondary --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $var=$v', round(d['ms_per_step'],3))" || exit 1
  done
done
done
