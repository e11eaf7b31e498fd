mkdir -p gpurun_out/r3
for rep in 1 2; do
for tiles in 256 512; do
for mr in 2048 1024 0; do
  for cfg in costt; do
    NEURST_GROUP_TILES=$tiles NEURST_GROUP_MIN_ROWS=$mr timeout -k 10 200 python bench.py --config configs/$cfg.yaml --no-cpu-baseline --no-kernel-timing --no-secondary --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg tiles=$tiles min_rows=$mr', round(d['ms_per_step'],3))" || exit 1
  done
done
done
done
