#!/bin/bash
# COSTT step against the number of frames the shrunk tensor is sized for (steady state: 200 steps after 30 of warm-up)
F="--config configs/costt.yaml --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-timing --no-secondary"
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run default A=1 && run min64 NEURST_SHRINK_MIN_BOUND=64 && run min128 NEURST_SHRINK_MIN_BOUND=128 && run min256 NEURST_SHRINK_MIN_BOUND=256 && run full NEURST_SHRINK_MIN_BOUND=500 && run default A=1
