#!/bin/bash
F="--config configs/costt.yaml --no-cpu-baseline --no-kernel-timing --no-secondary"
run() { echo "== $*"; timeout -k 10 120 python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run --steps 50 --warmup 5 && run --steps 50 --warmup 30 && run --steps 50 --warmup 100 && run --steps 200 --warmup 5 && run --steps 50 --warmup 5
