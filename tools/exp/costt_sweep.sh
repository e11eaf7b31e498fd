#!/bin/bash
# COSTT: the fused attention kernels on every shape they can take
F="--config configs/costt.yaml --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary"
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run base A=1 && run fused2 NEURST_MHA_FUSED=2 && run base A=1 && run fused2 NEURST_MHA_FUSED=2
