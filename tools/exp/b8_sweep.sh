#!/bin/bash
# batch 8 (64 pairs): tiled attention kernels (default rule: fused from 128 pairs) against the fused ones with their split
F="--batch 8 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary"
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run default A=1 && run fused2 NEURST_MHA_FUSED=2 && run default A=1 && run fused2 NEURST_MHA_FUSED=2
