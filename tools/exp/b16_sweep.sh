#!/bin/bash
# batch 16: which tile kernels take the 378-tile shapes
F="--batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary"
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run base A=1 && run thin64_400 NEURST_GEMM_THIN64_MAX_TILES=400 && run base A=1 && run thin64_800 NEURST_GEMM_THIN64_MAX_TILES=800 && run thin64_400 NEURST_GEMM_THIN64_MAX_TILES=400
