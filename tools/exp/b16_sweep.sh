#!/bin/bash
# batch 16: fused attention with a (batch, head) pair on one workgroup or on two (forward: query groups, backward: key blocks)
F="--batch 16 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary"
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run split0 NEURST_MHA_SPLIT=0 && run fwd NEURST_MHA_SPLIT=1 && run both NEURST_MHA_SPLIT=3 && run bwd NEURST_MHA_SPLIT=2 && run split0 NEURST_MHA_SPLIT=0 && run fwd NEURST_MHA_SPLIT=1 && run both NEURST_MHA_SPLIT=3
