#!/bin/bash
# On the GPU box (through gpurun): the five rocprofv3 passes behind profiles/<tag>_*  (tools/make_profiles.py <tag> turns them
# into the committed summaries).  Each pass is a run of its own - counters never share a run with --kernel-trace/--stats.
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r03'
set -o pipefail
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --serial-streams"
scmd="python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --serial-streams"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $scmd > $out/stats.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $cmd > $out/fetch.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $cmd > $out/write.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc MfmaUtil --output-format csv -d $out/mfma -o m -- $cmd > $out/mfma.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/lds -o l -- $cmd > $out/lds.log 2>&1
rc=$?
# rocprofv3 nests its files under a host-name directory: flatten
for d in stats fetch write mfma lds; do find $out/$d -mindepth 2 -name '*.csv' -exec mv {} $out/$d/ \; 2>/dev/null; done
ls $out/*
exit $rc
