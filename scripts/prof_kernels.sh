#!/bin/bash
# rocprofv3 kernel-trace summaries for one (partial) solve per instance; outputs under gpurun_out/prof_<tag>/
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p $out
MCF_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $R/scripts/prof_solve.py "$@" > $out/stdout.log 2> $out/stderr.log
rc=$?
tail -1 $out/stdout.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && cat $out/kernel_stats.csv | cut -c1-200
find $out -name '*.db' -delete; find $out -name '*_kernel_trace.csv' -delete
exit $rc
