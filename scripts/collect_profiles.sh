#!/bin/bash
# Round-end evidence: bench lines, rocprofv3 kernel summaries of the same commands (eager launches: rocprofv3 on
# this image faults inside hipGraphLaunch), time-to-optimal table, phase stamps.  Outputs under gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit=$?"
timeout -k 10 300 python bench.py --workload netgen_1m_16m > $O/bench_netgen_1m_16m.json 2> $O/bench_1m.err; echo "bench 1m exit=$?"
cd /tmp && export TMPDIR=/tmp
for tag in default netgen_1m_16m; do
  args="--no-cpu-baseline --no-hbm-point"
  [ $tag = netgen_1m_16m ] && args="$args --workload netgen_1m_16m"
  MCF_USE_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o run -- python3 $R/bench.py $args > $O/prof_${tag}_bench.json 2> $O/prof_$tag.err
  echo "rocprof $tag exit=$?"
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$tag.csv
  rm -rf $O/prof_$tag
done
cd $R
rm -f gpurun_out/solve_times.log
timeout -k 10 400 python scripts/solve_times.py > $O/solve_times.out 2>&1; echo "solve_times exit=$?"; cp gpurun_out/solve_times.json $O/solve_times.json
timeout -k 10 200 python scripts/stamps_pivot.py > $O/stamps_pivot.txt 2>&1
timeout -k 10 100 python scripts/stamps_small.py > $O/stamps_small.txt 2>&1
ls -la $O
