#!/bin/bash
# Round evidence: bench lines, rocprofv3 kernel summaries of the same commands, PMC passes (FETCH_SIZE / WRITE_SIZE,
# separate runs, --kernel-trace only), time-to-optimal table, phase stamps.  Outputs under gpurun_out/final/.
# rocprofv3 runs use eager launches (MCF_USE_GRAPH=0); see DESIGN.md section 5 for why.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
PART=${PART:-ABC}   # A: bench lines + rocprofv3 kernel summaries, B: PMC passes + solve times, C: stamps + windowed profile (one gpurun call each)
case $PART in *A*)
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-hbm-point > $O/bench_default_20steps.json 2> $O/bench_20.err; echo "bench 20 steps exit=$?"
timeout -k 10 300 python bench.py --workload netgen_1m_16m --no-hbm-point > $O/bench_netgen_1m_16m.json 2> $O/bench_1m.err; echo "bench 1m exit=$?"
timeout -k 10 500 python bench.py --workload netgen_6m_96m --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-point > $O/bench_netgen_6m_96m.json 2> $O/bench_4m.err; echo "bench 4m exit=$?"
cd /tmp && export TMPDIR=/tmp
for tag in default netgen_1m_16m netgen_6m_96m; do
  args="--no-cpu-baseline --no-hbm-point"
  [ $tag = netgen_1m_16m ] && args="$args --workload netgen_1m_16m"
  [ $tag = netgen_6m_96m ] && args="$args --workload netgen_6m_96m --steps 100 --warmup 10"
  MCF_USE_GRAPH=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o run -- python3 $R/bench.py $args > $O/prof_${tag}_bench.json 2> $O/prof_$tag.err
  echo "rocprof $tag exit=$?"
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$tag.csv
  rm -rf $O/prof_$tag
done
;; esac
case $PART in *B*)
cd /tmp && export TMPDIR=/tmp
for wl in netgen_1m_16m netgen_6m_96m netgen_8_08a; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=$O/pmc_${wl}_$ctr
    rm -rf $out
    MCF_USE_GRAPH=0 WL=$wl PIVOTS=64 REPS=10 timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 $R/scripts/pmc_sweep.py > $O/pmc_${wl}_$ctr.log 2>&1
    echo "pmc $wl $ctr exit=$?"; tail -1 $O/pmc_${wl}_$ctr.log
  done
  (cd $R && python scripts/pmc_summarize.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE > $O/pmc_$wl.txt)
  rm -rf $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE
done
# the tree update at 1 M nodes (candidate list, blocked preorder list): k_pivot / k_update_bpl over the first 3 000 pivots
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$O/pmc_update_1m_$ctr
  rm -rf $out
  MCF_USE_GRAPH=0 WL=netgen_1m_16m RULE=2 FULL_SWEEPS=0 PIVOTS=3000 REPS=1 timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 $R/scripts/pmc_sweep.py > $O/pmc_update_1m_$ctr.log 2>&1
  echo "pmc update 1m $ctr exit=$?"; tail -1 $O/pmc_update_1m_$ctr.log
done
(cd $R && python scripts/pmc_summarize.py $O/pmc_update_1m_FETCH_SIZE $O/pmc_update_1m_WRITE_SIZE > $O/pmc_update_netgen_1m_16m_candidate_list.txt)
rm -rf $O/pmc_update_1m_FETCH_SIZE $O/pmc_update_1m_WRITE_SIZE
cd $R
timeout -k 10 500 python scripts/solve_times.py > $O/solve_times.out 2>&1; echo "solve_times exit=$?"; cp gpurun_out/solve_times.json $O/solve_times.json
;; esac
case $PART in *C*)
cd $R
(cd $R/network_flow_solver_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMCF_STAMPS -o $R/scripts/libmcf_stamps.so mcf_engine.hip) && {
  timeout -k 10 200 python scripts/stamps_pivot.py > $O/stamps_pivot.txt 2>&1; echo "stamps pivot exit=$?"
  timeout -k 10 100 python scripts/stamps_small.py > $O/stamps_small.txt 2>&1; echo "stamps small exit=$?"
}
timeout -k 10 600 python scripts/late_phase_profile.py netgen_1m_16m 2 250000 > $O/late_phase_netgen_1m_16m.txt 2>&1; echo "late phase exit=$?"
;; esac
ls -la $O
