#!/bin/bash
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over scripts/pmc_sweep.py; summary under gpurun_out/pmc_r01g.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmc_$ctr
  rm -rf $out
  MCF_USE_GRAPH=0 PIVOTS=64 REPS=10 timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 $R/scripts/pmc_sweep.py > $R/gpurun_out/pmc_$ctr.log 2>&1
  echo "$ctr exit=$?"; tail -1 $R/gpurun_out/pmc_$ctr.log
done
cd $R
python scripts/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_r01g.txt
cat gpurun_out/pmc_r01g.txt
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name '*.csv' -size +200k -delete
