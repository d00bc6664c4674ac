#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per run, --kernel-trace only) for the sweep workloads; outputs under gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in netgen_1m_16m netgen_6m_96m; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=$O/pmc_${wl}_$ctr
    rm -rf $out
    MCF_USE_GRAPH=0 WL=$wl PIVOTS=64 REPS=10 timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 $R/scripts/pmc_sweep.py > $O/pmc_${wl}_$ctr.log 2>&1
    echo "pmc $wl $ctr exit=$?"; tail -1 $O/pmc_${wl}_$ctr.log
  done
  (cd $R && python scripts/pmc_summarize.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE > $O/pmc_$wl.txt)
  rm -rf $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE
  grep "k_price_v" $O/pmc_$wl.txt | cut -c60-250
done
