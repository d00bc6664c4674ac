#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
  for lib in prev new; do
    [ $lib = prev ] && export MCF_HIP_LIB=$R/scripts/libmcf_prev.so || unset MCF_HIP_LIB
    echo "== $lib"; python scripts/prof_solve.py netgen_1m_16m 0 4000; python scripts/prof_solve.py netgen_1m_16m 2 8000; python scripts/prof_solve.py netgen_8_16a 0 20000
  done
done
