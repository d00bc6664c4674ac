#!/bin/bash
# same-box A/B of two builds of the library (scripts/libmcf_prev.so vs the in-tree one); args: instance names
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
names=${@:-netgen_8_14a netgen_8_12a}
for i in 1 2; do
  echo "== prev";  MCF_HIP_LIB=$R/scripts/libmcf_prev.so python scripts/quick_rates.py $names
  echo "== new";   python scripts/quick_rates.py $names
done
