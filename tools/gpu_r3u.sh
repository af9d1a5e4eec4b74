#!/bin/bash
# round 3, call u: ablation of the 13200-sample fused range launch in its two-workgroups-per-CU form (re / im planes)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
export SARX_MIXED_PLANES=1
for rep in 1 2; do
  echo "== planes base"; timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
  for x in 3 4 7; do
    echo "== planes MIX_ABL=$x"; SARX_LIB=$R/build/abl/libsarx_mixabl$x.so timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
  done
done
