#!/bin/bash
# round 3, call q: where the 13200-sample fused range launch spends its time - ablation builds of range_mixed.hip
# (MIX_ABL bits: 1 no global loads, 2 no global stores, 4 no butterflies / twiddles), library variants under build/abl/
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
for rep in 1 2; do
  echo "== base"; timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
  for x in 1 2 3 4 7; do
    echo "== MIX_ABL=$x"; SARX_LIB=$R/build/abl/libsarx_mixabl$x.so timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
  done
done
