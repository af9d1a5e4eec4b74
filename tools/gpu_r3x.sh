#!/bin/bash
# round 3, call x: where the 8192-sample fused range launch (generic range_pass_kernel, configs 3 / 5) spends its time - RG_ABL ablation builds
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  echo "== base"; timeout -k 10 120 python3 tools/run_pass.py 23 8192 40 || exit 1
  for x in 1 2 3 4 7; do
    echo "== RG_ABL=$x"; SARX_LIB=$R/build/abl/libsarx_rgabl$x.so timeout -k 10 120 python3 tools/run_pass.py 23 8192 40 || exit 1
  done
done
echo "== 4096"; timeout -k 10 120 python3 tools/run_pass.py 23 4096 80
for x in 3 4 7; do echo "== 4096 RG_ABL=$x"; SARX_LIB=$R/build/abl/libsarx_rgabl$x.so timeout -k 10 120 python3 tools/run_pass.py 23 4096 80 || exit 1; done
