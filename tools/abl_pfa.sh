cd /tmp && export TMPDIR=/tmp
for A in 0 1 2; do
  SARX_PFA_ABL=$A timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$A -- python3 $GRAFT_REPO_ROOT/tools/bench_native.py > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/abl_$A -name "*kernel_stats.csv" | head -1); echo "ABL=$A"; grep rader $f | cut -d, -f1-4
done
