#!/bin/bash
# first GPU call of round 2: new parity tests, the Infinity-Cache slab question, SQ counters of the range kernels
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== mallbench" ; timeout -k 10 300 tools/mallbench.bin > gpurun_out/mallbench.log 2>&1; echo "rc $?"
echo "== new tests"; timeout -k 10 900 python -m pytest tests/test_gpu_benchsize.py tests/test_gpu_batch64.py -x -q -m gpu > gpurun_out/t_new.log 2>&1; echo "rc $?"; tail -5 gpurun_out/t_new.log
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "rc $?"; tail -c 600 gpurun_out/bench_a.err
