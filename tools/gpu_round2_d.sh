#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_anysize.py tests/test_gpu_rda.py tests/test_gpu_benchsize.py -q -m gpu -k "anysize or rda or slab" > gpurun_out/t_d.log 2>&1; echo "rc $?"; tail -12 gpurun_out/t_d.log
echo "== native"; timeout -k 10 300 python tools/bench_native.py > gpurun_out/native2.log 2>&1; echo "rc $?"; cat gpurun_out/native2.log
echo "== slab sweep 16384"
for M in 0 16 32 64 96 128; do
  echo "SARX_SLAB_MIB=$M"; SARX_SLAB_MIB=$M timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-batch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', round(d['value'],1),'frames/s', round(d['ms_per_step'],3),'ms', 'range launch ms', round(d['roofline']['launch_ms'],4))"
done
echo "== slab sweep 8192"
for M in 0 8 16 32 64; do
  echo "SARX_SLAB_MIB=$M"; SARX_SLAB_MIB=$M timeout -k 10 300 python bench.py --size 8192 --steps 40 --warmup 5 --no-cpu --no-batch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', round(d['value'],1),'frames/s', round(d['ms_per_step'],3),'ms')"
done
echo "== videosar/rda"; timeout -k 10 300 python tools/bench_videosar.py > gpurun_out/videosar2.log 2>&1; echo "rc $?"; cat gpurun_out/videosar2.log
