#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py -q -m gpu -x -k "look or batch64 or config5 or magnitude or two_ranks or rccl" > gpurun_out/t_h.log 2>&1; echo "rc $?"; tail -4 gpurun_out/t_h.log
timeout -k 10 300 python tools/bench_batch64.py 2>&1 | tail -1
SARX_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-batch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced comm (1 rank RCCL): ', round(d['value'],1), round(d['ms_per_step'],3), d.get('collective',{}).get('transport'))"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-batch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain: ', round(d['value'],1), round(d['ms_per_step'],3))"
