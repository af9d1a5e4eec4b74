cd $GRAFT_REPO_ROOT
for NT in 0 1; do for SZ in 16384 8192 4096; do
 echo -n "SARX_AZ_NT=$NT size $SZ: "; SARX_AZ_NT=$NT timeout -k 10 300 python bench.py --size $SZ --steps 40 --warmup 5 --no-cpu --no-batch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'frames/s', round(d['ms_per_step'],4),'ms', {k:v['ms'] for k,v in d['passes'].items() if k.startswith('az')})"
done; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchsize.py -q -m gpu -x -k "each_pass or full_scene or large" 2>&1 | tail -1
