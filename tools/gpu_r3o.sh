#!/bin/bash
# round 3, call o: final validation at HEAD: whole GPU suite, smoke, default-flag bench, the evidence bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3o; O=gpurun_out/r3o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "tests rc $?"; tail -4 $O/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc $?"
timeout -k 10 900 python bench.py --steps 20 --warmup 3 --stack all --passes > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; grep "pass\]" $O/bench.err
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu --stack multilook --batch-scene c3 > $O/bench_c3scene.json 2>/dev/null; echo "bench c3 rc $?"
python3 -c "
import json
for f in ('bench_default','bench','bench_c3scene'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), round(d['roofline']['frac'],3), round(d.get('roofline_rg_fft_phi2_pass',{}).get('frac',0),3), {k:round(v['value'],1) for k,v in d.items() if k.startswith('batch64')})"
