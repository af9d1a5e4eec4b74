#!/bin/bash
# round 4, call m: bench line with roofline from the solo leg; 2-rank rehearsal on the one GPU (gloo fallback, cpu_baseline on an N>1 line); force-comm path with lanes
set -o pipefail
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench default FAILED"; tail -20 $O/bench_default.err; exit 1; }
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("default: %.3f ms/step %.1f fps | solo leg %.3f ms | roofline %.3f (%.3f ms; %s...) | shared %.3f (%.3f ms) | frame bw %.3f/%.3f | batch64 %.1f | cpu %.4f" % (
  d['ms_per_step'], d['value'], d['one_frame_in_flight']['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['measured_in'][:40],
  d['roofline_shared']['frac'], d['roofline_shared']['launch_ms'], d['frame_bandwidth']['frac_of_peak_algorithmic'], d['frame_bandwidth']['frac_of_peak_moved'], d['batch64']['value'], d['cpu_baseline']['value']))
print(d['roofline']['traffic'], d['roofline']['traffic_source'][:60], d['env_switches'])
PY
SARX_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --steps 50 --warmup 3 --no-cpu --no-batch > $O/bench_forcecomm.json 2> $O/bench_forcecomm.err || { echo "force-comm FAILED"; tail -20 $O/bench_forcecomm.err; exit 1; }
python3 -c "import json; d=json.loads(open('$O/bench_forcecomm.json').read().strip().splitlines()[-1]); print('force-comm (gather path on one GPU): %.3f ms/step' % d['ms_per_step'], d.get('collective',{}).get('transport'))"
timeout -k 10 600 python bench.py --gpus 2 --steps 10 --warmup 2 --size 4096 --batch-frames 8 --batch-size 2048 > $O/bench_2rank.json 2> $O/bench_2rank.err || { echo "2-rank FAILED"; tail -30 $O/bench_2rank.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_2rank.json').read().strip().splitlines()[-1]); print('2 ranks on one GPU: n_gpus', d['n_gpus'], 'collective_ok', d.get('collective_ok'), 'value %.1f' % d['value'], 'cpu_baseline' in d, d['cpu_baseline']['sample'][-90:], 'batch64 %.1f' % d['batch64']['value'])"
