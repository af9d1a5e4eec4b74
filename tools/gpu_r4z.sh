#!/bin/bash
# round 4, call z: last validation (whole GPU suite, smoke, default bench) + SQ / traffic counters of the azimuth launches
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench default FAILED"; tail -20 $O/bench_default.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default: %.3f ms/step %.1f fps | one in flight %.3f ms | roofline %.3f shared %.3f phi2 %.3f | batch64 %.1f | cpu %.4f' % (d['ms_per_step'], d['value'], d['one_frame_in_flight']['ms_per_step'], d['roofline']['frac'], d['roofline_shared']['frac'], d['roofline_rg_fft_phi2_pass']['frac'], d['batch64']['value'], d['cpu_baseline']['value']))"
bash tools/pmc.sh r4az1 1 16384 > /dev/null 2>&1; echo "pmc az1 rc $?"
bash tools/pmc.sh r4az4 4 16384 > /dev/null 2>&1; echo "pmc az4 rc $?"
bash tools/pmc_traffic.sh r4az1 1 16384 > /dev/null 2>&1; echo "traffic az1 rc $?"
bash tools/pmc_traffic.sh r4az4 4 16384 > /dev/null 2>&1; echo "traffic az4 rc $?"
python3 tools/pmc_summary.py $O/pmc_az_kernels.json gpurun_out/pmc_r4az1_A gpurun_out/pmc_r4az1_B gpurun_out/pmc_r4az4_A gpurun_out/pmc_r4az4_B gpurun_out/traffic_r4az1_FETCH_SIZE gpurun_out/traffic_r4az1_WRITE_SIZE gpurun_out/traffic_r4az4_FETCH_SIZE gpurun_out/traffic_r4az4_WRITE_SIZE > /dev/null 2>&1; echo "summary rc $?"
python3 -c "
import json; d=json.load(open('$O/pmc_az_kernels.json'))
for k,v in d['kernels'].items(): print(k[:70], v.get('share_of_wave_lifetime'), v.get('hbm_bytes_per_launch'), v.get('per_wave'), v['dispatch_ms_under_profiler'])"
rm -rf gpurun_out/pmc_r4az* gpurun_out/traffic_r4az*
