#!/bin/bash
# round 4, call d: whole GPU suite on the pinned-pool / allreduce / advisor-fix build, two frames in flight experiment
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 300 python3 tools/bench_two_streams.py 16384 40 > $O/two_streams_16384.log 2>&1 || { echo "two_streams FAILED"; tail -20 $O/two_streams_16384.log; exit 1; }
cat $O/two_streams_16384.log
timeout -k 10 300 python3 tools/bench_two_streams.py 8192 100 > $O/two_streams_8192.log 2>&1 || { echo "two_streams FAILED"; tail -20 $O/two_streams_8192.log; exit 1; }
cat $O/two_streams_8192.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1 || { echo "GPU tests FAILED"; tail -40 $O/gputests.log; exit 1; }
tail -3 $O/gputests.log
