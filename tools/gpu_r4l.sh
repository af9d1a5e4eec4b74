#!/bin/bash
set -o pipefail
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 300 ./tools/mallpipe.bin > $O/mallpipe.log 2>&1 || { echo "mallpipe FAILED"; tail -20 $O/mallpipe.log; exit 1; }
grep -v "results ok" $O/mallpipe.log
